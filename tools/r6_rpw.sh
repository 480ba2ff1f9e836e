#!/bin/bash
# round 6: rays per wave of the neighbour-query kernel -- bit-exact tests, then kernel stats per setting
O=gpurun_out/r6_rpw.txt; : > $O
python -m pytest tests/test_gpu_render.py -x -q -k "grid or query or compact or golden or brute" >> $O 2>&1
cd /tmp; export TMPDIR=/tmp
export NPCD_RENDERS=30
for S in 128 64; do
for rpw in 1 2 4 8; do
  export NPCD_S=$S NPCD_QUERY_RPW=$rpw
  rm -rf /tmp/rp_$rpw
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$rpw -o r -- python3 $GRAFT_REPO_ROOT/tools/probes/gpu_dev_render_time.py > /tmp/rp_$rpw.log 2>&1
  echo "== S=$S rpw=$rpw: $(tail -1 /tmp/rp_$rpw.log)" >> $GRAFT_REPO_ROOT/$O
  python3 - >> $GRAFT_REPO_ROOT/$O <<PY
import csv
rows = list(csv.DictReader(open("/tmp/rp_$rpw/r_kernel_stats.csv")))
for r in rows:
    if any(k in r["Name"] for k in ("grid_query", "compact_ordered")):
        print("   ", r["Name"].split("(")[0][-40:], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
done
cd $GRAFT_REPO_ROOT; grep -v amdgpu.ids $O | tail -40
echo "== attention forward forms at per-GPU batch 8" >> $O
for cfg in "X=1" "NPCD_ATTN_ROWX32=1" "NPCD_ATTN_FWD=64"; do
echo "-- $cfg" >> $O
env $cfg python tools/probes/gpu_dev_b8.py 8 30 2>&1 | grep "wall" >> $O
done
tail -8 $O
