#!/bin/bash
# GPU box: re-tune the library GEMM choices with operands rotated through a large buffer (cold caches), then the bench step with the
# committed table and with the new one, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
NPCD_TUNE_OUT=tunableop_rot.csv NPCD_TUNE_ROTATE=${ROT:-1024} NPCD_TUNE_MS=${MS:-60} NPCD_TUNE_ITERS=${ITERS:-40} timeout 2400 python3 tools/tune_gemms.py 2>&1 | tail -3
wc -l $O/tunableop_rot.csv
for rep in 1 2; do
  for t in committed rot; do
    if [ $t = committed ]; then unset NPCD_TUNED_CSV; else export NPCD_TUNED_CSV=$O/tunableop_rot.csv; fi
    echo "== $t: $(timeout 900 python3 bench.py --steps 20 --warmup 5 --no-render --no-proxy --no-cfg5 --no-sampler --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  done
done
