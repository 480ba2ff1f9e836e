#!/bin/bash
# GPU box: one DDPM reverse step at batch 4 (eager / fused / HIP-graph replay) and the kernel breakdown of the same probe under rocprofv3
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/probes/gpu_dev_sampler_time.py 4 100 2>&1 | tail -4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python3 $R/tools/probes/gpu_dev_sampler_time.py 4 100 > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/sp/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:16]:
    print("%-80s calls %6d avg %7.1f us  %5.1f %%"%(r['Name'][:80], int(r['Calls']), float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
