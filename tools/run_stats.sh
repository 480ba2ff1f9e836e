#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel stats of the bench's timed region alone (no proxy / render / CPU legs, so that the per-kernel
# averages are those of the 13 full-batch steps) next to the bench line of the same run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stats; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sprof -o b -- python3 $R/bench.py --no-proxy --no-render --no-cpu-baseline --no-cfg5 --no-sampler > $O/bench_line.json 2> $O/err.log
cp /tmp/sprof/b_kernel_stats.csv $O/kernel_stats.csv
cd $R; python tools/kernel_stats_grep.py $O/kernel_stats.csv attn_
python - <<'PY'
import json
d=json.loads(open('gpurun_out/stats/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['per_kernel_ms'], d['roofline']['avg_ms'])
PY
