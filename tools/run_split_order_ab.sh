#!/bin/bash
# GPU box: the bench step with the left-over rows' small GEMM call in front of / behind the large one, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in small_first big_first; do
  if [ $m = small_first ]; then export NPCD_GEMM_SPLIT_SMALL_FIRST=1; else unset NPCD_GEMM_SPLIT_SMALL_FIRST; fi
  echo "== $m: $(timeout 900 python3 bench.py --steps 20 --warmup 5 --no-render --no-proxy --no-cfg5 --no-sampler --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
