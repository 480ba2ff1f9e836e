#!/bin/bash
# dev tool (GPU box): the full GPU suite, smoke, the default bench line and a rocprofv3 kernel-stats pass of the same command;
# results under gpurun_out/final/ (bench.json -> profiles/<tag>_bench.json; its kernel_stats.csv covers ALL bench legs -- the
# timed region's per-kernel averages come from tools/run_stats.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -o b -- python3 $R/bench.py > $O/bench_prof.json 2> $O/bench_prof.err
cp /tmp/fprof/b_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; ls -la $O
