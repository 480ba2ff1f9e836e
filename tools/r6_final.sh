#!/bin/bash
# round 6, final evidence pass: full GPU suite, the round's profile set, the rank-step profile, the default bench line
python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_gpu_tests_full.txt 2>&1; tail -3 gpurun_out/r6_gpu_tests_full.txt
bash tools/run_round_profiles.sh r6 > gpurun_out/r6_profiles.log 2>&1
bash tools/run_b8_profile.sh r6 8 > gpurun_out/r6_b8_profile.log 2>&1
for f in r6_attention_hbm_traffic_pmc.json r6_attention_sq_pmc.json r6_attention_sq_pmc_n2049.json r6_bench_timed_region.json r6_bench_timed_region_kernel_stats.csv r6_elementwise_hbm_traffic_pmc.json r6_render_kernel_stats_S128.csv r6_render_kernel_stats_S64.csv r6_render_sq_pmc.json r6_shade_sq_pmc.json; do cp gpurun_out/r6_profiles/$f profiles/; done
cp gpurun_out/r6_render/r6_render_hbm_traffic_pmc.json profiles/
python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r6_bench.json").read().strip().splitlines()[-1])
print("steps/s", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), d["roofline"]["stale_counters"], "hbm stale", d["roofline_hbm"]["stale_counters"], "query stale", d["render"]["roofline_query"].get("stale_counters"))
print("proxy8", d["strong_scaling_proxy"]["8"]); print("render", d["render"]["ms_per_view"])
PY
