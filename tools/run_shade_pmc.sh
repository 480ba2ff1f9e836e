cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shade_pmc; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/mem -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/mem.log 2>&1
python3 $R/tools/make_sq_pmc_json.py $O/sq/runc $O/sq.json shade_ | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='counters_mean_per_launch' and a!='wave_cycle_split'}, {a:round(b,2) for a,b in v.get('wave_cycle_split',{}).items()})
"
python3 $R/tools/pmc_summary.py $O/mem/runc/*counter_collection.csv 2>/dev/null | grep -i "shade_pairs" | head -12
tail -3 $O/mem.log
