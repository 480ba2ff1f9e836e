#!/bin/bash
# round-6 baseline: the own weight-gradient kernel at a rank's token count, the rank step eager / graph
mkdir -p gpurun_out
O=gpurun_out/r6_base.txt
: > $O
python tools/probes/gpu_dev_wgrad_own.py 4104 8208 >> $O 2>&1
python tools/probes/gpu_dev_smallbatch.py 8 >> $O 2>&1
NPCD_OWN_WGRAD=1 python tools/probes/gpu_dev_smallbatch.py 8 >> $O 2>&1
python tools/probes/gpu_dev_b8.py 8 30 >> $O 2>&1
python tools/probes/gpu_dev_step_graph.py 8 >> $O 2>&1
tail -40 $O
