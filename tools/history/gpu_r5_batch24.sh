#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final5; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -2
PART=c bash tools/gpu_final_r5.sh
cat $O/zoomed.txt | cut -c1-160
tail -30 $O/summit_kernels.txt
tail -5 $O/modes.txt
cat $O/multi_4ranks_one_gpu_c_loop.json | cut -c1-600
