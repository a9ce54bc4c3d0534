#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b55; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_series.py -x -q -m gpu 2>&1 | tail -1
timeout 2400 python tools/ab_patch.py -R tools/patches/r5_kbig_shard_counts_in_a_register.diff > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-260
