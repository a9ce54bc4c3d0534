#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b42; mkdir -p $O
timeout 2400 python tools/ab_patch.py -R tools/patches/r5_queue_shards.diff > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-260
