#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b43; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; grep -E " passed| failed| error" $O/pytest_full.txt | tail -2
timeout 2400 python tools/ab_patch.py -R tools/patches/r5_queue_shards.diff > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-260
