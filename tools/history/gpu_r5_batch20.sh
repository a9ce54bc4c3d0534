#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b20; mkdir -p $O
timeout 2000 python tools/ab_flush_copies.py > $O/ab_flush.txt 2>&1
cat $O/ab_flush.txt
