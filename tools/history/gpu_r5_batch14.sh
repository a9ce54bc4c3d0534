#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b14; mkdir -p $O
timeout 1500 python tools/march_bounds.py > $O/march_bounds.txt 2>&1
cat $O/march_bounds.txt
