#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b16; mkdir -p $O
timeout 1500 python tools/ab_short_sequences.py > $O/ab.txt 2>&1
cat $O/ab.txt
