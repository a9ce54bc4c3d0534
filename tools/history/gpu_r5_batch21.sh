#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b21; mkdir -p $O
timeout 2400 python tools/ab_flags.py -DMR_EARLYZ_PAIRS -DMR_FAR_GATE "-DMR_EARLYZ_PAIRS -DMR_FAR_GATE" > $O/ab_flags.txt 2>&1
cat $O/ab_flags.txt
