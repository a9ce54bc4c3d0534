#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b57; mkdir -p $O
timeout 2400 python tools/ab_flags.py -DHZ_QSHARDS_SECOND_ROUNDS > $O/ab_flags.txt 2>&1; cat $O/ab_flags.txt
