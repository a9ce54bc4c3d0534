#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b54; mkdir -p $O
timeout 2400 python tools/ab_flags.py --host -DHZ_QSHARDS_FIRST_ROUNDS > $O/ab_flags.txt 2>&1; cat $O/ab_flags.txt
