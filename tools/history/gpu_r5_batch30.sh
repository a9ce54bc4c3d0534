#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b30; mkdir -p $O
timeout 1500 python tools/stress_random.py 300 > $O/stress.txt 2>&1; tail -4 $O/stress.txt
HZ_TWO_PASS=1 HZ_HIZ=1 timeout 1200 python tools/stress_random.py 150 1000 >> $O/stress.txt 2>&1; tail -3 $O/stress.txt
