#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b30; mkdir -p $O
timeout 1800 python tools/stress_random.py 100 400 > $O/stress.txt 2>&1; tail -4 $O/stress.txt
STRESS_HIZ=1 timeout 1200 python tools/stress_random.py 100 250 > $O/stress_hiz.txt 2>&1; tail -3 $O/stress_hiz.txt
