#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; tail -3 $O/pytest_full.txt
for k in 1 2 3; do HZ_G=8,4,2 timeout 300 python tools/sector_b2b.py 2>&1 | tail -1; done | tee $O/sector_b2b.txt
timeout 600 python tools/sector_timing.py > $O/sector_timing.txt 2>&1; tail -12 $O/sector_timing.txt
