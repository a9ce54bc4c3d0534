#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | tail -2
HZ_HOST_TIMES=1 timeout 400 python tools/host_inclusive.py cfg3 sectors=0,3,4 > $O/host.txt 2>&1
grep "^cfg3:" $O/host.txt
for n in 4; do grep " $n sector" $O/host.txt | sed -n '5,7p;30,31p' | cut -c60-460; done
timeout 300 python tools/host_inclusive.py cfg2 2>&1 | grep "^cfg2:"
