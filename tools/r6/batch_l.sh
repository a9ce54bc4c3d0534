#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | tail -2
run() { echo "== $*"; env "$@" HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "kept"; }
{
for rep in 1 2 3; do
run HZ_HOST_FIRST=25
run HZ_HOST_FIRST=12.5
run HZ_HOST_FIRST=8
run HZ_HOST_FIRST=12.5 HZ_HOST_PREFILL=45
run HZ_HOST_FIRST=12.5 HZ_HOST_PREFILL=15
done
} > $O/first_sector.txt 2>&1
cat $O/first_sector.txt
HZ_HOST_TIMES=1 HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "host path" | sed -n '5,9p' | cut -c60-330
