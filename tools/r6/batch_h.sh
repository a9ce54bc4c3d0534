#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
HZ_VERTEX_CACHE=0 HZ_INIT_TIMES=1 HZ_DRAW_TIMES=1 HZ_HOST_TIMES=1 timeout 300 python tools/r6/first_call.py > $O/first_call.txt 2>&1
grep -v "draw:.* 0.0[0-9] ms" $O/first_call.txt | cut -c1-330
run() { echo "== $*"; env "$@" HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "kept"; }
{
run HZ_COPY_THREADS=24
run HZ_COPY_NODE=any
run HZ_COPY_THREADS=32
run HZ_COPY_THREADS=48
run HZ_COPY_THREADS=24
run HZ_COPY_NODE=any
} > $O/numa.txt 2>&1
cat $O/numa.txt
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -2
