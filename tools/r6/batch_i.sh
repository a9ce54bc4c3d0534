#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
HZ_VERTEX_CACHE=0 HZ_INIT_TIMES=1 HZ_HOST_TIMES=1 timeout 300 python tools/r6/first_call.py > $O/first_call.txt 2>&1
grep -v "draw:" $O/first_call.txt | cut -c1-330
run() { echo "== $*"; env "$@" HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "kept"; }
{
for rep in 1 2 3; do
run HZ_COPY_THREADS=24 HZ_COPY_NODE=gpu
run HZ_COPY_THREADS=24 HZ_COPY_NODE=any
run HZ_COPY_THREADS=32 HZ_COPY_NODE=gpu
run HZ_COPY_THREADS=32 HZ_COPY_NODE=any
done
} > $O/numa.txt 2>&1
cat $O/numa.txt
