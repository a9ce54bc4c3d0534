#!/bin/bash
# round 6, batch f: where do the pool's threads and pages have to be?  (in situ: tools/host_inclusive.py cfg3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
lscpu | grep -i "numa\|model name\|socket\|thread" > $O/lscpu.txt; cat /sys/kernel/mm/transparent_hugepage/enabled >> $O/lscpu.txt; cat $O/lscpu.txt
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist)
run() { echo "== $*"; env "$@" HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "kept"; }
{
run HZ_COPY_THREADS=24
run HZ_COPY_NODE=here HZ_COPY_THREADS=24
run HZ_COPY_NODE=here HZ_COPY_THREADS=48
echo "== taskset node0 ($N0), 24 threads"; HZ_VERTEX_CACHE=0 HZ_COPY_THREADS=24 timeout 300 taskset -c $N0 python tools/host_inclusive.py cfg3 2>&1 | grep kept
echo "== taskset node1 ($N1), 24 threads"; HZ_VERTEX_CACHE=0 HZ_COPY_THREADS=24 timeout 300 taskset -c $N1 python tools/host_inclusive.py cfg3 2>&1 | grep kept
echo "== taskset node0, 48 threads"; HZ_VERTEX_CACHE=0 HZ_COPY_THREADS=48 timeout 300 taskset -c $N0 python tools/host_inclusive.py cfg3 2>&1 | grep kept
run HZ_HOST_PREFILL=0
run HZ_HOST_PREFILL=100
run HZ_COPY_THREADS=64
run HZ_COPY_THREADS=96
} > $O/numa.txt 2>&1
cat $O/numa.txt
