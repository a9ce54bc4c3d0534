#!/bin/bash
# what the GPU box's host looks like (cores, vector extensions, transparent huge pages) and the host path as round 4 left it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
lscpu | head -30
nproc
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag /sys/kernel/mm/transparent_hugepage/shmem_enabled
free -g
uname -r
HZ_HOST_TIMES=1 timeout 600 python tools/host_inclusive.py cfg3
} > gpurun_out/probe_r5.txt 2>&1
tail -30 gpurun_out/probe_r5.txt
