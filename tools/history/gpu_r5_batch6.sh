#!/bin/bash
# round 5, sixth batch: what the box's host memory does (fill by thread count and node), the pool on the caller's node, 3 sectors
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b6; mkdir -p $O
gcc -O2 -fopenmp -o $O/hostfill tools/hostfill_bench.c horizonator_amd/csrc/hz_scatter.c -Ihorizonator_amd/csrc 2> $O/hostfill_build.txt && timeout 200 $O/hostfill > $O/hostfill.txt 2>&1
cat $O/hostfill.txt | head -40
numactl --hardware 2>/dev/null | head -12; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4
for node in "" here; do
  for t in "" 32; do
    echo "== HZ_COPY_NODE=$node HZ_COPY_THREADS=$t"
    HZ_COPY_NODE=$node HZ_COPY_THREADS=$t timeout 300 python tools/host_inclusive.py cfg3 sectors=1,2,3,4 2>&1 | grep "^cfg3:"
  done
done
