#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b40; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/atomic_one_address.hip -o $O/atomic_one_address 2> $O/build.txt && timeout 120 $O/atomic_one_address > $O/atomic_one_address.txt 2>&1
cat $O/atomic_one_address.txt
