#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b12; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/exact_seq.hip -o $O/exact_seq 2> $O/build.txt && timeout 600 $O/exact_seq 40 > $O/exact_seq.txt 2>&1
cat $O/exact_seq.txt
bash tools/gpu_scene_kernels.sh cfg3_zoom45_summit > $O/summit_kernels.txt 2>&1; tail -40 $O/summit_kernels.txt
