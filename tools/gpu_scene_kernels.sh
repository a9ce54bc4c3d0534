#!/bin/bash
# every kernel of a scene alone (HZ_SERIAL=1), three waited-for renders: tools/gpu_scene_kernels.sh <scene> [env ...]  (rocprofv3 --kernel-trace, tools/timeline.py)
cd /tmp; export TMPDIR=/tmp
sc=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl; env HZ_SERIAL=1 "$@" rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -- python3 $GRAFT_REPO_ROOT/tools/scene_times.py $sc 2>&1 | grep "$sc"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f --last 1 | head -40
rm -f $f
