#!/bin/bash
# what bounds a 1/8 sector's strips back to back: host time per call (tools/host_enqueue.py), and the kernels of a few strips in the
# middle of a series with their queues (rocprofv3 --kernel-trace of tools/sector_b2b.py, tools/timeline.py)
cd /tmp; export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/host_enqueue.py 2>&1 | grep -v amdgpu.ids
for e in "$@"; do
  echo "=== $e"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl; env $e HZ_G=8 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -- python3 $GRAFT_REPO_ROOT/tools/sector_b2b.py 2>&1 | grep "G="
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f --split k_pack_sparse | head -60
  rm -f $f
done
