#!/bin/bash
# pipelined timelines (tools/timeline.py) of bench.py under each of the given environments: tools/gpu_tl.sh "<env A>" "<env B>" ...
cd /tmp; export TMPDIR=/tmp
for e in "$@"; do
  echo "=== $e"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl; env $e rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra --no-host > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f | head -34
  rm -f $f
done
