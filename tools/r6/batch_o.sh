#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt40 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --zfar 40000 --no-cpu-baseline --no-extra --no-host --no-scenes > $GRAFT_REPO_ROOT/$O/bench40.json 2> $GRAFT_REPO_ROOT/$O/bench40.err
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find $O/kt40 -name "*_kernel_trace.csv" | head -1) > $O/timeline40.txt 2>&1
rm -rf $O/kt40
cut -c1-300 $O/bench40.json | head -2
wc -l $O/timeline40.txt
