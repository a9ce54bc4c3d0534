#!/bin/bash
# round 6, batch b: the host path on a time axis (kernel trace), priority on / off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for prio in 1 0; do
HZ_SHIP_PRIORITY=$prio timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$prio -- python3 $GRAFT_REPO_ROOT/tools/r6/host_trace_run.py > $GRAFT_REPO_ROOT/$O/kt$prio.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/r6/trace_tail.py $GRAFT_REPO_ROOT/$O/kt$prio 14 > $GRAFT_REPO_ROOT/$O/timeline_prio$prio.txt 2>&1
done
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" -delete
wc -l $O/timeline_prio*.txt
