#!/bin/bash
# round 6, batch c: k_ship beside the draws - how many workgroups, on one XCD or all
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "64 0" "2 0" "4 0" "16 0" "4 1" "16 1" "32 1"; do
set -- $v
HZ_SHIP_BLOCKS=$1 HZ_SHIP_XCD=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/tools/r6/host_trace_run.py > $GRAFT_REPO_ROOT/$O/kt.log 2>&1
echo "== blocks $1 one_xcd $2" >> $GRAFT_REPO_ROOT/$O/summary.txt
python3 $GRAFT_REPO_ROOT/tools/r6/trace_summary.py $GRAFT_REPO_ROOT/$O/kt >> $GRAFT_REPO_ROOT/$O/summary.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/$O/kt
done
cat $GRAFT_REPO_ROOT/$O/summary.txt
