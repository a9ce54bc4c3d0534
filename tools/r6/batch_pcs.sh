#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6pcs; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --output-format csv -d $GRAFT_REPO_ROOT/$O/pcs -- python3 $GRAFT_REPO_ROOT/tools/r6/pcs_run.py > $GRAFT_REPO_ROOT/$O/pcs.log 2>&1
echo "rc $?"; tail -5 $GRAFT_REPO_ROOT/$O/pcs.log
cd $GRAFT_REPO_ROOT
find $O/pcs -type f | head; du -sh $O/pcs 2>/dev/null
f=$(find $O/pcs -name "*pc_sampling*csv" | head -1); [ -n "$f" ] && { head -5 $f; wc -l $f; }
