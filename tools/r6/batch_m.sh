#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
run() { echo "== $*"; env "$@" HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "kept"; }
{
for rep in 1 2 3; do
run HZ_HOST_PREFILL=30 HZ_HOST_PREFILL_SERIES=0
run HZ_HOST_PREFILL=45 HZ_HOST_PREFILL_SERIES=30
run HZ_HOST_PREFILL=60 HZ_HOST_PREFILL_SERIES=45
run HZ_HOST_PREFILL=75 HZ_HOST_PREFILL_SERIES=60
run HZ_HOST_PREFILL=100 HZ_HOST_PREFILL_SERIES=100
done
} > $O/prefill.txt 2>&1
cat $O/prefill.txt
