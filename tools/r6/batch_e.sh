#!/bin/bash
# round 6, batch e: the landing + copy-engine host path: correctness, timings, a traced series
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_api.py tests/test_gpu_sequences.py -x -q -m gpu > $O/pytest_hostpath.txt 2>&1; echo "pytest rc $?" >> $O/pytest_hostpath.txt
tail -3 $O/pytest_hostpath.txt
HZ_VERTEX_CACHE=0 HZ_INIT_TIMES=1 HZ_HOST_TIMES=1 timeout 600 python tools/host_inclusive.py cfg3 > $O/cfg3_default.txt 2> $O/cfg3_default_times.txt
cat $O/cfg3_default.txt
HZ_VERTEX_CACHE=0 timeout 900 python tools/host_inclusive.py cfg3 sectors=0,1,2,3,4,6 env=HZ_HOST_SERIES_WHOLE=1 > $O/cfg3_variants.txt 2>&1
grep -v "^hz_hip" $O/cfg3_variants.txt
HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg2 > $O/cfg2.txt 2>&1; cat $O/cfg2.txt
cd /tmp; export TMPDIR=/tmp
HZ_VERTEX_CACHE=0 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/tools/r6/host_trace_run.py > $GRAFT_REPO_ROOT/$O/kt.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/r6/trace_tail.py $GRAFT_REPO_ROOT/$O/kt 22 > $GRAFT_REPO_ROOT/$O/timeline.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/$O/kt
