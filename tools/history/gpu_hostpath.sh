#!/bin/bash
# round 5: the new host path - its tests first, then the timings (sector sweep, timelines), then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_scatter.py tests/test_gpu_api.py tests/test_annot.py tests/test_naive_host_math.py -x -q -m gpu > gpurun_out/pytest_hostpath.txt 2>&1
tail -5 gpurun_out/pytest_hostpath.txt
grep -n -B5 -A40 "^___" gpurun_out/pytest_hostpath.txt | head -150
timeout 600 python tools/host_inclusive.py cfg3 sectors=0,1,2,4,8 > gpurun_out/host_inclusive_r5.txt 2>&1
HZ_HOST_TIMES=1 timeout 300 python tools/host_inclusive.py cfg3 >> gpurun_out/host_inclusive_r5.txt 2>&1
timeout 300 python tools/host_inclusive.py cfg2 >> gpurun_out/host_inclusive_r5.txt 2>&1
grep -v "^hz_hip host path" gpurun_out/host_inclusive_r5.txt | tail -20
grep "^hz_hip host path" gpurun_out/host_inclusive_r5.txt | sed -n '4,8p'
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_full.txt | tail -3
grep -n -B5 -A40 "^___" gpurun_out/pytest_full.txt | head -120
