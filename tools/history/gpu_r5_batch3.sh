#!/bin/bash
# round 5, third batch: kernel stores into pinned host memory vs the copy engine; the host path with the lighter k_pack_host;
# the C series loop with gloo ranks sharing the GPU; the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b3; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $O/zero_copy tools/zero_copy.hip 2> $O/zero_copy_build.txt && timeout 120 $O/zero_copy > $O/zero_copy.txt 2>&1
cat $O/zero_copy.txt
HZ_HOST_TIMES=1 timeout 300 python tools/host_inclusive.py cfg3 sectors=1,4 > $O/host_times.txt 2>&1
grep "^cfg3" $O/host_times.txt; grep "4 sector" $O/host_times.txt | sed -n '4,6p' | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_bench_multi.py tests/test_gpu_rccl.py -x -q -m gpu > $O/pytest_multi.txt 2>&1
tail -3 $O/pytest_multi.txt; grep -n -B5 -A40 "^___" $O/pytest_multi.txt | head -80
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1
grep -E "passed|failed|error" $O/pytest_full.txt | tail -3
grep -n -B5 -A40 "^___" $O/pytest_full.txt | head -100
