#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b25; mkdir -p $O
HZ_HOST_SECTORS=3 HZ_COPY_THREADS=2 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_multi.py > $O/mode_sectors3.txt 2>&1; tail -45 $O/mode_sectors3.txt | cut -c1-220
HZ_HOST_DENSE=1 timeout 600 python -m pytest tests/test_gpu_hostpath.py tests/test_fullsize_checksums.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/wave_timing.py > $O/wave_timing.txt 2>&1; tail -20 $O/wave_timing.txt | cut -c1-200
