#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b27; mkdir -p $O
HZ_WT_SAVE=$O/wave_t.npy timeout 300 python tools/wave_timing.py > $O/wave_timing.txt 2>&1; head -8 $O/wave_timing.txt | cut -c1-200
python tools/wave_schedule.py $O/wave_t.npy
HZ_HOST_SECTORS=3 HZ_COPY_THREADS=2 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_multi.py 2>&1 | tail -2
