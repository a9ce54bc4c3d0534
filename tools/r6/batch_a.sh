#!/bin/bash
# round 6, batch a: the device-shipped host path (hz_k_ship.h) - correctness, then timings by variant
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_api.py tests/test_gpu_sequences.py -x -q -m gpu > $O/pytest_hostpath.txt 2>&1; echo "pytest rc $?" >> $O/pytest_hostpath.txt
tail -3 $O/pytest_hostpath.txt
HZ_INIT_TIMES=1 HZ_HOST_TIMES=1 timeout 600 python tools/host_inclusive.py cfg3 > $O/cfg3_default.txt 2> $O/cfg3_default_times.txt
cat $O/cfg3_default.txt
timeout 900 python tools/host_inclusive.py cfg3 sectors=1,2,3,4,6 env=HZ_SHIP_PRIORITY=0 env=HZ_SHIP_BLOCKS=16 env=HZ_SHIP_BLOCKS=32 env=HZ_SHIP_BLOCKS=128 env=HZ_SHIP_BLOCKS=256 > $O/cfg3_variants.txt 2>&1
grep -v "^hz_hip" $O/cfg3_variants.txt
for t in 16 32 48; do HZ_COPY_THREADS=$t timeout 300 python tools/host_inclusive.py cfg3 > $O/cfg3_threads$t.txt 2>&1; grep "kept\|equals" $O/cfg3_threads$t.txt; done
timeout 300 python tools/host_inclusive.py cfg2 > $O/cfg2.txt 2>&1; cat $O/cfg2.txt
