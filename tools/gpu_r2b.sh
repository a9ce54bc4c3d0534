#!/bin/bash
# abridged transform: exactness tests, then k_march alone (serial, one round) with and without it,
# then the same with -fno-slp-vectorize
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fastmath.py -x -q > $O/pytest_fast.log 2>&1; echo "rc $?" >> $O/pytest_fast.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fastmath.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
HZ_SERIAL=1 HZ_TWO_PASS=0 HZ_NO_FAST_MATH=1 timeout 300 $B > $O/b_serial_one_plain.json 2> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 $B > $O/b_serial_one_fast.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 $B > $O/b_serial_two_fast.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B > $O/b_two_fast.json 2>> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B > $O/b_one_fast.json 2>> $O/err.log
touch horizonator_amd/csrc/hz_kernels.hip
make -s -C horizonator_amd/csrc HIPFLAGS_EXTRA=-fno-slp-vectorize > $O/make.log 2>&1
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 $B > $O/b_noslp_serial_one_fast.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=0 HZ_NO_FAST_MATH=1 timeout 300 $B > $O/b_noslp_serial_one_plain.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B > $O/b_noslp_two_fast.json 2>> $O/err.log
tail -3 $O/pytest_fast.log; tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
