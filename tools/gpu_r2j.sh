#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host --no-extra"
run() { tag=$1; shift; env "$@" HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 $B > $O/b_serial_$tag.json 2>> $O/err.log; env "$@" HZ_TWO_PASS=1 timeout 300 $B > $O/b_pipe_$tag.json 2>> $O/err.log; }
run v153_d20 HZ_DEFER_MAX=20
run v153_d8 HZ_DEFER_MAX=8
run v153_dm1 HZ_DEFER_MAX=-1
touch horizonator_amd/csrc/hz_kernels.hip
make -s -C horizonator_amd/csrc HIPFLAGS_EXTRA=-DHZ_MARCH_WAVES=4 > $O/make.log 2>&1
run v128_d20 HZ_DEFER_MAX=20
run v128_dm1 HZ_DEFER_MAX=-1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_checksums.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
