#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_cfg5.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host --no-extra"
for n in 32 48 64 96 128 192; do HZ_NEAR_CELLS=$n timeout 300 $B > $O/b_near$n.json 2>> $O/err.log; done
HZ_SERIAL=1 timeout 300 $B > $O/b_serial.json 2>> $O/err.log
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
