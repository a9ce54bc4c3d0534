#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
for dbg in 0 1 2; do for tp in 0 1; do
HZ_MARCH_DEBUG=$dbg HZ_SERIAL=1 HZ_TWO_PASS=$tp timeout 300 $B > $O/b_serial_tp${tp}_dbg${dbg}.json 2>> $O/err.log
done; done
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
