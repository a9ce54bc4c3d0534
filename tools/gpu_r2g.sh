#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_cfg5.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
timeout 300 $B > $O/b_default.json 2>> $O/err.log
for t in 4 8 16 24 32; do HZ_COPY_THREADS=$t timeout 300 $B > $O/b_threads$t.json 2>> $O/err.log; done
HZ_PLAIN_COPY=1 timeout 300 $B > $O/b_plain.json 2>> $O/err.log
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f host %s' % (d['ms_per_step'], d.get('host_inclusive')))
except Exception as e: print(' failed', e)
"; done
