#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_cfg5.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-host --no-extra"
timeout 300 $B > $O/b_hiz.json 2>> $O/err.log
HZ_NO_HIZ=1 timeout 300 $B > $O/b_nohiz.json 2>> $O/err.log
timeout 300 $B > $O/b_hiz2.json 2>> $O/err.log
HZ_NO_HIZ=1 timeout 300 $B > $O/b_nohiz2.json 2>> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B > $O/b_one.json 2>> $O/err.log
timeout 300 $B --config cfg2 > $O/b_cfg2.json 2>> $O/err.log
tail -3 $O/pytest.log | grep -v "Hostname\|Librccl\|version"
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
