#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for k in 1 2; do
timeout 300 $B > $O/b_wait$k.json 2>> $O/err.log
HZ_FAR_WAITS_NEAR=0 timeout 300 $B > $O/b_nowait$k.json 2>> $O/err.log
done
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
