#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
B="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-host --no-extra"
timeout 300 $B > $O/b_nfb3.json 2>> $O/err.log
timeout 300 $B > $O/b_nfb3b.json 2>> $O/err.log
for n in 2 4; do
touch horizonator_amd/csrc/hz_kernels.hip
make -s -C horizonator_amd/csrc HIPFLAGS_EXTRA=-DHZ_NFB=$n > $O/make$n.log 2>&1
timeout 300 $B > $O/b_nfb$n.json 2>> $O/err.log
timeout 300 $B > $O/b_nfb${n}b.json 2>> $O/err.log
done
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
