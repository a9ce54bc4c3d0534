#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2u; mkdir -p $O; rm -f $O/*
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for k in 1 2 3 4; do
  for v in base new; do
    if [ $v = new ]; then unset HORIZONATOR_AMD_LIB; else export HORIZONATOR_AMD_LIB=$GRAFT_REPO_ROOT/horizonator_amd/libhz_$v.so; fi
    timeout 300 $B > $O/b_${v}_$k.json 2>> $O/err.log
  done
done
unset HORIZONATOR_AMD_LIB
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], {k:round(v,3) for k,v in d['roofline']['other_kernels_ms'].items()}))
except Exception as e: print(' failed', e)
"; done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-200
grep -E "passed|failed" $O/pytest.txt | tail -2
