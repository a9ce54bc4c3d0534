#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 300 ./tools/build/valu_issue > $O/valu_issue.json 2> $O/valu_issue.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 $B > $O/b_serial_one.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 $B > $O/b_serial_two.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B > $O/b_two.json 2>> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B > $O/b_one.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=0 bash tools/pmc_groups.sh r2d_one "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" > $O/pmc_one.txt 2>&1
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms']))
except Exception as e: print(' failed', e)
"; done
grep k_march $O/pmc_one.txt
