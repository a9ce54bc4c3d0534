#!/bin/bash
# first GPU trip of round 2: issue-rate table, parity tests, one- vs two-round draws, serial profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 300 ./tools/build/valu_issue > $O/valu_issue.json 2> $O/valu_issue.err
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
HZ_TWO_PASS=0 HZ_RESOLVE_CLEARS=0 timeout 300 $B > $O/b_one_memset.json 2> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B > $O/b_one.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B > $O/b_two.json 2>> $O/err.log
HZ_TWO_PASS=1 HZ_NEAR_CELLS=64 timeout 300 $B --no-extra > $O/b_two_n64.json 2>> $O/err.log
HZ_TWO_PASS=1 HZ_NEAR_CELLS=256 timeout 300 $B --no-extra > $O/b_two_n256.json 2>> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B --config cfg2 > $O/b_cfg2_one.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B --config cfg2 > $O/b_cfg2_two.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 $B --no-extra > $O/b_serial_one.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 $B --no-extra > $O/b_serial_two.json 2>> $O/err.log
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial_one -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/kt_serial_one.json 2>> $GRAFT_REPO_ROOT/$O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial_two -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/kt_serial_two.json 2>> $GRAFT_REPO_ROOT/$O/err.log
HZ_TWO_PASS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_pipe_two -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/kt_pipe_two.json 2>> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -size +20M -delete
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s  40km %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms'], d.get('zfar_40km',{}).get('ms_per_step')))
except Exception as e: print(' failed', e)
"; done
