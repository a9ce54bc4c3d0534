#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_cfg5.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host"
HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 $B --no-extra > $O/b_serial_two.json 2>> $O/err.log
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 $B --no-extra > $O/b_serial_one.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B > $O/b_two.json 2>> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B > $O/b_one.json 2>> $O/err.log
HZ_TWO_PASS=1 timeout 300 $B --config cfg2 > $O/b_cfg2_two.json 2>> $O/err.log
HZ_TWO_PASS=0 timeout 300 $B --config cfg2 > $O/b_cfg2_one.json 2>> $O/err.log
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 HZ_TWO_PASS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial_two -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_serial_two.json 2>> $GRAFT_REPO_ROOT/$O/err.log
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -size +20M -delete
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  kern %.3f  other %s 40km %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms'], d.get('zfar_40km',{}).get('ms_per_step')))
except Exception as e: print(' failed', e)
"; done
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r2l/kt_serial_two/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("  %-40s calls %5s avg %10.1f us  min %9.1f max %9.1f  %5.1f%%"%(r['Name'][:40],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,float(r['Percentage'])))
PY
