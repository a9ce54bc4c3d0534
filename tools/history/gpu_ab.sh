#!/bin/bash
# A/B of library variants horizonator_amd/libhz_<name>.so (scratch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O; rm -f $O/*
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
V="$@"
for k in 1 2 3; do
  for v in $V; do
    export HORIZONATOR_AMD_LIB=$GRAFT_REPO_ROOT/horizonator_amd/libhz_$v.so
    timeout 300 $B > $O/b_${v}_$k.json 2>> $O/err.log
  done
done
cd /tmp; export TMPDIR=/tmp
for v in $V; do
  export HORIZONATOR_AMD_LIB=$GRAFT_REPO_ROOT/horizonator_amd/libhz_$v.so
  HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > /dev/null 2>> $GRAFT_REPO_ROOT/$O/err.log
done
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -delete
python3 - $V <<'PY'
import json,sys,glob,csv
for v in sys.argv[1:]:
    ms=[json.load(open(f))['ms_per_step'] for f in sorted(glob.glob('gpurun_out/ab/b_%s_*.json'%v))]
    print(v, 'pipelined ms/step', ' '.join('%.3f'%m for m in ms))
    f=sorted(glob.glob('gpurun_out/ab/kt_%s/*/*kernel_stats.csv'%v), key=lambda p: -len(open(p).read()))[0]
    for r in csv.DictReader(open(f)):
        n=r['Name'].split('(')[0].replace('void ','')
        if n.startswith('k_') and 'reset' not in n:
            print('    serial %-18s avg %8.1f min %8.1f max %8.1f' % (n[:18], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
