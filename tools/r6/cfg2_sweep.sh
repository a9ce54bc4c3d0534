#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
one() { env "$@" timeout 300 python tools/scenes.py --scenes cfg2,cfg1 --steps 16 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', ' '.join('%s %.4f' % (k, v['ms_per_render']) for k, v in d['scenes'].items() if 'ms_per_render' in v))"; }
for rep in 1 2; do
one HZ_X=0
one HZ_TWO_PASS=0
one HZ_HIZ=0
one HZ_HIZ=1
one HZ_NEAR_CELLS=32
one HZ_NEAR_CELLS=100
one HZ_PRETEST_MARCH=1
one HZ_ZONE_ROWS=16,8,2
one HZ_ZONE_ROWS=16,8,2 HZ_HIZ=0
done | tee $O/cfg2_sweep.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host --no-scenes > $GRAFT_REPO_ROOT/$O/bench_cfg2.json 2> $GRAFT_REPO_ROOT/$O/bench_cfg2.err
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find $O/kt -name "*_kernel_trace.csv" | head -1) > $O/timeline_cfg2.txt 2>&1
rm -rf $O/kt
cat $O/timeline_cfg2.txt
