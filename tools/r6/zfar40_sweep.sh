#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6v; mkdir -p $O
one() { env "$@" timeout 300 python tools/scenes.py --scenes cfg3_zfar40km --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', ' '.join('%s %.4f' % (k, v['ms_per_render']) for k, v in d['scenes'].items() if 'ms_per_render' in v))"; }
for rep in 1 2; do
one HZ_X=0
one HZ_TWO_PASS=0
one HZ_HIZ=0
one HZ_NEAR_CELLS=48
one HZ_NEAR_CELLS=80
one HZ_NEAR_CELLS=200
one HZ_NEAR_CELLS=300
one HZ_NEAR_CELLS=80 HZ_HIZ=0
one HZ_ZONE_ROWS=16,8,2
one HZ_ZONE_ROWS=8,4,2
done | tee $O/zfar40_sweep.txt
