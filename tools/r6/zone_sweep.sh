#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6r; mkdir -p $O
for rep in 1 2; do
for z in "default" "32,16,4" "16,8,4" "16,8,2" "8,8,2" "8,4,2" "4,4,2" "64,16,4"; do
  if [ "$z" = default ]; then unset HZ_ZONE_ROWS; else export HZ_ZONE_ROWS=$z; fi
  timeout 300 python tools/scenes.py --scenes cfg1,cfg2,mid_4000 --steps 16 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$z', ' '.join('%s %.4f' % (k, v['ms_per_render']) for k, v in d['scenes'].items() if 'ms_per_render' in v))"
done; done | tee $O/zone_sweep.txt
