#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "X=1" "HZ_PRETEST_MARCH=0"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg3,cfg3_rough,cfg3_summit,cfg3_zoom45,cfg3_zfar40km,cfg2,cfg1,cfg4_32,cfg5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v.get('ms_per_render',-1),4) for k,v in d['scenes'].items()})"; done
echo "== sectors"; python tools/sector_b2b.py 2>/dev/null | tail -1; HZ_PRETEST_MARCH=0 python tools/sector_b2b.py 2>/dev/null | tail -1
