#!/bin/bash
cd $GRAFT_REPO_ROOT
HZ_SERIAL=1 python tools/scene_times.py cfg3 cfg3_zoom45 cfg2 2>&1 | grep -v amdgpu.ids
HZ_SERIAL=1 HZ_TWO_PASS=1 python tools/scene_times.py cfg2 2>&1 | grep -v amdgpu.ids
for e in "X=0" "HZ_TWO_PASS=0" "HZ_TWO_PASS=1"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg1,mid_4000,cfg2,cfg3_zfar40km 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v['ms_per_render'],4) for k,v in d['scenes'].items()})"; done
