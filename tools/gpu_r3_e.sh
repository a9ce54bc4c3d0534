#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "X=0" "HZ_PRETEST=1" "HZ_NEAR_CELLS=512" "HZ_NEAR_CELLS=1000" "HZ_NEAR_CELLS=1000 HZ_PRETEST=1"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg3,cfg3_zoom45,cfg2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v.get('ms_per_render',-1),4) for k,v in d['scenes'].items()})"; done
HZ_SERIAL=1 HZ_PRETEST=1 python tools/scene_times.py cfg3_zoom45 2>&1 | grep -v amdgpu.ids
HZ_SERIAL=1 HZ_NEAR_CELLS=1000 python tools/scene_times.py cfg3_zoom45 2>&1 | grep -v amdgpu.ids
