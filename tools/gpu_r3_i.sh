#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "X=1" "HZ_PRETEST=2" "X=1" "HZ_PRETEST=2"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg3,cfg3_rough,cfg3_summit,cfg3_zoom45,cfg5 --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v.get('ms_per_render',-1),4) for k,v in d['scenes'].items()})"; done
