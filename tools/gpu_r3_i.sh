#!/bin/bash
cd $GRAFT_REPO_ROOT
HZ_TILES=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_fullsize_checksums.py tests/test_gpu_sequences.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for e in "X=1" "HZ_TILES=1" "X=1" "HZ_TILES=1"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg3,cfg3_zfar40km,cfg3_zoom45,cfg2,cfg1 --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v.get('ms_per_render',-1),4) for k,v in d['scenes'].items()})"; done
HZ_SERIAL=1 HZ_TILES=1 python tools/scene_times.py cfg3 cfg3_zoom45 2>&1 | grep -v amdgpu.ids
HZ_SERIAL=1 python tools/scene_times.py cfg3 cfg3_zoom45 2>&1 | grep -v amdgpu.ids
