#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_checksums.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for e in "X=1" "HZ_ZBOX=0" "X=1" "HZ_ZBOX=0"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg3,cfg3_rough,cfg3_summit,cfg3_zoom45,cfg3_zfar40km,cfg2,cfg4_32,cfg5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v.get('ms_per_render',-1),4) for k,v in d['scenes'].items()})"; done
echo "== sectors"; python tools/sector_b2b.py 2>/dev/null | tail -1; HZ_ZBOX=0 python tools/sector_b2b.py 2>/dev/null | tail -1
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for e in "X=1" "HZ_ZBOX=0" "X=1" "HZ_ZBOX=0"; do
 echo "$e: $(env $e $B 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
done
