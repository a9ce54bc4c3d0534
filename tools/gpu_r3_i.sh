#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "X=1" "HZ_TILES=1" "HZ_TILES=1 HZ_TILE_LIST=512" "HZ_TILES=1 HZ_TILE_LIST=1024"; do echo "== $e"; env $e python tools/scenes.py --scenes cfg3,cfg3_zfar40km,cfg3_zoom45,cfg2 --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v.get('ms_per_render',-1),4) for k,v in d['scenes'].items()})"; done
HZ_SERIAL=1 HZ_TILES=1 python tools/scene_times.py cfg3 cfg3_zoom45 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp
HZ_TILES=1 HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_tiles -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-host > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/kt_tiles/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    print(r["Name"].split("(")[0][:30], r["Calls"], "avg %.1f us min %.1f max %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf gpurun_out/kt_tiles
