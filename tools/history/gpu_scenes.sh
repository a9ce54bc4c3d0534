#!/bin/bash
# the scenes of tools/scenes.py under the library's defaults (with the waves' counters) and under forced
# settings of its two heuristics - the first round's reach (HZ_NEAR_PX: cells wider than that many pixels)
# and one / two rounds - one JSON line each into gpurun_out/scenes/
cd $GRAFT_REPO_ROOT
O=gpurun_out/scenes; rm -rf $O; mkdir -p $O
timeout 900 python tools/scenes.py --counters > $O/default.json 2> $O/default.err
for e in "HZ_NEAR_PX=10" "HZ_NEAR_PX=40" "HZ_TWO_PASS=0" "HZ_TWO_PASS=1"; do
  env $e timeout 900 python tools/scenes.py > $O/$e.json 2> $O/$e.err
done
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/scenes/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(os.path.basename(f), {k: (round(v["ms_per_render"], 3) if "ms_per_render" in v else v.get("error")) for k, v in d["scenes"].items()})
PY
