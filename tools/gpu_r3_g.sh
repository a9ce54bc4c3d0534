#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
bash tools/gpu_tests.sh
python tools/host_inclusive.py 2>&1 | grep -v amdgpu.ids
HZ_COPY_THREADS=24 python tools/host_inclusive.py 2>&1 | grep -v amdgpu.ids
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["host_inclusive"], d["cpu_baseline"], d.get("zfar_40km"))
print({k:(round(v.get("ms_per_render",-1),3), round(v.get("ps_per_triangle_vs_headline",-1),2)) for k,v in d["scenes"].items()})
print(d["roofline"])
PY
