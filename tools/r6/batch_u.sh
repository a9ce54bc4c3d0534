#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6u/bench_k20.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("single", d.get("single_panorama_latency_ms"))
print("init", d["config"].get("init"))
h = d["host_inclusive"]; print({k: h[k] for k in ("ms", "first_call_ms", "moving_viewer_ms", "moving_viewer_used_vertex_cache", "ms_with_fresh_arrays_per_call", "ms_per_panorama_two_in_flight", "ms_all_calls", "equals_device_render", "two_in_flight_equals_device_render")})
print("zfar40", d["zfar_40km"]["ms_per_step"], "same_viewpoint", d["same_viewpoint"]["ms_per_step"])
print("valu", d["roofline"]["valu_issue"].get("whole_render"))
print({k: (round(v.get("ms_per_render", 0), 4), round(v.get("ps_per_triangle_vs_headline", 0), 2), round(v.get("init_s", 0), 3)) for k, v in d["scenes"].items()})
print("parity", d["parity"])
PY
