#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for z in 16 8 32; do
  a=$(HZ_EXP_Z16=$z HZ_SERIAL=1 python bench.py --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],4))")
  b=$(HZ_EXP_Z16=$z python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['parity']['bgr_sha_is_llvmpipe'])")
  echo "middle zones $z rows: k_march alone $a ms, render of a series $b"
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_series.py tests/test_gpu_cfg5.py -x -q -m gpu 2>&1 | tail -1
timeout 900 python tools/hiz_ab.py cfg3_zoom45 cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom10 --steps 10 --set "HZ_VERTEX_CACHE=0" 2>&1 | python tools/hiz_ab_table.py | grep "|" | cut -c1-70
