#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for z in 4 2 1; do
  a=$(HZ_EXP_Z4=$z HZ_SERIAL=1 python bench.py --zfar 40000 --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))")
  b=$(HZ_EXP_Z4=$z python bench.py --zfar 40000 --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['parity']['bgr_sha_is_llvmpipe'])")
  c=$(HZ_EXP_Z4=$z python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['parity']['bgr_sha_is_llvmpipe'])")
  echo "4-row zones with $z rows: 40 km: serial render, its k_march $a ms; render of a series $b; 600 km: render of a series $c"
done; done
