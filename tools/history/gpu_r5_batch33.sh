#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for fr in 64 32 16; do
  a=$(HZ_EXP_FAR_ROWS=$fr HZ_SERIAL=1 python bench.py --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],4))")
  b=$(HZ_EXP_FAR_ROWS=$fr python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['parity']['bgr_sha_is_llvmpipe'])")
  echo "far rows $fr: k_march alone $a ms, render of a series $b"
done; done
