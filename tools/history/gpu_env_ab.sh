#!/bin/bash
# A/B/C... of environments on one box: tools/gpu_env_ab.sh "<env A>" "<env B>" ... ; three alternating rounds of bench.py
# (40 renders back to back) under each; BENCH_ARGS = further bench.py arguments
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-scenes $BENCH_ARGS"
for k in 1 2 3; do
  for e in "$@"; do
    env $e $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', 'ms/render', round(d['ms_per_step'],4), '40km', round(d.get('zfar_40km',{}).get('ms_per_step',0),4), 'parity', d['parity']['bgr_sha_is_llvmpipe'])"
  done
done
