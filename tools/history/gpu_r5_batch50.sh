#!/bin/bash
cd $GRAFT_REPO_ROOT
HZ_EXP_SHARDS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_series.py tests/test_gpu_cfg5.py tests/test_fullsize_checksums.py tests/test_sharding.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for e in "" "HZ_EXP_SHARDS=1" "HZ_EXP_SHARDS=1 HZ_EXP_SECTOR_INLINE=32" "HZ_EXP_SHARDS=1 HZ_EXP_SECTOR_INLINE=16" "HZ_EXP_SECTOR_INLINE=16"; do
  echo "[$e] $(env $e python tools/sector_b2b.py 2>/dev/null | tail -1)"
done; done
for rep in 1 2; do
for e in "" "HZ_EXP_SHARDS=1"; do
  b=$(env $e python bench.py --zfar 40000 --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['parity']['bgr_sha_is_llvmpipe'])")
  echo "[$e] 40 km, render of a series: $b"
done; done
