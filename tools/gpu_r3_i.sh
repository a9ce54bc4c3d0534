#!/bin/bash
cd $GRAFT_REPO_ROOT
for env in "HZ_TILES=1" "HZ_TILES=1 HZ_TWO_PASS=1 HZ_TILE_LIST=5"; do
  echo "== $env"
  env $env timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_multi.py 2>&1 | grep -E "passed|failed|error" | tail -2
done
bash tools/gpu_tests.sh
