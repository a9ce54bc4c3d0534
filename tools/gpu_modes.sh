#!/bin/bash
# The GPU test suite once under each mode that runs DIFFERENT code from the default (about 1.5 minutes each on an MI355X;
# hz_options_t in include/hz_hip.h says what each switch does):
#   one stream; one round / two rounds forced on every scene (two rounds on small scenes make the first round the longer
#   one - that is how the conversion-waits-for-both-rounds bug of round 2 showed), with a short and a long reach;
#   the unabridged transform; no clearing conversion; dense host results; no work lists; reads before the atomics and coarse
#   depth forced on; first rounds by screen tile with short tile lists (the fall-back to k_big); the reach of zoomed views
#   tried after every first draw; host results in 3 sectors whatever the image, and in 1; round 6: the DEM decoded on the host,
#   no warm-up draw in horizonator_init, the pool's threads wherever the scheduler puts them.
# Round 4 ran 32 combinations (profiles/r4_modes.txt); the options that went in round 5 took 20 of them along.
# MODES="<env> ..." (one string per mode, separated by ';') runs a selection instead.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
rc=0
ALL=("HZ_SERIAL=1 HZ_TWO_PASS=1" "HZ_TWO_PASS=0" "HZ_TWO_PASS=1 HZ_NEAR_CELLS=8" "HZ_TWO_PASS=1 HZ_NEAR_CELLS=300 HZ_RESOLVE_CLEARS=0" \
     "HZ_NO_FAST_MATH=1" "HZ_HOST_DENSE=1" "HZ_NO_WORKLIST=1 HZ_TWO_PASS=1" "HZ_TWO_PASS=1 HZ_PRETEST_MARCH=1 HZ_HIZ=1" \
     "HZ_TILES=1 HZ_TWO_PASS=1 HZ_TILE_LIST=5" "HZ_HIZ=0 HZ_TILES=0 HZ_ADAPT=0" "HZ_ADAPT_HI=0 HZ_TWO_PASS=1" "HZ_HOST_SECTORS=3 HZ_COPY_THREADS=2" "HZ_HOST_SECTORS=1" \
     "HORIZONATOR_INGEST=host HORIZONATOR_NO_WARMUP=1 HZ_COPY_NODE=any")
if [ -n "$MODES" ]; then IFS=";" read -ra ALL <<< "$MODES"; fi
for env in "${ALL[@]}"; do
  echo "== $env"
  env $env timeout 900 python -m pytest tests -x -q -rf -m gpu --deselect tests/test_gpu_bench_multi.py 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|error" | tail -4
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
