#!/bin/bash
# The GPU test suite once under every global mode switch (about one minute each on an MI355X):
# one stream; one / two rounds forced on every scene (two rounds on small scenes make the first
# round the longer one - that is how the conversion-waits-for-both-rounds bug of round 2 showed);
# first rounds of 8 and 300 cells; the unabridged transform; no clearing conversion; plain copies;
# round 3: no work lists, reads before the atomics forced on, odd segment lengths and a padded launch grid;
# round 4: results for host memory whole instead of without the sky (HZ_HOST_DENSE), the sparse path with one host thread,
# a middle round forced on the suite's (small) scenes; the 8-row segments of narrow sectors, and odd ones, everywhere;
# first rounds by screen tile whatever the view, and never (the default: zoomed views); the reach of zoomed views never / always
# long, and tried after every first draw (HZ_ADAPT).
# MODES="<env> ..." (one string per mode, separated by ';') runs a selection instead.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
rc=0
ALL=("HZ_SERIAL=1" "HZ_TWO_PASS=0" "HZ_TWO_PASS=1" "HZ_TWO_PASS=1 HZ_SERIAL=1" "HZ_TWO_PASS=1 HZ_NEAR_CELLS=8" \
           "HZ_TWO_PASS=1 HZ_NEAR_CELLS=300" "HZ_TWO_PASS=1 HZ_RESOLVE_CLEARS=0" "HZ_NO_FAST_MATH=1" "HZ_RESOLVE_CLEARS=0" \
           "HZ_ALWAYS_WAIT_NEAR=1 HZ_TWO_PASS=1" "HZ_PLAIN_COPY=1" "HZ_COPY_THREADS=1" "HZ_HOST_DENSE=1" \
           "HZ_NO_WORKLIST=1" "HZ_NO_WORKLIST=1 HZ_TWO_PASS=1" "HZ_TWO_PASS=1 HZ_PRETEST_MARCH=1" "HZ_TWO_PASS=1 HZ_PRETEST=1 HZ_NEAR_PX=3" \
           "HZ_TWO_PASS=1 HZ_FAR_ROWS=5 HZ_EXP_XCD_PAD=1" "HZ_TILES=1" "HZ_TILES=1 HZ_TWO_PASS=1 HZ_TILE_LIST=5" "HZ_HIZ=1 HZ_TWO_PASS=1" "HZ_HIZ=1 HZ_TWO_PASS=1 HZ_NEAR_CELLS=16 HZ_SERIAL=1" "HZ_HIZ=0" "HZ_MID=1 HZ_TWO_PASS=1 HZ_MID_NEAR=8 HZ_MID_CELLS=40" "HZ_MID=1 HZ_TWO_PASS=1 HZ_MID_NEAR=16 HZ_MID_CELLS=64 HZ_SERIAL=1" \
           "HZ_Z16_ROWS=8" "HZ_TWO_PASS=1 HZ_Z16_ROWS=5 HZ_FAR_ROWS=9" "HZ_TILES=2 HZ_TWO_PASS=1" "HZ_TILES=0" \
           "HZ_ADAPT=0" "HZ_ADAPT=2 HZ_TWO_PASS=1" "HZ_ADAPT_HI=0 HZ_TWO_PASS=1")
if [ -n "$MODES" ]; then IFS=";" read -ra ALL <<< "$MODES"; fi
for env in "${ALL[@]}"; do
  echo "== $env"
  env $env timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_multi.py 2>&1 | grep -E "passed|failed|error" | tail -2
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
