#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "X=0" "HZ_FAR_ROWS=32" "HZ_FAR_ROWS=64" "HZ_FAR_ROWS=8" "HZ_NEAR_CELLS=64" "HZ_NEAR_CELLS=192" "HZ_TWO_PASS=0" "HZ_NO_WORKLIST=1"; do
  echo "== $e: $(env $e timeout 300 python tools/sector_b2b.py 2>/dev/null | tail -1)"
done
