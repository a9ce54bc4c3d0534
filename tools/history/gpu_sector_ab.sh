#!/bin/bash
# a sector's strips back to back (tools/sector_b2b.py) under each of the given environments, three times round: tools/gpu_sector_ab.sh "<env A>" "<env B>" ...
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for e in "$@"; do
    echo "[$e] $(env $e python3 tools/sector_b2b.py 2>&1 | grep 'G=')"
  done
done
