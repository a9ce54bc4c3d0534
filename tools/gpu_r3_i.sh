#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for e in "X=1" "HZ_RESOLVE_CONST=0" "X=1" "HZ_RESOLVE_CONST=0" "X=1" "HZ_RESOLVE_CONST=0"; do
 echo "$e $(env $e $B 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
done
