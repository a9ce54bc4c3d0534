#!/bin/bash
# bench three times (40 renders), serial kernel times once, then the GPU tests
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for k in 1 2 3; do timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined', round(d['ms_per_step'],3))"; done
HZ_SERIAL=1 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), {k:round(x,3) for k,x in d['roofline']['other_kernels_ms'].items()})"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
