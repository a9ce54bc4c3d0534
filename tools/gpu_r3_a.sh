#!/bin/bash
# round 3, first look: GPU tests, sector timing with and without work lists, pipelined bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt
tail -3 $O/pytest.txt
timeout 600 python tools/sector_timing.py > $O/sector_timing.txt 2>&1
HZ_NO_WORKLIST=1 timeout 600 python tools/sector_timing.py > $O/sector_timing_nolist.txt 2>&1
grep "^G=" $O/sector_timing.txt; echo; grep "^G=" $O/sector_timing_nolist.txt
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for k in 1 2 3; do timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined', round(d['ms_per_step'],3))"; done
HZ_SERIAL=1 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), {k:round(x,3) for k,x in d['roofline']['other_kernels_ms'].items()})"
