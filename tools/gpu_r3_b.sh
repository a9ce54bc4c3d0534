#!/bin/bash
# round 3: sector timing, where a sector's time goes (serial kernel trace), bench + the multi-rank loops on one GPU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3; grep -A40 "^___" $O/pytest.txt | head -80
timeout 600 python tools/sector_timing.py > $O/sector_timing.txt 2>&1
grep -E "^G=|fixed" $O/sector_timing.txt
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for k in 1 2 3; do timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined', round(d['ms_per_step'],3))"; done
HZ_SERIAL=1 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), {k:round(x,3) for k,x in d['roofline']['other_kernels_ms'].items()})"
for g in rotate root0; do timeout 600 python bench.py --gpus 4 --backend gloo --same-gpu --steps 8 --warmup 2 --no-cpu-baseline --no-host --no-extra --gather $g 2>$O/multi_$g.err | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('4 gloo ranks on one GPU, $g', round(d['ms_per_step'],3), d['gathered_panorama_equals_single_gpu_render'], d['config']['sector_widths'])"; done
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_sector -- python3 $GRAFT_REPO_ROOT/tools/sector_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*_domain_stats.csv" -delete
cat $(find $O/kt_sector -name "*kernel_stats.csv" | head -1) | cut -c1-200
