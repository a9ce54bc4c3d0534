#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra"
for e in "X=1" "BENCH_NO_KERNEL_EVENTS=1" "X=1" "BENCH_NO_KERNEL_EVENTS=1"; do
 echo "$e: $(env $e $B 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find gpurun_out/kt_tmp -name "*_kernel_trace.csv" | head -1) | grep "k_march<f grid 8576"
rm -rf gpurun_out/kt_tmp
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_sequences.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
