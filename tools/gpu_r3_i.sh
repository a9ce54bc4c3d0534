#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --config cfg5 --gpus 2 --backend gloo --same-gpu --steps 3 --warmup 1 --no-extra 2>gpurun_out/cfg5_multi.err | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5, 2 gloo ranks on one GPU:', round(d['ms_per_step'],2), d['gathered_panorama_equals_single_gpu_render'], d['config']['sector_widths'], d['config']['workload'][:80])"
grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*\|oracle:\|caller_stubs" gpurun_out/cfg5_multi.err | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_multi.py tests/test_gpu_rccl.py -x -q -k "strip or sparse or rank or rccl or sector" 2>&1 | grep -E "passed|failed" | tail -2
