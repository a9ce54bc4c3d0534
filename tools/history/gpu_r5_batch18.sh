#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_series.py -x -q -m gpu 2>&1 | tail -2
for k in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline > $O/bench_k20_$k.json 2> $O/bench.err
python3 -c "
import json
d=json.loads(open('$O/bench_k20_$k.json').read())
print('K20 ms', d['ms_per_step'], 'same_viewpoint', d['same_viewpoint']['ms_per_step'], '40km', d.get('zfar_40km',{}).get('ms_per_step'), 'kernel_ms', d['roofline']['kernel_ms'], d['parity']['bgr_sha_is_llvmpipe'])"
done
HZ_SERIAL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('serial: ms', d['ms_per_step'], 'k_march', d['roofline']['kernel_ms'], d['roofline'].get('other_kernels_ms'))"

