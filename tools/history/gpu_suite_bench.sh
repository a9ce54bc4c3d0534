#!/bin/bash
# the GPU suite, then the driver's bench command; logs under gpurun_out/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_full.txt | tail -3
grep -n -B5 -A40 "^___" gpurun_out/pytest_full.txt | head -120
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_k20.json 2> gpurun_out/bench_k20.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_k20.json').read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'host', {k:v for k,v in d.items() if 'host' in k})
print('roofline', d['roofline'].get('frac'), d['roofline'].get('kernel_ms'))
PY
