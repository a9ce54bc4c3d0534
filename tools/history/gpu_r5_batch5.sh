#!/bin/bash
# round 5, fifth batch: grouped copies on high-priority streams; serial kernel times cold and from the vertex cache
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hostpath.py -x -q -m gpu > $O/pytest_quick.txt 2>&1
tail -2 $O/pytest_quick.txt; grep -n -B5 -A30 "^___" $O/pytest_quick.txt | head -60
HZ_HOST_TIMES=1 timeout 300 python tools/host_inclusive.py cfg3 sectors=1,2,4,8 > $O/host.txt 2>&1
grep "^cfg3:" $O/host.txt
for n in 1 2 4 8; do grep " $n sector" $O/host.txt | sed -n '5,6p' | cut -c60-440; done
HZ_SERIAL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host --no-scenes > $O/bench_serial.json 2> $O/bench_serial.err
python3 -c "
import json; d=json.loads(open('$O/bench_serial.json').read()); r=d['roofline']
print('serial cold ms', d['ms_per_step'], 'k_march', r['kernel_ms'], r['other_kernels_ms']); print('cached', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host --no-scenes > $O/bench_k20.json 2> $O/bench_k20.err
python3 -c "
import json; d=json.loads(open('$O/bench_k20.json').read()); r=d['roofline']
print('pipelined cold ms', d['ms_per_step'], 'k_march', r['kernel_ms']); print('cached', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})"
