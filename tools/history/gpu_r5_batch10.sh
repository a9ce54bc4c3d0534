#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b10; mkdir -p $O
gcc -O2 -fopenmp -o $O/hostfill tools/hostfill_bench.c horizonator_amd/csrc/hz_scatter.c -Ihorizonator_amd/csrc -lm 2> $O/hostfill_build.txt && timeout 150 $O/hostfill > $O/hostfill.txt 2>&1
grep "node -1" $O/hostfill.txt | head -12
PART=a bash tools/gpu_final_r5.sh
python3 -c "
import json
d=json.loads(open('gpurun_out/final5/bench_k20.json').read())
print('K20 ms', d['ms_per_step'], 'value', d['value'], 'parity', d.get('parity'))
print('same_viewpoint', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})
print('host', {k:v for k,v in d.get('host_inclusive',{}).items() if k not in ('what','two_in_flight')})
print('40km', d.get('zfar_40km',{}).get('ms_per_step'), 'roofline', {k:d['roofline'].get(k) for k in ('frac','kernel_ms','frac_whole_render','traffic')})
print('scenes', {k:round(v.get('ms_per_render',0),3) for k,v in d.get('scenes',{}).items() if isinstance(v,dict)})
print('cpu', d.get('cpu_baseline',{}).get('value'))
print('k50', json.loads(open('gpurun_out/final5/bench_k50.json').read())['ms_per_step'])"
grep "^cfg" gpurun_out/final5/host_inclusive.txt
