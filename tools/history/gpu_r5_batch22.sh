#!/bin/bash
cd $GRAFT_REPO_ROOT
PART=a bash tools/gpu_final_r5.sh
python3 -c "
import json
d=json.loads(open('gpurun_out/final5/bench_k20.json').read())
print('K20 ms', d['ms_per_step'], 'value', d['value'], 'parity', {k:v for k,v in d.get('parity').items() if k!='what'})
print('same_viewpoint', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})
print('host', {k:v for k,v in d.get('host_inclusive',{}).items() if k not in ('what','two_in_flight')})
print('40km', d.get('zfar_40km',{}).get('ms_per_step'), 'roofline', {k:d['roofline'].get(k) for k in ('frac','kernel_ms','frac_whole_render','traffic')})
print('scenes', {k:round(v.get('ms_per_render',0),3) for k,v in d.get('scenes',{}).items() if isinstance(v,dict)})
print('cpu', d.get('cpu_baseline',{}).get('value'))
print('k50', json.loads(open('gpurun_out/final5/bench_k50.json').read())['ms_per_step'])"
grep "^cfg" gpurun_out/final5/host_inclusive.txt
