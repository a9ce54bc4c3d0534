#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b17; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_series.py tests/test_gpu_api.py tests/test_gpu_cfg5.py tests/test_fullsize_checksums.py tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | tail -5
for k in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline > $O/bench_k20_$k.json 2> $O/bench.err
python3 -c "
import json
d=json.loads(open('$O/bench_k20_$k.json').read())
print('K20 ms', d['ms_per_step'], 'parity', {k:v for k,v in d['parity'].items() if k!='what'})
print('same_viewpoint', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})
print('40km', d.get('zfar_40km',{}).get('ms_per_step'), 'roofline', {k:d['roofline'].get(k) for k in ('frac','kernel_ms','frac_whole_render')})"
done
HZ_SERIAL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('serial: ms', d['ms_per_step'], 'k_march', d['roofline']['kernel_ms'], d['roofline'].get('other_kernels_ms'))"
timeout 900 python tools/hiz_ab.py cfg3_zoom45 cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom10 --steps 10 --set "HZ_VERTEX_CACHE=0" 2>&1 | python tools/hiz_ab_table.py | grep "|\|same_bytes"
