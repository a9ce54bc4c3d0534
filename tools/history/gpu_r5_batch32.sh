#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b32; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; grep -E " passed| failed| error" $O/pytest_full.txt | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('$O/bench_k20.json').read())
print('K20 ms', d['ms_per_step'], 'value', d['value'], 'parity', {k:v for k,v in d.get('parity').items() if k!='what'})
print('same_viewpoint', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})
print('host', {k:v for k,v in d.get('host_inclusive',{}).items() if k in ('ms','ms_with_fresh_arrays_per_call','ms_per_panorama_two_in_flight','equals_device_render')})
print('valu', d['roofline'].get('valu_issue',{}).get('wave_instructions'))"
timeout 600 python bench.py > $O/bench_default.json 2>> $O/bench.err; echo "default bench rc $?"; python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read()); print('default: steps', d['steps'], 'ms', d['ms_per_step'])"
