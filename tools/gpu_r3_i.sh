#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_tests.sh
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real
python3 -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['steps'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['host_inclusive']['ms'], d['host_inclusive']['ms_with_fresh_arrays_per_call'], {k:round(v.get('ms_per_render',-1),3) for k,v in d['scenes'].items()})"
