#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b13; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/exact_seq.hip -o $O/exact_seq 2> $O/build.txt && timeout 600 $O/exact_seq 40 > $O/exact_seq.txt 2>&1
head -16 $O/exact_seq.txt; tail -1 $O/exact_seq.txt
timeout 900 python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_exactness.py tests/test_gpu_parity.py tests/test_gpu_series.py tests/test_gpu_api.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline > $O/bench_k20.json 2> $O/bench.err
python3 -c "
import json
d=json.loads(open('$O/bench_k20.json').read())
print('K20 ms', d['ms_per_step'], 'parity', {k:v for k,v in d['parity'].items() if k!='what'})
print('same_viewpoint', {k:v for k,v in d.get('same_viewpoint',{}).items() if k!='what'})
print('40km', d.get('zfar_40km',{}).get('ms_per_step'), 'roofline', {k:d['roofline'].get(k) for k in ('frac','kernel_ms','frac_whole_render')})"
HZ_SERIAL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('serial: ms', d['ms_per_step'], 'k_march', d['roofline']['kernel_ms'])"
HZ_SERIAL=1 bash tools/pmc_groups.sh r5_mix2 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" -- --no-host --no-scenes > $O/pmc_mix.txt 2>&1
grep "k_march" $O/pmc_mix.txt | cut -c1-400
