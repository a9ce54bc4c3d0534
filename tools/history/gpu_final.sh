#!/bin/bash
# the measurements DESIGN.md and profiles/ quote: run on the GPU box, results under gpurun_out/final
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O   # (gpurun merges into the local gpurun_out/: remove the local copy first as well)
timeout 300 ./tools/build/valu_issue > $O/valu_issue.json 2> $O/valu_issue.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
HZ_TWO_PASS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host > $O/bench_one_round.json 2>> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host --config cfg2 > $O/bench_cfg2.json 2>> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host --config cfg1 > $O/bench_cfg1.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_serial_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_pipelined_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
HZ_SERIAL=1 HZ_TWO_PASS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial_one_round -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_serial_one_round_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*_domain_stats.csv" -delete
HZ_SERIAL=1 bash tools/collect_pmc.sh r2_final > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r2_final.json $O/ 2>/dev/null
HZ_SERIAL=1 bash tools/pmc_groups.sh r2_mix "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" -- --no-host > $O/pmc_mix.txt 2>&1
cp gpurun_out/pmc_r2_mix.json $O/ 2>/dev/null
python tools/host_inclusive.py > $O/host_inclusive.txt 2>&1
HZ_TWO_PASS=1 python tools/wave_timing.py > $O/wave_timing_two.txt 2>&1
python tools/cfg5_check.py > $O/cfg5.txt 2>&1
python tools/cfg4_batch.py --repeat 2 > $O/cfg4.txt 2>&1
python tools/cfg4_batch.py --repeat 2 --zfar 40000 --ranges > $O/cfg4_40km.txt 2>&1
python tools/sector_timing.py > $O/sector_timing.txt 2>&1
ls $O; tail -2 $O/cfg5.txt $O/cfg4.txt $O/sector_timing.txt 2>/dev/null | cut -c1-400; python3 -c "
import json
d=json.load(open('$O/bench.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','host_inclusive','zfar_40km')})[:1500])"
