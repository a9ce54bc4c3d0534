#!/bin/bash
# the core of tools/gpu_final_r4.sh once more on the round's last build (the bench lines, both kernel traces, the timeline, both
# counter runs, the host path, the scenes): gpurun_out/final4b; tools/collect_profiles_r4.py --core turns it into profiles/
cd $GRAFT_REPO_ROOT
O=gpurun_out/final4b; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host --no-scenes > $O/bench_k20.json 2>> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra > $O/bench_k40.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_serial_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_pipelined_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find $O/kt_pipelined -name "*_kernel_trace.csv" | head -1) > $O/pipelined_timeline.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*_domain_stats.csv" -delete
HZ_SERIAL=1 bash tools/collect_pmc.sh r4_final > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r4_final.json $O/ 2>/dev/null
HZ_SERIAL=1 bash tools/pmc_groups.sh r4_mix "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" -- --no-host --no-scenes > $O/pmc_mix.txt 2>&1
cp gpurun_out/pmc_r4_mix.json $O/ 2>/dev/null
mkdir -p $O/scenes
timeout 900 python tools/scenes.py --counters > $O/scenes/default.json 2> $O/scenes/default.err
python tools/host_inclusive.py > $O/host_inclusive.txt 2>&1
ls $O; python3 -c "
import json
d=json.load(open('$O/bench.json')); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','host_inclusive')})[:600])
print('k20', json.load(open('$O/bench_k20.json'))['ms_per_step'], 'k40', json.load(open('$O/bench_k40.json'))['ms_per_step'])"
grep -E "k_march|k_big|k_resolve4|k_hiz" $(ls -S $O/kt_serial/*/*_kernel_stats.csv | head -1) | cut -c1-160
