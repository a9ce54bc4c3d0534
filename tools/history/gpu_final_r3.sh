#!/bin/bash
# the measurements DESIGN.md and profiles/ quote for round 3: run on the GPU box, results under gpurun_out/final3
cd $GRAFT_REPO_ROOT
O=gpurun_out/final3; [ -z "$FINAL_R3_SHORT" ] && rm -rf $O; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host --no-extra > $O/bench_k40.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_serial_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_pipelined_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find $O/kt_pipelined -name "*_kernel_trace.csv" | head -1) > $O/pipelined_timeline.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*_domain_stats.csv" -delete
HZ_SERIAL=1 bash tools/collect_pmc.sh r3_final > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r3_final.json $O/ 2>/dev/null
HZ_SERIAL=1 bash tools/pmc_groups.sh r3_mix "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" -- --no-host > $O/pmc_mix.txt 2>&1
cp gpurun_out/pmc_r3_mix.json $O/ 2>/dev/null
bash tools/gpu_scenes.sh > $O/scenes_summary.txt 2>&1
cp -r gpurun_out/scenes $O/
[ -n "$FINAL_R3_SHORT" ] && exit 0      # (FINAL_R3_SHORT=1: the bench lines, traces, counters and scenes only)
python tools/host_inclusive.py > $O/host_inclusive.txt 2>&1
python tools/sector_timing.py > $O/sector_timing.txt 2>&1
python tools/sector_b2b.py > $O/sector_b2b.txt 2>&1
HZ_G=2 python tools/sector_b2b.py >> $O/sector_b2b.txt 2>&1
timeout 900 python tools/hiz_ab.py cfg3_zoom10 cfg3_zoom45 cfg3_zoom90 cfg3_zoom180 cfg3 cfg3_zfar40km cfg5 --steps 8 --set "HZ_HIZ=0" --set "HZ_HIZ=1" --set "" > $O/coarse_depth.txt 2> $O/coarse_depth.err
timeout 900 python tools/hiz_ab.py cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45 cfg3_zoom10 --steps 10 --set "HZ_NEAR_CELLS=256" --set "HZ_NEAR_CELLS=384" --set "HZ_NEAR_CELLS=512" --set "HZ_HIZ=0" > $O/coarse_depth_reach.txt 2> $O/coarse_depth_reach.err
timeout 900 python tools/hiz_ab.py cfg3 cfg3_rough cfg3_summit cfg3_valley cfg2 --steps 30 --set "HZ_HIZ=0" --set "" --set "HZ_HIZ=0" --set "" > $O/coarse_depth_series.txt 2> $O/coarse_depth_series.err
timeout 2400 python tools/experiments.py > $O/r3_experiments.json 2> $O/r3_experiments.err
for g in rotate root0; do timeout 600 python bench.py --gpus 4 --backend gloo --same-gpu --steps 8 --warmup 2 --no-cpu-baseline --no-host --no-extra --gather $g 2>>$O/multi.err | grep "^{" > $O/multi_4ranks_one_gpu_$g.json; done
timeout 300 python bench.py --gpus 1 --exchange-anyway --steps 20 --warmup 4 --no-cpu-baseline --no-host --no-extra 2>>$O/multi.err | grep "^{" > $O/exchange_anyway.json
ls $O; grep -E "^G=|fixed" $O/sector_timing.txt | cut -c1-330; cat $O/sector_b2b.txt | grep "G="; cat $O/host_inclusive.txt | grep cfg; python3 -c "
import json
d=json.load(open('$O/bench.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step','host_inclusive','zfar_40km')})[:900])
print(json.load(open('$O/bench_k40.json'))['ms_per_step'])"
