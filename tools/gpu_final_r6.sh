#!/bin/bash
# the measurements DESIGN.md and profiles/ quote for round 6: run on the GPU box, results under gpurun_out/final6
# (tools/collect_profiles_r6.py turns them into the files under profiles/).  PART=a|b|c picks a third (a GPU lease is short).
cd $GRAFT_REPO_ROOT
O=gpurun_out/final6; mkdir -p $O
PART=${PART:-abc}
if [[ $PART == *a* ]]; then
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
timeout 300 python bench.py --steps 50 --warmup 6 --no-cpu-baseline --no-host --no-scenes > $O/bench_k50.json 2>> $O/bench.err
HZ_VERTEX_CACHE=0 HZ_HOST_TIMES=1 timeout 400 python tools/host_inclusive.py cfg3 > $O/host_inclusive.txt 2>&1
HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg3 sectors=1,2,3,4 >> $O/host_inclusive.txt 2>&1
HZ_VERTEX_CACHE=0 timeout 300 python tools/host_inclusive.py cfg2 >> $O/host_inclusive.txt 2>&1
HZ_VERTEX_CACHE=0 HZ_HOST_DENSE=1 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "^cfg3" >> $O/host_inclusive.txt
HZ_VERTEX_CACHE=0 timeout 300 python tools/first_call.py 2>&1 | grep "^buffers" > $O/first_call.txt
timeout 900 python tools/init_times.py cfg3 cfg5 > $O/init_times.txt 2>&1
HORIZONATOR_INGEST=host timeout 900 python tools/init_times.py cfg3 cfg5 > $O/init_times_host_ingest.txt 2>&1
timeout 300 tools/build/pcie_beside > $O/pcie_beside.txt 2>&1
fi
if [[ $PART == *b* ]]; then
cd /tmp; export TMPDIR=/tmp
HZ_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_serial_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-host > $GRAFT_REPO_ROOT/$O/kt_pipelined_bench.json 2>> $GRAFT_REPO_ROOT/$O/bench.err
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_host -- python3 $GRAFT_REPO_ROOT/tools/host_trace_run.py > $GRAFT_REPO_ROOT/$O/kt_host.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py $(find $O/kt_pipelined -name "*_kernel_trace.csv" | head -1) > $O/pipelined_timeline.txt 2>&1
python3 tools/trace_tail.py $O/kt_host 22 > $O/host_call_timeline.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_memory_copy_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*_domain_stats.csv" -delete
# (counters of the RENDERS only: no throw-away draw in horizonator_init, no k_ingest)
HORIZONATOR_NO_WARMUP=1 HORIZONATOR_INGEST=host HZ_SERIAL=1 bash tools/collect_pmc.sh r6_final > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r6_final.json $O/ 2>/dev/null
HORIZONATOR_NO_WARMUP=1 HORIZONATOR_INGEST=host HZ_SERIAL=1 bash tools/pmc_groups.sh r6_mix "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" -- --no-host --no-scenes > $O/pmc_mix.txt 2>&1
cp gpurun_out/pmc_r6_mix.json $O/ 2>/dev/null
fi
if [[ $PART == *c* ]]; then
timeout 900 python tools/hiz_ab.py cfg3_zoom45 cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom10 --steps 10 --set "HZ_VERTEX_CACHE=0" --set "HZ_VERTEX_CACHE=1" 2>&1 | python tools/hiz_ab_table.py | grep "|\|same_bytes" > $O/zoomed.txt
timeout 600 python bench.py --gpus 4 --backend gloo --same-gpu --steps 8 --warmup 2 --no-cpu-baseline --no-host --no-extra --loop c 2>>$O/multi.err | grep "^{" > $O/multi_4ranks_one_gpu_c_loop.json
BENCH_HOST_TIMES=1 timeout 300 python bench.py --gpus 1 --exchange-anyway --steps 20 --warmup 4 --no-cpu-baseline --no-host --no-extra 2>>$O/multi.err | grep "^{" > $O/exchange_anyway.json
python tools/sector_timing.py > $O/sector_timing.txt 2>&1
bash tools/gpu_scene_kernels.sh cfg3_zoom45_summit > $O/summit_kernels.txt 2>&1
fi
ls $O
