#!/bin/bash
cd $GRAFT_REPO_ROOT
PART=b bash tools/gpu_final_r5.sh
tail -5 gpurun_out/final5/pmc_traffic.txt; tail -5 gpurun_out/final5/pmc_mix.txt
head -30 gpurun_out/final5/pipelined_timeline.txt
head -40 gpurun_out/final5/host_call_timeline.txt
