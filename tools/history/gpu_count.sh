#!/bin/bash
# VALU instructions of every kernel of a render (one --pmc pass), then the pipelined and serial timings
cd $GRAFT_REPO_ROOT
HZ_SERIAL=1 bash tools/pmc_groups.sh cnt "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" -- --no-host 2>/dev/null | grep -E "k_march|k_big" | cut -c1-200
bash tools/gpu_quick.sh 2>&1 | head -4
