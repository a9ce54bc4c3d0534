#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_pmc.sh <tag> [bench args]
# PMC passes only (no trace domains with --pmc), one counter group per run.
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT
cd /tmp; export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
         "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-host "$@" >/dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
read SEGFRAC READS TERRAIN < <(python3 tools/touched_segments.py 2>/dev/null | tail -1)
python3 tools/collect_pmc.py $OUT gpurun_out/pmc_$TAG.json cfg3 600000 16000 4000 $READS $TERRAIN
