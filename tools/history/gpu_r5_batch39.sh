#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b39; mkdir -p $O
for az in "-22.5,22.5" "-5,5"; do
HZ_WT_AZ="$az" HZ_WT_SAVE=$O/wave_zoom.npy timeout 300 python tools/wave_timing.py > $O/wave_timing_zoom.txt 2>&1; grep -v "^  File\|^    " $O/wave_timing_zoom.txt | head -24 | cut -c1-200
python tools/wave_schedule.py $O/wave_zoom.npy | head -6
done
