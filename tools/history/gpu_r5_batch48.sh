#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b48; mkdir -p $O
HZ_WT_ZFAR=40000 timeout 300 python tools/wave_timing.py > $O/wave_timing_40km.txt 2>&1; grep -v "^  File\|^    " $O/wave_timing_40km.txt | head -8 | cut -c1-200
HZ_WT_SECTOR=8,0 timeout 300 python tools/wave_timing.py > $O/wave_timing_sector.txt 2>&1; grep -v "^  File\|^    " $O/wave_timing_sector.txt | head -8 | cut -c1-200
