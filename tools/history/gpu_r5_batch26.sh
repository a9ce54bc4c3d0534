#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b26; mkdir -p $O
timeout 1500 python tools/hiz_ab.py cfg3_zoom45 cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom10 --steps 10 --set "HZ_VERTEX_CACHE=0" --set "HZ_VERTEX_CACHE=0 HZ_EXP_INLINE_MAX=32" --set "HZ_VERTEX_CACHE=0 HZ_EXP_INLINE_MAX=16" 2>&1 | python tools/hiz_ab_table.py | grep "|\|same_bytes" > $O/zoomed_inline.txt
cat $O/zoomed_inline.txt | cut -c1-120
HZ_WT_DEBUG=1 timeout 300 python tools/wave_timing.py > $O/wave_timing.txt 2>&1; grep -v "^  File\|^    " $O/wave_timing.txt | tail -12 | cut -c1-200
HZ_HOST_SECTORS=3 HZ_COPY_THREADS=2 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_multi.py 2>&1 | tail -3
