#!/bin/bash
# round 5, eighth batch: the hybrid prefill; zoomed views with second rounds by tile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_sequences.py -x -q -m gpu > $O/pytest_quick.txt 2>&1
tail -2 $O/pytest_quick.txt; grep -n -B5 -A30 "^___" $O/pytest_quick.txt | head -60
HZ_HOST_TIMES=1 timeout 400 python tools/host_inclusive.py cfg3 sectors=1,2,3,4,6 > $O/host.txt 2>&1
grep "^cfg3:" $O/host.txt
for n in 2 3 4; do grep " $n sector" $O/host.txt | sed -n '5,6p' | cut -c60-460; done
for p in 30 60; do echo "== prefill $p"; HZ_HOST_PREFILL=$p timeout 300 python tools/host_inclusive.py cfg3 sectors=3,4 2>&1 | grep "^cfg3:"; done
timeout 300 python tools/host_inclusive.py cfg2 2>&1 | grep "^cfg2:"
timeout 600 python tools/hiz_ab.py cfg3_zoom45_summit cfg3_zoom10 cfg3_zoom45_south --steps 10 --set "HZ_VERTEX_CACHE=0" --set "HZ_VERTEX_CACHE=0 HZ_TILES2=1" 2>&1 | python tools/hiz_ab_table.py | grep "|\|same_bytes" > $O/zoomed_tiles2.txt
cat $O/zoomed_tiles2.txt
