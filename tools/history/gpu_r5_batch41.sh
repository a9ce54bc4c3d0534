#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b41; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1; grep -E " passed| failed| error" $O/pytest_full.txt | tail -2; grep -B30 "Error\|assert" $O/pytest_full.txt | head -60 | cut -c1-200
timeout 900 python tools/hiz_ab.py cfg3_zoom45 cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom10 --steps 10 --set "HZ_VERTEX_CACHE=0" 2>&1 | python tools/hiz_ab_table.py | grep "|" | cut -c1-75
for k in 1 2; do python bench.py --steps 20 --warmup 5 --no-host --no-scenes --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('K20 ms', round(d['ms_per_step'],4), '40km', round(d['zfar_40km']['ms_per_step'],4), 'same view', round(d['same_viewpoint']['ms_per_step'],4), d['parity']['bgr_sha_is_llvmpipe'])"; done
HZ_SERIAL=1 python bench.py --steps 10 --warmup 3 --no-host --no-scenes --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('serial: ms', round(d['ms_per_step'],4), 'k_march', round(d['roofline']['kernel_ms'],4), d['roofline'].get('other_kernels_ms'))"
