#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_cfg5.py > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 $B --exchange-anyway --no-host > $O/b_exch1.json 2> $O/err_exch1.log
timeout 300 $B --exchange-anyway --no-host --no-extra --wire packed > $O/b_exch1_packed.json 2> $O/err_exch1p.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --same-gpu > $O/b_gloo2.json 2> $O/err_gloo2.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --backend gloo --same-gpu > $O/b_gloo4.json 2> $O/err_gloo4.log
timeout 300 $B > $O/b_default.json 2> $O/err_default.log
tail -3 $O/pytest.log
for f in $O/b_*.json; do echo $f; python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print(' ms/step %.3f  verified %s wire %s resends %s 40km %s' % (d['ms_per_step'], d.get('gathered_panorama_equals_single_gpu_render'), d['config'].get('wire_bytes_per_rank'), d['config'].get('strip_resends'), d.get('zfar_40km',{}).get('ms_per_step')))
except Exception as e: print(' failed', e)
"; done
tail -5 $O/err_exch1.log $O/err_gloo2.log | cut -c1-300
