#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_cfg5.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/init_times.py cfg3 cfg5 > $O/init_device.txt 2>&1; cat $O/init_device.txt
HORIZONATOR_INGEST=host timeout 900 python tools/init_times.py cfg3 cfg5 > $O/init_host.txt 2>&1; cat $O/init_host.txt
