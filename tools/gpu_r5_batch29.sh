#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b29; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hostpath.py tests/test_fullsize_checksums.py -x -q -m gpu 2>&1 | tail -2
for p in 30 100 60 30 100; do echo "== HZ_HOST_PREFILL_LAST=$p"; HZ_HOST_PREFILL_LAST=$p HZ_HOST_TIMES=1 timeout 300 python tools/host_inclusive.py cfg3 2>&1 | grep "^cfg3:\|blobs in place" | tail -3 | cut -c1-330; done
timeout 200 python tools/host_inclusive.py cfg2 2>&1 | grep "^cfg2:"
