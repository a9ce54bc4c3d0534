#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_exactness.py -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -25
