#!/bin/bash
# the GPU test suite; the summary and the first failure's report
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/pytest_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_full.txt | tail -3
grep -n -B5 -A60 "^___" gpurun_out/pytest_full.txt | head -150
