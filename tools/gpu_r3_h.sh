#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python tools/experiments.py > gpurun_out/r3_experiments.json 2> gpurun_out/r3_experiments.err
tail -25 gpurun_out/r3_experiments.err | cut -c1-330
