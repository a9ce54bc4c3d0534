#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b31; mkdir -p $O
timeout 1500 python tools/ab_flags_zoomed.py -DMR_PIPE_PRETEST > $O/ab_zoomed.txt 2>&1; cat $O/ab_zoomed.txt | cut -c1-250
timeout 1500 python tools/ab_flags.py -DMR_PIPE_PRETEST > $O/ab_flags.txt 2>&1; cat $O/ab_flags.txt
