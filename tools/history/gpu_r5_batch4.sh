#!/bin/bash
# round 5, fourth batch: zero-copy host path against the copy engine, sector counts, thread counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_sequences.py -x -q -m gpu > $O/pytest_quick.txt 2>&1
tail -2 $O/pytest_quick.txt; grep -n -B5 -A30 "^___" $O/pytest_quick.txt | head -60
for z in 1 0; do
  HZ_HOST_ZERO_COPY=$z HZ_HOST_TIMES=1 timeout 300 python tools/host_inclusive.py cfg3 sectors=1,2,4,8 > $O/host_zero$z.txt 2>&1
  echo "== HZ_HOST_ZERO_COPY=$z"; grep "^cfg3:" $O/host_zero$z.txt
  for n in 1 4 8; do grep " $n sector" $O/host_zero$z.txt | sed -n '5,6p' | cut -c60-420; done
done
for t in 16 32 48; do echo "== threads $t"; HZ_COPY_THREADS=$t timeout 300 python tools/host_inclusive.py cfg3 sectors=4,8 2>&1 | grep "^cfg3:"; done
timeout 300 python tools/host_inclusive.py cfg2 2>&1 | grep "^cfg2:"
