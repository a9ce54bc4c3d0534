#!/bin/bash
# round 5, seventh batch: one call in 4 sectors on a time axis (kernels and copies), the host's fill and scatter rates
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/host_inclusive.py cfg3 sectors=4 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $O/trace.log
python3 - <<'PY'
import csv, glob
O='gpurun_out/r5b7'
ev=[]
for f in glob.glob(O+'/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:34], 'q'+r.get('Queue_Id','?')))
for f in glob.glob(O+'/trace/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')[12:], ''))
ev.sort()
packs=[i for i,e in enumerate(ev) if 'k_pack_host' in e[2]]
# the synchronous calls come first (12 calls x 4 packs): take the 9th call
i0=packs[8*4]
t0=ev[i0][0]-700000
out=open(O+'/trace_one_call.txt','w')
for e in ev:
    if t0 <= e[0] <= t0+4500000:
        out.write("%9.1f .. %9.1f (%7.1f us) %s %s\n" % ((e[0]-t0)/1e3, (e[1]-t0)/1e3, (e[1]-e[0])/1e3, e[2], e[3]))
out.close()
print(open(O+'/trace_one_call.txt').read()[:7000])
PY
gcc -O2 -fopenmp -o $O/hostfill tools/hostfill_bench.c horizonator_amd/csrc/hz_scatter.c -Ihorizonator_amd/csrc -lm 2> $O/hostfill_build.txt && timeout 150 $O/hostfill > $O/hostfill.txt 2>&1
grep "node -1\|node  0" $O/hostfill.txt | head -14
gcc -O2 -fopenmp -ffp-contract=off -o $O/scatter_bench tools/scatter_bench.c horizonator_amd/csrc/hz_scatter.c -Ihorizonator_amd/csrc -lm 2> $O/scatter_build.txt && timeout 150 $O/scatter_bench > $O/scatter_bench.txt 2>&1
cat $O/scatter_bench.txt
