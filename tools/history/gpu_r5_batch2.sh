#!/bin/bash
# round 5, second batch: the quick tests, then host path timings (threads, sectors), a copy/kernel trace of one call,
# the vertex cache on / off, the zoomed views with k_clip<true>
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_scatter.py tests/test_gpu_api.py tests/test_gpu_sequences.py tests/test_naive_host_math.py tests/test_kernel_resources.py -x -q -m "gpu or not gpu" > $O/pytest_quick.txt 2>&1
tail -3 $O/pytest_quick.txt; grep -n -B5 -A40 "^___" $O/pytest_quick.txt | head -100
for t in default 32 48 96; do
  if [ $t = default ]; then unset HZ_COPY_THREADS; else export HZ_COPY_THREADS=$t; fi
  timeout 300 python tools/host_inclusive.py cfg3 sectors=1,4 2>&1 | grep -v "equals the device render: True" | sed "s/^/threads=$t /"
done > $O/host_threads.txt 2>&1
unset HZ_COPY_THREADS
cat $O/host_threads.txt
HZ_HOST_TIMES=1 timeout 300 python tools/host_inclusive.py cfg3 sectors=1,4 > $O/host_times.txt 2>&1
grep "^hz_hip host path" $O/host_times.txt | sed -n '5,7p;20,22p'
# the copies and the kernels of a few calls on one time axis
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/host_inclusive.py cfg3 sectors=4 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R $O/trace | head -20
python3 - <<'PY'
import csv, glob, os
O='gpurun_out/r5b2'
k=glob.glob(O+'/trace/**/*kernel_trace.csv', recursive=True); m=glob.glob(O+'/trace/**/*memory_copy_trace.csv', recursive=True)
print(k, m)
ev=[]
for f in k:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], 'q'+r.get('Queue_Id','?')))
for f in m:
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')+' '+r.get('Source_Agent_Id','')+'->'+r.get('Destination_Agent_Id',''), ''))
ev.sort()
# the last sync call: find the last k_pack_host cluster
packs=[i for i,e in enumerate(ev) if 'k_pack_host' in e[2]]
if packs:
    # take the 4 packs of one call somewhere in the middle of the run
    i0=packs[len(packs)//2 - (len(packs)//2)%4]
    t0=ev[i0][0]-1500000
    out=open(O+'/trace_one_call.txt','w')
    for e in ev:
        if t0 <= e[0] <= t0+6000000:
            out.write("%10.1f .. %10.1f (%8.1f us) %s %s\n" % ((e[0]-t0)/1e3, (e[1]-t0)/1e3, (e[1]-e[0])/1e3, e[2], e[3]))
    out.close()
    print(open(O+'/trace_one_call.txt').read()[:6000])
PY
# the vertex cache: a series of 20 renders of the headline view, cold and cached; zoomed views with and without it
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host --no-scenes > $O/bench_k20.json 2> $O/bench_k20.err
python3 -c "
import json; d=json.loads(open('$O/bench_k20.json').read()); print('cold ms', d['ms_per_step'], 'same_viewpoint', d.get('same_viewpoint'), 'parity', d.get('parity',{}).get('bgr_sha_is_llvmpipe'), '40km', d.get('zfar_40km',{}).get('ms_per_step'))"
timeout 900 python tools/hiz_ab.py cfg3_zoom45 cfg3_zoom45_east cfg3_zoom45_south cfg3_zoom45_summit cfg3_zoom45_valley cfg3_zoom45_rough cfg3_zoom10 --steps 10 --set "HZ_VERTEX_CACHE=0" --set "HZ_VERTEX_CACHE=1" 2>&1 | python tools/hiz_ab_table.py | grep "|\|same_bytes" > $O/zoomed.txt
cat $O/zoomed.txt
