#!/usr/bin/env python3
"""One horizonator_render_offscreen() call on a time axis: the kernels and the device-to-host copies of a rocprofv3
--kernel-trace --memory-copy-trace run of tools/host_inclusive.py (profiles/r5_host_inclusive.txt).

    python tools/host_timeline.py <rocprofv3 output dir> [call number among the synchronous calls, default 8]
"""
import csv
import glob
import sys

out = sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", ""), ""))
ev.sort()
packs = [i for i, e in enumerate(ev) if "k_pack_host" in e[2]]
if not packs:
    sys.exit("no k_pack_host in the trace")
# a call's first kernel: the small fill that zeroes its cursor words
fills = [i for i, e in enumerate(ev) if "fillBuffer" in e[2] and e[1] - e[0] < 20000]
starts = [i for i in fills if any(p > i and ev[p][0] - ev[i][0] < 3000000 for p in packs)]
i0 = starts[min(nth, len(starts) - 1)]
t0 = ev[i0][0]
i1 = starts[starts.index(i0) + 1] if starts.index(i0) + 1 < len(starts) else len(ev)
print("us since the call's first command; kernels with their hardware queue, copies = the copy engine (device to host)")
for e in ev[i0:i1]:
    print("%9.1f .. %9.1f (%7.1f us) %s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
