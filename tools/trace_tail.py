"""the kernels (and copies) of a rocprofv3 --kernel-trace run on a time axis, from the n-th k_pack_host from the END on"""
import csv, glob, sys
out = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", ""), ""))
ev.sort()
packs = [i for i, e in enumerate(ev) if "k_pack_host" in e[2]]
i0 = packs[-back] if len(packs) >= back else 0
while i0 > 0 and ev[i0][0] - ev[i0-1][1] < 200000 and i0 > packs[-back] - 40: i0 -= 1
t0 = ev[i0][0]
for e in ev[i0:]:
    print("%9.1f .. %9.1f (%7.1f us) %s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
