#!/usr/bin/env python3
"""Timeline of a pipelined run from a rocprofv3 --kernel-trace csv: for a few renders in the
middle, every kernel's start/end relative to the first one's start, with its queue."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# renders: split at k_resolve4 ends
res = [i for i, n in enumerate(names) if "k_resolve4" in n]
mid = len(res)//2
lo, hi = res[mid-2], res[mid+1]
t0 = int(rows[lo]["Start_Timestamp"])
qs = {}
for r in rows[lo-8:hi+1]:
    q = r["Queue_Id"]; qs.setdefault(q, len(qs))
    n = r["Kernel_Name"].split("(")[0][:14]
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print("%10.1f .. %10.1f  (%7.1f us)  q%d  %-14s grid %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3,
          (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, qs[q], n, g))
