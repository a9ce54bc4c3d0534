#!/usr/bin/env python3
"""Timeline of a pipelined run from a rocprofv3 --kernel-trace csv: for a few renders in the
middle, every kernel's start/end relative to the first one's start, with its queue."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# renders: split at k_resolve4 ends
split = "k_resolve4"
if "--split" in sys.argv:                               # (sectors end in k_pack_sparse)
    i = sys.argv.index("--split"); split = sys.argv[i + 1]; del sys.argv[i:i + 2]
res = [i for i, n in enumerate(names) if split in n]
mid = len(res)//2
lo, hi = res[mid-2], res[mid+1]
if len(sys.argv) > 2 and sys.argv[2] == "--last":      # the last N renders and the drain
    lo, hi = res[-int(sys.argv[3])-1] + 8, len(rows) - 1
if len(sys.argv) > 2 and sys.argv[2] == "--from":      # from the K-th conversion on
    lo, hi = res[int(sys.argv[3])] + 8, res[int(sys.argv[3]) + int(sys.argv[4])]
t0 = int(rows[lo]["Start_Timestamp"])
qs = {}
for r in rows[max(lo-8, 0):hi+1]:
    q = r["Queue_Id"]; qs.setdefault(q, len(qs))
    n = r["Kernel_Name"].split("(")[0][:14]
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print("%10.1f .. %10.1f  (%7.1f us)  q%d  %-14s grid %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3,
          (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, qs[q], n, g))
