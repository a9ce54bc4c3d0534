"""the last jobs of a traced series (tools/r6/host_trace_run.py): k_ship durations, the period between their starts, and
how long the kernels of the draws beside them took"""
import csv, glob, sys, collections
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]))
ev.sort()
ships = [e for e in ev if e[2].startswith("k_ship")]
last = ships[-6:]
print("k_ship (last 6): us " + " ".join("%.0f" % ((e[1]-e[0])/1e3) for e in last) + "; period between starts: " + " ".join("%.0f" % ((b[0]-a[0])/1e3) for a, b in zip(last, last[1:])))
t0 = last[0][0]
agg = collections.defaultdict(list)
for e in ev:
    if e[0] >= t0 and not e[2].startswith(("k_ship", "__amd")): agg[e[2] + (" long" if e[2].startswith("k_march") and e[1]-e[0] > 400000 else "")].append((e[1]-e[0])/1e3)
for k, v in sorted(agg.items()): print("   %-34s n=%2d mean %7.1f us  max %7.1f" % (k, len(v), sum(v)/len(v), max(v)))
