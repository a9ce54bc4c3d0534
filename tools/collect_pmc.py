"""Aggregate rocprofv3 --pmc runs of bench.py into profiles/pmc_<tag>.json.

Run on the GPU box (see tools/collect_pmc.sh), FETCH_SIZE and WRITE_SIZE in
separate passes as MI355X_MICROARCH.md prescribes.  On gfx950 FETCH_SIZE counts
half the bytes of a streaming read; the factor is checked in the same run on
the conversion kernel, whose reads are known exactly: 8 B per pixel of every
256-pixel row segment that holds terrain (tools/touched_segments.py counts them;
argv[7] = that many bytes) - and its writes: 7 B/pixel + 8 B per terrain pixel
cleared (argv[8] = terrain fraction).  The runtime's own fill/copy kernels are
context creation, not part of a render: listed, not summed."""
import collections
import csv
import glob
import json
import sys

indir, outpath, config, zfar, W, H = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
resolve_reads = float(sys.argv[7]) if len(sys.argv) > 7 else 8.0*W*H
terrain_fraction = float(sys.argv[8]) if len(sys.argv) > 8 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(indir + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("k_big<"): name = "k_big"          # (k_big<SHARDS>: whole panoramas run <false>)
        if name.startswith("k_march<"):          # k_march<COUNTERS, HIZ>: false, false = the plain production instance;
            args = [a.strip() for a in name[name.index("<")+1:name.rindex(">")].split(",")]
            name = "k_march_coarse_depth" if len(args) > 1 and args[1] == "true" else "k_march"    # (second rounds of a series / of zoomed views)
        if name == "k_march":
            # a two-round draw launches k_march twice: the strips next to the viewer (small grid), then the rest
            name += "_near_round" if int(r.get("Grid_Size", 0)) < 1000000 else ""
        elif name == "k_big" or name == "k_clip":
            pass
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
count = {k: max(len(v) for v in cs.values()) for k, cs in agg.items()}
res = mean.get("k_resolve4<true>", mean.get("k_resolve<true>", mean.get("k_resolve", {})))
read_factor = None
if "FETCH_SIZE" in res:
    read_factor = resolve_reads / (res["FETCH_SIZE"] * 1024.0)
out = {"config": config, "zfar": zfar, "W": W, "H": H, "fetch_size_read_factor_measured_on_k_resolve": read_factor,
       "conversion_reads_bytes_expected": resolve_reads,
       "write_size_check_on_k_resolve": (res.get("WRITE_SIZE", 0) * 1024.0) / ((7.0 + 8.0*(terrain_fraction or 0.0)) * W * H) if res else None, "kernels": {}}
for k, cs in mean.items():
    if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    renders = count.get("k_resolve4<true>", count.get("k_resolve<true>", count.get("k_resolve", 1)))
    out["kernels"][k] = {"FETCH_SIZE_KB": cs["FETCH_SIZE"], "WRITE_SIZE_KB": cs["WRITE_SIZE"],
                         "hbm_bytes_per_launch": (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0,
                         "launches_per_render": count[k] / renders,
                         "hbm_bytes_per_render": (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0 * count[k] / renders,
                         **{c: v for c, v in cs.items() if c not in ("FETCH_SIZE", "WRITE_SIZE")}}
dom = max((k for k in ("k_march_coarse_depth", "k_march", "k_scatter") if k in out["kernels"]), key=lambda k: count[k])
out["kernel"] = dom
out["hbm_bytes_per_launch"] = out["kernels"][dom]["hbm_bytes_per_launch"]
out["hbm_bytes_per_render_all_kernels"] = sum(v["hbm_bytes_per_render"] for k, v in out["kernels"].items() if k.startswith("k_"))
json.dump(out, open(outpath, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for k, v in out["kernels"].items():
    print(k, {a: round(b) for a, b in v.items()})
