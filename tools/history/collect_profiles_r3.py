#!/usr/bin/env python3
"""gpurun_out/final3 (written on the GPU box by tools/gpu_final_r3.sh) -> the files under profiles/ that DESIGN.md
quotes for round 3.  Copies, except r3_scenes.json, which puts the five runs of tools/gpu_scenes.sh into one file
with a table and a summary of the default run."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final3")
P = os.path.join(ROOT, "profiles")


def copy(src, dst):
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
    print(dst)


def largest(pattern):
    """(rocprofv3 writes one stats file per process of the command: the bench's own is the longest)"""
    files = glob.glob(os.path.join(O, pattern))
    big = max(os.path.getsize(f) for f in files)
    return max((f for f in files if os.path.getsize(f) >= 0.9*big), key=os.path.getmtime)     # (the newest: a short rerun leaves older ones behind)


copy("bench.json", "r3_final_cfg3_bench.json")
copy("bench_k40.json", "r3_final_cfg3_bench_k40.json")
shutil.copy(largest("kt_serial/*/*_kernel_stats.csv"), os.path.join(P, "r3_serial_cfg3_kernel_stats.csv"))
shutil.copy(largest("kt_pipelined/*/*_kernel_stats.csv"), os.path.join(P, "r3_pipelined_cfg3_kernel_stats.csv"))
copy("pipelined_timeline.txt", "r3_pipelined_timeline.txt")
copy("pmc_r3_final.json", "pmc_r3_final_cfg3.json")
copy("pmc_r3_final.json", "pmc_latest.json")
copy("pmc_r3_mix.json", "pmc_r3_instruction_mix_cfg3.json")
copy("host_inclusive.txt", "r3_host_inclusive.txt")
copy("r3_experiments.json", "r3_experiments.json")
with open(os.path.join(P, "r3_coarse_depth.txt"), "w") as f:
    f.write(open(os.path.join(O, "coarse_depth.txt")).read())
    for part in ("coarse_depth_reach.txt",          # seven zoomed views x the first round's reach
                 "coarse_depth_series.txt"):        # whole panoramas in a series of 30 renders, without / with, twice
        if os.path.exists(os.path.join(O, part)):
            f.write(open(os.path.join(O, part)).read())
print("r3_coarse_depth.txt")
with open(os.path.join(P, "r3_sector_timing.txt"), "w") as f:
    f.write(open(os.path.join(O, "sector_timing.txt")).read())
    f.write("\ntools/sector_b2b.py: one sector rendered back to back as sparse strips, ms per strip\n")
    f.write(open(os.path.join(O, "sector_b2b.txt")).read())
with open(os.path.join(P, "r3_multi_rank_loops_on_one_gpu.jsonl"), "w") as f:
    for name in ("multi_4ranks_one_gpu_rotate.json", "multi_4ranks_one_gpu_root0.json", "exchange_anyway.json"):
        f.write(open(os.path.join(O, name)).read().strip() + "\n")

runs = {}
for path in sorted(glob.glob(os.path.join(O, "scenes", "*.json"))):
    runs[os.path.basename(path)[:-5]] = json.load(open(path))
table = {}
for run, d in runs.items():
    for scene, rec in d["scenes"].items():
        table.setdefault(scene, {})[run] = round(rec["ms_per_render"], 4) if "ms_per_render" in rec else rec.get("error")
default = runs["default"]["scenes"]
head = default["cfg3"]
summary = {}
for scene, rec in default.items():
    c = rec.get("counters") or {}
    summary[scene] = {
        "ms_per_render": round(rec["ms_per_render"], 4), "ps_per_triangle": round(rec["ps_per_triangle"], 3),
        "vs_headline": round(rec["ps_per_triangle"] / head["ps_per_triangle"], 2),
        "early_z_kill_rate": c.get("early_z_kill_rate"), "triangles_set_up": c.get("triangles_set_up"), "to_k_big": c.get("to_k_big"),
        "pixel_centres_tested_in_the_waves": c.get("pixel_centres_tested_in_the_waves"),
    }
per_mpix = lambda r: r["ms_per_render"] / (r["image"][0] * r["image"][1] / 1e6)
summary["cfg3_zoom45"]["explanation"] = (
    "a 45 degree view at 16000 px has the angular resolution of a 128000-px panorama: every triangle covers 64 times the pixels of "
    "the headline's.  Its second round keeps coarse depth (hz_k_hiz.h: profiles/r3_coarse_depth.txt has the same view without: "
    "the early depth test then only reaches boxes of up to 4x2 pixels and kills 7 %% instead of 78 %%).  What is left is per fragment, "
    "not per triangle; per megapixel of output it is %.1fx the headline's." % (per_mpix(default["cfg3_zoom45"]) / per_mpix(head)))
summary["cfg2"]["explanation"] = (
    "5.4x fewer triangles than the headline but only 4x fewer pixels and the same number of kernel launches: per-pixel work "
    "(fragments, the conversion of 16 Mpix) and launch latencies do not shrink with the triangle count; per megapixel of output it is "
    "%.1fx the headline's." % (per_mpix(default["cfg2"]) / per_mpix(head)))
out = {
    "what": "tools/scenes.py on one MI355X (tools/gpu_scenes.sh): the scenes the kernel was not tuned on, under the library's defaults "
            "(with the marching waves' own counters) and with its two heuristics forced - HZ_NEAR_PX: the first round takes the cells "
            "wider than that many pixels (default 20); HZ_TWO_PASS: one / two rounds (default: two from 6 Mpix on with a far clip three "
            "reaches of the first round away)",
    "runs": runs, "ms_per_render_table": table, "summary_default_run": summary,
}
json.dump(out, open(os.path.join(P, "r3_scenes.json"), "w"), indent=1)
print("r3_scenes.json")
for scene, row in table.items():
    print("  %-14s" % scene, row)
