#!/usr/bin/env python3
"""gpurun_out/final5 (written on the GPU box by tools/gpu_final_r5.sh) -> the files under profiles/ that DESIGN.md quotes
for round 5.  Copies and concatenations; nothing is computed here except profiles/r5_k_march_cycles.json (tools/k_march_cycles.py)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final5")
P = os.path.join(ROOT, "profiles")


def copy(src, dst):
    if not os.path.exists(os.path.join(O, src)):
        print("MISSING", src)
        return
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
    print(dst)


def largest(pattern):
    """(rocprofv3 writes one stats file per process of the command: the bench's own is the longest)"""
    files = glob.glob(os.path.join(O, pattern))
    if not files:
        return None
    big = max(os.path.getsize(f) for f in files)
    return max((f for f in files if os.path.getsize(f) >= 0.9*big), key=os.path.getmtime)


copy("bench_k20.json", "r5_final_cfg3_bench_k20.json")
copy("bench_k50.json", "r5_final_cfg3_bench_k50.json")
for kind in ("serial", "pipelined"):
    f = largest("kt_%s/*/*_kernel_stats.csv" % kind)
    if f:
        shutil.copy(f, os.path.join(P, "r5_%s_cfg3_kernel_stats.csv" % kind)); print("r5_%s_cfg3_kernel_stats.csv" % kind)
copy("pipelined_timeline.txt", "r5_pipelined_timeline.txt")
copy("pmc_r5_final.json", "pmc_r5_final_cfg3.json")
copy("pmc_r5_final.json", "pmc_latest.json")
copy("pmc_r5_mix.json", "pmc_r5_instruction_mix_cfg3.json")
if os.path.exists(os.path.join(P, "pmc_r5_instruction_mix_cfg3.json")):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k_march_cycles.py")], capture_output=True, text=True)
    if r.returncode == 0:
        open(os.path.join(P, "r5_k_march_cycles.json"), "w").write(r.stdout); print("r5_k_march_cycles.json")
    else:
        print("k_march_cycles.py:", r.stderr[-400:])
with open(os.path.join(P, "r5_host_inclusive.txt"), "w") as f:
    f.write("round 5, horizonator_render_offscreen() into host memory (tools/gpu_final_r5.sh on one MI355X box): tools/host_inclusive.py - median of\n"
            "10 calls after 2 warm-ups into kept buffers, 7 into fresh numpy arrays, a series with two panoramas in flight - with each call's own\n"
            "account of its time (HZ_HOST_TIMES=1), then by number of sectors, 8000 x 2000, and the dense path; at the end one call on a time axis\n"
            "(rocprofv3 --kernel-trace --memory-copy-trace, tools/host_timeline.py).\n\n")
    for part in ("host_inclusive.txt", "host_call_timeline.txt"):
        if os.path.exists(os.path.join(O, part)):
            f.write(open(os.path.join(O, part)).read() + "\n")
print("r5_host_inclusive.txt")
copy("zoomed.txt", "r5_zoomed_views.txt")
copy("modes.txt", "r5_modes.txt")
copy("sector_timing.txt", "r5_sector_timing.txt")
with open(os.path.join(P, "r5_multi_rank_loops_on_one_gpu.jsonl"), "w") as f:
    for name in ("multi_4ranks_one_gpu_c_loop.json", "exchange_anyway.json"):
        if os.path.exists(os.path.join(O, name)):
            f.write(open(os.path.join(O, name)).read().strip() + "\n")
if os.path.exists(os.path.join(O, "pytest_full.txt")):
    lines = [l for l in open(os.path.join(O, "pytest_full.txt")).read().strip().splitlines() if " passed" in l or " failed" in l or " error" in l]
    open(os.path.join(P, "r5_gpu_suite.txt"), "w").write("python -m pytest tests -x -q -m gpu on the MI355X box (tools/gpu_final_r5.sh):\n" + "\n".join(lines[-3:]) + "\n")
    print("r5_gpu_suite.txt")
