#!/usr/bin/env python3
"""gpurun_out/final6 (written on the GPU box by tools/gpu_final_r6.sh) -> the files under profiles/ that DESIGN.md quotes
for round 6.  Copies and concatenations; nothing is computed here except profiles/r6_k_march_cycles.json (tools/k_march_cycles.py)."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final6")
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


copy("bench_k20.json", "r6_final_cfg3_bench_k20.json")
copy("bench_k50.json", "r6_final_cfg3_bench_k50.json")
for kind in ("serial", "pipelined"):
    f = largest("kt_%s/*/*_kernel_stats.csv" % kind)
    if f:
        shutil.copy(f, os.path.join(P, "r6_%s_cfg3_kernel_stats.csv" % kind)); print("r6_%s_cfg3_kernel_stats.csv" % kind)
copy("pipelined_timeline.txt", "r6_pipelined_timeline.txt")
copy("pmc_r6_final.json", "pmc_r6_final_cfg3.json")
copy("pmc_r6_final.json", "pmc_latest.json")
copy("pmc_r6_mix.json", "pmc_r6_instruction_mix_cfg3.json")
if os.path.exists(os.path.join(P, "pmc_r6_instruction_mix_cfg3.json")):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k_march_cycles.py")], capture_output=True, text=True)
    if r.returncode == 0:
        open(os.path.join(P, "r6_k_march_cycles.json"), "w").write(r.stdout); print("r6_k_march_cycles.json")
    else:
        print("k_march_cycles.py:", r.stderr[-400:])
with open(os.path.join(P, "r6_host_inclusive.txt"), "w") as f:
    f.write("round 6, horizonator_render_offscreen() into host memory (tools/gpu_final_r6.sh on one MI355X box): tools/host_inclusive.py - median of\n"
            "10 calls after 2 warm-ups into kept buffers, 7 into fresh numpy arrays, a series with two panoramas in flight - with each call's own\n"
            "account of its time (HZ_HOST_TIMES=1), then by number of sectors, 8000 x 2000, and the dense path; the first calls of four contexts\n"
            "(tools/first_call.py); at the end single calls and a series on a time axis (rocprofv3 --kernel-trace --memory-copy-trace of\n"
            "tools/host_trace_run.py, tools/trace_tail.py).  Cold draws throughout (HZ_VERTEX_CACHE=0).\n\n")
    for part in ("host_inclusive.txt", "first_call.txt", "host_call_timeline.txt"):
        if os.path.exists(os.path.join(O, part)):
            f.write(open(os.path.join(O, part)).read() + "\n")
print("r6_host_inclusive.txt")
copy("zoomed.txt", "r6_zoomed_views.txt")
copy("pcie_beside.txt", "r6_pcie_beside_final_box.txt")
with open(os.path.join(P, "r6_init_times.txt"), "w") as f:
    f.write("round 6, horizonator_init() on one MI355X box (tools/init_times.py: HZ_INIT_TIMES=1, two inits per configuration - the first also\n"
            "pays for tiles written a moment before and, the very first, for the HIP runtime's start-up): the tiles' bytes through pinned\n"
            "memory and k_ingest (the default), then HORIZONATOR_INGEST=host (round 1's decode on the host).\n\n")
    for part in ("init_times.txt", "init_times_host_ingest.txt"):
        if os.path.exists(os.path.join(O, part)):
            f.write("== " + part + "\n" + open(os.path.join(O, part)).read() + "\n")
print("r6_init_times.txt")
copy("sector_timing.txt", "r6_sector_timing.txt")
with open(os.path.join(P, "r6_multi_rank_loops_on_one_gpu.jsonl"), "w") as f:
    for name in ("multi_4ranks_one_gpu_c_loop.json", "exchange_anyway.json"):
        if os.path.exists(os.path.join(O, name)):
            f.write(open(os.path.join(O, name)).read().strip() + "\n")
if os.path.exists(os.path.join(O, "pytest_full.txt")):
    lines = [l for l in open(os.path.join(O, "pytest_full.txt")).read().strip().splitlines() if " passed" in l or " failed" in l or " error" in l]
    open(os.path.join(P, "r6_gpu_suite.txt"), "w").write("python -m pytest tests -x -q -m gpu on the MI355X box (tools/gpu_final_r6.sh):\n" + "\n".join(lines[-3:]) + "\n")
    print("r6_gpu_suite.txt")
