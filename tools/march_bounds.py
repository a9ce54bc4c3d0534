#!/usr/bin/env python3
"""What bounds k_march?  (round 5: 9 % fewer vector instructions made it 2 % faster.)  Builds of the tree with
-DHZ_EXPERIMENTS and, for one, every vertex transformed TWICE; each timed alone on the chip (HZ_SERIAL=1: the
second round's marching kernel between HIP events) and in a series of 20 renders, with the survivors of the cull
dropped before (HZ_MARCH_DEBUG=1) or after (=2) the early depth test - WRONG pictures, bounds only.

    python tools/march_bounds.py > gpurun_out/r5_march_bounds.txt"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import experiments as ex

BENCH = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-host", "--no-scenes", "--zfar", "600000"]


def run(root, env):
    out = {}
    for serial in (1, 0):
        e = dict(os.environ, **env)
        if serial:
            e["HZ_SERIAL"] = "1"
        r = subprocess.run([sys.executable] + BENCH, cwd=root, env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            return {"error": r.stderr[-300:]}
        d = json.loads(line[0])
        if serial:
            out["k_march alone, ms"] = round(d["roofline"]["kernel_ms"], 4)
        else:
            out["render of a series, ms"] = round(d["ms_per_step"], 4)
            if "same_viewpoint" in d:
                out["the same view again, ms"] = round(d["same_viewpoint"]["ms_per_step"], 4)
            if "zfar_40km" in d:
                out["far clip 40 km, ms"] = round(d["zfar_40km"]["ms_per_step"], 4)
            if "single_panorama_latency_ms" in d:
                out["one panorama waited for, ms"] = round(d["single_panorama_latency_ms"]["value"], 4)
    return out


def main():
    builds = [("experiments", "-DHZ_EXPERIMENTS", 3), ("twice", "-DHZ_EXPERIMENTS -DHZ_EXP_TRANSFORM_TWICE", 3),
              ("waves5", "-DMR_WAVES_PER_EU=5", 1), ("shipped", "", 1)]
    for name, flags, nenv in builds:
        root, err = ex.variant(name, flags)
        if root is None:
            print(name, "build failed:", err)
            continue
        for env in ({}, {"HZ_MARCH_DEBUG": "2"}, {"HZ_MARCH_DEBUG": "1"})[:nenv]:
            print(f"{flags:48s} {' '.join(f'{k}={v}' for k, v in env.items()) or '-':18s}", run(root, env), flush=True)


if __name__ == "__main__":
    main()
