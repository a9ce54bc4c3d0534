#!/usr/bin/env python3
"""The reference's own device code - vertex.glsl, geometry.glsl, fragment.glsl, unmodified, read
from /root/reference at run time - timed on Mesa llvmpipe on THIS machine's cores, for BASELINE's
single-context configurations.  Container only (needs /root/reference and Mesa's swrast driver;
the GPU box has neither): the numbers go into profiles/llvmpipe_timing.json and BASELINE.md with
the machine they were taken on.

What is timed is the reference's per-frame GL work through our own GL host (oracle/glsl_golden.c:
the reference's glClear + glDrawElements of its index buffer, and its two glReadPixels) - not its
CPU readback conversion (flip + depth->range, reference horizonator-lib.c:949-1047), which
horizonator-lib.c cannot be built for here; BASELINE.md section 2 has that split
from the survey's build.

    python tools/llvmpipe_timing.py [--configs cfg1,cfg2,cfg3] [--reps 3] [--threads N]
"""
import argparse, hashlib, json, os, platform, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, oracle
from oracle import glsl_run

CONFIGS = {"cfg1": (600, 2000, 500), "cfg2": (1800, 8000, 2000), "cfg3": (4200, 16000, 4000)}
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg1,cfg2,cfg3")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=0, help="LP_NUM_THREADS (0: every core; llvmpipe caps its pool at 16)")
    ap.add_argument("--zfar", default="40000,600000")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "llvmpipe_timing.json"))
    a = ap.parse_args()
    if not glsl_run.available():
        sys.exit("needs /root/reference and oracle/_ref/glsl_golden (the build container)")
    cores = os.cpu_count()
    threads = a.threads or cores
    cpu = ""
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip(); break
    os.environ["HZ_GL_TIMING"] = str(a.reps)
    rows = []
    for name in a.configs.split(","):
        R, W, H = CONFIGS[name]
        d = hzutil.dem_dir_for(LAT, LON, R)
        od = oracle.Dem(LAT, LON, d, radius_cells=R)
        m = od.mosaic()
        for zfar in [float(z) for z in a.zfar.split(",")]:
            v = od.view(LAT, LON, W, H, -180.0, 180.0, znear=100.0, zfar=zfar)
            t0 = time.time()
            g = glsl_run.render(m, v, W, H, threads=threads)
            wall = time.time() - t0
            reps = [(float(x), float(y)) for x, y in re.findall(r"draw_s ([0-9.]+) readback_s ([0-9.]+)", g["log"])]
            draw = sorted(x for x, _ in reps)[len(reps) // 2]
            rb = sorted(y for _, y in reps)[len(reps) // 2]
            o = oracle.render(m, v, W, H, want=("bgr", "z24"))
            row = {"config": name, "R": R, "W": W, "H": H, "zfar": zfar, "triangles": 2 * (2 * R - 1) ** 2,
                   "draw_s_each": [x for x, _ in reps], "readback_s_each": [y for _, y in reps],
                   "draw_s": draw, "readback_s": rb, "mpix_per_s_draw_plus_readback": W * H / (draw + rb) / 1e6,
                   "mtri_per_s": 2 * (2 * R - 1) ** 2 / draw / 1e6, "whole_process_s": wall,
                   "equals_oracle": bool(np.array_equal(g["bgr"], o["bgr"]) and np.array_equal(g["z24"], o["z24"])),
                   "bgr_sha256": hashlib.sha256(g["bgr"].tobytes()).hexdigest()}
            rows.append(row)
            print(json.dumps(row), flush=True)
    gl = re.search(r"renderer: (.*)", g["log"])
    out = {"what": "reference vertex/geometry/fragment.glsl on Mesa llvmpipe through oracle/glsl_golden.c: glClear+glDrawElements and the two glReadPixels of one frame, median of the repetitions",
           "machine": {"cpu": cpu, "cores": cores, "LP_NUM_THREADS": threads, "platform": platform.platform()},
           "renderer": gl.group(1) if gl else None, "rows": rows}
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
