#!/usr/bin/env python3
"""The measurements behind DESIGN.md section 4's design decisions, each reproducible with the switch named
beside it: run on the GPU box, writes one JSON document.

    python tools/experiments.py > gpurun_out/r3_experiments.json

For every row: the benchmark workload (cfg3: 7x7 SRTM3 tiles, 16000x4000, 360 degrees, zfar 600 km; rows
marked zfar 40 km: the API's default far clip), 40 renders back to back, three times -> the median ms per
render; and once with every kernel alone on the chip (HZ_SERIAL=1) -> the stage times of HIP events.
Rows whose switch makes the picture WRONG say so: they bound what a different design could gain, they are
not candidates - and the library that ships does not know their switches: those rows run in a copy of the
tree built with -DHZ_EXPERIMENTS.  Build-time variants (another register budget, two framebuffers) are
built into copies of the tree under /tmp likewise."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = ["bench.py", "--steps", "40", "--warmup", "6", "--no-cpu-baseline", "--no-host", "--no-extra"]


def run(env, root=ROOT, zfar=None, repeat=3):
    e = dict(os.environ, **env)
    args = BENCH + (["--zfar", str(zfar)] if zfar else [])
    out = {"ms_per_render": [], "serial": None}
    for k in range(repeat):
        r = subprocess.run([sys.executable] + args, cwd=root, env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        # (rows whose switch draws a WRONG picture: bench.py's parity gates then make its exit code non-zero - the line is what counts here)
        if not line:
            out["error"] = r.stderr[-400:]
            return out
        out["ms_per_render"].append(json.loads(line[0])["ms_per_step"])
    r = subprocess.run([sys.executable] + args, cwd=root, env=dict(e, HZ_SERIAL="1"), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        d = json.loads(line[0])
        out["serial"] = {"ms_per_render": d["ms_per_step"], "k_march_second_round_ms": d["roofline"]["kernel_ms"],
                         **{k: v for k, v in d["roofline"]["other_kernels_ms"].items()}}
    out["median_ms_per_render"] = sorted(out["ms_per_render"])[len(out["ms_per_render"]) // 2]
    return out


def variant(name, flags):
    """a copy of the tree built with extra hipcc flags"""
    dst = os.path.join("/tmp", "hz_variant_" + name)
    shutil.rmtree(dst, ignore_errors=True)
    shutil.copytree(ROOT, dst, ignore=shutil.ignore_patterns("gpurun_out", ".git", "build", "*.so", "__pycache__", "profiles"))
    r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(dst, "horizonator_amd", "csrc"), "HIPFLAGS_EXTRA=" + flags], capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr[-400:]
    subprocess.run(["make", "-s", "-C", os.path.join(dst, "oracle")], capture_output=True, text=True)
    return dst, None


ROWS = [
    ("as shipped", {}, None, "the reference point of every other row"),
    ("one round forced", {"HZ_TWO_PASS": "0"}, None, "no first round, no early depth test"),
    ("k_big: fragments dropped instead of written (WRONG picture)", {"HZ_EXP_FB_BIG": "1"}, None,
     "upper bound of what k_big's atomics cost: nothing of the near field reaches the framebuffer (the early depth test then finds no occluders, the conversion skips most segments)"),
    ("k_big: plain stores instead of atomic minima (WRONG picture)", {"HZ_EXP_FB_BIG": "2"}, None,
     "what a first round that OWNED its pixels (tile-binned, depth in LDS, one store per pixel: north_star's design) could gain at the very most"),
    ("k_march: its own fragments dropped (WRONG picture)", {"HZ_EXP_FB_MARCH": "1"}, None,
     "upper bound of what the marching waves' atomics cost"),
    ("k_march: plain stores (WRONG picture)", {"HZ_EXP_FB_MARCH": "2"}, None, "the same with the store kept"),
    ("both: plain stores (WRONG picture)", {"HZ_EXP_FB_MARCH": "2", "HZ_EXP_FB_BIG": "2"}, None, "no atomic anywhere"),
    ("the second round's waves do not look before their atomics", {"HZ_PRETEST_MARCH": "0"}, None, "round 2's behaviour: every fragment of the marching waves is an atomic (default for framebuffers of up to 256 MB)"),
    ("north_star's tile-binned rasteriser with depth in LDS for the large triangles (hz_k_tile.h)", {"HZ_TILES": "1"}, None,
     "64x64-pixel tiles: batches of up to 64 triangles drawn into 32 KB of LDS (LDS atomic minima) and merged into the framebuffer with one atomic minimum per touched pixel; first rounds only; byte-identical"),
    ("the same, zfar 40 km", {"HZ_TILES": "1"}, 40000.0, "where the near field is most of the work"),
    ("no coarse depth in a series of renders", {"HZ_HIZ": "0"}, None,
     "the second rounds without the tables of hz_k_hiz.h (default: zoomed views always, whole panoramas when they are part of a series): the build before them"),
    ("coarse depth forced", {"HZ_HIZ": "1"}, None, "in a series the same as shipped; it differs for the first renders only"),
    ("as shipped, zfar 40 km", {}, 40000.0, "the API's default far clip"),
    ("k_big: plain stores, zfar 40 km (WRONG picture)", {"HZ_EXP_FB_BIG": "2"}, 40000.0, "with the 40 km far clip k_big is the longest kernel"),
    ("two rounds forced, zfar 40 km", {"HZ_TWO_PASS": "1"}, 40000.0, ""),
    ("no coarse depth, zfar 40 km", {"HZ_HIZ": "0"}, 40000.0, ""),
    ("as shipped, once more", {}, None, "the first row again at the end of the list: what the box drifts by during the run"),
]
VARIANTS = [
    ("five marching waves per SIMD", "waves5", "-DMR_WAVES_PER_EU=5", "96 registers per wave instead of 105"),
    ("three marching waves per SIMD", "waves3", "-DMR_WAVES_PER_EU=3", ""),
    ("two framebuffers and queue sets instead of three", "nfb2", "-DHZ_NFB=2", "the first round of panorama k+1 then waits for the conversion of k-1"),
    ("four framebuffers", "nfb4", "-DHZ_NFB=4", ""),
]


def wrong_picture(env):
    return any(k.startswith("HZ_EXP_FB") or k == "HZ_MARCH_DEBUG" for k in env)


def main():
    exp_root, exp_err = variant("experiments", "-DHZ_EXPERIMENTS")
    doc = {"workload": "bench.py cfg3 (7x7 SRTM3 tiles, 16000x4000, 360 degrees, zfar 600 km unless a row says otherwise), 40 renders back to back",
           "how": "python tools/experiments.py on one MI355X; every row: the environment switch (or build flag) that reproduces it",
           "rows": []}
    # the wrong-picture rows run in another build (their branches cost the marching kernel registers, and with them the second
    # co-resident wave of its neighbours: DESIGN.md section 4): that build, unswitched, is their reference point
    rec = {"what": "as shipped, built with -DHZ_EXPERIMENTS: the reference point of the WRONG-picture rows", "switch": "-", "zfar_m": 600000.0,
           "build": "make -C horizonator_amd/csrc HIPFLAGS_EXTRA=-DHZ_EXPERIMENTS", "note": ""}
    rec.update(run({}, root=exp_root) if exp_root else {"error": exp_err})
    doc["rows"].append(rec)
    rec = dict(rec, what="the same, zfar 40 km", zfar_m=40000.0)
    rec.update(run({}, root=exp_root, zfar=40000.0) if exp_root else {"error": exp_err})
    doc["rows"].append(rec)
    rows, variants = ROWS, VARIANTS
    if os.environ.get("EXPERIMENTS_QUICK"):     # round 4: the rows that bound what the atomics cost, and the two rounds / coarse depth switches
        keep = ("as shipped", "one round forced", "k_big: plain stores instead", "k_march: plain stores", "both: plain stores", "k_big looks before",
                "no coarse depth in a series", "as shipped, zfar 40 km", "k_big: plain stores, zfar 40 km", "north_star's tile-binned", "the same, zfar 40 km")
        rows = [r for r in ROWS if r[0].startswith(keep)]
        variants = []
    for what, env, zfar, note in rows:
        rec = {"what": what, "switch": " ".join(f"{k}={v}" for k, v in env.items()) or "-", "zfar_m": zfar or 600000.0, "note": note}
        if wrong_picture(env):
            rec["build"] = "make -C horizonator_amd/csrc HIPFLAGS_EXTRA=-DHZ_EXPERIMENTS"
            rec.update(run(env, root=exp_root, zfar=zfar) if exp_root else {"error": exp_err})
        else:
            rec.update(run(env, zfar=zfar))
        doc["rows"].append(rec)
        print(what, rec.get("median_ms_per_render"), rec.get("serial"), file=sys.stderr, flush=True)
    for what, name, flags, note in variants:
        rec = {"what": what, "switch": "make -C horizonator_amd/csrc HIPFLAGS_EXTRA=" + flags, "zfar_m": 600000.0, "note": note}
        root, err = variant(name, flags)
        if root is None:
            rec["error"] = err
        else:
            rec.update(run({}, root=root))
            # the registers the variant's marching kernel got
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import test_kernel_resources as tk
                save = tk.ROOT
                tk.ROOT = root
                k = tk._kernels()
                tk.ROOT = save
                rec["k_march_resources"] = {"k_march<false,false>": tk._one(k, "k_marchILb0ELb0ELb0ELb0E"), "k_march<false,true> (coarse depth)": tk._one(k, "k_marchILb0ELb1ELb0ELb0E")}
            except Exception as e:
                rec["k_march_resources"] = repr(e)
        doc["rows"].append(rec)
        print(what, rec.get("median_ms_per_render"), rec.get("serial"), rec.get("k_march_resources"), file=sys.stderr, flush=True)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
