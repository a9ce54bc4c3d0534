#!/usr/bin/env python3
"""A/B on ONE box, zoomed views: the library as shipped against builds with extra hipcc flags - the seven zoomed views of
profiles/r5_zoomed_views.txt, ten renders each (tools/hiz_ab.py in each tree), twice, alternating.

    python tools/ab_flags_zoomed.py -DMR_PIPE_PRETEST"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import experiments as ex

SCENES = ["cfg3_zoom45", "cfg3_zoom45_east", "cfg3_zoom45_south", "cfg3_zoom45_summit", "cfg3_zoom45_valley", "cfg3_zoom45_rough", "cfg3_zoom10"]


def run(root):
    r = subprocess.run([sys.executable, "tools/hiz_ab.py", *SCENES, "--steps", "10", "--set", "HZ_VERTEX_CACHE=0"], cwd=root, capture_output=True, text=True)
    t = subprocess.run([sys.executable, "tools/hiz_ab_table.py"], cwd=root, input=r.stdout + r.stderr, capture_output=True, text=True).stdout
    ms = {}
    for l in t.splitlines():
        if "|" in l:
            f = [x.strip() for x in l.split("|")]
            ms[f[0].replace("cfg3_", "")] = float(f[2].split()[0])
    return ms


def main():
    trees = [("as shipped", ex.variant("shipped", "")[0])]
    for k, flags in enumerate(sys.argv[1:]):
        root, err = ex.variant(f"flags{k}", flags)
        assert root, err
        trees.append((flags, root))
    for k in range(2):
        for name, root in trees:
            ms = run(root)
            print(f"{name:28s} sum {sum(ms.values()):.3f} worst {max(ms.values()):.3f}  ", " ".join(f"{k}={v:.3f}" for k, v in ms.items()), flush=True)


if __name__ == "__main__":
    main()
