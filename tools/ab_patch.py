#!/usr/bin/env python3
"""A/B on ONE box: the library as shipped against a copy of the tree with a patch applied (-R: reversed) - the panorama
(k_march alone, a render of a series of 20, the same view again) and the seven zoomed views, twice, alternating.

    python tools/ab_patch.py -R tools/patches/r5_queue_shards.diff"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_flags_zoomed as zo
import experiments as ex
import march_bounds as mb


def main():
    reverse = sys.argv[1] == "-R"
    patch = os.path.abspath(sys.argv[2 if reverse else 1])
    shipped = ex.variant("shipped", "")[0]
    other, err = ex.variant("patched", "")
    assert shipped and other, err
    subprocess.run(["git", "apply"] + (["-R"] if reverse else []) + [patch], cwd=other, check=True)
    r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(other, "horizonator_amd", "csrc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-400:]
    trees = [("as shipped", shipped), (("without " if reverse else "with ") + os.path.basename(patch), other)]
    for k in range(2):
        for name, root in trees:
            ms = zo.run(root)
            print(f"{name:40s}", mb.run(root, {}), f"zoomed views: sum {sum(ms.values()):.3f} worst {max(ms.values()):.3f}", flush=True)


if __name__ == "__main__":
    main()
