#!/usr/bin/env python3
"""A/B on ONE box: the marching loop with four inlined copies of mr_flush (until round 5), with one, with two
(tools/patches/r5_*_flush_cop*.diff applied to copies of the tree; the patched kernels name v103 to stay at four waves per
SIMD, 104 registers, like the four copies):
k_march alone and a render of a series of 20, three times each, alternating.

    python tools/ab_flush_copies.py > gpurun_out/r5_ab_flush_copies.txt"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import experiments as ex
import march_bounds as mb


def patched(name, diff):
    root, err = ex.variant(name, "")
    assert root, err
    if diff:
        subprocess.run(["git", "apply", os.path.join(ex.ROOT, "tools", "patches", diff)], cwd=root, check=True)
        r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(root, "horizonator_amd", "csrc")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-400:]
    return root


def main():
    trees = [("four copies", patched("four", None)), ("one copy   ", patched("one", "r5_one_flush_copy.diff")),
             ("two copies ", patched("two", "r5_two_flush_copies.diff"))]
    for k in range(3):
        for name, root in trees:
            print(name, mb.run(root, {}), flush=True)


if __name__ == "__main__":
    main()
