#!/usr/bin/env python3
"""A/B on ONE box: the marching kernel's transform before and after the shortened sequences of round 5
(tools/patches/r5_short_sequences.diff reversed = before): k_march alone and a render of a series of 20, three
times each, alternating.

    python tools/ab_short_sequences.py > gpurun_out/r5_ab_short_sequences.txt"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import experiments as ex
import march_bounds as mb


def main():
    after, err = ex.variant("after", "")
    before, err2 = ex.variant("before", "")
    assert after and before, (err, err2)
    diff = os.path.join(ex.ROOT, "tools", "patches", "r5_short_sequences.diff")
    subprocess.run(["git", "apply", "-R", diff], cwd=before, check=True)
    # (-k: the self-test library of today does not compile against yesterday's hz_fast.h; the library itself does)
    lib = os.path.join(before, "horizonator_amd", "libhorizonator.so")
    os.remove(lib)
    r = subprocess.run(["make", "-s", "-k", "-j8", "-C", os.path.join(before, "horizonator_amd", "csrc")], capture_output=True, text=True)
    assert os.path.exists(lib), r.stderr[-400:]
    for k in range(3):
        print("before", mb.run(before, {}), flush=True)
        print("after ", mb.run(after, {}), flush=True)


if __name__ == "__main__":
    main()
