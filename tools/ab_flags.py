#!/usr/bin/env python3
"""A/B on ONE box: the library as shipped against builds of it with extra hipcc flags (build-time options of the kernels):
k_march alone and a render of a series of 20, three times each, alternating.

    python tools/ab_flags.py -DMR_EARLYZ_PAIRS -DMR_FAR_GATE "-DMR_EARLYZ_PAIRS -DMR_FAR_GATE" """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import experiments as ex
import march_bounds as mb


def host_call(root):
    """tools/host_inclusive.py cfg3 in that tree: ms per call into kept buffers"""
    import re
    import subprocess
    r = subprocess.run([sys.executable, "tools/host_inclusive.py", "cfg3"], cwd=root, capture_output=True, text=True)
    m = re.search(r"kept buffers ([0-9.]+) ms/call", r.stdout + r.stderr)
    return float(m.group(1)) if m else None


def main():
    host = "--host" in sys.argv
    if host:
        sys.argv.remove("--host")
    trees = [("as shipped", ex.variant("shipped", "")[0])]
    for k, flags in enumerate(sys.argv[1:]):
        root, err = ex.variant(f"flags{k}", flags)
        assert root, err
        trees.append((flags, root))
    for k in range(3):
        for name, root in trees:
            print(f"{name:40s}", mb.run(root, {}), *(["host call", host_call(root)] if host else []), flush=True)


if __name__ == "__main__":
    main()
