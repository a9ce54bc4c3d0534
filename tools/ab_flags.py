#!/usr/bin/env python3
"""A/B on ONE box: the library as shipped against builds of it with extra hipcc flags (build-time options of the kernels):
k_march alone and a render of a series of 20, three times each, alternating.

    python tools/ab_flags.py -DMR_EARLYZ_PAIRS -DMR_FAR_GATE "-DMR_EARLYZ_PAIRS -DMR_FAR_GATE" """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import experiments as ex
import march_bounds as mb


def main():
    trees = [("as shipped", ex.variant("shipped", "")[0])]
    for k, flags in enumerate(sys.argv[1:]):
        root, err = ex.variant(f"flags{k}", flags)
        assert root, err
        trees.append((flags, root))
    for k in range(3):
        for name, root in trees:
            print(f"{name:40s}", mb.run(root, {}), flush=True)


if __name__ == "__main__":
    main()
