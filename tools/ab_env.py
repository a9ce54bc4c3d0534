#!/usr/bin/env python3
"""A/B on ONE box by environment: the library as built, under each of the given sets of variables (the switches a context
reads when it is made), three times each, alternating: k_march alone, a render of a series of 20, the same view again, the
API's 40 km far clip, one panorama waited for (tools/march_bounds.py: run).

    python tools/ab_env.py "HZ_SX_ORDER=0" "HZ_SX_ORDER=1" """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import march_bounds as mb

sets = [dict(kv.split("=", 1) for kv in a.split()) for a in sys.argv[1:]] or [{}]
for k in range(3):
    for env in sets:
        print(f"{' '.join(f'{a}={b}' for a, b in env.items()) or '-':40s}", mb.run(ROOT, env), flush=True)
