"""tools/hiz_ab.py's lines as a table: scene | settings | ms per render | one render waited for | sha | kill rate"""
import json, sys
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    if "settings" in d:
        print(d["scene"], "|", d["settings"], "|", d["ms_per_render"], d["one_render_waited_for_ms"], d["sha"], d["kill_rate"], d.get("second_round_queue"))
    else:
        print(d)
