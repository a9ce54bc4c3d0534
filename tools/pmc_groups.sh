#!/bin/bash
# usage (GPU box, repo root): tools/pmc_groups.sh <tag> "<group1>" "<group2>" ... -- <bench args>
# one rocprofv3 --pmc pass of bench.py per counter group (no trace domains beside --pmc);
# per-kernel means of every counter end up in gpurun_out/pmc_<tag>.json
TAG=$1; shift
GROUPS_=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
[ "$1" == "--" ] && shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in "${GROUPS_[@]}"; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra "$@" >/dev/null 2>>$OUT/err.log
done
python3 - "$OUT" "$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.json" <<'PY'
import collections, csv, glob, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        # a kernel launched with different grids in one render (k_march: the strips next to the viewer,
        # then all the others) is kept apart by its grid size
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("k_big<"): name = "k_big"
        if name.startswith("k_march<"):          # k_march<COUNTERS, HIZ>: the second rounds of a series of renders run the instance with coarse depth
            args = [a.strip() for a in name[name.index("<")+1:name.rindex(">")].split(",")]
            name = "k_march_coarse_depth" if len(args) > 1 and args[1] == "true" else "k_march"
        agg[name + " grid " + str(r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, launches_seen=max(len(v) for v in cs.values())) for k, cs in agg.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(k, {a: round(b) for a, b in v.items()})
PY
rm -rf $OUT/*/*_counter_collection.csv
