#!/bin/bash
# Collect hardware counters for bench.py, one rocprofv3 run per counter set (each "set" is one quoted argument).
#   tools/pmc.sh <tag> "SQ_A SQ_B ..." "FETCH_SIZE" ...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  out=$root/gpurun_out/pmc_${tag}_$i
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out -- python3 $root/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-other-cases --no-traffic --no-graph > $out.log 2>&1 || { echo "set $i failed"; tail -3 $out.log; exit 1; }
  i=$((i+1))
done
python3 - "$root" "$tag" <<'PY'
import csv, collections, glob, sys
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmc_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pde_amd" not in k or "finalize" in k or "totals" in k:
            continue
        agg[k.split("(")[0].replace("void pde_amd::", "")[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c in sorted(v):
        vals = v[c]
        print(f"    {c:28s} {sum(vals)/len(vals):16.1f}  (n={len(vals)})")
PY
