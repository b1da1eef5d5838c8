#!/bin/bash
# rocprofv3 kernel statistics of selected tools/bench_all.py cases:  tools/prof_cases.sh <tag> <case filters...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/tools/bench_all.py "$@" > $out/${tag}.log 2>&1 || { tail -5 $out/${tag}.log; exit 1; }
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{out}/{tag}_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pde_amd" in r["Name"]:
            print(f"{r['Name'].replace('void pde_amd::','').replace('pde_amd::','')[:64]:64s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
