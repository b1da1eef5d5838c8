#!/bin/bash
# rocprofv3 kernel statistics of a python script: tools/prof_script.sh <tag> <script.py> [args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/"$@" > $out/${tag}.log 2>&1 || { tail -5 $out/${tag}.log; exit 1; }
grep -v "^[EW]2026" $out/${tag}.log | tail -3
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{out}/{tag}_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) >= 0.3:
            print(f"{r['Name'].replace('void ','').replace('pde_amd::','')[:84]:84s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
