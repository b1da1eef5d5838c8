#!/bin/bash
# tools/scratch/prof.sh <tag> <script> [args]: per-kernel averages of a python script
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/"$@" > $R/gpurun_out/prof_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $R/gpurun_out/prof_$tag.log; exit 1; }
python3 - "$R/gpurun_out/prof_$tag" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:25]:
        print("   %-110s %5s %9.1f us %5.1f%%" % (r["Name"].replace("void ", "")[:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
