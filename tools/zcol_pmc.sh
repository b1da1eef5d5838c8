#!/bin/bash
# Hardware counters of a tools/bench_all.py case for one library build: tools/zcol_pmc.sh "<case filter>" lib.so "<counters>" [kernel substring]
filt=$1; lib=$2; ctrs=$3; ksub=${4:-zcol}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
tag=pmc_$(basename $lib .so)
rm -rf $R/gpurun_out/$tag
PDE_AMD_LIB=$R/$lib timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/bench_all.py $filt --quick > $R/gpurun_out/$tag.log 2>&1 || { echo "pmc run failed"; tail -5 $R/gpurun_out/$tag.log; exit 1; }
python3 - "$R/gpurun_out/$tag" "$ksub" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sys.argv[2] in k:
            agg[k.split("(")[0].replace("void pde_amd::", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c in sorted(v):
        vals = v[c]
        print("    %-28s %16.1f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
