#!/bin/bash
# Run on the GPU box: per-kernel statistics of the training step's physics losses (tools/step_profile.py).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/step_stats
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step_stats -- python3 $root/tools/step_profile.py > $out/step_stats.log 2>&1 || { echo failed; tail -5 $out/step_stats.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(f"{sys.argv[1]}/step_stats/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
tail -1 $out/step_stats.log
