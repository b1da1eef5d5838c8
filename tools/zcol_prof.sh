#!/bin/bash
# Per-kernel averages of a tools/bench_all.py case for a list of library builds: tools/zcol_prof.sh "<case filter>" lib1.so lib2.so ...
filt=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf $R/gpurun_out/prof_$tag
  PDE_AMD_LIB=$R/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/bench_all.py $filt > $R/gpurun_out/prof_$tag.log 2>&1 || { echo "$tag failed"; tail -5 $R/gpurun_out/prof_$tag.log; exit 1; }
  echo "== $tag: $(grep 'fwd+bwd' $R/gpurun_out/prof_$tag.log | cut -c1-28,60-80)"
  python3 - "$R/gpurun_out/prof_$tag" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f))):
        if "pde_amd" in r["Name"] and float(r["Percentage"]) > 1:
            print("   %-100s %5s %9.1f us" % (r["Name"].replace("void pde_amd::", "").split("(")[0][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
