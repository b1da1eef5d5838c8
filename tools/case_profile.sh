#!/bin/bash
# Kernel durations and HBM-side traffic of one tools/bench_all.py case:  tools/case_profile.sh <tag> "<case filter>" [--fused]
#   three rocprofv3 runs (kernel trace + stats; FETCH_SIZE counters; WRITE_SIZE counters: the two do not fit one pass, and
#   counters are collected without any trace domain) -> gpurun_out/<tag>_traffic.json and a printed table
tag=$1; filt=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/tools/bench_all.py "$filt" "$@" > $out/${tag}_stats.log 2>&1 || { echo "stats run failed"; tail -5 $out/${tag}_stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 $root/tools/bench_all.py "$filt" "$@" > $out/${tag}_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $out/${tag}_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 $root/tools/bench_all.py "$filt" "$@" > $out/${tag}_write.log 2>&1 || { echo "write pass failed"; tail -5 $out/${tag}_write.log; exit 1; }
cd $root && python3 tools/case_report.py "$tag" "$filt" "$@"
