#!/bin/bash
# A/B timing of library builds on selected tools/bench_all.py cases: tools/ab_cases.sh "<case filter>" lib1.so lib2.so ...
filt=$1; shift
for pass in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  PDE_AMD_LIB=$lib timeout -k 10 200 python tools/bench_all.py $filt 2>&1 | tail -2 || exit 1
done
done
