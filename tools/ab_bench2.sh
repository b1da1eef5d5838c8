#!/bin/bash
# A/B timing of library builds on the GPU box, cold (4 input sets) and warm (1): tools/ab_bench2.sh lib1.so lib2.so ...
here=$(dirname "$0")
for lib in "$@"; do
  for rot in 4 1; do
    echo -n "== $lib rotate $rot: "
    PDE_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-cases --no-traffic --rotate $rot 2>/dev/null | python $here/ab_line.py || exit 1
  done
done
