#!/bin/bash
# A/B timing of library builds on the GPU box: tools/ab_bench.sh lib1.so lib2.so ...   (two passes, to see drift)
here=$(dirname "$0")
for pass in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  PDE_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-cases --no-traffic 2>/dev/null | python $here/ab_line.py || exit 1
done
done
