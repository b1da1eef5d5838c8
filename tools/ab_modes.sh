#!/bin/bash
# A/B timing of library builds on the GPU box in both step modes: tools/ab_modes.sh lib1.so lib2.so ...   (two passes, to see drift)
here=$(dirname "$0")
for pass in 1 2; do
for lib in "$@"; do
for mode in fused auto; do
  echo "== $lib $mode"
  PDE_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-cases --no-traffic --mode $mode 2>/dev/null | python $here/ab_line.py || exit 1
done
done
done
