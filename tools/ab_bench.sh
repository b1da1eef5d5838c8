#!/bin/bash
# A/B timing of library builds on the GPU box: tools/ab_bench.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "== $lib"
  PDE_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('ms/step %.4f  bwd %.4f fwd %.4f' % (d['ms_per_step'], d["roofline"]["median_ms"], d["roofline"]["fwd_median_ms"]))"
done
