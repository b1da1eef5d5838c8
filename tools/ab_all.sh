#!/bin/bash
# A/B of two library builds over every case of tools/bench_all.py (fused mode): tools/ab_all.sh old.so new.so
for lib in "$1" "$2" "$1" "$2"; do
  echo "== $lib"
  PDE_AMD_LIB=$lib timeout -k 10 300 python tools/bench_all.py --fused 2>&1 | grep "fwd+bwd" | cut -c1-28,64-78 | tr '\n' ';' || exit 1
  echo
done
