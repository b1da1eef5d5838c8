#!/bin/bash
# Diagnostic build of the library with only the Taylor-Green kernels: tools/build_variant.sh <name> [extra -D flags...]
name=$1; shift
cd "$(dirname "$0")/../pde_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared \
  -DPDE_AMD_DIAG_ONLY=Pde_tg2d "$@" -o ../../build/variants/$name.so *.hip 2>&1 | grep -E "error|spill" ; ls -la ../../build/variants/$name.so | awk '{print $5, $9}'
