#!/bin/bash
# Diagnostic build with the kernels of one PDE only: tools/build_variant_pde.sh <Pde_name> <variant name> [extra flags...]
pde=$1; name=$2; shift; shift
cd "$(dirname "$0")/../pde_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared \
  -DPDE_AMD_DIAG_ONLY=$pde "$@" -o ../../build/variants/$name.so *.hip 2>&1 | grep -E "error|spill" ; ls -la ../../build/variants/$name.so | awk '{print $5, $9}'
