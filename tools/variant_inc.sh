#!/bin/bash
# Diagnostic build of one PDE's kernels with edited traits: tools/variant_inc.sh <Pde_name> <variant name> '<sed script for its .inc>' [extra flags...]
# The sources are copied to build/variants/t_<name>/ and the sed script is applied to gen/pde_<pde>.inc there.
pde=$1; name=$2; script=$3; shift; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
top=$root/build/variants/t_$name
src=$top/pde_amd/csrc
rm -rf $top && mkdir -p $src && cp -r $root/pde_amd/csrc/. $src/ && cp -r $root/include $top/include
inc=$src/gen/pde_${pde#Pde_}.inc
sed -i -E "$script" $inc
cd $src && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared \
  -DPDE_AMD_DIAG_ONLY=$pde "$@" -Rpass-analysis=kernel-resource-usage -o $root/build/variants/$name.so *.hip 2>&1 | grep -E "error|bwd_march_q_kernel|VGPRs:|Spill|Occupancy|LDS Size" | grep -A5 "bwd_march_q_kernel" | grep -vE "^--" | head -24
ls -la $root/build/variants/$name.so | awk '{print $5, $9}'
