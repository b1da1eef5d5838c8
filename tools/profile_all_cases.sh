#!/bin/bash
# Per-case kernel durations and HBM-side traffic of the benchmark configurations beside Taylor-Green (run on the GPU box):
#   tools/profile_all_cases.sh <round tag>   -> gpurun_out/<tag>_<case>_traffic.json
tag=${1:-r02}
here=$(dirname "$0")
$here/case_profile.sh ${tag}_rb "RayleighBenard B=64" --fused > gpurun_out/${tag}_rb.log 2>&1 || exit 1
$here/case_profile.sh ${tag}_shear "ShearFlow B=64" --fused > gpurun_out/${tag}_shear.log 2>&1 || exit 1
$here/case_profile.sh ${tag}_gsfd "GrayScott fd4 B=64" --fused > gpurun_out/${tag}_gsfd.log 2>&1 || exit 1
$here/case_profile.sh ${tag}_heat3d "Heat3D" --fused > gpurun_out/${tag}_heat3d.log 2>&1 || exit 1
$here/case_profile.sh ${tag}_burgers3d "Burgers3D" --fused > gpurun_out/${tag}_burgers3d.log 2>&1 || exit 1
$here/case_profile.sh ${tag}_tg3d "TaylorGreen3D" --fused > gpurun_out/${tag}_tg3d.log 2>&1 || exit 1
