#!/bin/bash
# Every backward mode of the cases whose default is a measured choice:  tools/mode_sweep.sh [case filters...]
here=$(dirname "$0")
cases=("$@"); [ ${#cases[@]} -eq 0 ] && cases=("Heat3D" "ActiveMatter" "Burgers2DCH" "Advection3D" "SwiftHohenberg")
for c in "${cases[@]}"; do for m in stored recompute fused; do PDE_AMD_FORCE_FUSED=1 timeout -k 10 120 python $here/mode_probe.py "$c" $m 2>/dev/null | tail -1 || exit 1; done; done
