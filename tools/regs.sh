#!/bin/bash
# registers / spills / scratch / LDS / occupancy of the kernels of one HIP source (device-only
# compile with the Makefile's flags, LLVM's kernel-resource-usage remarks):
#   tools/regs.sh losses_cos.hip [kernel substring] [extra hipcc flags]
cd "$(dirname "$0")/../nicr_mt_scene_analysis_amd/csrc"
src=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function \
    --offload-device-only -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null "$@" 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk -v pat="$pat" '
/Function Name:/ {name=$NF}
/ VGPRs:/ {vg=$NF}
/TotalSGPRs:/ {sg=$NF}
/SGPRs Spill/ {ssp=$NF}
/AGPRs:/ {ag=$NF}
/VGPR Spill/ {sp=$NF}
/ScratchSize/ {scr=$NF}
/Occupancy/ {occ=$NF}
/LDS Size/ { if (name ~ pat) printf "%-70s vgpr %3s agpr %3s vspill %3s sgpr %3s sspill %3s scratch %4s occ %s lds %s\n", substr(name,1,70), vg, ag, sp, sg, ssp, scr, occ, $NF }'
