#!/bin/bash
# Code-generation survey of every kernel of the library (no GPU needed): instructions, SGPR spills
# to lane registers (v_writelane), scratch instructions, packed f32 VALU (v_pk_*_f32: slower on
# gfx950 than the scalar pairs), 64-bit divisions show up as instruction count.
#   tools/isa_survey.sh [min instructions to list, default 2500]
# Round 4 found with it: hash-probe loops unrolled into 37.9 k instructions (k_pq_count), 200-700
# spilled scalar plane offsets in the wide-column loss kernels, spill stores that were 8 % of a
# kernel's HBM writes.
cd "$(dirname "$0")/../nicr_mt_scene_analysis_amd/csrc"
MIN=${1:-2500}
tmp=$(mktemp -d)
for f in *.hip; do
    extra=""; [ "$f" = losses_multi.hip ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function $extra \
        --offload-device-only -S -o "$tmp/$f.s" "$f" 2>/dev/null &
done
wait
printf "%-18s %7s %6s %7s %6s  %s\n" file instrs spills scratch pk_f32 kernel
for f in *.hip; do
    awk -v f="$f" -v min="$MIN" '
        /^_ZN4nmsa[^ ]*:/ {name=$1}
        /^[ \t]+(s_|v_|ds_|global_|buffer_|flat_|scratch_)/ {n[name]++}
        /v_writelane/ {w[name]++}
        /^[ \t]+scratch_/ {sc[name]++}
        /v_pk_[a-z]*_f32/ {pk[name]++}
        END {for (k in n) if (n[k] >= min || w[k] > 0 || sc[k] > 0)
                 printf "%-18s %7d %6d %7d %6d  %s\n", f, n[k], w[k]+0, sc[k]+0, pk[k]+0, substr(k, 1, 90)}' "$tmp/$f.s"
done | sort -k2,2n
rm -rf "$tmp"
