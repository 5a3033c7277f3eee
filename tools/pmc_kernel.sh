#!/bin/bash
# Hardware-counter passes for one diagnosis program (GPU box):
#   bash tools/pmc_kernel.sh <out dir under gpurun_out> <kernel substring> <python script> [args...]
# Each counter group is its own rocprofv3 pass (--kernel-trace + --pmc only).
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; KSUB=$2; shift 2
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "VALUBusy MemUnitStalled" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -o p -- python3 "$@" > "$OUT/$tag.log" 2>&1 || echo "pass '$grp' failed"
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]
    print(f'{k:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})')
PY
