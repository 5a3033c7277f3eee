#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the kernels matching a substring, for one python tool:
#   gpurun -- 'bash tools/prof_fetch.sh tools/prof_nms.py k_nms_rows3'
# (separate --pmc passes with --kernel-trace only; FETCH_SIZE is reported in KB and counts a wide
# coalesced stream at half its bytes on gfx950: MI355X_MICROARCH.md — printed raw and doubled)
R=${GRAFT_REPO_ROOT:-$(pwd)}
script=$1; pat=$2
export TMPDIR=/tmp
out=$R/gpurun_out/prof_fetch
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- python3 $R/$script > $out/$c.log 2>&1)
done
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]
    m = sum(v) / len(v)
    print(f'{k[0]:50s} {k[1]:12s} mean {m:14.1f} KB  = {m * 1024 / 1e6:9.2f} MB (x2: {2 * m * 1024 / 1e6:9.2f} MB)  n={len(v)}')
PY
