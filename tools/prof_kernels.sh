#!/bin/bash
# per-kernel durations of one python tool on the GPU box (kernel trace + stats):
#   gpurun -- 'bash tools/prof_kernels.sh tools/prof_f4.py [kernel substring ...]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
script=$1; shift
export TMPDIR=/tmp
out=$R/gpurun_out/prof_kernels
rm -rf $out; mkdir -p $out
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $R/$script > $out/run.log 2>&1)
tail -2 $out/run.log | grep -v "^W20\|^E20" 
python3 - "$out" "$@" <<'PY'
import csv, glob, sys
pats = sys.argv[2:]
for path in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:40]:
        if pats and not any(p in r['Name'] for p in pats):
            continue
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us min {float(r['MinNs']) / 1e3:8.1f}")
PY
