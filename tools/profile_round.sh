#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01e'
# Three separate rocprofv3 passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes
# (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE — counters never combined with
# tracing domains other than --kernel-trace), then the summaries under profiles/<tag>_*.
set -e -o pipefail
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline"
echo "[profile] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ARGS > "$OUT/kt.log" 2>&1
echo "[profile] FETCH_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 $ARGS > "$OUT/fetch.log" 2>&1
echo "[profile] WRITE_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 $ARGS > "$OUT/write.log" 2>&1
echo "[profile] plain bench"
cd "$ROOT"
timeout -k 10 300 python3 bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"
STATS=$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)
FETCH=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
WRITE=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
echo "[profile] $STATS | $FETCH | $WRITE"
mkdir -p "$ROOT/gpurun_out/profiles_$TAG"
python3 tools/summarize_profile.py "$TAG" "$STATS" "$FETCH" "$WRITE" "$OUT/bench.log"
cp profiles/${TAG}_* "$ROOT/gpurun_out/profiles_$TAG/"
tail -c 1200 "$OUT/bench.log"
