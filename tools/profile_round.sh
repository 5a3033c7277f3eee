#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02a'
# For the headline step (bench.py) and for every leg of bench.py's `secondary` object
# (tools/profile_leg.py): three separate rocprofv3 passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes — kernel trace + stats; --pmc FETCH_SIZE;
# --pmc WRITE_SIZE (counters never combined with tracing domains other than --kernel-trace; the
# program itself follows `--`) — then the summaries under profiles/<tag>_* and
# profiles/<tag>_<leg>_*, and one plain bench.py run whose JSON line is kept next to them.
set -e -o pipefail
TAG=${1:-r02x}
LEGS=${2:-"cfg2_bf16 cfg3_losses cfg5_bf16 ce150 cos512 cos768 next_rows api cfg5_full"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

three_passes() {        # <name> <program args...>
    local name=$1; shift
    echo "[profile] $name: kernel trace"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/kt" -o kt -- python3 "$@" > "$OUT/$name.kt.log" 2>&1
    echo "[profile] $name: FETCH_SIZE"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/fetch" -o fetch -- python3 "$@" > "$OUT/$name.fetch.log" 2>&1
    echo "[profile] $name: WRITE_SIZE"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/write" -o write -- python3 "$@" > "$OUT/$name.write.log" 2>&1
}
summarize() {           # <name> <tag suffix> [bench log]
    local name=$1 suffix=$2 log=$3
    local stats fetch write
    stats=$(find "$OUT/$name/kt" -name '*kernel_stats.csv' | head -1)
    fetch=$(find "$OUT/$name/fetch" -name '*counter_collection.csv' | head -1)
    write=$(find "$OUT/$name/write" -name '*counter_collection.csv' | head -1)
    (cd "$ROOT" && python3 tools/summarize_profile.py "$TAG$suffix" "$stats" "$fetch" "$write" $log)
}

if [ -z "$LEGS_ONLY" ]; then
mkdir -p "$OUT/headline"
three_passes headline "$ROOT/bench.py" --steps 30 --warmup 10 --no-cpu-baseline --no-secondary
echo "[profile] plain bench"
(cd "$ROOT" && timeout -k 10 400 python3 bench.py > "$OUT/bench.log" 2> "$OUT/bench.err")
summarize headline "" "$OUT/bench.log"
fi
for leg in $LEGS; do
    mkdir -p "$OUT/$leg"
    three_passes "$leg" "$ROOT/tools/profile_leg.py" "$leg"
    summarize "$leg" "_$leg"
done
mkdir -p "$ROOT/gpurun_out/profiles_$TAG"
cp "$ROOT"/profiles/${TAG}_* "$ROOT/gpurun_out/profiles_$TAG/"
[ -z "$LEGS_ONLY" ] && tail -c 600 "$OUT/bench.log" || true
