#!/bin/bash
# Regenerates the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   pass 1: --kernel-trace --stats of the bench command      -> gpurun_out/prof/stats
#   pass 2: --kernel-trace --pmc FETCH_SIZE (own pass)        -> gpurun_out/prof/fetch
#   pass 3: --kernel-trace --pmc WRITE_SIZE (own pass)        -> gpurun_out/prof/write
# then tools/summarize_profile.py folds them into <tag>_trainstep_kernel_stats.csv / traffic_conv_igemm.json.
# Usage: tools/profile_bench.sh <tag> [bench.py arguments ...]     e.g. r02_a   or   r02_a_bf16 --dtype bf16
set -e
TAG=${1:-r02_x}
shift || true
EXTRA="$@"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python3 "$ROOT/bench.py" --steps 20 --warmup 3 $EXTRA > "$OUT/bench.json"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > "$OUT/bench_under_rocprof.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null
python3 "$ROOT/tools/summarize_profile.py" "$OUT" "$TAG" $EXTRA
