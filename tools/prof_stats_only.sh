#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command only (no PMC passes): tools/prof_stats_only.sh <tag> [bench args]
TAG=${1:-r04_x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_under_rocprof.json" 2>/dev/null
