#!/bin/bash
# A/B of two library builds inside one gpurun call: alternates `bench.py` between the product library and
# tools/lab/libacvae_<name>.so (tools/ab_build.py) and prints ms_per_step / kernel averages of every run.
# usage: tools/ab_bench.sh <name> [rounds] [bench args...]
NAME=${1:-prev}; ROUNDS=${2:-2}; shift 2 || true
cd "$(dirname "$0")/.."
for r in $(seq 1 $ROUNDS); do
  for which in product $NAME; do
    if [ $which = product ]; then LIBARG=""; else LIBARG="--lib $PWD/tools/lab/libacvae_$NAME.so"; fi
    python3 bench.py --no-cpu-baseline $LIBARG "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-8s ms_per_step %.3f  captions/s %.1f  fwd/dgrad launch %.4f ms  wgrad launch %.4f ms' % ('$which', d['ms_per_step'], d['value'], r['avg_launch_ms'], r['wgrad_avg_launch_ms']))"
  done
done
