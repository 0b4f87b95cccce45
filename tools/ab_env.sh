#!/bin/bash
# A/B of an environment switch inside one gpurun call: alternates `bench.py` with and without VAR=VALUE.
# usage: tools/ab_env.sh VAR=VALUE [rounds] [bench args...]
KV=${1:?VAR=VALUE}; ROUNDS=${2:-2}; shift 2 || true
cd "$(dirname "$0")/.."
for r in $(seq 1 $ROUNDS); do
  for which in default "$KV"; do
    if [ "$which" = default ]; then E=""; else E="$KV"; fi
    env $E python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-24s ms_per_step %.3f  captions/s %.1f  fwd/dgrad launch %.4f ms  wgrad launch %.4f ms' % ('$which', d['ms_per_step'], d['value'], r['avg_launch_ms'], r['wgrad_avg_launch_ms']))"
  done
done
