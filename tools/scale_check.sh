#!/bin/bash
# Scaling check of the train step: `bench.py` single-process, then under torch.distributed.run at N = 1, 2, 4, 8 (as many as
# the node has GPUs; the driver's launch line, one rank per GPU over RCCL).  Fails loudly when
#   * the N=1 launcher line differs by more than 3 % from the single-process line (the launcher path must cost nothing),
#   * config.rccl_ranks (dist.get_world_size() as seen by rank 0 around the timed region) or n_gpus is not N,
#   * a run prints no JSON line or exits non-zero.
#   * (N >= 2) RCCL kernels do not appear on exactly N ranks of a traced run (tools/overlap_report.py; OVERLAP=0 skips it).
# Prints one line per N with the value and value / (N x value(1)); the efficiency is for the reader, the records are in
# $OUT, incl. overlap_nN.txt: per gradient bucket what it overlapped and the exposed tail.  Usage: tools/scale_check.sh [steps] [warmup]   (OUT=dir, NS="1 2 4 8", TOL=0.03 from the environment)
set -u -o pipefail
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARM=${2:-5}; OUT=${OUT:-gpurun_out/scale_check}; TOL=${TOL:-0.03}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$OUT"
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')      # counting devices does not initialise the GPU
[ "$NGPU" -ge 1 ] || { echo "scale_check: no GPU visible" >&2; exit 2; }
NS=${NS:-"1 2 4 8"}
fail=0
json_of() { grep -E '^\{"metric"' "$1" | tail -1; }
field() { python3 -c 'import json,sys; d=json.loads(sys.argv[1]); v=d
for k in sys.argv[2].split("."): v=v[k]
print(v)' "$1" "$2"; }

echo "scale_check: $NGPU GPU(s), steps=$STEPS warmup=$WARM"
python3 bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline > "$OUT/single.log" 2> "$OUT/single.err" \
  || { echo "scale_check: FAIL single-process bench exited non-zero (see $OUT/single.err)" >&2; exit 1; }
J0=$(json_of "$OUT/single.log"); [ -n "$J0" ] || { echo "scale_check: FAIL no JSON line from the single-process run" >&2; exit 1; }
V0=$(field "$J0" value)
echo "single-process      : $V0 captions/s"
V1=""
for N in $NS; do
  if [ "$N" -gt "$NGPU" ]; then echo "N=$N: skipped ($NGPU GPU(s) on this node)"; continue; fi
  PORT=$((29500 + RANDOM % 2000))
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
    bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline > "$OUT/n$N.log" 2> "$OUT/n$N.err"
  rc=$?
  J=$(json_of "$OUT/n$N.log")
  if [ $rc -ne 0 ] || [ -z "$J" ]; then echo "N=$N: FAIL rc=$rc, JSON line ${J:+present}${J:-missing} (see $OUT/n$N.err)" >&2; fail=1; continue; fi
  V=$(field "$J" value); R=$(field "$J" config.rccl_ranks); G=$(field "$J" n_gpus)
  if [ "$R" != "$N" ] || [ "$G" != "$N" ]; then echo "N=$N: FAIL rccl_ranks=$R n_gpus=$G, expected $N" >&2; fail=1; fi
  if [ "$N" = 1 ]; then
    V1=$V
    python3 -c "import sys; a,b,t=map(float,sys.argv[1:]); d=abs(a-b)/b; print('N=1 launcher vs single-process: %+.2f %%' % (100*(a-b)/b)); sys.exit(d>t)" "$V" "$V0" "$TOL" \
      || { echo "N=1: FAIL launcher line differs by more than $TOL from the single-process line" >&2; fail=1; }
  fi
  if [ "$N" -ge 2 ] && [ "${OVERLAP:-1}" = 1 ] && command -v rocprofv3 > /dev/null; then
    # one short traced run per N: does the exchange hide behind the backward?  (tools/overlap_report.py asserts that RCCL
    # kernels appear on exactly N ranks and prints per bucket what it overlapped and the exposed tail)
    PORT=$((31500 + RANDOM % 2000)); TR="$OUT/trace_n$N"; rm -rf "$TR"
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d "$OLDPWD/$TR" -o t --output-format csv -- python3 -m torch.distributed.run \
        --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" "$OLDPWD/bench.py" --gpus "$N" --steps 6 --warmup 3 \
        --no-cpu-baseline > "$OLDPWD/$OUT/trace_n$N.log" 2>&1 )
    python3 tools/overlap_report.py "$TR" --gpus "$N" > "$OUT/overlap_n$N.txt" 2>&1 \
      && tail -n +1 "$OUT/overlap_n$N.txt" | grep -E "overlap_report|mean over" \
      || { echo "N=$N: FAIL overlap report (see $OUT/overlap_n$N.txt)" >&2; fail=1; }
  fi
  python3 -c "import sys; v,n,b=float(sys.argv[1]),int(sys.argv[2]),float(sys.argv[3]); print('N=%d ranks=%s          : %.1f captions/s, x%.2f of N=1 (%.1f %% of linear)' % (n, sys.argv[4], v, v/b, 100*v/(n*b)))" "$V" "$N" "${V1:-$V0}" "$R"
done
[ $fail -eq 0 ] && echo "scale_check: OK" || { echo "scale_check: FAILED" >&2; exit 1; }
