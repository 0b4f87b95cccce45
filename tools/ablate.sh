#!/bin/bash
# Builds ablation variants of the library (operand fetch path of the NT MFMA block switched off piece by piece,
# acvae_amd/csrc/mfma_tile.h ACVAE_ABL) into tools/abl/ and, on the GPU box, times the conv kernels with each.
# Usage: tools/ablate.sh build   (here, cross-compiles)      tools/ablate.sh run   (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/abl"
if [ "$1" = "build" ]; then
  for n in ${ABLS:-1 2 3 4 5}; do
    d="$ROOT/tools/abl/obj$n"; mkdir -p "$d"
    for f in "$ROOT"/acvae_amd/csrc/*.hip; do
      /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -DACVAE_ABL=$n -c "$f" -o "$d/$(basename "$f" .hip).o" &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/abl/libacvae_abl$n.so" "$d"/*.o
    rm -rf "$d"
  done
else
  export TMPDIR=/tmp
  export ACVAE_CONV_STRIP=0     # the NT ablation hooks live in the one-tap-per-stage kernel (nt_block); wgrad is unaffected
  cd /tmp
  for n in 0 ${ABLS:-1 2 3 4 5}; do
    lib="$ROOT/tools/abl/libacvae_abl$n.so"; [ "$n" = 0 ] && lib="$ROOT/acvae_amd/libacvae_hip.so"
    rm -rf /tmp/abl_tr
    ACVAE_DEV_LIB="$lib" rocprofv3 --kernel-trace -d /tmp/abl_tr -o t --output-format csv -- python3 "$ROOT/tools/bench_encoder.py" 32 1000 6 > /dev/null 2>&1 || true
    echo "== ablation $n"
    python3 "$ROOT/tools/conv_kernels.py" /tmp/abl_tr/t_kernel_trace.csv | grep -E "${ABL_GREP:-igemm|wgrad <2}" | head -14
  done
fi
