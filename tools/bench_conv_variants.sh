#!/bin/bash
# Build conv/gemm variants with different tile macros on the GPU box and time the encoder with each (dev tool).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "32 2" "16 2" "16 3" "16 4" "32 3"; do
  set -- $v
  BKV=$1; OCC=$2
  for f in attention conv decoder encoder gemm losses optim prof rnn; do
    EXTRA=""
    if [ $f = conv ] || [ $f = gemm ]; then
      /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -DACVAE_BK=$BKV -DACVAE_NT_OCC=$OCC -c acvae_amd/csrc/$f.hip -o /tmp/v_$f.o
    fi
  done
  cp acvae_amd/libacvae_hip.so /tmp/lib_orig.so 2>/dev/null || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o acvae_amd/libacvae_hip.so /tmp/v_conv.o /tmp/v_gemm.o acvae_amd/csrc/attention.o acvae_amd/csrc/decoder.o acvae_amd/csrc/encoder.o acvae_amd/csrc/losses.o acvae_amd/csrc/optim.o acvae_amd/csrc/prof.o acvae_amd/csrc/rnn.o
  echo "=== BK=$BKV OCC=$OCC"
  python tools/bench_encoder.py 32 1000 4
done
