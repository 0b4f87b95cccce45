#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "-DACVAE_WS_PRIO=3" "-DACVAE_WS_PRIO=1" "-DACVAE_ABL_NOSTASH" "-DACVAE_ABL_NOISSUE"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off $v -c acvae_amd/csrc/gemm.hip -o /tmp/v_gemm.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o acvae_amd/libacvae_hip.so /tmp/v_gemm.o acvae_amd/csrc/conv.o acvae_amd/csrc/attention.o acvae_amd/csrc/decoder.o acvae_amd/csrc/encoder.o acvae_amd/csrc/losses.o acvae_amd/csrc/optim.o acvae_amd/csrc/prof.o acvae_amd/csrc/rnn.o
  echo "=== variant: $v"
  python tools/bench_gemm.py | grep "NT 8192\|NT 4096\|NT 524288"
done
