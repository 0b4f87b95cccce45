#!/bin/bash
# Where does the Winograd kernel's time go?  Builds libacvae_abl{1..4}.so with parts of the main loop removed (results are
# wrong by construction) and times two layers with each: 1 = no activation reads / 2 = + no weight-fragment reads /
# 3 = + no global traffic / 4 = + no barrier; 5 = only the weight DMA removed / 6 = only the activation staging removed /
# 7 = weight DMA always from chunk 0 (cache-hot).  Run on the GPU box: bash tools/wino_ablate.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/acvae_amd/csrc"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function"
OBJS=$(ls *.o | grep -v conv_wino.o)
for n in ${ABLS:-1 2 3 4 5 6 7}; do
  DEF="-DWN_ABL=$n"; [ "$n" = "ilv" ] && DEF="-DWN_ILV=1"; [ "$n" = "prio" ] && DEF="-DWN_PRIO=1"
  /opt/rocm/bin/hipcc $FLAGS $DEF -c conv_wino.hip -o /tmp/conv_wino_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libacvae_abl$n.so $OBJS /tmp/conv_wino_abl$n.o
done
cd "$ROOT"
echo "== full"; python3 tools/wino_check.py time | grep -E "1000x64|125x8 512"
for n in ${ABLS:-1 2 3 4 5 6 7}; do echo "== ablation $n"; ACVAE_DEV_LIB=/tmp/libacvae_abl$n.so python3 tools/wino_check.py time | grep -E "1000x64|125x8 512"; done
