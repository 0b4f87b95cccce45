"""GPU: where a wavefront of conv_wino_kernel spends its cycles.  Builds the kernel with -DWN_TIMING=1 (cycle counter reads
around the phases of every chunk; the sums are written over the start of Y, so the output is garbage), runs two layers and
prints the average per chunk: issue of the DMA / loads, reads + transform + MFMAs, staging, vmcnt wait, barrier.
usage (GPU box): python tools/wino_timing.py"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "acvae_amd", "csrc")
LIB = "/tmp/libacvae_timing.so"
if "ACVAE_DEV_LIB" not in os.environ:
    flags = "-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-DWN_TIMING=1", "-c", os.path.join(CSRC, "conv_wino.hip"), "-o",
                           "/tmp/conv_wino_timing.o"])
    objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "conv_wino.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs,
                           "/tmp/conv_wino_timing.o"])
    os.environ["ACVAE_DEV_LIB"] = LIB
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=os.environ))

import torch  # noqa: E402
sys.path.insert(0, ROOT)
from acvae_amd import _lib  # noqa: E402

N = 32
for (H, W, Cin, Cout, act) in [(125, 8, 512, 512, True), (125, 8, 512, 512, False), (1000, 64, 64, 64, True), (250, 16, 256, 256, True)]:
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
    sc = torch.rand(Cin, device="cuda") + 0.5 if act else None
    sh = torch.randn(Cin, device="cuda") * 0.3 if act else None
    y = torch.empty(N, H, W, Cout, device="cuda")
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        _lib.call("acvae_conv3x3_fwd_wino", x, w, sc, sh, y, None, None, None, None, None, 0, None, ws, wsb, N, H, W, Cin, Cout,
                  _lib.current_stream())
    torch.cuda.synchronize()
    TW = W // 2
    R = 64 // TW
    nwg = N * -(-((H + 1) // 2) // R) * (Cout // 64)
    t = y.flatten()[:nwg * 64].view(nwg, 8, 8).double().cpu()
    nch = float(t[0, 0, 7])
    per = t[:, :, :5].mean(dim=(0, 1)) / nch
    names = ["issue DMA/loads", "reads+transform+MFMA", "staging", "vmcnt wait", "barrier"]
    print(f"{H}x{W} {Cin}->{Cout} act={int(act)}: {nwg} workgroups, {int(nch)} chunks; cycles per chunk (ideal MFMA: 2048 per wave, "
          f"4096 per SIMD): " + ", ".join(f"{n} {float(v):.0f}" for n, v in zip(names, per)) +
          f"; sum {float(per.sum()):.0f}; main loop {float(t[:, :, 5].mean()):.0f}, epilogue {float(t[:, :, 6].mean()):.0f} cycles", flush=True)
    bw = t[:, :, 4].mean(dim=0) / nch
    print("    barrier wait per wave:", " ".join(f"{float(v):.0f}" for v in bw), " compute per wave:",
          " ".join(f"{float(v):.0f}" for v in (t[:, :, 1].mean(dim=0) / nch)))

# ---- weight gradient: the sums are written over the start of dY at the end of the kernel
print("weight gradient:")
for (H, W, Cin, Cout, act) in [(125, 8, 512, 512, True), (1000, 64, 64, 64, True), (250, 16, 256, 256, False)]:
    x = torch.randn(N, H, W, Cin, device="cuda")
    dy = torch.randn(N, H, W, Cout, device="cuda")
    sc = torch.rand(Cin, device="cuda") + 0.5 if act else None
    sh = torch.randn(Cin, device="cuda") * 0.3 if act else None
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
    ws = torch.zeros(int(wsb), dtype=torch.uint8, device="cuda")
    for _ in range(1):
        _lib.call("acvae_conv3x3_wgrad_wino", dy, x, sc, sh, dw, ws, wsb, N, H, W, Cin, Cout, _lib.current_stream())
    torch.cuda.synchronize()
    nwg = 256
    t = dy.flatten()[:nwg * 64].view(nwg, 8, 8).double().cpu()
    ns = t[:, :, 4].clamp_min(1)
    per = (t[:, :, :4] / ns.unsqueeze(-1)).mean(dim=(0, 1))
    print(f"{H}x{W} {Cin}->{Cout} act={int(act)}: {float(ns.mean()):.0f} stages per workgroup; cycles per 16-tile stage (ideal MFMA: 4096 per "
          f"wave, 8192 per SIMD): load issue {float(per[0]):.0f}, reads+transform+MFMA {float(per[1]):.0f}, staging {float(per[2]):.0f}, "
          f"barrier {float(per[3]):.0f}; sum {float(per.sum()):.0f}", flush=True)
