"""GPU: Winograd F(2x2,3x3) convolution (acvae_conv3x3_fwd_wino / _dgrad_wino) against an fp64 conv2d and against the
implicit-GEMM kernel, element-wise error statistics and timing at the real layer shapes.
usage: python tools/wino_check.py [check] [time]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acvae_amd import _lib  # noqa: E402


def S():
    return _lib.current_stream()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def check(N, H, W, Cin, Cout, act, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    dy = torch.randn(N, Cout, H, W, generator=g)
    sc = sh = None
    if act:
        sc = torch.rand(Cin, generator=g) + 0.5
        sh = torch.randn(Cin, generator=g) * 0.3
        xin = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        xa = torch.where(xin > 0, xin.double(), torch.zeros((), dtype=torch.double))
    else:
        xa = x.double()
    y_ref = F.conv2d(xa.cuda(), w.double().cuda(), padding=1)
    dx_ref = F.conv_transpose2d(dy.double().cuda(), w.double().cuda(), padding=1)
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device="cuda")
    xd, wd, dyd = nhwc(x).cuda(), w.cuda().contiguous(), nhwc(dy).cuda()
    scd = None if sc is None else sc.cuda()
    shd = None if sh is None else sh.cuda()
    gamma = torch.ones(Cout, device="cuda"); beta = torch.zeros(Cout, device="cuda")
    out = {}
    for name in ("acvae_conv3x3_fwd", "acvae_conv3x3_fwd_wino"):
        y = torch.full((N, H, W, Cout), float("nan"), device="cuda")
        rm = torch.zeros(Cout, device="cuda"); rv = torch.ones(Cout, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda")
        bn = torch.empty(4, Cout, device="cuda")
        _lib.call(name, xd, wd, scd, shd, y, gamma, beta, rm, rv, nbt, 1, bn, ws, wsb, N, H, W, Cin, Cout, S())
        torch.cuda.synchronize()
        out[name] = (y, bn)
    ref = nhwc(y_ref)
    rms = float(ref.pow(2).mean().sqrt())
    e_d = float((out["acvae_conv3x3_fwd"][0].double() - ref).abs().max()) / rms
    e_w = float((out["acvae_conv3x3_fwd_wino"][0].double() - ref).abs().max()) / rms
    mean_ref = ref.mean(dim=(0, 1, 2)); var_ref = ref.var(dim=(0, 1, 2), unbiased=False)
    bnw = out["acvae_conv3x3_fwd_wino"][1].double()
    e_mean = float((bnw[2] - mean_ref).abs().max()) / rms
    e_istd = float((bnw[3] - 1 / torch.sqrt(var_ref + 1e-5)).abs().max())
    dxs = {}
    for name in ("acvae_conv3x3_dgrad", "acvae_conv3x3_dgrad_wino"):
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        _lib.call(name, dyd, wd, dx, ws, wsb, N, H, W, Cin, Cout, S())
        torch.cuda.synchronize()
        dxs[name] = dx
    dref = nhwc(dx_ref); drms = float(dref.pow(2).mean().sqrt())
    g_d = float((dxs["acvae_conv3x3_dgrad"].double() - dref).abs().max()) / drms
    g_w = float((dxs["acvae_conv3x3_dgrad_wino"].double() - dref).abs().max()) / drms
    # weight gradient
    xg = xa.cuda().requires_grad_(False)
    wref = torch.nn.grad.conv2d_weight(xg, (Cout, Cin, 3, 3), dy.double().cuda(), padding=1)
    wrms = float(wref.pow(2).mean().sqrt())
    we = {}
    for name in ("acvae_conv3x3_wgrad", "acvae_conv3x3_wgrad_wino"):
        dw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
        _lib.call(name, dyd, xd, scd, shd, dw, ws, wsb, N, H, W, Cin, Cout, S())
        torch.cuda.synchronize()
        we[name] = float((dw.double() - wref).abs().max()) / wrms
    print(f"    wgrad max|err|/rms direct {we['acvae_conv3x3_wgrad']:.2e} wino {we['acvae_conv3x3_wgrad_wino']:.2e}")
    g_w = max(g_w, we["acvae_conv3x3_wgrad_wino"])
    print(f"N={N} H={H} W={W} {Cin}->{Cout} act={int(act)}: fwd max|err|/rms direct {e_d:.2e} wino {e_w:.2e}; "
          f"bn mean {e_mean:.1e} invstd {e_istd:.1e}; dgrad direct {g_d:.2e} wino {g_w:.2e}", flush=True)
    return e_w, g_w


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def bench():
    N = 32
    for (H, W, Cin, Cout) in [(1000, 64, 64, 64), (500, 32, 64, 128), (500, 32, 128, 128), (250, 16, 128, 256),
                              (250, 16, 256, 256), (125, 8, 256, 512), (125, 8, 512, 512)]:
        x = torch.randn(N, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
        sc = torch.rand(Cin, device="cuda") + 0.5; sh = torch.randn(Cin, device="cuda") * 0.3
        y = torch.empty(N, H, W, Cout, device="cuda")
        wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
        ws = torch.empty(int(wsb), dtype=torch.uint8, device="cuda")
        gamma = torch.ones(Cout, device="cuda"); beta = torch.zeros(Cout, device="cuda")
        rm = torch.zeros(Cout, device="cuda"); rv = torch.ones(Cout, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda"); bn = torch.empty(4, Cout, device="cuda")
        gf = 2.0 * N * H * W * 9 * Cin * Cout / 1e9
        res = []
        for name in ("acvae_conv3x3_fwd", "acvae_conv3x3_fwd_wino"):
            ms = timeit(lambda: _lib.call(name, x, w, sc, sh, y, gamma, beta, rm, rv, nbt, 1, bn, ws, wsb, N, H, W, Cin,
                                          Cout, S()))
            res.append(ms)
        dyt = torch.randn(N, H, W, Cout, device="cuda")
        dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
        wres = []
        for name in ("acvae_conv3x3_wgrad", "acvae_conv3x3_wgrad_wino"):
            wres.append(timeit(lambda: _lib.call(name, dyt, x, sc, sh, dw, ws, wsb, N, H, W, Cin, Cout, S())))
        print(f"    wgrad direct {wres[0]:.3f} ms ({gf / wres[0]:.1f} TFLOP/s)  wino {wres[1]:.3f} ms ({gf / wres[1]:.1f} algorithmic, "
              f"{gf / 2.25 / wres[1]:.1f} executed)  incl. slab reduce")
        print(f"{H}x{W} {Cin}->{Cout}: direct {res[0]:.3f} ms ({gf / res[0]:.1f} TFLOP/s)  wino {res[1]:.3f} ms "
              f"({gf / res[1]:.1f} algorithmic TFLOP/s, {gf / 2.25 / res[1]:.1f} executed)  incl. weight transform + bn finalize",
              flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what:
        worst = 0.0
        for i, (N, H, W, Cin, Cout) in enumerate([(2, 8, 8, 64, 64), (2, 11, 64, 64, 64), (3, 9, 32, 64, 128),
                                                  (2, 37, 16, 128, 256), (2, 37, 8, 256, 512), (3, 21, 4, 512, 512),
                                                  (1, 250, 16, 128, 128), (2, 125, 8, 64, 64)]):
            for act in (False, True):
                e = check(N, H, W, Cin, Cout, act, seed=i)
                worst = max(worst, *e)
                assert all(x == x for x in e), "NaN in a Winograd result"
        print("worst", worst)
    if "time" in what:
        bench()
