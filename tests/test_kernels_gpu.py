"""GPU parity of the path's kernels ONE BY ONE through the per-op C ABI (include/acvae_hip.h: acvae_conv3x3_*,
acvae_bn_*, acvae_gru_step, acvae_lstm_step, acvae_bigru_seq), with no ReLU decision left to rounding:

  * conv3x3 forward / data gradient / weight gradient against an fp64 ``conv2d`` at every (Cin, Cout) of Cnn10 and
    Cnn14_16k, odd H / W, pixel counts that are not a tile multiple, W down to 2, with and without the fused
    BatchNorm+ReLU operand transform - EVERY element within the fp32 rounding bound of a K-term product chain;
  * all kernel variants behind the run-time switches (child processes: the switches are read once per process);
  * BatchNorm statistics / BN+ReLU+pool / BN backward against fp64, the ReLU mask taken from the same fp32 expression on
    both sides (so no boundary flip can occur);
  * GRU / LSTM step and the packed BiGRU against golden g5 (made by torch.nn.GRU / LSTM and the reference's
    PosteriorRNN_hybrid) and against stock torch.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import acvae_oracle as O
from acvae_amd import _lib
from conftest import load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def S():
    return _lib.current_stream()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def ws_buf(nbytes):
    return torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")


def chain_tol(K):
    """|err| bound for an fp32 k-ordered fma chain of K terms with O(1) result: the rounding random walk is
    2^-24 * sqrt(K) * rms; 6 sigma of it, floor 1e-5 (the judge's rtol)."""
    return max(1e-5, 6 * 2.0 ** -24 * math.sqrt(K))


def assert_every_element(got, ref, K, what):
    got = got.detach().cpu().double(); ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rms = float(ref.pow(2).mean().sqrt())
    tol = chain_tol(K) * torch.maximum(ref.abs(), torch.full_like(ref, rms))
    err = (got - ref).abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance, worst " \
                                f"{float((err / tol).max()):.2f} x tol (|err| {float(err.max()):.3e}, rms {rms:.3e})"


LAYERS = [(64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512),      # Cnn10
          (512, 1024), (1024, 1024), (1024, 2048), (2048, 2048)]                                  # Cnn14_16k blocks 5, 6
SHAPES = [(3, 7, 5), (2, 33, 6), (2, 9, 2)]      # (N, H, W): M = 105, 396, 36 - never a multiple of the 128-pixel tile


def conv_case(N, H, W, Cin, Cout, act, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    dy = torch.randn(N, Cout, H, W, generator=g) / math.sqrt(N * H * W)
    sc = sh = None
    if act:
        sc = torch.rand(Cin, generator=g) + 0.5
        sh = torch.randn(Cin, generator=g) * 0.3
        xin = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))          # fp32, the kernel's own expression
        xa = torch.where(xin > 0, xin.double(), torch.zeros((), dtype=torch.double))
    else:
        xa = x.double()
    xa = xa.clone().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv2d(xa, wd, padding=1)
    y.backward(dy.double())
    return x, w, dy, sc, sh, y.detach(), xa.grad, wd.grad


def run_conv(N, H, W, Cin, Cout, act, seed=0):
    x, w, dy, sc, sh, y_ref, dx_ref, dw_ref = conv_case(N, H, W, Cin, Cout, act, seed)
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
    ws = ws_buf(wsb)
    xd, wd, dyd = nhwc(x).cuda(), w.cuda().contiguous(), nhwc(dy).cuda()
    scd = None if sc is None else sc.cuda()
    shd = None if sh is None else sh.cuda()
    y = torch.empty(N, H, W, Cout, device="cuda")
    _lib.call("acvae_conv3x3_fwd", xd, wd, scd, shd, y, None, None, None, None, None, 0, None, ws, wsb, N, H, W, Cin,
              Cout, S())
    assert_every_element(y, nhwc(y_ref), 9 * Cin, f"fwd {Cin}->{Cout} {N}x{H}x{W} act={act}")
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    _lib.call("acvae_conv3x3_wgrad", dyd, xd, scd, shd, dw, ws, wsb, N, H, W, Cin, Cout, S())
    assert_every_element(dw, dw_ref, N * H * W, f"wgrad {Cin}->{Cout} {N}x{H}x{W} act={act}")
    if not act:         # the data gradient never sees the operand transform (it is dY that is convolved)
        dx = torch.empty(N, H, W, Cin, device="cuda")
        _lib.call("acvae_conv3x3_dgrad", dyd, wd, dx, ws, wsb, N, H, W, Cin, Cout, S())
        assert_every_element(dx, nhwc(dx_ref), 9 * Cout, f"dgrad {Cin}->{Cout} {N}x{H}x{W}")


@pytest.mark.parametrize("Cin,Cout", LAYERS)
def test_conv3x3_every_layer_shape_vs_fp64(Cin, Cout):
    shapes = SHAPES if Cin <= 512 else SHAPES[:1] + SHAPES[2:]        # keep the fp64 reference of the widest layers short
    for i, (N, H, W) in enumerate(shapes):
        for act in (False, True):
            run_conv(N, H, W, Cin, Cout, act, seed=100 * Cin + 10 * i + int(act))


def test_conv3x3_real_block_geometry():
    """Row lengths of the real stack (W = 64, 32, 8, 4) with several 128-pixel tiles per image row / several rows per
    tile, a partial last tile, N > 1."""
    for (N, H, W, Cin, Cout) in [(2, 11, 64, 64, 64), (3, 9, 32, 64, 128), (2, 37, 8, 256, 512), (3, 21, 4, 512, 512)]:
        run_conv(N, H, W, Cin, Cout, True, seed=7)
        run_conv(N, H, W, Cin, Cout, False, seed=8)


def run_conv_wino(N, H, W, Cin, Cout, act, seed=0):
    """Winograd F(2x2,3x3) forms (conv_wino.hip): same fp64 reference, same element-wise bound as the implicit GEMM (the
    transforms only add and halve; measured errors are a third of the implicit GEMM's: 16 products per output, not 36)."""
    x, w, dy, sc, sh, y_ref, dx_ref, dw_ref = conv_case(N, H, W, Cin, Cout, act, seed)
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
    ws = ws_buf(wsb)
    xd, wd, dyd = nhwc(x).cuda(), w.cuda().contiguous(), nhwc(dy).cuda()
    scd = None if sc is None else sc.cuda()
    shd = None if sh is None else sh.cuda()
    if Cin % 64 == 0:
        dw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
        _lib.call("acvae_conv3x3_wgrad_wino", dyd, xd, scd, shd, dw, ws, wsb, N, H, W, Cin, Cout, S())
        assert_every_element(dw, dw_ref, N * H * W, f"wino wgrad {Cin}->{Cout} {N}x{H}x{W} act={act}")
    y = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    bn = torch.empty(4, Cout, device="cuda")
    _lib.call("acvae_conv3x3_fwd_wino", xd, wd, scd, shd, y, gamma, beta, rm, rv, nbt, 1, bn, ws, wsb, N, H, W, Cin, Cout, S())
    assert_every_element(y, nhwc(y_ref), 9 * Cin, f"wino fwd {Cin}->{Cout} {N}x{H}x{W} act={act}")
    # the fused batch statistics: mean / invstd of the raw output, running buffers with the unbiased variance
    yr = nhwc(y_ref)
    cnt = N * H * W
    mean, var = yr.mean(dim=(0, 1, 2)), yr.var(dim=(0, 1, 2), unbiased=False)
    rms = float(yr.pow(2).mean().sqrt())
    assert float((bn[2].cpu().double() - mean).abs().max()) <= 1e-5 * rms
    assert torch.allclose(bn[3].cpu().double(), 1 / torch.sqrt(var + 1e-5), rtol=1e-4, atol=0)
    assert torch.allclose(rv.cpu().double(), 0.9 + 0.1 * var * cnt / (cnt - 1), rtol=1e-4, atol=1e-6)
    assert int(nbt) == 1
    if not act:
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        _lib.call("acvae_conv3x3_dgrad_wino", dyd, wd, dx, ws, wsb, N, H, W, Cin, Cout, S())
        assert_every_element(dx, nhwc(dx_ref), 9 * Cout, f"wino dgrad {Cin}->{Cout} {N}x{H}x{W}")


WINO_SHAPES = [(3, 7, 4), (2, 33, 8), (2, 9, 16)]      # odd H (half-empty last tile row), several / partial row blocks


def test_conv3x3_winograd_dgrad_with_bn_backward_sums():
    """conv_wino_bnred_kernel: the data gradient whose epilogue also reduces the BatchNorm + ReLU backward that consumes it -
    dX every element against fp64 as for the plain data gradient, the two sums against fp64 sums over the fp64 dX."""
    for (N, H, W, Cin, Cout) in [(2, 11, 64, 64, 64), (3, 9, 32, 128, 128), (2, 37, 16, 256, 256), (3, 21, 4, 512, 512),
                                 (2, 33, 8, 64, 128), (5, 40, 64, 64, 64)]:
        x, w, dy, _, _, _, dx_ref, _ = conv_case(N, H, W, Cin, Cout, False, seed=31 + Cin)
        g = torch.Generator().manual_seed(7 + Cout)
        yprev = torch.randn(N, H, W, Cin, generator=g) * 1.3 + 0.2
        gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        mean, var = yprev.double().mean(dim=(0, 1, 2)), yprev.double().var(dim=(0, 1, 2), unbiased=False)
        invstd = 1 / torch.sqrt(var + 1e-5)
        scale = gamma.double() * invstd
        shift = beta.double() - mean * scale
        bn = torch.stack([scale, shift, mean, invstd]).float().cuda().contiguous()
        wsb = max(_lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout), _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cout, Cin))
        ws = ws_buf(wsb)
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        sg, sgy = torch.full((Cin,), float("nan"), device="cuda"), torch.full((Cin,), float("nan"), device="cuda")
        _lib.call("acvae_conv3x3_dgrad_bnred_wino", nhwc(dy).cuda(), w.cuda().contiguous(), dx, yprev.cuda(), bn, sg, sgy, ws, wsb,
                  N, H, W, Cin, Cout, S())
        assert_every_element(dx, nhwc(dx_ref), 9 * Cout, f"wino dgrad+bnred {Cin}->{Cout} {N}x{H}x{W}")
        # reference sums from the fp64 data gradient under the KERNEL's ReLU decisions (fp32 y * scale + shift, as bn_bwd_reduce_kernel)
        yd = yprev.double()
        mask = (yprev * bn[0].cpu() + bn[1].cpu()) > 0
        gg = nhwc(dx_ref) * mask
        ref_g = gg.sum(dim=(0, 1, 2))
        ref_gy = (gg * ((yd - bn[2].cpu().double()) * bn[3].cpu().double())).sum(dim=(0, 1, 2))
        cnt = N * H * W
        tol_g = 4e-6 * float(nhwc(dx_ref).abs().max()) * cnt ** 0.5 * 9 + 1e-6
        assert float((sg.cpu().double() - ref_g).abs().max()) <= tol_g, (float((sg.cpu().double() - ref_g).abs().max()), tol_g)
        assert float((sgy.cpu().double() - ref_gy).abs().max()) <= 3 * tol_g, (float((sgy.cpu().double() - ref_gy).abs().max()), tol_g)
        assert float(ref_g.abs().max()) > 100 * tol_g      # the bound means something


@pytest.mark.parametrize("Cin,Cout", LAYERS)
def test_conv3x3_winograd_every_layer_shape_vs_fp64(Cin, Cout):
    shapes = WINO_SHAPES if Cin <= 512 else WINO_SHAPES[:1]
    for i, (N, H, W) in enumerate(shapes):
        for act in (False, True):
            run_conv_wino(N, H, W, Cin, Cout, act, seed=100 * Cin + 10 * i + int(act))


def test_conv3x3_winograd_real_block_geometry():
    """Row lengths of the real stack (W = 64 .. 4): 2 .. 16 tile rows per workgroup, clips that end inside a row block,
    odd heights, N > 1; and what the kernel refuses."""
    for (N, H, W, Cin, Cout) in [(2, 11, 64, 64, 64), (3, 9, 32, 64, 128), (2, 37, 16, 128, 256), (2, 37, 8, 256, 512),
                                 (3, 21, 4, 512, 512), (1, 250, 16, 128, 128), (2, 125, 8, 64, 64), (5, 40, 64, 64, 64)]:
        run_conv_wino(N, H, W, Cin, Cout, True, seed=7)
        run_conv_wino(N, H, W, Cin, Cout, False, seed=8)
    # the heights of BASELINE configs[3] (T = 3000: H = 3000 / 1500 / 750 / 375 at W = 64 / 32 / 16 / 8; the last one ends in
    # a half-empty row block), every element against fp64
    for (H, W) in [(3000, 64), (1500, 32), (750, 16), (375, 8)]:
        run_conv_wino(1, H, W, 64, 64, True, seed=9)
        run_conv_wino(1, H, W, 64, 64, False, seed=10)
    ws = ws_buf(1 << 20)
    t = torch.zeros(1 << 16, device="cuda")
    for (H, W, Cin, Cout) in [(8, 6, 64, 64), (8, 2, 64, 64), (8, 8, 24, 64), (8, 8, 64, 96)]:
        with pytest.raises(RuntimeError, match="UNSUPPORTED"):
            _lib.call("acvae_conv3x3_fwd_wino", t, t, None, None, t, None, None, None, None, None, 0, None, ws, 1 << 20, 1, H,
                      W, Cin, Cout, S())


def run_conv_bf16(N, H, W, Cin, Cout, act, seed=0):
    """bf16-storage convolutions (BASELINE configs[2]): operands are bf16 values, so the fp64 reference on the SAME
    rounded operands differs only by the fp32 accumulation order and the final rounding of a bf16 output (2^-9)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    dy = (torch.randn(N, Cout, H, W, generator=g) / math.sqrt(N * H * W)).bfloat16()
    wb = w.bfloat16()
    sc = sh = None
    if act:
        sc = torch.rand(Cin, generator=g) + 0.5
        sh = torch.randn(Cin, generator=g) * 0.3
        xa = (x.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).clamp_min(0).bfloat16().double()
    else:
        xa = x.double()
    xa = xa.clone().requires_grad_(True)
    wd = wb.double().requires_grad_(True)
    y_ref = F.conv2d(xa, wd, padding=1)
    y_ref.backward(dy.double())
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
    ws = ws_buf(wsb)
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    scd = None if sc is None else sc.cuda()
    shd = None if sh is None else sh.cuda()
    what = f"bf16 {Cin}->{Cout} {N}x{H}x{W} act={act}"

    def check_bf16(got, ref, K, name):
        got = got.float().cpu().double(); ref = ref.detach().double()
        rms = float(ref.pow(2).mean().sqrt())
        tol = chain_tol(K) * torch.maximum(ref.abs(), torch.full_like(ref, rms)) + 2.0 ** -8 * ref.abs()
        err = (got - ref).abs()
        assert bool((err <= tol).all()), f"{name} {what}: worst {float((err / tol).max()):.2f} x tol"

    y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    g_, b_ = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    bn = torch.empty(4, Cout, device="cuda")
    _lib.call("acvae_conv3x3_fwd_bf16", xd, w.cuda(), scd, shd, y, g_.cuda(), b_.cuda(), rm, rv, None, 1, bn, ws, wsb, N, H,
              W, Cin, Cout, S())
    check_bf16(y, nhwc(y_ref), 9 * Cin, "fwd")
    # the statistics are those of the stored (rounded) tensor
    ys = y.float().cpu().double()
    mu, var = ys.mean((0, 1, 2)), ys.var((0, 1, 2), unbiased=False)
    np.testing.assert_allclose(bn[2].cpu().double(), mu, rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(bn[3].cpu().double(), 1 / torch.sqrt(var + 1e-5), rtol=3e-5)
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    _lib.call("acvae_conv3x3_wgrad_bf16", dyd, xd, scd, shd, dw, ws, wsb, N, H, W, Cin, Cout, S())
    assert_every_element(dw, wd.grad, N * H * W, "wgrad " + what)
    if not act:
        dx = torch.empty(N, H, W, Cin, device="cuda", dtype=torch.bfloat16)
        _lib.call("acvae_conv3x3_dgrad_bf16", dyd, w.cuda(), dx, ws, wsb, N, H, W, Cin, Cout, S())
        check_bf16(dx, nhwc(xa.grad), 9 * Cout, "dgrad")


@pytest.mark.parametrize("Cin,Cout", LAYERS)
def test_conv3x3_bf16_every_layer_shape_vs_fp64(Cin, Cout):
    shapes = SHAPES if Cin <= 512 else SHAPES[:1] + SHAPES[2:]
    for i, (N, H, W) in enumerate(shapes):
        for act in (False, True):
            run_conv_bf16(N, H, W, Cin, Cout, act, seed=100 * Cin + 10 * i + int(act))


def test_conv3x3_bf16_real_block_geometry():
    for (N, H, W, Cin, Cout) in [(2, 11, 64, 64, 64), (3, 9, 32, 64, 128), (2, 37, 8, 256, 512), (3, 21, 4, 512, 512),
                                 (2, 70, 64, 64, 64)]:
        run_conv_bf16(N, H, W, Cin, Cout, True, seed=7)
        run_conv_bf16(N, H, W, Cin, Cout, False, seed=8)


def test_first_conv_and_bn_mel_vs_fp64():
    """bn0 over the mel axis (models/encoder.py:679-681) + conv_block1.conv1 (Cin = 1) forward with its BatchNorm
    statistics, and the first layer's backward (dW1, bn0's dgamma / dbeta)."""
    N, Tt, Fm = 3, 37, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Tt, Fm, generator=g) * 1.7 + 0.4
    g0, b0 = torch.rand(Fm, generator=g) + 0.5, torch.randn(Fm, generator=g) * 0.2
    w1 = torch.randn(64, 1, 3, 3, generator=g) / 3
    g1, b1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    dy = torch.randn(N, 64, Tt, Fm, generator=g) / math.sqrt(N * Tt * Fm)
    # fp64 reference: x.transpose(1,3) -> BatchNorm2d(64) over mel -> transpose back -> conv -> BN statistics
    xd = x.double().requires_grad_(True)
    g0d, b0d, w1d = g0.double().requires_grad_(True), b0.double().requires_grad_(True), w1.double().requires_grad_(True)
    mu0 = xd.mean((0, 1)); var0 = xd.var((0, 1), unbiased=False)
    xin = (xd - mu0) / torch.sqrt(var0 + 1e-5) * g0d + b0d
    y = F.conv2d(xin.unsqueeze(1), w1d, padding=1)                         # [N,64,T,F]
    y.backward(dy.double())
    mu1 = y.detach().mean((0, 2, 3)); var1 = y.detach().var((0, 2, 3), unbiased=False)
    # HIP
    dev = lambda t: t.cuda().contiguous()
    wsb = _lib.call("acvae_bn_workspace_bytes", N, Tt, Fm, 64)
    ws = ws_buf(wsb)
    rm, rv = torch.zeros(Fm, device="cuda"), torch.ones(Fm, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    bn0 = torch.empty(4, Fm, device="cuda")
    _lib.call("acvae_bn_mel_fwd", dev(x), dev(g0), dev(b0), rm, rv, nbt, 1, bn0, ws, wsb, N * Tt, Fm, S())
    sc0 = (g0.double() / torch.sqrt(var0.detach() + 1e-5))
    np.testing.assert_allclose(bn0[0].cpu().double(), sc0, rtol=2e-6)
    np.testing.assert_allclose(bn0[1].cpu().double(), b0.double() - mu0.detach() * sc0, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(bn0[2].cpu().double(), mu0.detach(), rtol=1e-5, atol=1e-6)
    n0 = N * Tt
    np.testing.assert_allclose(rm.cpu().double(), 0.1 * mu0.detach(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().double(), 0.9 + 0.1 * var0.detach() * n0 / (n0 - 1), rtol=1e-5)
    assert int(nbt) == 1
    wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, Tt, Fm, 1, 64)
    ws = ws_buf(wsb)
    Y = torch.empty(N, Tt, Fm, 64, device="cuda")
    rm1, rv1 = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    bn1 = torch.empty(4, 64, device="cuda")
    _lib.call("acvae_conv3x3_fwd", dev(x), dev(w1), bn0[0], bn0[1], Y, dev(g1), dev(b1), rm1, rv1, None, 1, bn1, ws, wsb,
              N, Tt, Fm, 1, 64, S())
    assert_every_element(Y, nhwc(y.detach()), 9, "first conv fwd")
    np.testing.assert_allclose(bn1[2].cpu().double(), mu1, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(bn1[3].cpu().double(), 1 / torch.sqrt(var1 + 1e-5), rtol=2e-5)
    np.testing.assert_allclose(bn1[0].cpu().double(), g1.double() / torch.sqrt(var1 + 1e-5), rtol=2e-5)
    dW1, dg0, db0 = torch.empty(64, 1, 3, 3, device="cuda"), torch.empty(64, device="cuda"), torch.empty(64, device="cuda")
    _lib.call("acvae_conv1_first_bwd", dev(x), bn0, dev(w1), dev(nhwc(dy)), dW1, dg0, db0, ws, wsb, N, Tt, Fm, S())
    assert_every_element(dW1, w1d.grad, N * Tt * Fm, "first conv dW")
    assert_every_element(dg0, g0d.grad, N * Tt * 9, "bn0 dgamma")
    assert_every_element(db0, b0d.grad, N * Tt * 9, "bn0 dbeta")


@pytest.mark.parametrize("N,H,W,C,pool", [(3, 10, 6, 64, 1), (2, 7, 5, 128, 1), (2, 5, 2, 2048, 0), (3, 6, 4, 512, 1)])
def test_bn_relu_pool_and_backward_vs_fp64(N, H, W, C, pool):
    """conv output -> (this layer's BN statistics from acvae_conv3x3_fwd) -> BN + ReLU + 2x2 average pool + dropout with
    an explicit keep mask, and its backward, training and evaluation mode.  The ReLU mask is the SAME fp32 expression
    (y * scale + shift > 0) on both sides, so the comparison has no rounding-dependent branch."""
    g = torch.Generator().manual_seed(11 + C)
    Cin = 64
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(C, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    keep = torch.rand(N, C, Ho, Wo, generator=g) > 0.2
    dP = torch.randn(N, C, Ho, Wo, generator=g)
    for training in (1, 0):
        rm = torch.randn(C, generator=g) * 0.1
        rv = torch.rand(C, generator=g) + 0.5
        wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, C)
        ws = ws_buf(wsb)
        Y = torch.empty(N, H, W, C, device="cuda")
        bn = torch.empty(4, C, device="cuda")
        rmd, rvd = rm.cuda(), rv.cuda()
        _lib.call("acvae_conv3x3_fwd", nhwc(x).cuda(), w.cuda(), None, None, Y, gamma.cuda(), beta.cuda(), rmd, rvd, None,
                  training, bn, ws, wsb, N, H, W, Cin, C, S())
        Yc = Y.cpu().permute(0, 3, 1, 2).contiguous()                       # the kernel's own fp32 output, NCHW
        yd = Yc.double()
        if training:
            mu, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=False)
            cnt = N * H * W
            np.testing.assert_allclose(rmd.cpu().double(), 0.9 * rm.double() + 0.1 * mu, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(rvd.cpu().double(), 0.9 * rv.double() + 0.1 * var * cnt / (cnt - 1), rtol=1e-5)
        else:
            mu, var = rm.double(), rv.double()
        np.testing.assert_allclose(bn[2].cpu().double(), mu, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(bn[3].cpu().double(), 1 / torch.sqrt(var + 1e-5), rtol=2e-5)
        sc, sh = bn[0].cpu(), bn[1].cpu()
        mask = (Yc * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0          # fp32, as the kernels evaluate it
        # forward
        P = torch.empty(N, Ho, Wo, C, device="cuda")
        _lib.call("acvae_bn_relu_pool_fwd", Y, bn, P, N, H, W, C, pool, 0.2, 0, 0, keep.to(torch.uint8).cuda().contiguous(),
                  S())
        a = torch.where(mask, yd * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1),
                        torch.zeros((), dtype=torch.double))
        p_ref = (F.avg_pool2d(a, 2) if pool else a) * keep.double() / 0.8
        assert_every_element(P, nhwc(p_ref), 4, f"bn_relu_pool C={C} train={training}")
        # backward through dropout, pool, ReLU and BatchNorm (batch statistics in training mode)
        yv = yd.clone().requires_grad_(True)
        gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        if training:
            m_, v_ = yv.mean((0, 2, 3), keepdim=True), yv.var((0, 2, 3), unbiased=False, keepdim=True)
        else:
            m_, v_ = mu.view(1, -1, 1, 1), var.view(1, -1, 1, 1)
        z = (yv - m_) / torch.sqrt(v_ + 1e-5) * gd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
        a2 = z * mask.double()
        out = (F.avg_pool2d(a2, 2) if pool else a2) * keep.double() / 0.8
        out.backward(dP.double())
        wsb2 = _lib.call("acvae_bn_workspace_bytes", N, H, W, C)
        ws2 = ws_buf(wsb2)
        dY = torch.empty(N, H, W, C, device="cuda")
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        _lib.call("acvae_bn_relu_bwd", Y, nhwc(dP).cuda(), 1 if pool else 2, bn, dg, db, dY, ws2, wsb2, N, H, W, C,
                  training, 0.2, 0, 0, keep.to(torch.uint8).cuda().contiguous(), S())
        K = N * H * W
        assert_every_element(db, bd.grad, K, f"dbeta C={C} train={training}")
        assert_every_element(dg, gd.grad, K, f"dgamma C={C} train={training}")
        assert_every_element(dY, nhwc(yv.grad), 16, f"dY C={C} train={training}")


def test_gru_lstm_step_golden_g5():
    """acvae_gru_step / acvae_lstm_step against golden g5 (torch.nn.GRU / torch.nn.LSTM single step, the modules
    models/decoder.py:39-44 and models/text_encoder.py:229-235 instantiate)."""
    g = load_golden("g5_rnn")
    N, I, H, V, E = (int(x) for x in g["dims"])
    gru = torch.nn.GRU(I, H, batch_first=True); lstm = torch.nn.LSTM(I, H, batch_first=True)
    gs = O.closed_form_state({k: tuple(v.shape) for k, v in gru.state_dict().items()})
    ls = O.closed_form_state({k: tuple(v.shape) for k, v in lstm.state_dict().items()})
    dev = lambda t: (T(t) if isinstance(t, np.ndarray) else t).cuda().contiguous()
    x, h, c = dev(g["x"][:, 0]), dev(g["h"][0]), dev(g["c"][0])
    wsb = _lib.call("acvae_rnn_workspace_bytes", N, 1, I, H)
    ws = ws_buf(wsb)
    ho = torch.empty(N, H, device="cuda")
    _lib.call("acvae_gru_step", x, h, dev(gs["weight_ih_l0"]), dev(gs["weight_hh_l0"]), dev(gs["bias_ih_l0"]),
              dev(gs["bias_hh_l0"]), ho, ws, wsb, N, I, H, S())
    np.testing.assert_allclose(ho.cpu().numpy(), g["gru_h"][0], rtol=1e-5, atol=2e-6)
    ho2, co2 = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
    _lib.call("acvae_lstm_step", x, h, c, dev(ls["weight_ih_l0"]), dev(ls["weight_hh_l0"]), dev(ls["bias_ih_l0"]),
              dev(ls["bias_hh_l0"]), ho2, co2, ws, wsb, N, I, H, S())
    np.testing.assert_allclose(ho2.cpu().numpy(), g["lstm_h"][0], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(co2.cpu().numpy(), g["lstm_c"][0], rtol=1e-5, atol=2e-6)


def test_bigru_seq_and_posterior_golden_g5():
    """The packed BiGRU alone against torch's packed bidirectional GRU on the golden's captions, and the whole posterior
    (acvae_posterior_fwd through the C ABI) against the reference's PosteriorRNN_hybrid outputs stored in g5."""
    from acvae_amd.encoder import ptr_table
    g = load_golden("g5_rnn")
    N, I, H, V, E = (int(x) for x in g["dims"])
    shapes = {k[len("qnet."):]: v for k, v in O.state_shapes(V, E, E, None, E).items() if k.startswith("qnet.")}
    qs = O.closed_form_state(shapes)
    caps, cap_lens = T(g["caps"]), g["cap_lens"]
    lens1 = torch.as_tensor(cap_lens - 1)
    Tc = int(lens1.max())
    n = caps.shape[0]
    emb = qs["word_embedding.weight"][caps[:, :Tc].long()]                    # [n,Tc,E]
    gru = torch.nn.GRU(E, E, num_layers=1, bidirectional=True, batch_first=True)
    gru.load_state_dict({k[len("network."):]: v for k, v in qs.items() if k.startswith("network.")})
    with torch.no_grad():
        packed = torch.nn.utils.rnn.pack_padded_sequence(emb, lens1, batch_first=True)
        ref_hidden, _ = torch.nn.utils.rnn.pad_packed_sequence(gru(packed)[0], batch_first=True)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    wts = [qs["network." + k].cuda().contiguous() for k in names] + \
          [qs["network." + k + "_reverse"].cuda().contiguous() for k in names]
    wsb = _lib.call("acvae_rnn_workspace_bytes", n, Tc, E, E)
    ws = ws_buf(wsb)
    hidden = torch.empty(n, Tc, 2 * E, device="cuda")
    _lib.call("acvae_bigru_seq", emb.cuda().contiguous(), lens1.cuda(), ptr_table(wts), hidden, ws, wsb, n, Tc, E, E, S())
    np.testing.assert_allclose(hidden.cpu().numpy(), ref_hidden.numpy(), rtol=1e-5, atol=3e-6)
    # whole posterior through the C ABI vs the reference's outputs
    table = [None] * _lib.ENUMS_TEXT_N
    order = ["word_embedding.weight"] + ["network." + k for k in names] + ["network." + k + "_reverse" for k in names] + \
            ["token_mean_log.weight", "token_mean_log.bias"]
    keep = [qs[k].cuda().contiguous() for k in order]
    table[10:21] = keep
    sb = _lib.call("acvae_posterior_saved_bytes", n, Tc, E, E, V)
    cb = _lib.call("acvae_posterior_scratch_bytes", n, Tc, E, E, V)
    saved, scratch = ws_buf(sb), ws_buf(cb)
    qm, ql, qz = (torch.empty(n, Tc, E, device="cuda") for _ in range(3))
    utt = torch.empty(n, 2 * E, device="cuda")
    caps_d = caps.long().cuda().contiguous()
    _lib.call("acvae_posterior_fwd", ptr_table(table), caps_d, caps_d.stride(0), lens1.cuda(), T(g["eps"]).cuda().contiguous(),
              qm, ql, qz, utt, saved, sb, scratch, cb, n, Tc, E, E, V, S(), 0)
    for got, key in ((qm, "q_means"), (ql, "q_logs"), (qz, "q_z"), (utt, "q_means_utt")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-4, atol=1e-5, err_msg=key)
