"""GPU parity of the individual HIP ops, through the C ABI, against the oracle and the golden vectors
generated from the reference (tests/golden/g1..g5).  fp32 tolerances: rtol 1e-4 / atol 1e-5 where the
summation order differs from the reference (hoisted attention projections, MFMA accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import acvae_oracle as O
from acvae_amd import _lib
from conftest import load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev(x):
    return (T(x) if isinstance(x, np.ndarray) else x).cuda().contiguous()


def close(a, b, rtol=1e-4, atol=1e-5):
    a = torch.as_tensor(a).detach().cpu().double(); b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    ok = err <= atol + rtol * b.abs()
    assert bool(ok.all()), f"max abs err {float(err.max()):.3e}, {int((~ok).sum())}/{ok.numel()} out of tolerance"


def S():
    return _lib.current_stream()


# ------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize("M,N,K", [(32, 2048, 1024), (32, 512, 512), (7, 50, 64), (672, 5000, 512), (1984, 512, 512),
                                   (300, 64, 192), (129, 130, 36), (96, 52, 50), (1000, 257, 31)])
def test_gemm_nt(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g); B = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g)
    ref = (A.double() @ B.double().T + bias.double()).float()
    a, b, bi = dev(A), dev(B), dev(bias)
    c = torch.zeros(M, N, device="cuda")
    _lib.call("acvae_gemm_nt", a, K, b, K, bi, c, N, M, N, K, 0, S())
    close(c, ref, 1e-4, 1e-4)
    _lib.call("acvae_gemm_nt", a, K, b, K, None, c, N, M, N, K, 1, S())     # accumulate, no bias
    close(c, ref + (A.double() @ B.double().T).float(), 1e-4, 2e-4)


def test_gemm_nt_strided_rows():
    # decode-step addressing: rows are column t of batch-major [N,Tc,*] buffers
    g = torch.Generator().manual_seed(3)
    N_, Tc, K, Nout = 32, 5, 128, 96
    X = torch.randn(N_, Tc, K, generator=g); W = torch.randn(Nout, K, generator=g)
    x, w = dev(X), dev(W)
    out = torch.zeros(N_, Tc, Nout, device="cuda")
    t = 3
    _lib.call("acvae_gemm_nt", x.data_ptr() + t * K * 4, Tc * K, w, K, None, out.data_ptr() + t * Nout * 4, Tc * Nout,
              N_, Nout, K, 0, S())
    close(out[:, t], X[:, t] @ W.T, 1e-4, 1e-4)
    assert float(out[:, :t].abs().max()) == 0.0 and float(out[:, t + 1:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(5000, 512, 672), (512, 1536, 672), (64, 576, 4096), (128, 128, 100000),
                                   (50, 64, 28), (130, 66, 333)])
def test_gemm_tn(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g); B = torch.randn(K, N, generator=g)
    ref = (A.double().T @ B.double()).float()
    a, b = dev(A), dev(B)
    wsb = _lib.call("acvae_gemm_tn_workspace_bytes", M, N, K)
    ws = torch.empty(max(wsb, 4) // 4, device="cuda")
    c = torch.full((M, N), 7.0, device="cuda")
    _lib.call("acvae_gemm_tn", a, M, b, N, c, N, M, N, K, 0, ws, wsb, S())
    tol = 1e-4 * max(1.0, (K / 1000.0) ** 0.5)
    close(c, ref, 1e-4, tol * 10)
    _lib.call("acvae_gemm_tn", a, M, b, N, c, N, M, N, K, 1, ws, wsb, S())
    close(c, 2 * ref, 1e-4, tol * 20)


def test_transpose():
    x = torch.randn(77, 130)
    o = torch.zeros(130, 77, device="cuda")
    _lib.call("acvae_transpose", dev(x), 130, o, 77, 77, 130, S())
    assert torch.equal(o.cpu(), x.T)


# ------------------------------------------------------------------ G1 attention
def _attn_hip(st, h_dec, h_enc, lens):
    N, S_, E = h_enc.shape
    Hd = h_dec.shape[1]
    W, b, v = st["a.h2attn.weight"], st["a.h2attn.bias"], st["a.v"]
    A = W.shape[0]
    wd, we = dev(W[:, :Hd]), dev(W[:, Hd:])
    enc = dev(h_enc)
    encproj = torch.empty(N, S_, A, device="cuda")
    _lib.call("acvae_gemm_nt", enc, E, we, E, dev(b), encproj, A, N * S_, A, E, 0, S())
    qproj = torch.empty(N, A, device="cuda")
    _lib.call("acvae_gemm_nt", dev(h_dec), Hd, wd, Hd, None, qproj, A, N, A, Hd, 0, S())
    ctx = torch.empty(N, E, device="cuda"); w = torch.empty(N, S_, device="cuda")
    _lib.call("acvae_attn_fwd", qproj, A, 0, encproj, enc, dev(lens), dev(v), ctx, E, 0, w, S_, 0, N, 1, S_, A, E, None, 0,
              S(), 0)
    return ctx, w, (qproj, encproj, enc, dev(v))


def test_g1_attention_golden():
    g = load_golden("g1_attention")
    for ci in range(int(g["ncases"])):
        N, S_, E, Hd, A = (int(x) for x in g[f"c{ci}_dims"])
        st = {"a." + k: v for k, v in O.closed_form_state({"v": (A,), "h2attn.weight": (A, E + Hd),
                                                           "h2attn.bias": (A,)}).items()}
        ctx, w, _ = _attn_hip(st, T(g[f"c{ci}_h_dec"]), T(g[f"c{ci}_h_enc"]), T(g[f"c{ci}_lens"]))
        close(ctx, g[f"c{ci}_ctx"]); close(w, g[f"c{ci}_weights"])


def test_attention_backward_vs_oracle_autograd():
    g = torch.Generator().manual_seed(9)
    N, Tq, S_, E, Hd, A = 3, 4, 13, 64, 48, 40
    st = {"a." + k: v for k, v in O.closed_form_state({"v": (A,), "h2attn.weight": (A, E + Hd),
                                                       "h2attn.bias": (A,)}).items()}
    h_enc = torch.randn(N, S_, E, generator=g, requires_grad=True)
    lens = torch.tensor([13, 5, 1])
    W, b, v = st["a.h2attn.weight"], st["a.h2attn.bias"], st["a.v"].clone().requires_grad_(True)
    Q = torch.randn(N, Tq, A, generator=g, requires_grad=True)           # projected queries
    P = (h_enc @ W[:, Hd:].T + b)                                         # encproj
    P.retain_grad()
    score = (torch.tanh(Q.unsqueeze(2) + P.unsqueeze(1)) @ v)             # [N,Tq,S]
    mask = torch.arange(S_).view(1, 1, -1) < lens.view(-1, 1, 1)
    wts = torch.softmax(score.masked_fill(~mask, -1e10), -1)
    ctx = wts @ h_enc
    dctx = torch.randn(N, Tq, E, generator=g)
    # gradient wrt enc only through the ctx sum (encproj grad reported separately)
    (ctx * dctx).sum().backward()
    denc_direct = (wts.detach().transpose(1, 2) @ dctx)                   # [N,S,E]
    q, p, e = dev(Q.detach()), dev(P.detach()), dev(h_enc.detach())
    c = torch.empty(N, Tq, E, device="cuda"); w = torch.empty(N, Tq, S_, device="cuda")
    _lib.call("acvae_attn_fwd", q, Tq * A, A, p, e, dev(lens), dev(v.detach()), c, Tq * E, E, w, Tq * S_, S_, N, Tq,
              S_, A, E, None, 0, S(), 0)
    close(c, ctx.detach()); close(w, wts.detach())
    dq = torch.empty(N, Tq, A, device="cuda")
    dP = torch.zeros(N, S_, A, device="cuda"); dH = torch.zeros(N, S_, E, device="cuda")
    dv = torch.zeros(N, A, device="cuda")
    wsb = _lib.call("acvae_attn_bwd_workspace_bytes", N, Tq, S_, A)
    ws = torch.empty(wsb // 4, device="cuda")
    _lib.call("acvae_attn_bwd", dev(dctx), Tq * E, E, q, Tq * A, A, p, e, dev(lens), dev(v.detach()), w, Tq * S_, S_,
              dq, Tq * A, A, dP, dH, dv, ws, wsb, N, Tq, S_, A, E, S())
    close(dq, Q.grad, 1e-4, 1e-5); close(dP, P.grad, 1e-4, 1e-5)
    close(dH, denc_direct, 1e-4, 1e-5); close(dv.sum(0), v.grad, 1e-4, 1e-5)


# ------------------------------------------------------------------ G2 reparam + KL, G3 CE, argmax
def test_g2_reparam_kl_golden():
    g = load_golden("g2_reparam_kl")
    N, Tt, E = g["mu_q"].shape
    rows = N * Tt
    ml = dev(torch.cat([T(g["mu_q"]), T(g["lv_q"])], -1).reshape(rows, 2 * E))
    mean = torch.empty(rows, E, device="cuda"); logv = torch.empty_like(mean); z = torch.empty_like(mean)
    z2 = torch.zeros(rows, 3 * E, device="cuda")
    _lib.call("acvae_reparam_fwd", ml, 2 * E, dev(g["eps"]), E, mean, logv, z, E, z2.data_ptr() + 2 * E * 4, 3 * E,
              rows, E, S())
    close(z.view(N, Tt, E), g["z"], 1e-6, 1e-6); close(z2[:, 2 * E:], z, 0, 0); close(mean.view(N, Tt, E), g["mu_q"], 0, 0)
    npart = _lib.call("acvae_kl_partials", rows * E)
    part = torch.empty(npart, device="cuda"); out = torch.empty(1, device="cuda")
    mu_q, lv_q, mu_p, lv_p = (dev(g[k]) for k in ("mu_q", "lv_q", "mu_p", "lv_p"))
    _lib.call("acvae_gauss_kl_fwd", mu_q, lv_q, mu_p, lv_p, part, out, rows, E, S())
    close(out[0], g["kl"], 1e-6, 1e-5)
    # backward vs autograd of the oracle restatement
    ts = [T(g[k]).clone().requires_grad_(True) for k in ("mu_q", "lv_q", "mu_p", "lv_p")]
    (O.normal_kl_loss(*ts) * 0.5).backward()
    outs = [torch.empty(N, Tt, E, device="cuda") for _ in range(4)]
    _lib.call("acvae_gauss_kl_bwd", mu_q, lv_q, mu_p, lv_p, torch.tensor([0.5], device="cuda"), *outs, rows, E, S())
    for o, t in zip(outs, ts):
        close(o, t.grad, 1e-5, 1e-7)
    # reparam backward
    dz = torch.randn(rows, E); dm = torch.randn(rows, E); dl = torch.randn(rows, E)
    mlr = torch.cat([T(g["mu_q"]), T(g["lv_q"])], -1).reshape(rows, 2 * E).clone().requires_grad_(True)
    zz = T(g["eps"]).reshape(rows, E) * torch.exp(.5 * mlr[:, E:]) + mlr[:, :E]
    ((zz * dz).sum() + (mlr[:, :E] * dm).sum() + (mlr[:, E:] * dl).sum()).backward()
    dml = torch.empty(rows, 2 * E, device="cuda")
    _lib.call("acvae_reparam_bwd", dev(dz), E, dev(dm), dev(dl), E, logv, E, dev(g["eps"]), E, dml, 2 * E, rows, E, S())
    close(dml, mlr.grad, 1e-5, 1e-6)


def test_g3_ce_golden_and_backward():
    g = load_golden("g3_ce")
    logits, targets, lens1 = T(g["logits"]), T(g["targets"]), T(g["lens1"])
    N, Tt, V = logits.shape
    lg, tg, l1 = dev(logits), dev(targets), dev(lens1)
    am = torch.empty(N, Tt, dtype=torch.long, device="cuda"); mlp = torch.empty(N, Tt, device="cuda")
    lse = torch.empty(N, Tt, device="cuda")
    _lib.call("acvae_row_logsoftmax_argmax", lg, Tt * V, V, am, mlp, lse, Tt, 1, N, Tt, V, S())
    lp = torch.log_softmax(logits, -1)
    assert torch.equal(am.cpu(), lp.argmax(-1)); close(mlp, lp.max(-1).values, 1e-5, 1e-6)
    rows = torch.empty(N, Tt, device="cuda"); out = torch.empty(1, device="cuda")
    for smooth, key in ((0.1, "ls_packed"), (0.0, "ce_packed")):
        _lib.call("acvae_ls_ce_fwd", lg, Tt * V, V, tg, Tt, l1, lse, smooth, 1, rows, out, N, Tt, V, S())
        close(out[0], g[key], 1e-6, 1e-5)
    _lib.call("acvae_ls_ce_fwd", lg, Tt * V, V, tg, Tt, l1, lse, 0.1, 1, rows, out, N, Tt, V, S())
    close(out[0], g["ls_packed"], 1e-6, 1e-5)       # masked mean == packed mean (A13 vs A8)
    lr = logits.clone().requires_grad_(True)
    (O.masked_ce(lr, targets, g["lens1"], 0.1, "mean") * 1.7).backward()
    dl = torch.empty(N, Tt, V, device="cuda")
    _lib.call("acvae_ls_ce_bwd", lg, Tt * V, V, tg, Tt, l1, lse, 0.1, 1, torch.tensor([1.7], device="cuda"), None, dl,
              N, Tt, V, S())
    close(dl, lr.grad, 1e-4, 1e-7)
    mse_check(out)


def test_g3b_masked_losses_of_losses_py_golden():
    """A13 (losses/loss.py:12-70): MaskedCrossEntropyLoss / MaskedLabelSmoothingLoss on the HIP path against the
    reference's own classes (golden g3b: values for none / mean / sum and d loss / d logits for mean / sum)."""
    from acvae_amd.train_util import MaskedCrossEntropyLoss, MaskedLabelSmoothingLoss
    g = load_golden("g3b_masked_losses")
    targets, lens = dev(g["targets"]), g["lens"]
    for name, mk in (("ce", lambda red: MaskedCrossEntropyLoss(reduction=red)),
                     ("ls", lambda red: MaskedLabelSmoothingLoss(smoothing=0.1, reduction=red))):
        for red in ("none", "mean", "sum"):
            x = dev(g["logits"]).requires_grad_(True)
            val = mk(red)({"logits": x, "targets": targets, "lens": lens})
            close(val, g[f"{name}_{red}"], 1e-5, 1e-5)
            if red != "none":
                val.backward()
                close(x.grad, g[f"{name}_{red}_dlogits"], 1e-4, 1e-7)
            else:                                            # row-wise upstream gradient through reduction="none"
                w = torch.linspace(0.5, 1.5, val.numel()).reshape(val.shape)
                (val * w.cuda()).sum().backward()
                xr = T(g["logits"]).clone().requires_grad_(True)
                (O.masked_ce(xr, T(g["targets"]), lens, 0.0 if name == "ce" else 0.1, "none") * w).sum().backward()
                close(x.grad, xr.grad, 1e-4, 1e-7)


def mse_check(out):
    a, b = torch.randn(33, 70), torch.randn(33, 70)
    part = torch.empty(_lib.call("acvae_kl_partials", a.numel()), device="cuda")
    _lib.call("acvae_mse_fwd", dev(a), dev(b), part, out, a.numel(), S())
    close(out[0], F.mse_loss(a, b), 1e-6, 1e-6)
    da = torch.empty(33, 70, device="cuda"); db = torch.empty(33, 70, device="cuda")
    _lib.call("acvae_mse_bwd", dev(a), dev(b), torch.tensor([2.0], device="cuda"), da, db, a.numel(), S())
    close(da, 2.0 * 2 * (a - b) / a.numel(), 1e-5, 1e-8); close(db, -da.cpu(), 0, 0)


@pytest.mark.parametrize("N,Tq,S_,A,E", [(16, 1, 187, 512, 512), (5, 3, 33, 64, 128), (3, 1, 17, 64, 64), (32, 2, 62, 512, 512),
                                         (2, 1, 300, 96, 256)])
def test_attention_split_over_frames_matches_one_workgroup_per_row(N, Tq, S_, A, E):
    """acvae_attn_fwd with few query rows splits a row's frames over workgroups and combines the per-split softmax pieces
    (models/attn_model.py:29-45 is one softmax over S) when the caller hands over a workspace: against the one-workgroup-per-
    row kernel (no workspace / ACVAE_FLAG_NO_ATTN_SPLIT) on ragged lengths - weights and context to fp32 summation-order
    tolerance, masked frames exactly 0, rows summing to 1; twice in a row on one workspace (its counters reset themselves) and
    on two streams at once with a workspace each.  The workspace is the CALLER's: the library allocates nothing."""
    g = torch.Generator().manual_seed(N * 1000 + S_)
    f = lambda *s: torch.randn(*s, generator=g).cuda()
    q, p, e, v = f(N, Tq, A), f(N, S_, A), f(N, S_, E), f(A)
    lens = torch.randint(1, S_ + 1, (N,), generator=g); lens[0] = S_
    if N > 2:
        lens[1] = 1
    lens_d = lens.cuda()
    wsb = _lib.call("acvae_attn_fwd_workspace_bytes", N, Tq, S_, A, E)
    assert wsb > 1024
    ws_a = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    ws_b = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    def run(ws, stream=None, flags=0):
        c, w = torch.empty(N, Tq, E, device="cuda"), torch.empty(N, Tq, S_, device="cuda")
        st = S() if stream is None else stream.cuda_stream
        _lib.call("acvae_attn_fwd", q, Tq * A, A, p, e, lens_d, v, c, Tq * E, E, w, Tq * S_, S_, N, Tq, S_, A, E, ws,
                  0 if ws is None else ws.numel(), st, flags)
        return c, w
    c0, w0 = run(None)
    c0b, w0b = run(ws_a, flags=_lib.FLAG_NO_ATTN_SPLIT)
    assert torch.equal(c0, c0b) and torch.equal(w0, w0b)
    with pytest.raises(RuntimeError, match="EWORKSPACE"):
        _lib.call("acvae_attn_fwd", q, Tq * A, A, p, e, lens_d, v, c0, Tq * E, E, w0, Tq * S_, S_, N, Tq, S_, A, E, ws_a, 1024,
                  S(), 0)
    c1, w1 = run(ws_a)
    c2, w2 = run(ws_a)
    torch.cuda.synchronize()
    assert int(ws_a[:1024].view(torch.int32).abs().max()) == 0        # the counters are zero again
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    c3, w3 = run(ws_b, s2)
    c4, w4 = run(ws_a)
    torch.cuda.synchronize()
    for c, w in ((c1, w1), (c2, w2), (c3, w3), (c4, w4)):
        assert torch.equal(c, c1) and torch.equal(w, w1)              # the same arithmetic whichever workgroup combines
        torch.testing.assert_close(w, w0, rtol=2e-5, atol=1e-7)
        torch.testing.assert_close(c, c0, rtol=1e-4, atol=2e-6)
        for n in range(N):
            assert float(w[n, :, int(lens[n]):].abs().max()) == 0.0 if int(lens[n]) < S_ else True
        torch.testing.assert_close(w.sum(-1), torch.ones(N, Tq, device="cuda"), rtol=0, atol=1e-5)


def test_tanh_of_the_attention_scores_elementwise():
    """tanh_att (csrc/common.h: 1 - 2 / (e^{2x} + 1) on the hardware exp2 / rcp) against tanh in fp64, element by element,
    over [-20, 20] and around zero (|x| < 1e-4, where the form cancels and its error is absolute, not relative): bound
    2.5e-7 absolute (measured 2.1e-7: one rounding each of the argument product, exp2, rcp and the fma; a library built with ACVAE_EXACT_TANH=1 uses tanhf and passes with a relative bound too).
    models/attn_model.py:33 applies tanh to every score term."""
    xs = torch.cat([torch.linspace(-20, 20, 400001), torch.linspace(-1e-4, 1e-4, 20001), torch.tensor([0.0, 44.0, -44.0, 90.0, -90.0]),
                    torch.randn(100000, generator=torch.Generator().manual_seed(3)) * 3]).float()
    y = torch.empty_like(xs, device="cuda")
    _lib.call("acvae_tanh_att", xs.cuda(), y, xs.numel(), S())
    ref = torch.tanh(xs.double())
    err = (y.cpu().double() - ref).abs()
    assert float(err.max()) <= 2.5e-7, float(err.max())
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
    small = xs.abs() < 1e-4
    assert float(err[small].max()) <= 2e-7, float(err[small].max())     # exp2's and rcp's ulp each weigh 6e-8 at e = 1


def test_loss_assembly_equals_the_tensor_expression():
    """combine_losses(ce, kl, mse, kl_weight, alpha) = ce + kl_weight * kl + alpha * mse as the runner writes it
    (runners/pytorch_runner_vae.py:315-320): the same fp32 value bit for bit, and the gradients (g, g kl_weight, g alpha);
    mse=None leaves the last term out."""
    from acvae_amd.train_util import combine_losses
    g = torch.Generator().manual_seed(11)
    for kl_w, alpha, with_mse in ((0.5, 1.0, True), (0.37, 2.5, True), (1.0, 0.0, False)):
        vals = [torch.randn((), generator=g).mul(10).cuda().requires_grad_(True) for _ in range(3)]
        ce, kl, mse = vals
        out = combine_losses(ce, kl, mse if with_mse else None, kl_w, alpha)
        ref = ce.detach() + kl_w * kl.detach()
        if with_mse:
            ref = ref + alpha * mse.detach()
        assert torch.equal(out.detach(), ref)
        (out * 3.0).backward()
        assert float(ce.grad) == 3.0 and float(kl.grad) == np.float32(3.0) * np.float32(kl_w)
        if with_mse:
            assert float(mse.grad) == np.float32(3.0) * np.float32(alpha)
        else:
            assert mse.grad is None
