"""Loss modules of the training loop on the HIP path: mirrors of ``utils/train_util.py``
(``LabelSmoothingLoss`` :234-251, ``Normal_kl_loss`` :253-266, length-mask helpers :198-231) and of the
masked dict-style losses of ``losses/loss.py`` (:12-70).  Same constructor arguments and forward
signatures; the arithmetic is in libacvae_hip.so (acvae_ls_ce_*, acvae_gauss_kl_*, acvae_mse_*).
"""
import torch
import torch.nn as nn

from . import _lib


def _dev_scalar(dev):
    return torch.empty(1, device=dev)


class _CEFn(torch.autograd.Function):
    """Label-smoothed CE over rows (n,t) of logits [N,T,V]; row valid iff t < lens1[n] (None: all)."""

    @staticmethod
    def forward(ctx, logits, targets, lens1, smoothing, reduction):
        _lib.require_cuda(logits)
        logits = logits.float()
        if not logits.is_contiguous():   # the backward writes dlogits with the same (contiguous) strides
            logits = logits.contiguous()
        N, T, V = logits.shape
        dev = logits.device
        targets = targets.to(device=dev, dtype=torch.long)
        if targets.stride(-1) != 1:
            targets = targets.contiguous()
        lens_d = None if lens1 is None else torch.as_tensor(lens1).to(device=dev, dtype=torch.long).contiguous()
        lse = torch.empty(N, T, device=dev)
        rows = torch.empty(N, T, device=dev)
        out = _dev_scalar(dev)
        st = _lib.current_stream()
        _lib.call("acvae_row_logsoftmax_argmax", logits, logits.stride(0), logits.stride(1), None, None, lse, T, 1, N,
                  T, V, st)
        _lib.call("acvae_ls_ce_fwd", logits, logits.stride(0), logits.stride(1), targets, targets.stride(0), lens_d,
                  lse, float(smoothing), reduction, rows, out, N, T, V, st)
        ctx.logits, ctx.targets, ctx.lens_d, ctx.lse = logits, targets, lens_d, lse
        ctx.smoothing, ctx.reduction = float(smoothing), reduction
        return rows if reduction == 0 else out[0]

    @staticmethod
    def backward(ctx, g):
        logits = ctx.logits
        N, T, V = logits.shape
        dl = torch.empty(N, T, V, device=logits.device)
        g = g.contiguous().float()
        _lib.call("acvae_ls_ce_bwd", logits, logits.stride(0), logits.stride(1), ctx.targets, ctx.targets.stride(0),
                  ctx.lens_d, ctx.lse, ctx.smoothing, ctx.reduction, None if ctx.reduction == 0 else g.reshape(1),
                  g if ctx.reduction == 0 else None, dl, N, T, V, _lib.current_stream())
        return dl, None, None, None, None


_RED = {"none": 0, "mean": 1, "sum": 2}


def masked_label_smoothing_ce(logits, targets, lens1, smoothing=0.0, reduction="mean"):
    """CE over the unpadded tokens of batch-major logits [N,T,V] without materialising the packed copy."""
    return _CEFn.apply(logits, targets, lens1, smoothing, _RED[reduction])


class LabelSmoothingLoss(nn.Module):
    """utils/train_util.py:234-251 — forward(logit [R,V] packed rows, target [R]) -> mean over rows."""

    def __init__(self, classes, smoothing=0.0, device=0, dim=-1):
        super().__init__()
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.cls = classes
        self.dim = dim
        self.device = device

    def forward(self, logit, target):
        return _CEFn.apply(logit.unsqueeze(1), target.reshape(-1, 1), None, self.smoothing, 1)

    def masked(self, logits, targets, lens1):
        """Same value as packing first (runner :89-95) then forward(): mean over tokens t < lens1[n]."""
        return _CEFn.apply(logits, targets, lens1, self.smoothing, 1)


class MaskedCrossEntropyLoss(nn.Module):
    """losses/loss.py:12-37 — forward({"logits","targets","lens"}), reduction in none|mean|sum."""

    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = reduction
        self.smoothing = 0.0

    def forward(self, output):
        return _CEFn.apply(output["logits"], output["targets"], output["lens"], self.smoothing, _RED[self.reduction])


class MaskedLabelSmoothingLoss(MaskedCrossEntropyLoss):
    """losses/loss.py:39-70."""

    def __init__(self, smoothing=0.0, dim=-1, reduction="mean"):
        super().__init__(reduction)
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.dim = dim


class _KLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu1, lv1, mu2, lv2):
        _lib.require_cuda(mu1, lv1, mu2, lv2)
        ts = [t.contiguous().float() for t in (mu1, lv1, mu2, lv2)]
        E = ts[0].shape[-1]
        rows = ts[0].numel() // E
        dev = ts[0].device
        part = torch.empty(_lib.call("acvae_kl_partials", rows * E), device=dev)
        out = _dev_scalar(dev)
        _lib.call("acvae_gauss_kl_fwd", *ts, part, out, rows, E, _lib.current_stream())
        ctx.ts, ctx.rows, ctx.E = ts, rows, E
        return out[0]

    @staticmethod
    def backward(ctx, g):
        outs = [torch.empty_like(t) if ctx.needs_input_grad[i] else None for i, t in enumerate(ctx.ts)]
        _lib.call("acvae_gauss_kl_bwd", *ctx.ts, g.contiguous().float().reshape(1), *outs, ctx.rows, ctx.E,
                  _lib.current_stream())
        return tuple(outs)


class Normal_kl_loss(nn.Module):
    """utils/train_util.py:253-266 — KL(N(mu1,e^lv1) || N(mu2,e^lv2)), sum over the last dim, mean over
    ALL leading positions (padded ones included)."""

    def __init__(self, device=0, dim=-1):
        super().__init__()
        self.dim = dim
        self.device = device

    def forward(self, mu1, lv1, mu2, lv2):
        return _KLFn.apply(mu1, lv1, mu2, lv2)


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _lib.require_cuda(a, b)
        a, b = a.contiguous().float(), b.contiguous().float()
        part = torch.empty(_lib.call("acvae_kl_partials", a.numel()), device=a.device)
        out = _dev_scalar(a.device)
        _lib.call("acvae_mse_fwd", a, b, part, out, a.numel(), _lib.current_stream())
        ctx.a, ctx.b = a, b
        return out[0]

    @staticmethod
    def backward(ctx, g):
        da = torch.empty_like(ctx.a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(ctx.b) if ctx.needs_input_grad[1] else None
        _lib.call("acvae_mse_bwd", ctx.a, ctx.b, g.contiguous().float().reshape(1), da, db, ctx.a.numel(),
                  _lib.current_stream())
        return da, db


class _CombineLossFn(torch.autograd.Function):
    """loss = ce + w_kl * kl (+ w_mse * mse) on device scalars (runners/pytorch_runner_vae.py:315-320): one launch forward, one
    backward - the same arithmetic as the tensor expression, without its dozen scalar kernels on the step's critical path."""

    @staticmethod
    def forward(ctx, ce, kl, mse, w_kl, w_mse):
        _lib.require_cuda(ce, kl)
        out = _dev_scalar(ce.device)
        _lib.call("acvae_loss_combine_fwd", ce.reshape(1), kl.reshape(1), None if mse is None else mse.reshape(1), float(w_kl),
                  float(w_mse), out, _lib.current_stream())
        ctx.w = (float(w_kl), float(w_mse), mse is not None)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        w_kl, w_mse, has_mse = ctx.w
        gs = torch.empty(3, device=g.device)
        _lib.call("acvae_loss_combine_bwd", g.contiguous().float().reshape(1), w_kl, w_mse, gs[0:1], gs[1:2],
                  gs[2:3] if has_mse else None, _lib.current_stream())
        return gs[0], gs[1], (gs[2] if has_mse else None), None, None


def combine_losses(ce, kl, mse=None, kl_weight=1.0, alpha=0.0):
    """ce + kl_weight * kl + alpha * mse as the runner forms it, fused (device scalars in, device scalar out)."""
    return _CombineLossFn.apply(ce, kl, mse, kl_weight, alpha if mse is not None else 0.0)


class MSELoss(nn.Module):
    """nn.MSELoss() as used for the global constraint (runners/pytorch_runner_vae.py:220,317)."""

    def forward(self, a, b):
        return _MSEFn.apply(a, b)


# ---- length helpers (utils/train_util.py:198-231); host-side index logic, used by callers of the modules
def generate_length_mask(lens):
    lens = torch.as_tensor(lens)
    T = int(lens.max())
    return torch.arange(T).unsqueeze(0) < lens.view(-1, 1)
