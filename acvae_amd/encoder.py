"""Audio encoder: mirror of the reference's ``models/encoder.py`` ``Cnn10`` (:651-707) and
``ConvBlock`` (:606-649) on the HIP path.

Same constructor ``Cnn10(inputdim, embed_size, **kwargs)``, same ``forward(input, lens)`` returning
``{"audio_embeds", "audio_embeds_pooled", "state", "audio_embeds_lens"}``, same state-dict names and
shapes (so reference checkpoints load), same initialisation (``init_layer`` / ``init_bn`` :593-604).
The torch.nn sub-modules are parameter containers only — their ``forward`` is never called; all
arithmetic runs in libacvae_hip.so (``acvae_encoder_fwd`` / ``acvae_encoder_bwd``).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib

_SCRATCH = {}
_BLOCK_HOOK = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p)   # acvae_encoder_bwd_hooked's block_done


def scratch_buffer(nbytes, device, tag=None):
    """Grow-only scratch per (device, stream): contents are dead between library calls, and calls on one stream
    are ordered, so one buffer per stream is enough (the posterior runs on a side stream beside the encoder).  A call
    that leaves work behind on a second stream (acvae_decode_bwd) takes its own buffer (``tag``)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        _SCRATCH[key] = buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return buf


def ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def init_layer(layer):
    """models/encoder.py:593-599"""
    nn.init.xavier_uniform_(layer.weight)
    if getattr(layer, "bias", None) is not None:
        layer.bias.data.fill_(0.)


def init_bn(bn):
    """models/encoder.py:601-604"""
    bn.bias.data.fill_(0.)
    bn.weight.data.fill_(1.)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        init_layer(self.conv1); init_layer(self.conv2); init_bn(self.bn1); init_bn(self.bn2)

    def forward(self, *a, **k):
        raise RuntimeError("ConvBlock is a parameter container on the HIP path; call the encoder's forward")


def _bn_tensors(bn):
    return [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]


class _Cnn10Fn(torch.autograd.Function):
    """One autograd node for the whole encoder: forward and backward are one library call each."""

    @staticmethod
    def forward(ctx, mod, feats, *weights):
        _lib.require_cuda(feats)
        feats = feats.contiguous().float()
        N, T, F = feats.shape
        tensors = mod._param_table()
        dev = feats.device
        arch = mod._arch()
        saved_b = _lib.call("acvae_encoder_saved_bytes", arch, N, T, F)
        scratch_b = _lib.call("acvae_encoder_scratch_bytes", arch, N, T, F)
        if saved_b < 0:
            raise RuntimeError(f"{type(mod).__name__}: unsupported input shape {tuple(feats.shape)} "
                               f"(need F=64, T>={mod.TIME_DIV})")
        saved = torch.empty(saved_b, dtype=torch.uint8, device=dev)
        scratch = scratch_buffer(scratch_b, dev)
        S, C = T // mod.TIME_DIV, mod.OUT_CHANNELS
        ae = torch.empty(N, S, C, device=dev)
        pooled = torch.empty(N, C, device=dev)
        training = bool(mod.training)
        masks = None
        if training and mod.dropout_masks is not None:
            masks = [m.to(device=dev, dtype=torch.uint8).contiguous() for m in mod.dropout_masks]
            if len(masks) != mod.N_BLOCKS + 2:
                raise ValueError(f"dropout_masks must hold the {mod.N_BLOCKS + 2} masks of the forward in call order")
        seed = mod._next_seed() if training else 0
        mt = ptr_table(masks) if masks is not None else None
        _lib.call("acvae_encoder_fwd", ptr_table(tensors), feats, ae, pooled, saved, saved_b, scratch, scratch_b, arch,
                  N, T, F, int(training), float(mod.p_block), float(mod.p_fc), seed, mt, _lib.current_stream())
        ctx.mod, ctx.feats, ctx.saved, ctx.masks, ctx.seed, ctx.training = mod, feats, saved, masks, seed, training
        ctx.arch = arch
        if mod.keep_saved:                       # test aid: relu_masks() reads the decisions out of this buffer
            mod._last_saved = (saved, arch, N, T, F)
        ctx.mark_non_differentiable(pooled)
        return ae, pooled

    @staticmethod
    def backward(ctx, d_ae, _d_pooled):
        mod, feats = ctx.mod, ctx.feats
        N, T, F = feats.shape
        tensors = mod._param_table()
        grads = [None] * len(tensors)
        outs = []
        for i, t in enumerate(tensors):
            if t.dtype.is_floating_point and t.requires_grad and i < len(tensors) - 2:
                grads[i] = mod._grad_buffer(t)
        d_ae = d_ae.contiguous().float()
        scratch_b = _lib.call("acvae_encoder_scratch_bytes", ctx.arch, N, T, F)
        scratch = scratch_buffer(scratch_b, feats.device)
        mt = ptr_table(ctx.masks) if ctx.masks is not None else None
        hook, failed = None, []
        if mod._grad_ready_cb is not None:
            # host callback after each ConvBlock's gradient kernels are queued: the data-parallel exchange starts the
            # all-reduce of the deepest block's bucket while the shallower blocks still run
            def block_done(block, _user):
                try:
                    mod._grad_ready_cb(("encoder_block", int(block)))
                except BaseException as exc:          # an exception cannot cross the C frame: re-raised below
                    failed.append(exc)
            hook = _BLOCK_HOOK(block_done)
        _lib.call("acvae_encoder_bwd_hooked", ptr_table(tensors), ptr_table(grads), feats, d_ae, ctx.saved,
                  ctx.saved.numel(), scratch, scratch_b, ctx.arch, N, T, F, int(ctx.training), float(mod.p_block), ctx.seed, mt,
                  _lib.current_stream(), ctypes.cast(hook, ctypes.c_void_p) if hook is not None else None, None)
        if failed:
            raise failed[0]
        ctx.saved = None
        if mod._grad_ready_cb is not None:
            mod._grad_ready_cb("encoder")
        for w in mod._weights():
            outs.append(next((g for t, g in zip(tensors, grads) if t is w), None))
        return (None, None, *outs)


class _PannsCnn(nn.Module):
    """Shared plumbing of the PANNs CNN encoders: bn0 over the mel axis, N_BLOCKS ConvBlocks, a pooled head."""
    ARCH, N_BLOCKS, TIME_DIV, OUT_CHANNELS, HEAD = 0, 4, 16, 512, "embed_pooled"

    def __init__(self, inputdim, embed_size, **kwargs):
        super().__init__()
        self.inputdim = inputdim          # BaseEncoder attributes
        self.embed_size = embed_size
        self.bn0 = nn.BatchNorm2d(64)
        chans = [1, 64, 128, 256, 512, 1024, 2048]
        for b in range(1, self.N_BLOCKS + 1):
            setattr(self, f"conv_block{b}", ConvBlock(chans[b - 1], chans[b]))
        setattr(self, self.HEAD, nn.Linear(self.OUT_CHANNELS, self.OUT_CHANNELS, bias=True))
        init_bn(self.bn0)
        init_layer(getattr(self, self.HEAD))
        self.p_block, self.p_fc = 0.2, 0.5      # F.dropout probabilities, encoder.py:684-698 / :929-950
        # "f32" (the reference's arithmetic, BASELINE configs[1]) or "bf16": BASELINE configs[2] "bf16 forward / fp32
        # loss" - the conv stack stores its activations in bf16 and multiplies on the bf16 MFMA pipe; parameters,
        # BatchNorm statistics, gradients of the parameters and both outputs stay fp32 (ACVAE_ENC_BF16)
        self.compute_dtype = kwargs.get("compute_dtype", "f32")
        self.dropout_masks = None               # optional explicit keep-masks (parity tests)
        self._seed_base, self._calls = None, 0
        self.keep_saved = False                 # test aid: keep the last forward's activation buffer for relu_masks()
        self._last_saved = None
        self._grad_views = None                 # {param: flat-gradient view}, set by the train-step harness
        self._grad_ready_cb = None              # called with "encoder" when the backward has written all grads

    # ---- plumbing
    def _arch(self):
        if self.compute_dtype not in ("f32", "bf16"):
            raise ValueError(f"compute_dtype must be 'f32' or 'bf16', not {self.compute_dtype!r}")
        return self.ARCH | (_lib.ENC_BF16 if self.compute_dtype == "bf16" else 0)

    def _blocks(self):
        return [getattr(self, f"conv_block{b}") for b in range(1, self.N_BLOCKS + 1)]

    def _head(self):
        return getattr(self, self.HEAD)

    def _param_table(self):
        t = _bn_tensors(self.bn0)
        for blk in self._blocks():
            t += [blk.conv1.weight, blk.conv2.weight] + _bn_tensors(blk.bn1) + _bn_tensors(blk.bn2)
        t += [self._head().weight, self._head().bias]
        return t

    def _weights(self):
        return [t for t in self._param_table()[:-2] if isinstance(t, nn.Parameter)]

    def _grad_buffer(self, p):
        if self._grad_views is not None and p in self._grad_views:
            return self._grad_views[p].detach()     # fresh alias: autograd adopts it as .grad without a copy
        return torch.empty_like(p)

    def relu_masks(self):
        """The ReLU decisions of the last forward (``keep_saved = True`` must have been set before it), one bool tensor
        [N,C,H,W] per BN+ReLU site in forward order - exactly the masks the backward applies (acvae_encoder_relu_mask)."""
        if self._last_saved is None:
            raise RuntimeError("relu_masks(): set keep_saved = True before the forward")
        saved, arch, N, T, F = self._last_saved
        chans = [64, 128, 256, 512, 1024, 2048]
        out, h, w = [], T, F
        for b in range(self.N_BLOCKS):
            for which in range(2):
                m = torch.empty(N, chans[b], h, w, dtype=torch.uint8, device=saved.device)
                _lib.call("acvae_encoder_relu_mask", saved, saved.numel(), arch, N, T, F, 2 * b + which, m,
                          _lib.current_stream())
                out.append(m.bool())
            if b < self.N_BLOCKS - 1 or self.ARCH == 0:
                h, w = h // 2, w // 2
        return out

    def _next_seed(self):
        if self._seed_base is None:
            self._seed_base = int(torch.initial_seed()) & 0x7FFFFFFFFFFF
        self._calls += 1
        return (self._seed_base * 1000003 + self._calls) & 0x7FFFFFFFFFFFFFFF

    def forward(self, input, lens):
        """input: [batch, time, 64] log-mel; lens: frame counts (numpy array / list / tensor)."""
        lens = torch.as_tensor(lens)
        lens //= self.TIME_DIV            # in place on the caller's array, as the reference does (:677-678 / :913-914)
        lens_dev = _lib.h2d(lens, input.device, torch.long)   # staged before the kernels are queued
        ae, pooled = _Cnn10Fn.apply(self, input, *self._weights())
        return {"audio_embeds": ae, "audio_embeds_pooled": pooled, "state": None, "audio_embeds_lens": lens,
                "audio_embeds_lens_dev": lens_dev}


class Cnn10(_PannsCnn):
    """PANNs CNN10 audio encoder (reference ``models/encoder.py:651-707``)."""


class Cnn14_16k(_PannsCnn):
    """PANNs CNN14 (16 kHz) audio encoder (reference ``models/encoder.py:871-964``): six ConvBlocks
    up to 2048 channels, the sixth pooled (1,1), time / 32, ``fc1`` (2048 x 2048) as the pooled head.  With a 512-wide
    decoder ``Hybrid_VAEModel`` adds the ``ln`` projection 2048 -> 512 (``models/vae_model.py:695-697``)."""
    ARCH, N_BLOCKS, TIME_DIV, OUT_CHANNELS, HEAD = 1, 6, 32, 2048, "fc1"
