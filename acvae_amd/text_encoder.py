"""Mirror of ``models/text_encoder.py``: ``PosteriorRNN_hybrid`` (:156-216) and ``PriorRNN`` (:218-268)
with the reference's constructor keywords, parameter names and initialisation (``init`` :25-41)."""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .attn_model import Seq2SeqAttention
from .encoder import ptr_table, scratch_buffer


def _init_weights(m):
    """models/text_encoder.py:29-41 (the Linear branch is the one that fires here)"""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


class PosteriorBaseEncoder(nn.Module):
    def __init__(self, word_dim, embed_size, vocab_size):
        super().__init__()
        self.word_dim, self.embed_size, self.vocab_size = word_dim, embed_size, vocab_size
        self.word_embedding = nn.Embedding(vocab_size, word_dim)

    def init(self):
        for m in self.modules():
            m.apply(_init_weights)


class PriorBaseEncoder(nn.Module):
    def __init__(self, word_dim, audiofeats_size, embed_size, vocab_size):
        super().__init__()
        self.word_dim, self.embed_size, self.vocab_size = word_dim, embed_size, vocab_size
        self.audiofeats_size = audiofeats_size
        self.word_embedding = nn.Embedding(vocab_size, word_dim)

    def init(self):
        for m in self.modules():
            m.apply(_init_weights)


def _check_rnn_kwargs(kwargs, what, rnn_default):
    if kwargs.get("num_layers", 1) != 1:
        raise NotImplementedError(f"{what}: the HIP path implements num_layers=1")
    if kwargs.get("rnn_type", rnn_default) != rnn_default:
        raise NotImplementedError(f"{what}: the HIP path implements rnn_type={rnn_default}")


class _PosteriorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, caps_d, lens1_d, eps_q, Tc, *weights):
        N = caps_d.shape[0]
        E, Hq, V = mod.embed_size, mod.hidden_size, mod.vocab_size
        dev = eps_q.device
        params = mod._text_table()
        saved_b = _lib.call("acvae_posterior_saved_bytes", N, Tc, E, Hq, V)
        scratch_b = _lib.call("acvae_posterior_scratch_bytes", N, Tc, E, Hq, V)
        saved = torch.empty(saved_b, dtype=torch.uint8, device=dev)
        scratch = scratch_buffer(scratch_b, dev)
        qm, ql, qz = (torch.empty(N, Tc, E, device=dev) for _ in range(3))
        utt = torch.empty(N, 2 * Hq, device=dev)
        _lib.persist_status(dev)                 # the device's status words are registered before the first persistent launch
        _lib.call("acvae_posterior_fwd", ptr_table(params), caps_d, caps_d.stride(0), lens1_d, eps_q, qm, ql, qz, utt,
                  saved, saved_b, scratch, scratch_b, N, Tc, E, Hq, V, _lib.current_stream(), _lib.call_flags())
        ctx.set_materialize_grads(False)         # unused outputs arrive as None in backward (no zero-fill launches)
        ctx.mod, ctx.saved, ctx.dims = mod, saved, (N, Tc, E, Hq, V)
        # an OUTPUT kept as a plain ctx attribute forms a tensor -> grad_fn -> ctx -> tensor cycle that is never collected
        ctx.save_for_backward(lens1_d, eps_q, ql)
        return qm, ql, qz, utt

    @staticmethod
    def backward(ctx, d_qm, d_ql, d_qz, d_utt):
        mod = ctx.mod
        lens1_d, eps_q, ql = ctx.saved_tensors
        N, Tc, E, Hq, V = ctx.dims
        params = mod._text_table()
        grads = [None] * len(params)
        for i, p in enumerate(params):
            if p is not None and p.requires_grad and 10 <= i <= 20:
                grads[i] = mod._grad_buffer(p)
        c = lambda t: None if t is None else t.contiguous().float()
        scratch_b = _lib.call("acvae_posterior_scratch_bytes", N, Tc, E, Hq, V)
        scratch = scratch_buffer(scratch_b, eps_q.device)
        _lib.call("acvae_posterior_bwd", ptr_table(params), ptr_table(grads), lens1_d, eps_q, ql, c(d_qm),
                  c(d_ql), c(d_qz), c(d_utt), ctx.saved, ctx.saved.numel(), scratch, scratch_b, N, Tc, E, Hq, V,
                  _lib.current_stream(), _lib.call_flags())
        ctx.saved = None
        owner = mod._owner() if mod._owner is not None else None
        if owner is not None and owner._grad_ready_cb is not None:
            owner._grad_ready_cb("text")            # decode backward always precedes this node (it consumes d_q_z)
        outs = [next((g for p, g in zip(params, grads) if p is w), None) for w in mod._weights()]
        return (None, None, None, None, None, *outs)


class PosteriorRNN_hybrid(PosteriorBaseEncoder):
    """q(z_t | caption): BiGRU over the packed caption -> token_mean_log -> (mean, logvar) -> reparam;
    utterance vector = mean_with_lens + max_with_lens of the BiGRU outputs."""

    def __init__(self, word_dim, embed_size, vocab_size, **kwargs):
        super().__init__(word_dim, embed_size, vocab_size)
        self.hidden_size = kwargs.get("hidden_size", 256)
        self.bidirectional = kwargs.get("bidirectional", True)
        self.num_layers = kwargs.get("num_layers", 1)
        self.dropout = kwargs.get("dropout", 0.3)
        self.rnn_type = kwargs.get("rnn_type", "GRU")
        _check_rnn_kwargs(kwargs, "PosteriorRNN_hybrid", "GRU")
        if not self.bidirectional:
            raise NotImplementedError("PosteriorRNN_hybrid: the HIP path implements the bidirectional GRU")
        if word_dim != embed_size:
            raise NotImplementedError("PosteriorRNN_hybrid: word_dim must equal embed_size")
        # dropout only acts between stacked layers; with num_layers=1 torch ignores it (and warns)
        self.network = nn.GRU(word_dim, self.hidden_size, num_layers=1, bidirectional=True, batch_first=True)
        self.token_mean_log = nn.Linear(2 * self.hidden_size, 2 * embed_size)
        self.init()
        self._owner = None          # Hybrid_VAEModel sets this so the parameter table covers the whole text side

    def _text_table(self):
        if self._owner is not None:
            return self._owner()._text_table()
        t = [None] * _lib.ENUMS_TEXT_N
        n = self.network
        t[10:21] = [self.word_embedding.weight, n.weight_ih_l0, n.weight_hh_l0, n.bias_ih_l0, n.bias_hh_l0,
                    n.weight_ih_l0_reverse, n.weight_hh_l0_reverse, n.bias_ih_l0_reverse, n.bias_hh_l0_reverse,
                    self.token_mean_log.weight, self.token_mean_log.bias]
        return t

    def _weights(self):
        return [p for p in self._text_table()[10:21]]

    def _grad_buffer(self, p):
        owner = self._owner() if self._owner is not None else None
        views = getattr(owner, "_grad_views", None) if owner is not None else None
        if views is not None and p in views:
            return views[p].detach()
        return torch.empty_like(p)

    def forward(self, x, lengths, eps=None):
        """x: caption ids [N,L] (float or long, as the collate fn pads with float zeros); lengths: cap_lens.
        eps (optional): the N(0,1) draw to use; default torch.randn on the CPU generator (text_encoder.py:196)."""
        dev = self.token_mean_log.weight.device
        lengths = np.asarray(lengths) - 1
        Tc = int(lengths.max())
        N = x.shape[0]
        caps_d = _lib.h2d(x, dev, torch.long).contiguous()
        lens1_d = _lib.h2d(lengths, dev, torch.long)
        if eps is None:
            eps = torch.randn(N, Tc, self.embed_size)
        eps = _lib.h2d(eps, dev).contiguous()
        qm, ql, qz, utt = _PosteriorFn.apply(self, caps_d, lens1_d, eps, Tc, *self._weights())
        return {"q_means": qm, "q_logs": ql, "q_z": qz, "q_means_utt": utt, "q_logs_utt": None, "q_z_utt": None}


class PriorRNN(PriorBaseEncoder):
    """p(z_t | w_<t, z_<t, audio): word-embedding attention over the audio memory + LSTM + mean_log_out +
    reparameterisation (text_encoder.py:218-268).  The per-step arithmetic runs inside acvae_decode_fwd."""

    def __init__(self, word_dim, audiofeats_size, embed_size, vocab_size, **kwargs):
        super().__init__(word_dim, audiofeats_size, embed_size, vocab_size)
        self.hidden_size = kwargs.get("hidden_size", 256)
        self.bidirectional = kwargs.get("bidirectional", False)
        self.num_layers = kwargs.get("num_layers", 1)
        self.dropout = kwargs.get("dropout", 0.3)
        self.rnn_type = kwargs.get("rnn_type", "LSTM")
        _check_rnn_kwargs(kwargs, "PriorRNN", "LSTM")
        if self.bidirectional:
            raise NotImplementedError("PriorRNN: the HIP path implements the unidirectional LSTM")
        if self.hidden_size != embed_size:
            # init_hidden sizes the LSTM state with embed_size (text_encoder.py:240-245)
            raise ValueError("PriorRNN needs hidden_size == embed_size (SURVEY §8)")
        self.word_attn = Seq2SeqAttention(audiofeats_size, word_dim, audiofeats_size)
        self.network = nn.LSTM(word_dim + audiofeats_size + embed_size, self.hidden_size, num_layers=1,
                               bidirectional=False, batch_first=True)
        self.mean_log_out = nn.Linear(self.hidden_size, 2 * embed_size)
        self.init()
        self._owner = None

    def init_hidden(self, bs, device):
        z = lambda: torch.zeros(self.num_layers, bs, self.embed_size, device=device)
        return (z(), z())

    def forward(self, word, enc_mem, hiddens_state, last_z, lens, eps=None):
        """One prior step (inference, no gradient): models/text_encoder.py:247-268.  word [N,1], enc_mem [N,S,E],
        hiddens_state (h, c) each [1,N,E], last_z [N,E], lens [N] -> {"mean","log","hiddens_state","z"}.
        eps: optional N(0,1) draw; default torch.randn on the CPU generator (:259)."""
        if self._owner is None:
            raise RuntimeError("PriorRNN.forward needs the parameters of its Hybrid_VAEModel")
        owner = self._owner()
        _lib.require_cuda(enc_mem)
        dev = enc_mem.device
        enc_mem = enc_mem.contiguous().float()
        N, S, E = enc_mem.shape
        V = self.vocab_size
        w = word.reshape(-1).to(device=dev, dtype=torch.long).contiguous()
        h_prev = hiddens_state[0].reshape(N, E).to(dev).contiguous().float()
        c_prev = hiddens_state[1].reshape(N, E).to(dev).contiguous().float()
        last_z = last_z.reshape(N, E).to(dev).contiguous().float()
        lens_d = torch.as_tensor(lens).to(device=dev, dtype=torch.long).contiguous()
        if eps is None:
            eps = torch.randn(N, E)
        eps = eps.to(dev).contiguous().float()
        with torch.no_grad():
            encproj = owner._encproj(1, enc_mem)
            mean, logv, z, h, c = (torch.empty(N, E, device=dev) for _ in range(5))
            attw = torch.empty(N, S, device=dev)
            H, A = owner.decoder.model.hidden_size, owner.decoder.attn.attn_size
            sb = _lib.call("acvae_step_scratch_bytes", N, S, E, H, A, V)
            scratch = scratch_buffer(sb, dev)
            _lib.call("acvae_prior_step_fwd", ptr_table(owner._text_table()), w, enc_mem, lens_d, encproj, h_prev,
                      c_prev, last_z, eps, mean, logv, z, h, c, attw, scratch, sb, N, S, E, V, _lib.current_stream())
        return {"mean": mean, "log": logv, "hiddens_state": (h.unsqueeze(0), c.unsqueeze(0)), "z": z}
