"""Mirror of ``models/attn_model.py`` ``Seq2SeqAttention`` (:6-46): same constructor, parameter names
(``h2attn.weight/bias``, ``v``) and forward(h_dec, h_enc, src_lens) -> (ctx, weights).  Inside the
decode loop the attention runs in the fused drivers (acvae_decode_fwd); this standalone forward uses the
same kernels (acvae_gemm_nt for the two halves of h2attn, acvae_attn_fwd) and is inference-only."""
import torch
import torch.nn as nn

from . import _lib


class Seq2SeqAttention(nn.Module):
    def __init__(self, hs_enc, hs_dec, attn_size):
        super().__init__()
        self.h2attn = nn.Linear(hs_enc + hs_dec, attn_size)
        self.v = nn.Parameter(torch.randn(attn_size))
        nn.init.kaiming_uniform_(self.h2attn.weight)
        self.hs_enc, self.hs_dec, self.attn_size = hs_enc, hs_dec, attn_size

    @torch.no_grad()
    def forward(self, h_dec, h_enc, src_lens):
        _lib.require_cuda(h_dec, h_enc)
        h_dec, h_enc = h_dec.contiguous().float(), h_enc.contiguous().float()
        N, S, E = h_enc.shape
        Hd, A = h_dec.shape[1], self.attn_size
        W = self.h2attn.weight                       # [A, hs_dec + hs_enc]; cat order [h_dec; h_enc] (:31)
        dev = h_enc.device
        st = _lib.current_stream()
        lens = torch.as_tensor(src_lens).to(device=dev, dtype=torch.long).contiguous()
        encproj = torch.empty(N, S, A, device=dev)
        qproj = torch.empty(N, A, device=dev)
        _lib.call("acvae_gemm_nt", h_enc, E, W.data_ptr() + Hd * 4, Hd + E, self.h2attn.bias, encproj, A, N * S, A, E,
                  0, st)
        _lib.call("acvae_gemm_nt", h_dec, Hd, W, Hd + E, None, qproj, A, N, A, Hd, 0, st)
        ctx = torch.empty(N, E, device=dev)
        weights = torch.empty(N, S, device=dev)
        ws, ws_b = _lib.attn_fwd_workspace(N, 1, S, A, E, dev)
        _lib.call("acvae_attn_fwd", qproj, A, 0, encproj, h_enc, lens, self.v, ctx, E, 0, weights, S, 0, N, 1, S, A, E,
                  ws, ws_b, st, _lib.call_flags())
        return ctx, weights
