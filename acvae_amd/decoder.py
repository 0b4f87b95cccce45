"""Mirror of ``models/decoder.py``: ``BaseDecoder`` (:10-25), ``RNNDecoder`` (:28-98) constructor
surface and ``VAERNNBahdanauAttnDecoder`` (:164-203).  Parameter names/shapes/initialisation follow the
reference (``word_embeddings``, ``model`` = nn.GRU container, ``classifier``, ``attn``).  The step
arithmetic lives in libacvae_hip.so and is driven by Hybrid_VAEModel through acvae_decode_fwd/bwd."""
import weakref

import torch
import torch.nn as nn

from . import _lib
from .attn_model import Seq2SeqAttention
from .encoder import ptr_table, scratch_buffer


class _ProjTableFn(torch.autograd.Function):
    """table[V,E] = Embedding.weight[V,D0] . Linear.weight[E,D0]^T + Linear.bias: the projected pretrained embeddings of
    ``load_word_embeddings(.., projection=True)`` (models/decoder.py:58-64) as ONE product per forward instead of a
    Linear per looked-up word; the decode loop then gathers rows of `table` like a plain embedding, and the table's
    gradient (the usual embedding scatter) is split here into d Embedding / d Linear."""

    @staticmethod
    def forward(ctx, owner, emb, w, b):
        _lib.require_cuda(emb, w, b)
        V, D0 = emb.shape
        E = w.shape[0]
        table = torch.empty(V, E, device=emb.device)
        _lib.call("acvae_gemm_nt", emb, D0, w, D0, b, table, E, V, E, D0, 0, _lib.current_stream())
        ctx.owner = None if owner is None else weakref.ref(owner)   # no ctx -> model -> table cache -> ctx cycle
        ctx.save_for_backward(emb, w)
        return table

    @staticmethod
    def backward(ctx, d_table):
        emb, w = ctx.saved_tensors
        owner = None if ctx.owner is None else ctx.owner()
        V, D0 = emb.shape
        E = w.shape[0]
        d_table = d_table.contiguous().float()
        st = _lib.current_stream()
        buf = (lambda p: owner._grad_buffer(p)) if owner is not None else torch.empty_like
        seq = owner.decoder.word_embeddings if owner is not None else None
        d_emb = d_w = d_b = None
        if ctx.needs_input_grad[1]:                     # d Embedding = d_table . W  (W^T as the NT product's second operand)
            wt = torch.empty(D0, E, device=w.device)
            _lib.call("acvae_transpose", w, D0, wt, E, E, D0, st)
            d_emb = buf(seq[0].weight) if seq is not None else torch.empty_like(emb)
            _lib.call("acvae_gemm_nt", d_table, E, wt, E, None, d_emb, D0, V, D0, E, 0, st)
        if ctx.needs_input_grad[2]:                     # d W[E,D0] = d_table^T . Embedding
            d_w = buf(seq[1].weight) if seq is not None else torch.empty_like(w)
            wsb = _lib.call("acvae_gemm_tn_workspace_bytes", E, D0, V)
            ws = scratch_buffer(max(int(wsb), 16), w.device, tag="projemb")
            _lib.call("acvae_gemm_tn", d_table, E, emb, D0, d_w, D0, E, D0, V, 0, ws, wsb, st)
        if ctx.needs_input_grad[3]:
            d_b = buf(seq[1].bias) if seq is not None else torch.empty(E, device=w.device)
            cb = _lib.call("acvae_colsum_workspace_bytes", E)
            cws = scratch_buffer(int(cb), w.device, tag="projemb_b")
            _lib.call("acvae_colsum", d_table, V, E, d_b, cws, cb, st)
        if owner is not None:
            owner.decoder._table_cache = None       # the table (and its graph) is spent
            if owner._grad_ready_cb is not None:
                owner._grad_ready_cb("projemb")     # the last decode-side gradients are queued now
        return None, d_emb, d_w, d_b


class BaseDecoder(nn.Module):
    def __init__(self, embed_size, vocab_size, enc_mem_size):
        super().__init__()
        self.embed_size = embed_size
        self.vocab_size = vocab_size
        self.enc_mem_size = enc_mem_size
        self.word_embeddings = nn.Embedding(vocab_size, embed_size)


class RNNDecoder(BaseDecoder):
    def __init__(self, vocab_size, enc_mem_size, **kwargs):
        embed_size = kwargs.get("embed_size", 256)
        super().__init__(embed_size, vocab_size, enc_mem_size)
        dropout_p = kwargs.get("dropout", 0.0)
        hidden_size = kwargs.get("hidden_size", 256)
        num_layers = kwargs.get("num_layers", 1)
        bidirectional = kwargs.get("bidirectional", False)
        rnn_type = kwargs.get("rnn_type", "GRU")
        self.dropoutlayer = nn.Dropout(dropout_p)
        self.model = getattr(nn, rnn_type)(input_size=embed_size + enc_mem_size, hidden_size=hidden_size,
                                           num_layers=num_layers, batch_first=True, bidirectional=bidirectional)
        self.classifier = nn.Linear(hidden_size * (bidirectional + 1), vocab_size)
        nn.init.kaiming_uniform_(self.word_embeddings.weight)
        nn.init.kaiming_uniform_(self.classifier.weight)

    def load_word_embeddings(self, embeddings, tune=True, **kwargs):
        """models/decoder.py:50-64"""
        assert embeddings.shape[0] == self.vocab_size, "vocabulary size mismatch!"
        embeddings = torch.as_tensor(embeddings).float()
        self.word_embeddings.weight = nn.Parameter(embeddings.to(self.word_embeddings.weight.device))
        for para in self.word_embeddings.parameters():
            para.requires_grad = tune
        if embeddings.shape[1] != self.embed_size:
            assert "projection" in kwargs, "embedding size mismatch!"
            if kwargs["projection"]:
                self.word_embeddings = nn.Sequential(
                    self.word_embeddings,
                    nn.Linear(embeddings.shape[1], self.embed_size).to(self.word_embeddings.weight.device))

    def embedding_table(self):
        """The [V, embed_size] table the decode loop gathers word vectors from: the embedding weight itself, or - with
        projected pretrained embeddings - Embedding.weight . Linear.weight^T + bias, computed once per model forward."""
        we = self.word_embeddings
        if not isinstance(we, nn.Sequential):
            return we.weight
        owner = self._owner() if getattr(self, "_owner", None) is not None else None
        token = getattr(owner, "_forward_token", None)       # refreshed at every model forward (and reused by its backward)
        # valid for ONE model forward and only while the three parameters are what they were when it was built (an
        # optimiser step, load_state_dict or an in-place edit bumps _version); callers outside a model forward (the
        # step-wise decoder / prior API) have no token and always rebuild
        # (the grad mode is NOT part of the key: the backward of that forward runs with grad mode off and must find the
        # forward's table, whose grad_fn routes the table's gradient to the three parameters)
        key = (token, we[0].weight._version, we[1].weight._version, we[1].bias._version, we[0].weight.data_ptr(),
               we[1].weight.data_ptr())
        hit = getattr(self, "_table_cache", None)
        if hit is not None and token is not None and hit[0] == key:
            return hit[1]
        table = _ProjTableFn.apply(owner, we[0].weight, we[1].weight, we[1].bias)
        self._table_cache = (key, table) if token is not None else None
        return table

    def train(self, mode=True):
        self._table_cache = None
        return super().train(mode)

    def init_hidden(self, bs):
        """models/decoder.py:94-98"""
        bidirectional = self.model.bidirectional
        return torch.zeros((bidirectional + 1) * self.model.num_layers, bs, self.model.hidden_size)


class VAERNNBahdanauAttnDecoder(RNNDecoder):
    """GRU over [emb; ctx; z] with Bahdanau attention on the previous hidden state (decoder.py:164-203)."""

    def __init__(self, vocab_size, enc_mem_size, **kwargs):
        super().__init__(vocab_size, enc_mem_size * 2, **kwargs)
        if kwargs.get("rnn_type", "GRU") != "GRU" or kwargs.get("num_layers", 1) != 1 or kwargs.get("bidirectional", False):
            raise NotImplementedError("the HIP path implements the 1-layer unidirectional GRU decoder")
        # `dropout` (default 0.0): nn.Dropout on the word embedding (:33,184); applied inside the fused decode loop in
        # training mode (Hybrid_VAEModel draws the keep masks on the CPU generator in the reference's call order)
        attn_size = kwargs.get("attn_size", self.model.hidden_size)
        self.attn = Seq2SeqAttention(enc_mem_size, self.model.hidden_size, attn_size)
        self.mem_size = enc_mem_size
        self._owner = None          # weakref to the Hybrid_VAEModel (set by it)
        if self.embed_size != enc_mem_size:
            # the reference sizes the GRU input as embed + 2*enc_mem but feeds [emb(E); ctx(mem); z(E)] (:171,:188)
            raise ValueError("VAERNNBahdanauAttnDecoder needs embed_size == enc_mem_size (SURVEY §8)")

    def forward(self, **kwargs):
        """One decode step (inference, no gradient): models/decoder.py:175-203.  word [N,1] (or [N]), state [1,N,H],
        enc_mem [N,S,E], enc_mem_lens [N], z [N,E] -> {"state","output","logits","weights","rnn_input"}.
        Training goes through Hybrid_VAEModel.forward, where the whole loop is one fused call."""
        if self._owner is None:
            raise RuntimeError("VAERNNBahdanauAttnDecoder.forward needs the parameters of its Hybrid_VAEModel "
                               "(the HIP library addresses the text side as one table)")
        owner = self._owner()
        enc_mem = kwargs["enc_mem"]
        _lib.require_cuda(enc_mem)
        dev = enc_mem.device
        enc_mem = enc_mem.contiguous().float()
        N, S, E = enc_mem.shape
        H, A, V = self.model.hidden_size, self.attn.attn_size, self.vocab_size
        w = kwargs["word"].reshape(-1).to(device=dev, dtype=torch.long).contiguous()
        h_prev = kwargs["state"].reshape(N, H).to(dev).contiguous().float()
        z = kwargs["z"].reshape(N, E).to(dev).contiguous().float()
        lens = torch.as_tensor(kwargs["enc_mem_lens"]).to(device=dev, dtype=torch.long).contiguous()
        with torch.no_grad():
            encproj = owner._encproj(0, enc_mem)
            logits = torch.empty(N, V, device=dev); h_out = torch.empty(N, H, device=dev)
            attw = torch.empty(N, S, device=dev); rnn_in = torch.empty(N, 3 * E, device=dev)
            sb = _lib.call("acvae_step_scratch_bytes", N, S, E, H, A, V)
            scratch = scratch_buffer(sb, dev)
            _lib.call("acvae_decoder_step_fwd", ptr_table(owner._text_table()), w, h_prev, enc_mem, lens, encproj, z,
                      logits, h_out, attw, rnn_in, scratch, sb, N, S, E, H, A, V, _lib.current_stream())
        return {"state": h_out.unsqueeze(0), "output": h_out.unsqueeze(1), "logits": logits.unsqueeze(1),
                "weights": attw, "rnn_input": rnn_in.unsqueeze(1)}
