"""Mirror of ``models/decoder.py``: ``BaseDecoder`` (:10-25), ``RNNDecoder`` (:28-98) constructor
surface and ``VAERNNBahdanauAttnDecoder`` (:164-203).  Parameter names/shapes/initialisation follow the
reference (``word_embeddings``, ``model`` = nn.GRU container, ``classifier``, ``attn``).  The step
arithmetic lives in libacvae_hip.so and is driven by Hybrid_VAEModel through acvae_decode_fwd/bwd."""
import torch
import torch.nn as nn

from . import _lib
from .attn_model import Seq2SeqAttention
from .encoder import ptr_table, scratch_buffer


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
        if embeddings.shape[1] != self.embed_size:
            raise NotImplementedError("projected pretrained embeddings are outside the HIP path (embedding size "
                                      "must equal embed_size)")
        self.word_embeddings.weight = nn.Parameter(embeddings.to(self.word_embeddings.weight.device))
        for para in self.word_embeddings.parameters():
            para.requires_grad = tune

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
