"""Mirror of ``models/decoder.py``: ``BaseDecoder`` (:10-25), ``RNNDecoder`` (:28-98) constructor
surface and ``VAERNNBahdanauAttnDecoder`` (:164-203).  Parameter names/shapes/initialisation follow the
reference (``word_embeddings``, ``model`` = nn.GRU container, ``classifier``, ``attn``).  The step
arithmetic lives in libacvae_hip.so and is driven by Hybrid_VAEModel through acvae_decode_fwd/bwd."""
import torch
import torch.nn as nn

from .attn_model import Seq2SeqAttention


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
        if kwargs.get("dropout", 0.0) != 0.0:
            raise NotImplementedError("word-embedding dropout > 0 is not on the HIP path (reference default 0.0)")
        attn_size = kwargs.get("attn_size", self.model.hidden_size)
        self.attn = Seq2SeqAttention(enc_mem_size, self.model.hidden_size, attn_size)
        self.mem_size = enc_mem_size
        if self.embed_size != enc_mem_size:
            # the reference sizes the GRU input as embed + 2*enc_mem but feeds [emb(E); ctx(mem); z(E)] (:171,:188)
            raise ValueError("VAERNNBahdanauAttnDecoder needs embed_size == enc_mem_size (SURVEY §8)")

    def forward(self, **kwargs):
        raise NotImplementedError(
            "single-step decoder calls are fused into acvae_decode_fwd on the HIP path; call "
            "Hybrid_VAEModel.forward (training or method='greedy' inference)")
