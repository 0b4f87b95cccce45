"""CPU oracle for the AC-VAE training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU (fp32) restatement of the reference's algorithm
for the path named by BASELINE.json:north_star.  It is NOT part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / CPU baseline.  The product
path (``acvae_amd``) never imports anything from ``oracle/``.

Parity is PINNED: ``oracle/make_golden.py`` imports the reference itself (through
the two-line in-memory shim of SURVEY.md §8(c)) in the build container, fills both
models with the same closed-form parameters, and writes input/output vectors to
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors.  The reference ships no tests or golden vectors of its own
(SURVEY.md §4), so these generated fixtures are the pin.

Every function cites the reference file:line it follows (paths relative to the
reference root).  The code is written functionally over a flat ``state`` dict
whose keys are the reference's state-dict names, so the same dict drives the
reference, this oracle and the HIP path.
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

PAD_IDX, START_IDX, END_IDX, MAX_LENGTH = 0, 1, 2, 20  # models/word_model.py:19-22


# ----------------------------------------------------------------------------
# model description + closed-form parameters
# ----------------------------------------------------------------------------
ENCODERS = {"Cnn10": dict(channels=(64, 128, 256, 512), head="embed_pooled", div=16, pool_last=True),
            # models/encoder.py:871-964: six blocks, the last one pooled (1,1), time // 32
            "Cnn14_16k": dict(channels=(64, 128, 256, 512, 1024, 2048), head="fc1", div=32, pool_last=False)}


def state_shapes(vocab_size: int, embed: int = 512, hidden: int = 512, attn: Optional[int] = None,
                 q_hidden: Optional[int] = None, enc_embed: int = 512, encoder: str = "Cnn10",
                 proj_embed: Optional[int] = None) -> Dict[str, tuple]:
    """Shapes of every state-dict entry of Hybrid_VAEModel(Cnn10, VAERNNBahdanauAttnDecoder,
    PosteriorRNN_hybrid, PriorRNN) in the reference's registration order.

    models/encoder.py:606-670 (Cnn10/ConvBlock), models/decoder.py:30-48,166-173,
    models/attn_model.py:16-18, models/text_encoder.py:157-176,220-238,
    models/vae_model.py:676-698.  Constraints: decoder embed == enc_mem size,
    prior hidden == embed.
    """
    E, H, V = embed, hidden, vocab_size
    A = attn if attn is not None else H
    Hq = q_hidden if q_hidden is not None else E
    s: Dict[str, tuple] = {}

    def bn(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,); s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    bn("encoder.bn0", 64)
    cin = 1
    arch = ENCODERS[encoder]
    for b, cout in enumerate(arch["channels"], start=1):
        p = f"encoder.conv_block{b}"
        s[p + ".conv1.weight"] = (cout, cin, 3, 3)
        s[p + ".conv2.weight"] = (cout, cout, 3, 3)
        bn(p + ".bn1", cout); bn(p + ".bn2", cout)
        cin = cout
    s[f"encoder.{arch['head']}.weight"] = (cin, cin); s[f"encoder.{arch['head']}.bias"] = (cin,)
    mem = E  # decoder is built with enc_mem_size = encoder embed_size (runner :44-48)
    if proj_embed is None:
        s["decoder.word_embeddings.weight"] = (V, E)
    else:       # load_word_embeddings(.., projection=True): Sequential(Embedding(V, D0), Linear(D0, E)), decoder.py:58-64
        s["decoder.word_embeddings.0.weight"] = (V, proj_embed)
        s["decoder.word_embeddings.1.weight"] = (E, proj_embed); s["decoder.word_embeddings.1.bias"] = (E,)
    s["decoder.model.weight_ih_l0"] = (3 * H, E + 2 * mem)
    s["decoder.model.weight_hh_l0"] = (3 * H, H)
    s["decoder.model.bias_ih_l0"] = (3 * H,); s["decoder.model.bias_hh_l0"] = (3 * H,)
    s["decoder.classifier.weight"] = (V, H); s["decoder.classifier.bias"] = (V,)
    s["decoder.attn.v"] = (A,)
    s["decoder.attn.h2attn.weight"] = (A, mem + H); s["decoder.attn.h2attn.bias"] = (A,)
    s["qnet.word_embedding.weight"] = (V, E)
    for suf in ("", "_reverse"):
        s["qnet.network.weight_ih_l0" + suf] = (3 * Hq, E)
        s["qnet.network.weight_hh_l0" + suf] = (3 * Hq, Hq)
        s["qnet.network.bias_ih_l0" + suf] = (3 * Hq,)
        s["qnet.network.bias_hh_l0" + suf] = (3 * Hq,)
    s["qnet.token_mean_log.weight"] = (2 * E, 2 * Hq); s["qnet.token_mean_log.bias"] = (2 * E,)
    s["pnet.word_embedding.weight"] = (V, E)
    s["pnet.word_attn.v"] = (E,)
    s["pnet.word_attn.h2attn.weight"] = (E, 2 * E); s["pnet.word_attn.h2attn.bias"] = (E,)
    s["pnet.network.weight_ih_l0"] = (4 * E, 3 * E)
    s["pnet.network.weight_hh_l0"] = (4 * E, E)
    s["pnet.network.bias_ih_l0"] = (4 * E,); s["pnet.network.bias_hh_l0"] = (4 * E,)
    s["pnet.mean_log_out.weight"] = (2 * E, E); s["pnet.mean_log_out.bias"] = (2 * E,)
    s["mean_log_out.weight"] = (2 * E, E); s["mean_log_out.bias"] = (2 * E,)
    if enc_embed != E:
        s["ln.weight"] = (E, enc_embed); s["ln.bias"] = (E,)
    return s


def closed_form_tensor(name: str, shape: tuple, tensor_id: int) -> torch.Tensor:
    """Deterministic closed-form fill shared by oracle, reference harness and HIP tests
    (weights are never committed, only inputs/outputs)."""
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    wave = np.sin(0.7368 * i + 1.3 * tensor_id + 0.37) * 0.6 + np.sin(0.1193 * i * (1 + 0.01 * tensor_id) + 0.5) * 0.4
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":
        v = 1.0 + 0.3 * wave
    elif leaf == "running_mean":
        v = 0.1 * wave
    elif len(shape) == 1 and (".bn" in name) and leaf == "weight":
        v = 1.0 + 0.2 * wave
    elif leaf == "v":
        v = wave
    elif len(shape) == 1:
        v = 0.1 * wave
    else:
        fan_in = int(np.prod(shape[1:]))
        v = math.sqrt(3.0 / fan_in) * 0.9 * wave
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def closed_form_state(shapes: Dict[str, tuple]) -> Dict[str, torch.Tensor]:
    return {k: closed_form_tensor(k, shp, i) for i, (k, shp) in enumerate(shapes.items())}


def synthetic_batch(B: int, T: int, V: int, L: int = 22, F_: int = 64, seed: int = 1, ragged: bool = False):
    """Seeded Clotho-shaped batch: feats f32[B,T,F], caps f32[B,L] sorted by
    caption length descending (datasets/caption_dataset.py:278-318), numpy length arrays."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, F_, generator=g)
    if ragged:
        cap_lens = torch.randint(8 if L >= 16 else 3, L + 1, (B,), generator=g).sort(descending=True).values
        cap_lens[0] = L
        feat_lens = (T * (0.5 + 0.5 * torch.rand(B, generator=g))).long()
        feat_lens[torch.randint(0, B, (1,), generator=g)] = T
    else:
        cap_lens = torch.full((B,), L, dtype=torch.long)
        feat_lens = torch.full((B,), T, dtype=torch.long)
    caps = torch.zeros(B, L)
    for b in range(B):
        n = int(cap_lens[b])
        caps[b, 0] = START_IDX
        caps[b, 1:n - 1] = torch.randint(4, V, (n - 2,), generator=g).float()
        caps[b, n - 1] = END_IDX
        feats[b, int(feat_lens[b]):] = 0.0
    return feats, caps, feat_lens.numpy().copy(), cap_lens.numpy().copy()


# ----------------------------------------------------------------------------
# utils/train_util.py:198-231
# ----------------------------------------------------------------------------
def generate_length_mask(lens):
    lens = torch.as_tensor(lens)
    T = int(lens.max())
    return torch.arange(T).unsqueeze(0) < lens.view(-1, 1)


def mean_with_lens(features, lens):
    lens = torch.as_tensor(lens)
    mask = generate_length_mask(lens)
    return (features * mask.unsqueeze(-1)).sum(1) / lens.unsqueeze(1)


def max_with_lens(features, lens):
    mask = generate_length_mask(lens)
    fm = features.clone()
    fm[~mask] = float("-inf")
    return fm.max(1).values


# ----------------------------------------------------------------------------
# A1  Cnn10   models/encoder.py:606-707
# ----------------------------------------------------------------------------
def _dropout(x, p, training, masks, record):
    """F.dropout on CPU == x * bernoulli_(1-p) bool mask * (1/(1-p)) (checked bit-exact).
    `masks`: explicit list consumed in call order; `record`: list the drawn masks are appended to."""
    if not training or p == 0.0:
        return x
    if masks is not None:
        m = masks.pop(0)
    else:
        m = torch.empty_like(x, dtype=torch.bool).bernoulli_(1 - p)
    if record is not None:
        record.append(m.clone())
    return x * m * (1.0 / (1.0 - p))


def _bn(state, p, x, training):
    return F.batch_norm(x, state[p + ".running_mean"], state[p + ".running_var"],
                        state[p + ".weight"], state[p + ".bias"], training, 0.1, 1e-5)


def _bn_track(state, p, training):
    if training:
        state[p + ".num_batches_tracked"] += 1


def _relu_site(z, site, relu_probe, relu_force):
    """F.relu with two test hooks (both None in every normal use): `relu_probe` (a list) receives the pre-activation of
    each site in call order; `relu_force` ({site: bool tensor}) replaces the site's mask z > 0 - used to evaluate the
    oracle under an explicit assignment of the mask bits whose pre-activation is within fp32 rounding distance of 0,
    where two summation orders legitimately disagree (tests: relu_mask_cases)."""
    if relu_probe is not None:
        relu_probe.append(z.detach().clone())
    if relu_force is not None and site in relu_force:
        return z * relu_force[site].to(z.dtype)
    return F.relu(z)


def cnn10_forward(state, feats, feat_lens, training=True, masks=None, record=None, prefix="encoder",
                  mutate_lens=True, relu_probe=None, relu_force=None):
    """models/encoder.py:672-707 (Cnn10) and :906-964 (Cnn14_16k, recognised by its conv_block6 / fc1 entries).
    Returns dict(audio_embeds[N,S,C], audio_embeds_pooled[N,C], audio_embeds_lens i64[N], state None)."""
    arch = ENCODERS["Cnn14_16k" if prefix + ".fc1.weight" in state else "Cnn10"]
    nblocks = len(arch["channels"])
    x = feats.unsqueeze(1)                                     # :676
    lens = torch.as_tensor(feat_lens)
    if not mutate_lens:
        lens = lens.clone()
    lens //= arch["div"]                                       # :678 / :914 (in place: F11)
    x = x.transpose(1, 3)
    x = _bn(state, prefix + ".bn0", x, training); _bn_track(state, prefix + ".bn0", training)
    x = x.transpose(1, 3)
    for b in range(1, nblocks + 1):                            # :683-690 / :928-939, ConvBlock.forward :633-649
        p = f"{prefix}.conv_block{b}"
        x = F.conv2d(x, state[p + ".conv1.weight"], None, 1, 1)
        x = _relu_site(_bn(state, p + ".bn1", x, training), 2 * b - 2, relu_probe, relu_force)
        _bn_track(state, p + ".bn1", training)
        x = F.conv2d(x, state[p + ".conv2.weight"], None, 1, 1)
        x = _relu_site(_bn(state, p + ".bn2", x, training), 2 * b - 1, relu_probe, relu_force)
        _bn_track(state, p + ".bn2", training)
        if b < nblocks or arch["pool_last"]:
            x = F.avg_pool2d(x, kernel_size=(2, 2))
        else:
            x = F.avg_pool2d(x, kernel_size=(1, 1))            # Cnn14_16k block 6, :938
        x = _dropout(x, 0.2, training, masks, record)
    x = torch.mean(x, dim=3)                                   # :691  [N,512,S]
    x1 = torch.max(x, dim=2).values                            # :693 (unmasked)
    x2 = torch.mean(x, dim=2)
    out = _dropout(x1 + x2, 0.5, training, masks, record)
    head = f"{prefix}.{arch['head']}"
    out = F.relu(F.linear(out, state[head + ".weight"], state[head + ".bias"]))
    emb = _dropout(out, 0.5, training, masks, record)
    return {"audio_embeds": x.transpose(1, 2).contiguous(), "audio_embeds_pooled": emb,
            "state": None, "audio_embeds_lens": lens}


# ----------------------------------------------------------------------------
# A3  Seq2SeqAttention   models/attn_model.py:20-46
# ----------------------------------------------------------------------------
def seq2seq_attention(state, prefix, h_dec, h_enc, src_lens):
    N, S, _ = h_enc.shape
    hd = h_dec.unsqueeze(1).repeat(1, S, 1)
    attn_out = torch.tanh(F.linear(torch.cat((hd, h_enc), dim=-1),                 # cat order [h_dec; h_enc] :31
                                   state[prefix + ".h2attn.weight"], state[prefix + ".h2attn.bias"]))
    score = (attn_out @ state[prefix + ".v"])                                        # [N,S]
    mask = torch.arange(S).unsqueeze(0) < torch.as_tensor(src_lens).view(-1, 1)
    score = score.masked_fill(~mask, -1e10)                                          # :41
    weights = torch.softmax(score, dim=-1)
    ctx = (weights.unsqueeze(1) @ h_enc).squeeze(1)
    return ctx, weights


# ----------------------------------------------------------------------------
# recurrent cells (torch.nn.GRU / LSTM formulas, gate order r,z,n / i,f,g,o)
# ----------------------------------------------------------------------------
def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    H = h.shape[-1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    H = h.shape[-1]
    i = torch.sigmoid(g[:, :H]); f = torch.sigmoid(g[:, H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H]); o = torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


# ----------------------------------------------------------------------------
# A2  PosteriorRNN_hybrid   models/text_encoder.py:182-216
# ----------------------------------------------------------------------------
def posterior_hybrid_forward(state, caps, cap_lens, eps=None, prefix="qnet"):
    lengths = torch.as_tensor(np.asarray(cap_lens)) - 1                      # :186
    x = F.embedding(caps[:, :-1].long(), state[prefix + ".word_embedding.weight"])
    N, Tfull, _ = x.shape
    Tc = int(lengths.max())                                                  # pad_packed_sequence length
    Hq = state[prefix + ".network.weight_hh_l0"].shape[1]
    outs = []
    for suf, order in (("", range(Tc)), ("_reverse", range(Tc - 1, -1, -1))):
        w_ih, w_hh = state[prefix + ".network.weight_ih_l0" + suf], state[prefix + ".network.weight_hh_l0" + suf]
        b_ih, b_hh = state[prefix + ".network.bias_ih_l0" + suf], state[prefix + ".network.bias_hh_l0" + suf]
        h = x.new_zeros(N, Hq)
        out = [None] * Tc
        for t in order:                                                       # packed semantics: rows with t>=len frozen, output 0
            valid = (lengths > t).unsqueeze(1)
            hn = gru_cell(x[:, t], h, w_ih, w_hh, b_ih, b_hh)
            h = torch.where(valid, hn, h)
            out[t] = torch.where(valid, hn, torch.zeros_like(hn))
        outs.append(torch.stack(out, 1))
    hidden_o = torch.cat(outs, dim=-1)                                        # [N,Tc,2Hq]
    ml = F.linear(hidden_o, state[prefix + ".token_mean_log.weight"], state[prefix + ".token_mean_log.bias"])
    E = ml.shape[-1] // 2
    means, logs = ml[:, :, :E], ml[:, :, E:]
    if eps is None:
        eps = torch.randn(means.shape)                                        # :196 CPU generator (F9)
    z = eps * torch.exp(.5 * logs) + means
    hidden = mean_with_lens(hidden_o, lengths) + max_with_lens(hidden_o, lengths)   # :199-201
    return {"q_means": means, "q_logs": logs, "q_z": z, "q_means_utt": hidden, "q_logs_utt": None,
            "q_z_utt": None, "_eps": eps}


# ----------------------------------------------------------------------------
# A4  PriorRNN   models/text_encoder.py:247-268 ; A5  VAERNNBahdanauAttnDecoder  models/decoder.py:175-203
# ----------------------------------------------------------------------------
def prior_step(state, word, enc_mem, hc, last_z, lens, eps=None, prefix="pnet"):
    x = F.embedding(word.long(), state[prefix + ".word_embedding.weight"]).squeeze(1)
    ctx, _ = seq2seq_attention(state, prefix + ".word_attn", x, enc_mem, lens)
    h, c = lstm_cell(torch.cat([x, ctx, last_z], dim=-1), hc[0], hc[1],
                     state[prefix + ".network.weight_ih_l0"], state[prefix + ".network.weight_hh_l0"],
                     state[prefix + ".network.bias_ih_l0"], state[prefix + ".network.bias_hh_l0"])
    ml = F.linear(h, state[prefix + ".mean_log_out.weight"], state[prefix + ".mean_log_out.bias"])
    half = ml.shape[-1] // 2
    mean, log = ml[:, :half], ml[:, half:]
    if eps is None:
        eps = torch.randn(mean.shape)                                         # :259
    z = eps * torch.exp(.5 * log) + mean
    return {"mean": mean, "log": log, "hiddens_state": (h, c), "z": z, "_eps": eps}


def decoder_step(state, word, h, enc_mem, enc_mem_lens, z, prefix="decoder", dropout_p=0.0, training=False,
                 keep=None, record=None):
    """VAERNNBahdanauAttnDecoder.forward, models/decoder.py:175-203.  `dropout_p` / `training`: the word-embedding
    nn.Dropout of :33,184 (default 0.0); `keep` replays its mask [N,E], `record` (a list) receives the drawn one."""
    if prefix + ".word_embeddings.0.weight" in state:      # projected pretrained embeddings (decoder.py:58-64)
        emb = F.linear(F.embedding(word.long(), state[prefix + ".word_embeddings.0.weight"]),
                       state[prefix + ".word_embeddings.1.weight"], state[prefix + ".word_embeddings.1.bias"]).squeeze(1)
    else:
        emb = F.embedding(word.long(), state[prefix + ".word_embeddings.weight"]).squeeze(1)
    if training and dropout_p > 0.0:
        if keep is None:
            keep = torch.empty(emb.shape, dtype=torch.bool).bernoulli_(1 - dropout_p)
        if record is not None:
            record.append(keep.clone())
        emb = emb * keep * (1.0 / (1.0 - dropout_p))
    ctx, w = seq2seq_attention(state, prefix + ".attn", h, enc_mem, enc_mem_lens)
    rnn_input = torch.cat((emb, ctx, z), dim=-1)
    h2 = gru_cell(rnn_input, h, state[prefix + ".model.weight_ih_l0"], state[prefix + ".model.weight_hh_l0"],
                  state[prefix + ".model.bias_ih_l0"], state[prefix + ".model.bias_hh_l0"])
    logits = F.linear(h2, state[prefix + ".classifier.weight"], state[prefix + ".classifier.bias"])
    return {"state": h2, "output": h2, "logits": logits, "weights": w, "rnn_input": rnn_input}


# ----------------------------------------------------------------------------
# A6/A7/A12  Hybrid_VAEModel   models/vae_model.py:700-894 ; word_model.py:173-207
# ----------------------------------------------------------------------------
def sample_next_word(logits, method="greedy", temp=1, noise=None):
    """CaptionModel.sample_next_word, models/word_model.py:173-207.  `noise` (optional) replays the branch's draw:
    the Gumbel noise g [N,V] (:189-191) or, for the multinomial branch, the Exp(1) tensor q [N,V] that
    torch.multinomial(prob, 1) draws internally (argmax of prob / q: ATen's one-sample path).  Returns
    (w_t, logprob of w_t, the noise used)."""
    logprobs = torch.log_softmax(logits, dim=1)
    if method == "greedy":
        lp, w_t = torch.max(logprobs, 1)
        return w_t.detach().long(), lp, None
    if method == "gumbel":
        if noise is None:
            U = torch.rand(logprobs.size())
            noise = -torch.log(-torch.log(U + 1e-20) + 1e-20)
        _logprob = torch.log_softmax((logprobs + noise) / temp, dim=-1)
        _, w_t = torch.max(_logprob.data, 1)
        return w_t.detach().long(), logprobs.gather(1, w_t.unsqueeze(-1)).squeeze(1), noise
    prob_prev = torch.exp(logprobs / temp)
    if noise is None:
        noise = torch.empty_like(prob_prev).exponential_(1)
    w_t = torch.argmax(prob_prev / noise, dim=-1, keepdim=True)           # == torch.multinomial(prob_prev, 1)
    return w_t.view(-1).detach().long(), logprobs.gather(1, w_t).squeeze(1), noise


def _embed_size(state):
    """embed_size of the decoder / prior (models/vae_model.py:676: mean_log_out = Linear(E, 2E)); with projected pretrained
    embeddings (Sequential(Embedding, Linear)) the Embedding's width is NOT it, and the GRU's hidden size need not be."""
    return state["mean_log_out.weight"].shape[1]


def hybrid_forward(state, feats, feat_lens, caps=None, cap_lens=None, *, ss_ratio=1.0, dis_ratio=0,
                   training=True, method="greedy", temp=1, max_length=MAX_LENGTH, noise=None, record=None,
                   mutate_lens=True, dec_dropout=0.0):
    """4-input form = train_forward, 2-input form = inference_forward(greedy).
    `noise` (optional): dict(dropout=[masks...], eps_q=[N,Tc,E], eps_p=[Tc,N,E]) to replay; else drawn
    from torch's CPU generator in the reference's call order.  `record` receives the drawn noise."""
    masks = list(noise["dropout"]) if noise is not None and "dropout" in noise else None
    rec_masks: List[torch.Tensor] = []
    relu_probe = [] if record is not None else None
    enc = cnn10_forward(state, feats, feat_lens, training, masks, rec_masks, mutate_lens=mutate_lens,
                        relu_probe=relu_probe, relu_force=None if noise is None else noise.get("relu_force"))
    if "ln.weight" in state:                                                  # vae_model.py:743-744
        enc["audio_embeds"] = F.linear(enc["audio_embeds"], state["ln.weight"], state["ln.bias"])
    mem, mem_lens = enc["audio_embeds"], enc["audio_embeds_lens"]
    N = mem.shape[0]
    E = _embed_size(state)
    H = state["decoder.model.weight_hh_l0"].shape[1]
    out: Dict[str, object] = {}
    train = caps is not None
    if train:
        q = posterior_hybrid_forward(state, caps, cap_lens, None if noise is None else noise["eps_q"])
        out.update({k: q[k] for k in ("q_means", "q_logs", "q_z", "q_means_utt", "q_logs_utt")})
        steps = int(max(cap_lens)) - 1                                        # :703
    else:
        steps = max_length
    seqs = torch.full((N, steps), END_IDX, dtype=torch.long)                  # prepare_output :762-790
    logits, outputs, slp, attw = [], [], [], []
    p_means, p_logs, p_z, eps_p, sample_noise, dec_keep = [], [], [], [], [], []
    h = mem.new_zeros(N, H)
    hc = (mem.new_zeros(N, E), mem.new_zeros(N, E))                           # PriorRNN.init_hidden :240-245
    last_z = mem.new_zeros(N, E)
    unfinished = None
    for t in range(steps):
        if train and random.random() < ss_ratio:                              # :826
            word = caps[:, t].long()
        elif t == 0:
            word = torch.full((N,), START_IDX, dtype=torch.long)
        else:
            word = seqs[:, t - 1]
        e = None if noise is None else noise["eps_p"][t]
        pr = prior_step(state, word.unsqueeze(1), mem, hc, last_z, mem_lens, e)
        eps_p.append(pr["_eps"])
        if train:                                                             # :800-808
            z = q["q_z"][:, t, :]
            if dis_ratio != 0 and torch.rand(1) <= dis_ratio:
                z = pr["z"]
        else:
            z = pr["z"]
        dk = None if noise is None or noise.get("dec_keep") is None else noise["dec_keep"][t]
        d = decoder_step(state, word.unsqueeze(1), h, mem, mem_lens, z, dropout_p=dec_dropout, training=training,
                         keep=dk, record=dec_keep)
        sn = None if noise is None or noise.get("sample_noise") is None else noise["sample_noise"][t]
        w_t, lp, sn_used = sample_next_word(d["logits"], method, temp, sn)     # word_model.py:173-207
        if sn_used is not None:
            sample_noise.append(sn_used)
        seqs[:, t] = w_t
        logits.append(d["logits"]); outputs.append(d["output"]); slp.append(lp); attw.append(d["weights"])
        p_means.append(pr["mean"]); p_logs.append(pr["log"]); p_z.append(pr["z"])
        h, hc, last_z = d["state"], pr["hiddens_state"], pr["z"]
        if not train:                                                         # :711-720
            unfinished_t = seqs[:, t] != END_IDX
            unfinished = unfinished_t if t == 0 else unfinished * unfinished_t
            seqs[:, t][~unfinished] = END_IDX
            if unfinished.sum() == 0:
                break
    nst = len(logits)
    out["seqs"] = seqs
    out["logits"] = torch.stack(logits, 1); out["outputs"] = torch.stack(outputs, 1)
    out["sampled_logprobs"] = torch.stack(slp, 1); out["attn_weights"] = torch.stack(attw, 2)
    out["p_means"] = torch.stack(p_means, 1); out["p_logs"] = torch.stack(p_logs, 1); out["p_z"] = torch.stack(p_z, 1)
    out["state"], out["hiddens_state"], out["last_z"] = h, hc, last_z
    out["_steps_run"] = nst
    if train:
        lens1 = torch.as_tensor(np.asarray(cap_lens)) - 1
        hidden = mean_with_lens(out["outputs"], lens1) + max_with_lens(out["outputs"], lens1)   # :722-725
        out["p_means_utt"] = F.linear(hidden, state["mean_log_out.weight"], state["mean_log_out.bias"])
        out["p_logs_utt"] = None
    out["audio_embeds"], out["audio_embeds_pooled"], out["audio_embeds_lens"] = mem, enc["audio_embeds_pooled"], mem_lens
    if record is not None:
        record["dropout"] = rec_masks
        record["eps_q"] = q["_eps"] if train else None
        record["eps_p"] = torch.stack(eps_p, 0)
        record["relu_z"] = relu_probe
        record["dec_keep"] = torch.stack(dec_keep, 0) if dec_keep else None
        record["sample_noise"] = torch.stack(sample_noise, 0) if sample_noise else None
    return out


# ----------------------------------------------------------------------------
# N1  Hybrid_VAEModel.beam_search   models/vae_model.py:896-995 (validation, beam_size=3)
# ----------------------------------------------------------------------------
def beam_search(state, feats, feat_lens, beam_size=3, max_length=MAX_LENGTH, eps=None):
    """Instance-by-instance beam search; returns seqs i64 [N,max_length] (beam 0 of each clip; `done_beams` is
    never filled in the reference, :986-995).  eps (optional): [N, max_length, beam, E] replay of the randn draws."""
    enc = cnn10_forward(state, feats, feat_lens, training=False)
    if "ln.weight" in state:
        enc["audio_embeds"] = F.linear(enc["audio_embeds"], state["ln.weight"], state["ln.bias"])
    mem_all, lens_all = enc["audio_embeds"], enc["audio_embeds_lens"]
    N = mem_all.shape[0]
    E = _embed_size(state)
    H = state["decoder.model.weight_hh_l0"].shape[1]
    V = state["decoder.classifier.weight"].shape[0]
    seqs_out = torch.full((N, max_length), END_IDX, dtype=torch.long)
    for i in range(N):
        mem = mem_all[i].unsqueeze(0).repeat(beam_size, 1, 1)
        lens = lens_all[i].repeat(beam_size)
        h = mem.new_zeros(beam_size, H)
        hc = (mem.new_zeros(beam_size, E), mem.new_zeros(beam_size, E))
        last_z = mem.new_zeros(beam_size, E)
        top_k_logprobs = mem.new_zeros(beam_size)
        seqs = None
        for t in range(max_length):
            if t == 0:
                w = torch.full((beam_size,), START_IDX, dtype=torch.long)
            else:
                w = next_w
                h = h[prev]; hc = (hc[0][prev], hc[1][prev]); last_z = last_z[prev]
            pr = prior_step(state, w.unsqueeze(1), mem, hc, last_z, lens, None if eps is None else eps[i, t])
            d = decoder_step(state, w.unsqueeze(1), h, mem, lens, pr["z"])
            logprobs = torch.log_softmax(d["logits"], dim=1)
            logprobs = top_k_logprobs.unsqueeze(1).expand_as(logprobs) + logprobs
            top_k_logprobs, top_k_words = logprobs.view(-1).topk(beam_size, 0, True, True)
            prev = torch.div(top_k_words, V, rounding_mode="trunc")
            next_w = top_k_words % V
            seqs = next_w.unsqueeze(1) if t == 0 else torch.cat([seqs[prev], next_w.unsqueeze(1)], dim=1)
            h, hc, last_z = d["state"], pr["hiddens_state"], pr["z"]
        seqs_out[i] = seqs[0]
    return seqs_out


# ----------------------------------------------------------------------------
# N3  CaptionModel.diverse_beam_search   models/word_model.py:297-394 with the Hybrid_VAEModel hooks
#     (prepare_dbs_decoder_input / dbs_step / dbs_process_step, models/vae_model.py:997-1040)
# ----------------------------------------------------------------------------
def diverse_beam_search(state, feats, feat_lens, beam_size=5, group_size=5, diversity_lambda=0.5, temperature=1.0,
                        group_nbest=True, max_length=MAX_LENGTH):
    """Group g of a clip runs one step behind group g-1 (global step t = local step + g); at a local step its
    log-probabilities are lowered by lambda x (how often the earlier groups chose each word at that local step).
    Per group `bdash = beam_size // group_size` beams; finished beams are scored by logprob / length.  Returns seqs
    i64 [N, beam_size (group_nbest) or group_size, max_length], <end>-filled.  The randn draws of the prior come
    from torch's CPU generator in call order (clip, t, group)."""
    enc = cnn10_forward(state, feats, feat_lens, training=False)
    if "ln.weight" in state:
        enc["audio_embeds"] = F.linear(enc["audio_embeds"], state["ln.weight"], state["ln.bias"])
    mem_all, lens_all = enc["audio_embeds"], enc["audio_embeds_lens"]
    N = mem_all.shape[0]
    E = _embed_size(state)
    H = state["decoder.model.weight_hh_l0"].shape[1]
    V = state["decoder.classifier.weight"].shape[0]
    bdash = beam_size // group_size
    out = torch.full((N, beam_size if group_nbest else group_size, max_length), END_IDX, dtype=torch.long)
    for i in range(N):
        mem = mem_all[i].unsqueeze(0).repeat(bdash, 1, 1)
        lens = lens_all[i].repeat(bdash)
        seq = [torch.zeros(bdash, 0, dtype=torch.long) for _ in range(group_size)]
        score = [mem.new_zeros(bdash) for _ in range(group_size)]
        done = [[] for _ in range(group_size)]
        carry = [None] * group_size          # (h, (hp, cp), last_z, next_word, parent beam) of each group
        for t in range(max_length + group_size - 1):
            for g in range(group_size):
                lt = t - g
                if lt < 0 or lt > max_length - 1:
                    continue
                if lt == 0:
                    w = torch.full((bdash,), START_IDX, dtype=torch.long)
                    h = mem.new_zeros(bdash, H)
                    hc = (mem.new_zeros(bdash, E), mem.new_zeros(bdash, E))
                    last_z = mem.new_zeros(bdash, E)
                else:
                    h0, hc0, z0, w, parent = carry[g]
                    h, hc, last_z = h0[parent], (hc0[0][parent], hc0[1][parent]), z0[parent]
                pr = prior_step(state, w.unsqueeze(1), mem, hc, last_z, lens)
                d = decoder_step(state, w.unsqueeze(1), h, mem, lens, pr["z"])
                lp = torch.log_softmax(d["logits"], dim=1)
                lp = torch.log_softmax(lp / temperature, dim=1)
                if g > 0:                                            # add_diversity, word_model.py:298-312
                    counts = torch.zeros(V)
                    for earlier in range(g):
                        for b in range(bdash):
                            counts[seq[earlier][b, lt]] += 1
                    lp = lp - counts.unsqueeze(0) * diversity_lambda
                lp = score[g].unsqueeze(1) + lp
                flat = lp[0] if lt == 0 else lp.reshape(-1)
                top, words = flat.topk(bdash, 0, True, True)
                score[g] = top
                parent = torch.div(words, V, rounding_mode="floor")
                nxt = words % V
                seq[g] = torch.cat([seq[g][parent] if lt > 0 else seq[g], nxt.unsqueeze(1)], dim=1)
                ended = seq[g][:, lt] == END_IDX
                if t == max_length + g - 1:
                    ended[:] = True
                for b in range(bdash):
                    if ended[b]:
                        done[g].append({"seq": seq[g][b].clone(), "score": score[g][b].item() / (lt + 1)})
                score[g][ended] -= 1000
                carry[g] = (d["state"], pr["hiddens_state"], pr["z"], nxt, parent)
        done = [sorted(d_, key=lambda x: -x["score"])[:bdash] for d_ in done]
        chosen = sum(done, []) if group_nbest else [d_[0] for d_ in done]
        for r, beam in enumerate(chosen):
            out[i, r, :len(beam["seq"])] = beam["seq"]
    return out


# ----------------------------------------------------------------------------
# A8/A9/A13 losses ; A10 loss assembly + optimiser
# ----------------------------------------------------------------------------
def label_smoothing_loss(logit, target, classes, smoothing):
    """utils/train_util.py:243-251 (packed rows)."""
    pred = logit.log_softmax(dim=-1)
    true_dist = torch.full_like(pred, smoothing / (classes - 1))
    true_dist.scatter_(1, target.unsqueeze(1).long(), 1.0 - smoothing)
    return torch.mean(torch.sum(-true_dist * pred, dim=-1))


def normal_kl_loss(mu1, lv1, mu2, lv2):
    """utils/train_util.py:259-266 (unmasked mean over N*T: F8)."""
    kl = lv2 / 2. - lv1 / 2. + ((torch.exp(lv1) + (mu1 - mu2) ** 2.) / (2. * torch.exp(lv2))) - .5
    return kl.sum(-1).mean()


def masked_ce(logits, targets, lens, smoothing=0.0, reduction="mean"):
    """losses/loss.py:18-37 (smoothing==0 -> CrossEntropyLoss) and :47-70 (LabelSmoothingLoss)."""
    c = logits.size(-1)
    preds = logits.log_softmax(dim=-1)
    if smoothing == 0.0:
        loss = -preds.gather(-1, targets.long().unsqueeze(-1)).squeeze(-1)
    else:
        td = torch.full_like(preds, smoothing / (c - 1))
        td.scatter_(-1, targets.long().unsqueeze(-1), 1.0 - smoothing)
        loss = torch.sum(-td * preds, dim=-1)
    mask = generate_length_mask(lens)
    loss = loss * mask
    if reduction == "none":
        return loss
    return loss.sum() / mask.sum() if reduction == "mean" else loss.sum()


def pack_rows(x, lens1):
    """Rows of pack_padded_sequence(x, lens1, batch_first=True).data (time-major over valid rows);
    runners/pytorch_runner_vae.py:89-95."""
    lens1 = np.asarray(lens1)
    return torch.cat([x[:int((lens1 > t).sum()), t] for t in range(int(lens1.max()))], 0)


def train_loss(out, caps, cap_lens, vocab, smoothing=0.1, kl_weight=0.5, alpha=1.0):
    """runners/pytorch_runner_vae.py:315-318."""
    lens1 = np.asarray(cap_lens) - 1
    ce = label_smoothing_loss(pack_rows(out["logits"], lens1), pack_rows(caps[:, 1:], lens1), vocab, smoothing)
    kl = normal_kl_loss(out["q_means"], out["q_logs"], out["p_means"], out["p_logs"])
    loss = ce + kl_weight * kl
    mse = None
    if alpha is not None:
        mse = F.mse_loss(out["q_means_utt"], out["p_means_utt"])
        loss = loss + alpha * mse
    return loss, ce, kl, mse


def relu_mask_cases(zs, tau=2e-6, max_bits=16, max_flips=3):
    """The ReLU mask assignments two correct fp32 implementations may legitimately produce for the pre-activations
    `zs` (list of tensors, one per ReLU site): a bit whose |z| < tau (z is BatchNorm output, O(1); two summation orders
    of the convolution move it by a few 1e-7) can fall on either side of 0, every other bit is z > 0.  Returns
    (lazy iterator over {site: mask} dicts, number of ambiguous bits): the natural assignment first, then every
    assignment that flips 1, 2, .. max_flips of the ambiguous bits."""
    import itertools
    base = {i: (z > 0) for i, z in enumerate(zs)}
    amb = [(i, int(j)) for i, z in enumerate(zs) for j in torch.nonzero(z.abs().flatten() < tau).flatten()]
    if len(amb) > max_bits:
        raise ValueError(f"{len(amb)} pre-activations within {tau} of zero: choose other test data")

    def gen():
        for nflip in range(min(len(amb), max_flips) + 1):
            for which in itertools.combinations(range(len(amb)), nflip):
                m = {i: v.clone() for i, v in base.items()}
                for a in which:
                    site, j = amb[a]
                    m[site].view(-1)[j] = ~m[site].view(-1)[j]
                yield m
    return gen(), len(amb)


def trainable_keys(state):
    return [k for k, v in state.items() if v.dtype.is_floating_point and "running_" not in k]


class OracleTrainer:
    """One optimiser step exactly as runners/pytorch_runner_vae.py:311-324 does it:
    zero_grad, forward, loss, backward, clip_grad_norm_(max_grad_norm), Adam.step."""

    def __init__(self, state, vocab, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 smoothing=0.1, kl_weight=0.5, alpha=1.0, dec_dropout=0.0):
        self.state, self.vocab = state, vocab
        self.dec_dropout = dec_dropout
        self.keys = trainable_keys(state)
        for k in self.keys:
            state[k].requires_grad_(True)
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.smoothing, self.kl_weight, self.alpha = smoothing, kl_weight, alpha
        self.m = {k: torch.zeros_like(state[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(state[k]) for k in self.keys}
        self.t = 0

    def step(self, feats, feat_lens, caps, cap_lens, ss_ratio=1.0, dis_ratio=0, noise=None, record=None,
             apply_update=True):
        st = self.state
        for k in self.keys:
            st[k].grad = None
        out = hybrid_forward(st, feats, np.array(feat_lens).copy(), caps, cap_lens, ss_ratio=ss_ratio,
                             dis_ratio=dis_ratio, training=True, noise=noise, record=record,
                             dec_dropout=self.dec_dropout)
        loss, ce, kl, mse = train_loss(out, caps, cap_lens, self.vocab, self.smoothing, self.kl_weight, self.alpha)
        loss.backward()
        grads = {k: st[k].grad for k in self.keys if st[k].grad is not None}
        raw = {k: g.clone() for k, g in grads.items()}
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)       # clip_grad_norm_
        if apply_update:
            self.t += 1
            b1, b2 = self.betas
            bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
            with torch.no_grad():
                for k, g in grads.items():                                    # torch.optim.Adam (no decay, no amsgrad)
                    g = g * coef
                    self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                    self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                    denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                    st[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)
        return {"loss": loss.detach(), "ce": ce.detach(), "kl": kl.detach(),
                "mse": None if mse is None else mse.detach(), "grad_norm": total, "clip_coef": coef,
                "grads": raw, "out": out}
