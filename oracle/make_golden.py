"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF (build container only).

    python -B oracle/make_golden.py

The reference ships no tests or fixtures, so parity is pinned by running its own
classes here, on closed-form parameters (``acvae_oracle.closed_form_state`` — weights are never
stored) and seeded synthetic inputs, and storing inputs + noise + outputs as small .npz files.
Only data is written: no reference source text, in any encoding, leaves /root/reference.

Noise (dropout masks, eps of both reparameterisations) is drawn by torch's CPU generator inside the
reference; the oracle makes the identical generator calls in the identical order, so running it with
the same seed yields the very same draws, which are what get stored ("noise_*" entries).  The stored
outputs are always the REFERENCE's.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import acvae_oracle as O  # noqa: E402
import ref_shim  # noqa: E402
from check_oracle_vs_reference import load_state_into, ref_train_step  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items() if v is not None})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def pack_masks(masks):
    d = {}
    for i, m in enumerate(masks):
        d[f"noise_drop{i}_bits"] = np.packbits(npy(m).astype(np.uint8).reshape(-1))
        d[f"noise_drop{i}_shape"] = np.array(m.shape)
    return d


def g1_attention(ref):
    """G1: Seq2SeqAttention.forward (models/attn_model.py:20-46)."""
    g = torch.Generator().manual_seed(101)
    out = {}
    for ci, (N, S, E, Hd, A) in enumerate([(3, 4, 32, 32, 32), (4, 31, 64, 48, 40), (5, 62, 512, 512, 512),
                                           (2, 187, 512, 512, 512)]):
        m = ref.attn_model.Seq2SeqAttention(E, Hd, A)
        st = O.closed_form_state({"v": (A,), "h2attn.weight": (A, E + Hd), "h2attn.bias": (A,)})
        m.load_state_dict(st)
        h_dec = torch.randn(N, Hd, generator=g); h_enc = torch.randn(N, S, E, generator=g)
        lens = torch.randint(1, S + 1, (N,), generator=g); lens[0] = S
        if N > 2:
            lens[1] = 1
        with torch.no_grad():
            ctx, w = m(h_dec, h_enc, lens)
        out.update({f"c{ci}_h_dec": h_dec, f"c{ci}_h_enc": h_enc, f"c{ci}_lens": lens, f"c{ci}_ctx": ctx,
                    f"c{ci}_weights": w, f"c{ci}_dims": np.array([N, S, E, Hd, A])})
    out["ncases"] = np.array(4)
    save("g1_attention", **out)


def g2_reparam_kl(ref):
    """G2: eps*exp(.5*logvar)+mu (models/text_encoder.py:196-197,259-262) and Normal_kl_loss
    (utils/train_util.py:259-266) incl. the unmasked padded rows (F8)."""
    g = torch.Generator().manual_seed(102)
    N, T, E = 5, 7, 96
    mu_q = torch.randn(N, T, E, generator=g); lv_q = 0.5 * torch.randn(N, T, E, generator=g)
    mu_p = torch.randn(N, T, E, generator=g); lv_p = 0.5 * torch.randn(N, T, E, generator=g)
    eps = torch.randn(N, T, E, generator=g)
    z = eps * torch.exp(.5 * lv_q) + mu_q
    kl = ref.train_util.Normal_kl_loss(device="cpu")(mu_q, lv_q, mu_p, lv_p)
    save("g2_reparam_kl", mu_q=mu_q, lv_q=lv_q, mu_p=mu_p, lv_p=lv_p, eps=eps, z=z, kl=kl)


def g3_ce(ref):
    """G3: LabelSmoothingLoss (utils/train_util.py:243-251) on packed rows and torch CrossEntropyLoss
    (runner :222-227); ragged cap_lens.  The masked [bs,max_len,C] forms of losses/loss.py:18-70 are pinned
    by the reference itself in G3b (g3b_masked_losses)."""
    g = torch.Generator().manual_seed(103)
    N, T, V = 6, 9, 257
    logits = 3 * torch.randn(N, T, V, generator=g)
    lens1 = torch.tensor([9, 9, 7, 4, 2, 1])
    targets = torch.randint(0, V, (N, T), generator=g)
    pk = torch.nn.utils.rnn.pack_padded_sequence(logits, lens1, batch_first=True).data
    tg = torch.nn.utils.rnn.pack_padded_sequence(targets.float(), lens1, batch_first=True).data
    ls = ref.train_util.LabelSmoothingLoss(V, smoothing=0.1, device="cpu")(pk, tg)
    ls0 = ref.train_util.LabelSmoothingLoss(V, smoothing=0.0, device="cpu")(pk, tg)
    ce = torch.nn.CrossEntropyLoss()(pk, tg.long())
    save("g3_ce", logits=logits, targets=targets, lens1=lens1, ls_packed=ls, ls0_packed=ls0, ce_packed=ce)


def g3b_masked_losses():
    """G3b: the reference's own ``losses/loss.py`` classes - CrossEntropyLoss (:12-37) and
    LabelSmoothingLoss (:39-70) on the dict ``{"logits" [bs,max_len,C], "targets" [bs,max_len], "lens" [bs]}`` - for
    reduction none / mean / sum, ragged lens.  Values AND input gradients (d loss / d logits of the mean and sum
    forms) are the reference's."""
    L = ref_shim.load_losses()
    g = torch.Generator().manual_seed(113)
    N, T, V = 6, 9, 257
    logits = 3 * torch.randn(N, T, V, generator=g)
    lens = torch.tensor([9, 9, 7, 4, 2, 1])
    targets = torch.randint(0, V, (N, T), generator=g)
    out = dict(logits=logits, targets=targets, lens=lens)
    for name, mk in (("ce", lambda red: L.CrossEntropyLoss(reduction=red)),
                     ("ls", lambda red: L.LabelSmoothingLoss(smoothing=0.1, reduction=red))):
        for red in ("none", "mean", "sum"):
            x = logits.clone().requires_grad_(True)
            val = mk(red)({"logits": x, "targets": targets, "lens": lens})
            out[f"{name}_{red}"] = val.detach().clone()
            if red != "none":
                val.backward()
                out[f"{name}_{red}_dlogits"] = x.grad.clone()
    save("g3b_masked_losses", **out)


def g4_encoder(ref):
    """G4: Cnn10.forward (models/encoder.py:672-707), train mode (batch-stat BN + dropout, running-stat
    update) and eval mode, tiny T."""
    out = {}
    for ci, (B, T) in enumerate([(2, 32), (3, 64), (2, 80)]):
        shapes = {k: v for k, v in O.state_shapes(10).items() if k.startswith("encoder.")}
        full = O.closed_form_state(O.state_shapes(10))
        st = {k[len("encoder."):]: full[k].clone() for k in shapes}
        m = ref.encoder.Cnn10(64, 512)
        m.load_state_dict(st)
        g = torch.Generator().manual_seed(104 + ci)
        feats = torch.randn(B, T, 64, generator=g) * 2 + 0.5
        lens = np.array([T] + [int(T * 0.7)] * (B - 1))
        m.train()
        torch.manual_seed(40 + ci)
        with torch.no_grad():
            r = m(feats, lens.copy())
        ost = {k: full[k].clone() for k in shapes}
        rec = []
        torch.manual_seed(40 + ci)
        with torch.no_grad():
            o = O.cnn10_forward(ost, feats, lens.copy(), True, None, rec)
        assert (r["audio_embeds"] - o["audio_embeds"]).abs().max() < 1e-5
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        out.update({f"c{ci}_feats": feats, f"c{ci}_lens": lens, f"c{ci}_train_audio_embeds": r["audio_embeds"],
                    f"c{ci}_train_pooled": r["audio_embeds_pooled"], f"c{ci}_train_lens": r["audio_embeds_lens"],
                    f"c{ci}_bn0_running_mean": sd["bn0.running_mean"], f"c{ci}_bn0_running_var": sd["bn0.running_var"],
                    f"c{ci}_b4bn2_running_mean": sd["conv_block4.bn2.running_mean"],
                    f"c{ci}_b4bn2_running_var": sd["conv_block4.bn2.running_var"],
                    f"c{ci}_b1bn1_running_var": sd["conv_block1.bn1.running_var"],
                    f"c{ci}_nbt": sd["bn0.num_batches_tracked"]})
        out.update({f"c{ci}_{k}": v for k, v in pack_masks(rec).items()})
        m.load_state_dict(st)
        m.eval()
        with torch.no_grad():
            r = m(feats, lens.copy())
        out.update({f"c{ci}_eval_audio_embeds": r["audio_embeds"], f"c{ci}_eval_pooled": r["audio_embeds_pooled"]})
    out["ncases"] = np.array(3)
    save("g4_encoder", **out)


def g5_rnn(ref):
    """G5: single GRU / LSTM step as the decoder / prior use torch.nn.GRU / LSTM (models/decoder.py:39-44,
    models/text_encoder.py:229-235) and the packed BiGRU posterior (text_encoder.py:182-216)."""
    g = torch.Generator().manual_seed(105)
    N, I, H = 5, 48, 32
    gru = torch.nn.GRU(I, H, batch_first=True); lstm = torch.nn.LSTM(I, H, batch_first=True)
    gs = O.closed_form_state({k: tuple(v.shape) for k, v in gru.state_dict().items()})
    ls = O.closed_form_state({k: tuple(v.shape) for k, v in lstm.state_dict().items()})
    gru.load_state_dict(gs); lstm.load_state_dict(ls)
    x = torch.randn(N, 1, I, generator=g); h = torch.randn(1, N, H, generator=g); c = torch.randn(1, N, H, generator=g)
    with torch.no_grad():
        _, gh = gru(x, h)
        _, (lh, lc) = lstm(x, (h, c))
    V, E = 30, 32
    q = ref.text_encoder.PosteriorRNN_hybrid(word_dim=E, embed_size=E, vocab_size=V, hidden_size=E, dropout=0.0)
    qs = O.closed_form_state({k: tuple(v.shape) for k, v in q.state_dict().items()})
    q.load_state_dict(qs)
    cap_lens = np.array([9, 7, 7, 4, 2])
    caps = torch.zeros(5, 9)
    for b, n in enumerate(cap_lens):
        caps[b, :n] = torch.randint(1, V, (int(n),), generator=g).float()
    torch.manual_seed(55)
    with torch.no_grad():
        qo = q(caps, cap_lens)
    torch.manual_seed(55)
    eps = torch.randn(qo["q_means"].shape)
    save("g5_rnn", x=x, h=h, c=c, gru_h=gh, lstm_h=lh, lstm_c=lc, caps=caps, cap_lens=cap_lens, eps=eps,
         q_means=qo["q_means"], q_logs=qo["q_logs"], q_z=qo["q_z"], q_means_utt=qo["q_means_utt"],
         dims=np.array([N, I, H, V, E]))


GRAD_KEYS = ("encoder.bn0.weight", "encoder.conv_block1.conv1.weight", "encoder.conv_block2.conv2.weight",
             "encoder.conv_block4.bn2.bias", "decoder.attn.v", "decoder.attn.h2attn.weight",
             "decoder.model.weight_ih_l0", "decoder.classifier.bias", "decoder.word_embeddings.weight",
             "pnet.mean_log_out.weight", "pnet.network.weight_hh_l0", "pnet.word_attn.h2attn.bias",
             "qnet.network.weight_ih_l0_reverse", "qnet.token_mean_log.weight", "mean_log_out.weight", "ln.weight")
OUT_KEYS = ("logits", "outputs", "seqs", "sampled_logprobs", "attn_weights", "p_means", "p_logs", "p_z",
            "q_means", "q_logs", "q_z", "q_means_utt", "p_means_utt")


def train_fixture(ref, name, B, T, V, E, L, ragged, dis, seed, keep_tensors=True, keep_noise=True, encoder="Cnn10",
                  dec_dropout=0.0, proj_embed=None):
    shapes = O.state_shapes(V, E, E, None, E, 512 if encoder == "Cnn10" else 2048, encoder=encoder, proj_embed=proj_embed)
    state = O.closed_form_state(shapes)
    model = ref_shim.build_reference_model(ref, V, E, E, encoder=encoder, dec_dropout=dec_dropout, proj_embed=proj_embed)
    load_state_into(model, state)
    model.train()
    feats, caps, feat_lens, cap_lens = O.synthetic_batch(B, T, V, L, seed=seed, ragged=ragged)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    torch.manual_seed(seed); random.seed(seed)
    rout, rl, rg = ref_train_step(ref, model, feats, feat_lens.copy(), caps, cap_lens, V, 1.0, dis, optimizer=opt)
    ostate = {k: v.clone() for k, v in state.items()}
    rec = {}
    torch.manual_seed(seed); random.seed(seed)
    res = O.OracleTrainer(ostate, V, dec_dropout=dec_dropout).step(feats, feat_lens.copy(), caps, cap_lens, 1.0, dis,
                                                                   record=rec)
    assert abs(float(res["loss"]) - float(rl["loss"])) < 1e-4 * max(1, abs(float(rl["loss"]))), (res["loss"], rl["loss"])
    d = dict(dims=np.array([B, T, V, E, L]), dec_dropout=np.array(float(dec_dropout)),
             proj_embed=np.array(0 if proj_embed is None else proj_embed), seed=np.array(seed), ragged=np.array(int(ragged)), dis_ratio=np.array(float(dis)),
             feat_lens=feat_lens, cap_lens=cap_lens, caps=caps,
             loss=rl["loss"], ce=rl["ce"], kl=rl["kl"], mse=rl["mse"], grad_norm=rl["grad_norm"])
    if keep_tensors:
        d["feats"] = feats
        d.update({"out_" + k: rout[k] for k in OUT_KEYS})
        d.update({"grad_" + k: rg[k] for k in GRAD_KEYS if k in rg})
        d.update({"grad_" + k: rg[k] for k in rg if k.startswith("decoder.word_embeddings.") and "." in k[24:]})
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for k in ("encoder.conv_block1.conv1.weight", "decoder.attn.v", "pnet.mean_log_out.bias",
                  "encoder.conv_block3.bn1.running_mean", "encoder.bn0.running_var"):
            d["post_" + k] = sd[k]
    if keep_noise:
        d["noise_eps_q"] = rec["eps_q"]; d["noise_eps_p"] = rec["eps_p"]
        if rec.get("dec_keep") is not None:
            d["noise_dec_keep"] = rec["dec_keep"]
        d.update(pack_masks(rec["dropout"]))
    save(name, **d)
    return model, state, feats, feat_lens


def g7_decode(ref):
    """G7: greedy and N=5 z-samples-per-clip decode (models/vae_model.py:880-894,700-721; replication as
    runners/pytorch_runner_vae.py:101-104), eval mode."""
    V, E = 50, 64
    shapes = O.state_shapes(V, E, E, None, E, 512)
    state = O.closed_form_state(shapes)
    model = ref_shim.build_reference_model(ref, V, E, E)
    load_state_into(model, state)
    model.eval()
    feats, _, feat_lens, _ = O.synthetic_batch(3, 96, V, 8, seed=7, ragged=True)
    d = dict(dims=np.array([3, 96, V, E]), feats=feats, feat_lens=feat_lens)
    for tag, rep in (("greedy1", 1), ("greedy5", 5)):
        f = feats.repeat(rep, 1, 1)
        l_ = [int(x) for x in feat_lens for _ in range(rep)]                  # runner :102-104 (B>1 mismatch kept)
        torch.manual_seed(70 + rep)
        with torch.no_grad():
            ro = model(f, list(l_), method="greedy", beam_size=rep)
        rec = {}
        torch.manual_seed(70 + rep)
        with torch.no_grad():
            oo = O.hybrid_forward({k: v.clone() for k, v in state.items()}, f, list(l_), training=False, record=rec)
        assert torch.equal(ro["seqs"], oo["seqs"])
        d[tag + "_seqs"] = ro["seqs"]; d[tag + "_lens"] = np.array(l_)
        eps = rec["eps_p"]                                                    # [steps_run,N,E]; pad to max_length
        full = torch.zeros(O.MAX_LENGTH, eps.shape[1], eps.shape[2]); full[:eps.shape[0]] = eps
        d[tag + "_noise_eps_p"] = full; d[tag + "_steps_run"] = np.array(eps.shape[0])
        d[tag + "_logits0"] = ro["logits"][:, 0]
    save("g7_decode", **d)


def g14_sampling(ref):
    """G14: the non-greedy branches of sample_next_word (models/word_model.py:188-203) through the
    reference's inference path (models/vae_model.py:880-894): method="sample" (multinomial, several temperatures,
    3 clips at once) and method="gumbel" (the reference's branch only runs with ONE clip per call: with more its
    sampled_logprobs assignment raises, word_model.py:197 / :165 - so one clip per call here).  The fixture keeps the
    reference's token ids and log-probabilities, and the noise the run drew: the prior's eps per step and the [N,V]
    Gumbel / Exp(1) tensor per step (re-drawn here from the same seed in the same order; the oracle replaying them
    must return the reference's tokens, asserted)."""
    V, E = 50, 64
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = ref_shim.build_reference_model(ref, V, E, E)
    load_state_into(model, state)
    model.eval()
    feats, _, feat_lens, _ = O.synthetic_batch(3, 96, V, 8, seed=14, ragged=True)
    d = dict(dims=np.array([3, 96, V, E]), feats=feats, feat_lens=feat_lens)
    cases = [("sample_t1", "sample", 1.0, slice(0, 3)), ("sample_t07", "sample", 0.7, slice(0, 3)),
             ("sample_t15", "sample", 1.5, slice(0, 3)), ("gumbel_t1_c0", "gumbel", 1.0, slice(0, 1)),
             ("gumbel_t05_c1", "gumbel", 0.5, slice(1, 2)), ("gumbel_t2_c2", "gumbel", 2.0, slice(2, 3))]
    for ci, (tag, method, temp, sl) in enumerate(cases):
        l_ = feat_lens[sl].copy()
        f = feats[sl][:, :int(l_.max())]                   # a batch is padded to ITS longest clip (collate_fn)
        torch.manual_seed(140 + ci)
        with torch.no_grad():
            ro = model(f, l_.copy(), method=method, temp=temp)
        rec = {}
        torch.manual_seed(140 + ci)
        with torch.no_grad():
            oo = O.hybrid_forward({k: v.clone() for k, v in state.items()}, f, l_.copy(), training=False, method=method,
                                  temp=temp, record=rec)
        assert torch.equal(ro["seqs"], oo["seqs"]), (tag, ro["seqs"], oo["seqs"])
        n, steps = f.shape[0], rec["eps_p"].shape[0]
        eps = torch.zeros(O.MAX_LENGTH, n, E); eps[:steps] = rec["eps_p"]
        sn = torch.ones(O.MAX_LENGTH, n, V); sn[:steps] = rec["sample_noise"]
        d[tag + "_seqs"] = ro["seqs"]; d[tag + "_logprobs"] = ro["sampled_logprobs"][:, :steps]
        d[tag + "_steps_run"] = np.array(steps); d[tag + "_temp"] = np.array(temp)
        d[tag + "_noise_eps_p"] = eps; d[tag + "_sample_noise"] = sn
        d[tag + "_clips"] = np.array([sl.start, sl.stop])
    d["cases"] = np.array([c[0] for c in cases]); d["methods"] = np.array([c[1] for c in cases])
    save("g14_sampling", **d)


def g9_beam(ref):
    """G9: validation beam search, beam_size=3 (models/vae_model.py:896-995), eval mode."""
    V, E, beam = 50, 64, 3
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = ref_shim.build_reference_model(ref, V, E, E)
    load_state_into(model, state)
    model.eval()
    feats, _, feat_lens, _ = O.synthetic_batch(3, 96, V, 8, seed=9, ragged=True)
    torch.manual_seed(90)
    with torch.no_grad():
        ro = model(feats, feat_lens.copy(), method="beam", beam_size=beam)
    torch.manual_seed(90)                       # the only generator calls are pnet's randn([beam,E]) per clip, per step
    eps = torch.stack([torch.stack([torch.randn(beam, E) for _ in range(O.MAX_LENGTH)]) for _ in range(3)])
    with torch.no_grad():
        oo = O.beam_search({k: v.clone() for k, v in state.items()}, feats, feat_lens.copy(), beam, O.MAX_LENGTH, eps)
    assert torch.equal(ro["seqs"], oo), (ro["seqs"], oo)
    save("g9_beam", dims=np.array([3, 96, V, E, beam]), feats=feats, feat_lens=feat_lens, eps=eps, seqs=ro["seqs"])


DBS_CASES = [dict(beam_size=4, group_size=2), dict(beam_size=6, group_size=3, diversity_lambda=0.8, temperature=1.5,
             group_nbest=False), dict(), dict(beam_size=6, group_size=2, diversity_lambda=2.0)]


def g11_dbs(ref):
    """N3: the reference's diverse beam search (word_model.py:297-394 + vae_model.py:997-1040) on two clips, four
    argument sets; a second weight set raises the <end> logit so that beams finish early.  The prior's randn draws
    come from the CPU generator seeded per case; the oracle must reproduce the token ids exactly."""
    V, E, ML = 40, 64, 7
    g = torch.Generator().manual_seed(111)
    feats = torch.randn(2, 96, 64, generator=g)
    feat_lens = np.array([96, 80])
    out = {}
    for tag, bump in (("a", 0.0), ("b", 0.3)):
        state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
        state["decoder.classifier.bias"] = state["decoder.classifier.bias"].clone()
        state["decoder.classifier.bias"][O.END_IDX] += bump
        model = ref_shim.build_reference_model(ref, V, E, E)
        load_state_into(model, state)
        model.eval()
        for ci, kw in enumerate(DBS_CASES):
            torch.manual_seed(40 + ci)
            with torch.no_grad():
                rs = model(feats, feat_lens.copy(), method="dbs", max_length=ML, **kw)["seqs"]
            torch.manual_seed(40 + ci)
            with torch.no_grad():
                os_ = O.diverse_beam_search({k: v.clone() for k, v in state.items()}, feats, feat_lens.copy(),
                                            max_length=ML, **kw)
            assert torch.equal(rs, os_), (tag, kw)
            out[f"seqs_{tag}{ci}"] = rs
    ends = sum(int((v[..., :-1] == O.END_IDX).any()) for k, v in out.items() if k.startswith("seqs_b"))
    assert ends > 0, "the <end>-biased weights should finish some beams early"
    save("g11_dbs", dims=np.array([2, 96, V, E, ML]), feats=feats, feat_lens=feat_lens, end_bump=np.array([0.0, 0.3]),
         **out)


def g12_cnn14(ref):
    """N4: Cnn14_16k.forward (models/encoder.py:906-964), train mode (batch-stat BN, 8 dropout sites, running-stat
    update) and eval mode, tiny T; the oracle restatement is checked against it on the way."""
    out = {}
    shapes = {k: v for k, v in O.state_shapes(10, enc_embed=2048, encoder="Cnn14_16k").items() if k.startswith("encoder.")}
    full = O.closed_form_state(shapes)
    st = {k[len("encoder."):]: full[k].clone() for k in shapes}
    for ci, (B, T) in enumerate([(2, 64), (3, 96)]):
        m = ref.encoder.Cnn14_16k(64, 2048)
        m.load_state_dict(st)
        g = torch.Generator().manual_seed(124 + ci)
        feats = torch.randn(B, T, 64, generator=g) * 2 + 0.5
        lens = np.array([T] + [int(T * 0.7)] * (B - 1))
        m.train()
        torch.manual_seed(60 + ci)
        with torch.no_grad():
            r = m(feats, lens.copy())
        ost = {k: full[k].clone() for k in shapes}
        rec = []
        torch.manual_seed(60 + ci)
        with torch.no_grad():
            o = O.cnn10_forward(ost, feats, lens.copy(), True, None, rec)
        assert (r["audio_embeds"] - o["audio_embeds"]).abs().max() < 1e-5
        assert (r["audio_embeds_pooled"] - o["audio_embeds_pooled"]).abs().max() < 1e-5
        assert len(rec) == 8 and torch.equal(r["audio_embeds_lens"], o["audio_embeds_lens"])
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        out.update({f"c{ci}_feats": feats, f"c{ci}_lens": lens, f"c{ci}_train_audio_embeds": r["audio_embeds"],
                    f"c{ci}_train_pooled": r["audio_embeds_pooled"], f"c{ci}_train_lens": r["audio_embeds_lens"],
                    f"c{ci}_b6bn2_running_mean": sd["conv_block6.bn2.running_mean"],
                    f"c{ci}_b6bn2_running_var": sd["conv_block6.bn2.running_var"],
                    f"c{ci}_b5bn1_running_var": sd["conv_block5.bn1.running_var"]})
        out.update({f"c{ci}_{k}": v for k, v in pack_masks(rec).items()})
        m.load_state_dict(st)
        m.eval()
        with torch.no_grad():
            r = m(feats, lens.copy())
        out.update({f"c{ci}_eval_audio_embeds": r["audio_embeds"], f"c{ci}_eval_pooled": r["audio_embeds_pooled"]})
    out["ncases"] = np.array(2)
    save("g12_cnn14_encoder", **out)


def g10_host():
    """Batch / evaluation contract, produced by the reference's own host code: collate_fn (caption_dataset.py:278-318)
    on a seeded ragged training batch and an evaluation batch, Vocabulary (build_vocab.py:9-28) pickled, and
    BaseRunner._convert_idx2sentence (base_runner.py:146-157) on id rows with/without <start>/<end>."""
    import pickle
    host = ref_shim.load_host_side()
    g = torch.Generator().manual_seed(110)
    words = ["<pad>", "<start>", "<end>", "<unk>"] + [f"w{i}" for i in range(16)]
    vocab = host.build_vocab.Vocabulary()
    for w in words:
        vocab.add_word(w)
    T = [9, 5, 12, 7]
    L = [5, 8, 3, 8]
    items = []
    for i in range(4):
        feat = torch.randn(T[i], 6, generator=g)
        cap = torch.cat([torch.tensor([1]), torch.randint(4, 20, (L[i] - 2,), generator=g), torch.tensor([2])])
        items.append((feat, cap, f"clip{i}"))
    feats = np.zeros((4, max(T), 6), np.float32); caps = np.zeros((4, max(L)), np.int64)
    for i, (f, c, _) in enumerate(items):
        feats[i, :T[i]] = f.numpy(); caps[i, :L[i]] = c.numpy()
    out = host.caption_dataset.collate_fn([0, 1], 1)(list(items))
    ev = host.caption_dataset.collate_fn([1, ])([(k, f) for f, _, k in items[:3]])
    rows = np.array([[1, 5, 6, 7, 2, 2, 2, 2], [1, 9, 9, 4, 10, 11, 12, 13], [2, 5, 6, 7, 8, 9, 10, 11],
                     [5, 1, 6, 2, 7, 8, 2, 9]])
    # CaptionSampler (caption_dataset.py:199-224): pair order, plain and shuffled under random.seed(77)
    import types as _types
    ncaps = [2, 1, 3, 1, 2]
    info = [{"audio_id": f"clip{a}", "captions": [{"tokens": "x"}] * n} for a, n in enumerate(ncaps)]
    src = _types.SimpleNamespace(_caption_info=info)
    plain = list(host.caption_dataset.CaptionSampler(src))
    random.seed(77)
    shuffled = list(host.caption_dataset.CaptionSampler(src, shuffle=True))
    subset = list(host.caption_dataset.CaptionSampler(src, audio_subset_indices=[4, 2]))
    sents = [host.base_runner.BaseRunner._convert_idx2sentence(r, vocab) for r in rows]
    sents_zh = [" ".join(host.base_runner.BaseRunner._convert_idx2sentence(r, vocab, zh=True)) for r in rows]
    save("g10_host", in_feats=feats, in_caps=caps, in_T=np.array(T), in_L=np.array(L),
         tr_feats=out[0], tr_caps=out[1], tr_keys=np.array(out[2]), tr_feat_lens=out[3], tr_cap_lens=out[4],
         tr_caps_is_float=np.array(out[1].dtype == torch.float32),
         ev_keys=np.array(ev[0]), ev_feats=ev[1], ev_feat_lens=ev[2],
         words=np.array(words), vocab_pickle=np.frombuffer(pickle.dumps(vocab), dtype=np.uint8),
         rows=rows, sentences=np.array(sents), sentences_zh=np.array(sents_zh),
         sampler_ncaps=np.array(ncaps), sampler_plain=np.array(plain), sampler_shuffled77=np.array(shuffled),
         sampler_subset42=np.array(subset))


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "host":          # only the host-side fixture
        g10_host()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "losses":        # only the losses/loss.py fixture
        g3b_masked_losses()
        return
    ref = ref_shim.load()
    if len(sys.argv) > 1 and sys.argv[1] == "dbs":
        g11_dbs(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sampling":
        g14_sampling(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "decdrop":
        train_fixture(ref, "g15_train_step_decdrop", 3, 48, 40, 64, 7, True, 0, seed=46, dec_dropout=0.3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "projemb":
        train_fixture(ref, "g16_train_step_projemb", 3, 48, 40, 64, 7, True, 0, seed=56, proj_embed=24)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cnn14":
        g12_cnn14(ref)
        train_fixture(ref, "g13_train_step_cnn14", 2, 64, 40, 64, 6, True, 0, seed=36, keep_tensors=False,
                      encoder="Cnn14_16k")
        return
    g10_host()
    g11_dbs(ref)
    g12_cnn14(ref)
    train_fixture(ref, "g13_train_step_cnn14", 2, 64, 40, 64, 6, True, 0, seed=36, keep_tensors=False, encoder="Cnn14_16k")
    g1_attention(ref); g2_reparam_kl(ref); g3_ce(ref); g3b_masked_losses(); g4_encoder(ref); g5_rnn(ref)
    train_fixture(ref, "g6_train_step", 4, 64, 50, 64, 8, True, 0, seed=6)
    train_fixture(ref, "g6b_train_step_dis", 3, 48, 40, 64, 6, True, 0.7, seed=16)
    train_fixture(ref, "g6c_train_step_e512", 2, 64, 300, 512, 7, False, 0, seed=26, keep_tensors=False)
    train_fixture(ref, "g15_train_step_decdrop", 3, 48, 40, 64, 7, True, 0, seed=46, dec_dropout=0.3)
    train_fixture(ref, "g16_train_step_projemb", 3, 48, 40, 64, 7, True, 0, seed=56, proj_embed=24)
    g7_decode(ref)
    g14_sampling(ref)
    g9_beam(ref)
    # G8: BASELINE config 1 shape; scalars only, noise re-drawn in the test from the stored seed.
    train_fixture(ref, "g8_config1_scalars", 8, 500, 5000, 512, 22, False, 0, seed=8, keep_tensors=False,
                  keep_noise=False)


if __name__ == "__main__":
    main()
