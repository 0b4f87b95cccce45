"""CPU: the oracle restatement against the golden vectors generated from the reference itself
(oracle/make_golden.py).  This is what pins the oracle."""
import random

import numpy as np
import torch

import acvae_oracle as O
from conftest import load_golden, unpack_masks

T = torch.from_numpy


def close(a, b, rtol=1e-4, atol=1e-5):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def test_g1_attention():
    g = load_golden("g1_attention")
    for ci in range(int(g["ncases"])):
        N, S, E, Hd, A = (int(x) for x in g[f"c{ci}_dims"])
        st = O.closed_form_state({"v": (A,), "h2attn.weight": (A, E + Hd), "h2attn.bias": (A,)})
        st = {"a." + k: v for k, v in st.items()}
        ctx, w = O.seq2seq_attention(st, "a", T(g[f"c{ci}_h_dec"]), T(g[f"c{ci}_h_enc"]), T(g[f"c{ci}_lens"]))
        close(ctx, g[f"c{ci}_ctx"]); close(w, g[f"c{ci}_weights"])


def test_g2_reparam_kl():
    g = load_golden("g2_reparam_kl")
    z = T(g["eps"]) * torch.exp(.5 * T(g["lv_q"])) + T(g["mu_q"])
    close(z, g["z"], 1e-6, 1e-6)
    close(O.normal_kl_loss(T(g["mu_q"]), T(g["lv_q"]), T(g["mu_p"]), T(g["lv_p"])), g["kl"], 1e-6, 1e-6)


def test_g3_ce():
    g = load_golden("g3_ce")
    logits, targets, lens1 = T(g["logits"]), T(g["targets"]), g["lens1"]
    V = logits.shape[-1]
    pk = O.pack_rows(logits, lens1); tg = O.pack_rows(targets, lens1)
    close(O.label_smoothing_loss(pk, tg, V, 0.1), g["ls_packed"], 1e-6, 1e-6)
    close(O.label_smoothing_loss(pk, tg, V, 0.0), g["ce_packed"], 1e-6, 1e-6)
    close(O.label_smoothing_loss(pk, tg, V, 0.0), g["ls0_packed"], 1e-6, 1e-6)
    # masked mean over valid tokens equals the packed mean (A13 vs A8)
    close(O.masked_ce(logits, targets, lens1, 0.1, "mean"), g["ls_packed"], 1e-6, 1e-6)


def test_g3b_masked_losses_of_losses_py():
    """A13: the oracle's masked CE / label-smoothed CE against the values the reference's own losses/loss.py classes
    produced (oracle/make_golden.py:g3b_masked_losses), all three reductions."""
    g = load_golden("g3b_masked_losses")
    logits, targets, lens = T(g["logits"]), T(g["targets"]), g["lens"]
    for name, sm in (("ce", 0.0), ("ls", 0.1)):
        for red in ("none", "mean", "sum"):
            x = logits.clone().requires_grad_(True)
            val = O.masked_ce(x, targets, lens, sm, red)
            close(val, g[f"{name}_{red}"], 1e-6, 1e-6)
            if red != "none":
                val.backward()
                close(x.grad, g[f"{name}_{red}_dlogits"], 1e-5, 1e-8)


def test_g4_encoder():
    g = load_golden("g4_encoder")
    full = O.closed_form_state(O.state_shapes(10))
    for ci in range(int(g["ncases"])):
        st = {k: v.clone() for k, v in full.items() if k.startswith("encoder.")}
        masks = unpack_masks(g, f"c{ci}_")
        lens = g[f"c{ci}_lens"].copy()
        o = O.cnn10_forward(st, T(g[f"c{ci}_feats"]), lens, True, list(masks), None)
        close(o["audio_embeds"], g[f"c{ci}_train_audio_embeds"]); close(o["audio_embeds_pooled"], g[f"c{ci}_train_pooled"])
        assert np.array_equal(o["audio_embeds_lens"].numpy(), g[f"c{ci}_train_lens"])
        assert np.array_equal(lens, g[f"c{ci}_train_lens"])          # in-place //=16 on the caller's array (F11)
        close(st["encoder.bn0.running_mean"], g[f"c{ci}_bn0_running_mean"])
        close(st["encoder.bn0.running_var"], g[f"c{ci}_bn0_running_var"])
        close(st["encoder.conv_block4.bn2.running_var"], g[f"c{ci}_b4bn2_running_var"])
        close(st["encoder.conv_block1.bn1.running_var"], g[f"c{ci}_b1bn1_running_var"])
        assert int(st["encoder.bn0.num_batches_tracked"]) == int(g[f"c{ci}_nbt"]) == 1
        st = {k: v.clone() for k, v in full.items() if k.startswith("encoder.")}
        o = O.cnn10_forward(st, T(g[f"c{ci}_feats"]), g[f"c{ci}_lens"].copy(), False)
        close(o["audio_embeds"], g[f"c{ci}_eval_audio_embeds"]); close(o["audio_embeds_pooled"], g[f"c{ci}_eval_pooled"])


def test_relu_mask_hooks_of_the_oracle():
    """The test hooks used by the GPU encoder-gradient tests: probing reports one pre-activation per ReLU site, forcing
    the natural masks reproduces F.relu bit for bit, flipping one listed bit changes the gradients."""
    full = O.closed_form_state(O.state_shapes(10))
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(2, 32, 64, generator=g)

    def run(force):
        st = {k: v.clone() for k, v in full.items() if k.startswith("encoder.")}
        for k in O.trainable_keys(st):
            st[k].requires_grad_(True)
        probe = []
        torch.manual_seed(1)
        o = O.cnn10_forward(st, feats, [32, 32], True, None, None, relu_probe=probe, relu_force=force)
        o["audio_embeds"].square().sum().backward()
        return o["audio_embeds"].detach(), st["encoder.conv_block1.conv1.weight"].grad.clone(), probe

    y0, g0, probe = run(None)
    assert len(probe) == 8 and probe[0].shape == (2, 64, 32, 64)
    cases, nbits = O.relu_mask_cases(probe, tau=1e-5, max_flips=1)
    cases = list(cases)
    assert nbits >= 1 and len(cases) == 1 + nbits           # natural + one assignment per single flipped bit
    y1, g1, _ = run(cases[0])
    assert torch.equal(y0, y1) and torch.equal(g0, g1)
    _, g2, _ = run(cases[1])
    assert not torch.equal(g0, g2)


def test_g5_rnn():
    g = load_golden("g5_rnn")
    N, I, H, V, E = (int(x) for x in g["dims"])
    gs = O.closed_form_state({"weight_ih_l0": (3 * H, I), "weight_hh_l0": (3 * H, H), "bias_ih_l0": (3 * H,),
                              "bias_hh_l0": (3 * H,)})
    h2 = O.gru_cell(T(g["x"])[:, 0], T(g["h"])[0], *gs.values())
    close(h2, g["gru_h"][0])
    ls = O.closed_form_state({"weight_ih_l0": (4 * H, I), "weight_hh_l0": (4 * H, H), "bias_ih_l0": (4 * H,),
                              "bias_hh_l0": (4 * H,)})
    lh, lc = O.lstm_cell(T(g["x"])[:, 0], T(g["h"])[0], T(g["c"])[0], *ls.values())
    close(lh, g["lstm_h"][0]); close(lc, g["lstm_c"][0])
    shapes = {k[len("qnet."):]: v for k, v in O.state_shapes(V, E, E, None, E).items() if k.startswith("qnet.")}
    qs = {"qnet." + k: v for k, v in O.closed_form_state(shapes).items()}
    q = O.posterior_hybrid_forward(qs, T(g["caps"]), g["cap_lens"], T(g["eps"]))
    for k in ("q_means", "q_logs", "q_z", "q_means_utt"):
        close(q[k], g[k])


def _train_case(name, tensors=True, encoder="Cnn10", gn_tol=1e-4):
    g = load_golden(name)
    B, Tt, V, E, L = (int(x) for x in g["dims"])
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512 if encoder == "Cnn10" else 2048, encoder=encoder))
    seed = int(g["seed"])
    feats, caps, feat_lens, cap_lens = O.synthetic_batch(B, Tt, V, L, seed=seed, ragged=bool(int(g["ragged"])))
    assert np.array_equal(cap_lens, g["cap_lens"]) and np.array_equal(feat_lens, g["feat_lens"])
    if "noise_eps_q" in g:
        noise = dict(dropout=unpack_masks(g), eps_q=T(g["noise_eps_q"]), eps_p=T(g["noise_eps_p"]))
    else:
        noise = None
    torch.manual_seed(seed); random.seed(seed)   # rand(1) for dis_ratio / replay of noise when not stored
    res = O.OracleTrainer(state, V).step(feats, feat_lens.copy(), caps, cap_lens, 1.0, float(g["dis_ratio"]), noise=noise)
    for k in ("loss", "ce", "kl", "mse"):
        assert abs(float(res[k]) - float(g[k])) <= 1e-4 * max(1.0, abs(float(g[k]))), (k, float(res[k]), float(g[k]))
    assert abs(float(res["grad_norm"]) - float(g["grad_norm"])) <= gn_tol * float(g["grad_norm"])
    if tensors:
        out = res["out"]
        assert np.array_equal(out["seqs"].numpy(), g["out_seqs"])
        for k in ("logits", "outputs", "attn_weights", "p_means", "p_logs", "p_z", "q_means", "q_z", "q_means_utt",
                  "p_means_utt", "sampled_logprobs"):
            close(out[k].detach(), g["out_" + k])
        for k in [k for k in g if k.startswith("grad_") and k != "grad_norm"]:
            close(res["grads"][k[5:]], g[k], 1e-3, 1e-5)
        for k in [k for k in g if k.startswith("post_")]:
            close(state[k[5:]].detach(), g[k], 1e-3, 1e-4)


def test_g6_train_step():
    _train_case("g6_train_step")


def test_g6c_train_step_e512():
    _train_case("g6c_train_step_e512", tensors=False)


def test_g12_cnn14_encoder():
    """N4: Cnn14_16k.forward (models/encoder.py:906-964) against the reference's own outputs."""
    g = load_golden("g12_cnn14_encoder")
    shapes = {k: v for k, v in O.state_shapes(10, enc_embed=2048, encoder="Cnn14_16k").items() if k.startswith("encoder.")}
    full = O.closed_form_state(shapes)
    for ci in range(int(g["ncases"])):
        st = {k: v.clone() for k, v in full.items()}
        masks = unpack_masks(g, f"c{ci}_")
        assert len(masks) == 8
        lens = g[f"c{ci}_lens"].copy()
        o = O.cnn10_forward(st, T(g[f"c{ci}_feats"]), lens, True, list(masks), None)
        close(o["audio_embeds"], g[f"c{ci}_train_audio_embeds"]); close(o["audio_embeds_pooled"], g[f"c{ci}_train_pooled"])
        assert o["audio_embeds"].shape[1:] == (g[f"c{ci}_feats"].shape[1] // 32, 2048)
        assert np.array_equal(lens, g[f"c{ci}_train_lens"])          # in-place //= 32 on the caller's array
        close(st["encoder.conv_block6.bn2.running_mean"], g[f"c{ci}_b6bn2_running_mean"])
        close(st["encoder.conv_block6.bn2.running_var"], g[f"c{ci}_b6bn2_running_var"])
        close(st["encoder.conv_block5.bn1.running_var"], g[f"c{ci}_b5bn1_running_var"])
        st = {k: v.clone() for k, v in full.items()}
        o = O.cnn10_forward(st, T(g[f"c{ci}_feats"]), g[f"c{ci}_lens"].copy(), False)
        close(o["audio_embeds"], g[f"c{ci}_eval_audio_embeds"]); close(o["audio_embeds_pooled"], g[f"c{ci}_eval_pooled"])


def test_g13_train_step_cnn14_with_ln():
    # blocks 5/6 normalise over 16 / 8 values at this size: the gradient norm is more sensitive to summation order
    _train_case("g13_train_step_cnn14", tensors=False, encoder="Cnn14_16k", gn_tol=1e-3)


def test_g7_decode_token_exact():
    g = load_golden("g7_decode")
    _, _, V, E = (int(x) for x in g["dims"])
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    for tag, rep in (("greedy1", 1), ("greedy5", 5)):
        f = T(g["feats"]).repeat(rep, 1, 1)
        with torch.no_grad():
            o = O.hybrid_forward({k: v.clone() for k, v in state.items()}, f, list(g[tag + "_lens"]), training=False,
                                 noise=dict(eps_p=T(g[tag + "_noise_eps_p"])))
        assert np.array_equal(o["seqs"].numpy(), g[tag + "_seqs"])


def test_g14_sampling_token_exact():
    """A6: the oracle's sample_next_word (gumbel / multinomial) replaying the noise the reference run drew."""
    g = load_golden("g14_sampling")
    _, _, V, E = (int(x) for x in g["dims"])
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    for tag, method in zip(g["cases"], g["methods"]):
        a, b = (int(x) for x in g[tag + "_clips"])
        lens = g["feat_lens"][a:b].copy()
        f = T(g["feats"])[a:b, :int(lens.max())]
        with torch.no_grad():
            o = O.hybrid_forward({k: v.clone() for k, v in state.items()}, f, lens, training=False, method=str(method),
                                 temp=float(g[tag + "_temp"]),
                                 noise=dict(eps_p=T(g[tag + "_noise_eps_p"]), sample_noise=T(g[tag + "_sample_noise"])))
        assert np.array_equal(o["seqs"].numpy(), g[tag + "_seqs"]), tag
        steps = int(g[tag + "_steps_run"])
        close(o["sampled_logprobs"][:, :steps], g[tag + "_logprobs"], 1e-5, 1e-6)


def test_g9_beam_search_token_exact():
    g = load_golden("g9_beam")
    _, _, V, E, beam = (int(x) for x in g["dims"])
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    with torch.no_grad():
        seqs = O.beam_search(state, T(g["feats"]), g["feat_lens"].copy(), beam, O.MAX_LENGTH, T(g["eps"]))
    assert np.array_equal(seqs.numpy(), g["seqs"])


DBS_CASES = [dict(beam_size=4, group_size=2), dict(beam_size=6, group_size=3, diversity_lambda=0.8, temperature=1.5,
             group_nbest=False), dict(), dict(beam_size=6, group_size=2, diversity_lambda=2.0)]


def dbs_state(V, E, bump):
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    state["decoder.classifier.bias"] = state["decoder.classifier.bias"].clone()
    state["decoder.classifier.bias"][O.END_IDX] += float(bump)
    return state


def test_g11_diverse_beam_search_token_exact():
    """N3: reference = word_model.py:297-394 + vae_model.py:997-1040 (golden made by running it)."""
    g = load_golden("g11_dbs")
    _, _, V, E, ML = (int(x) for x in g["dims"])
    for tag, bump in zip("ab", g["end_bump"]):
        state = dbs_state(V, E, bump)
        for ci, kw in enumerate(DBS_CASES):
            torch.manual_seed(40 + ci)
            with torch.no_grad():
                seqs = O.diverse_beam_search(state, T(g["feats"]), g["feat_lens"].copy(), max_length=ML, **kw)
            assert np.array_equal(seqs.numpy(), g[f"seqs_{tag}{ci}"]), (tag, kw)
