"""Validate the oracle restatement against the imported reference (build container only).

    python -B oracle/check_oracle_vs_reference.py

Runs the reference model and ``acvae_oracle`` on the same closed-form parameters, the same seeded
batch and the same torch RNG seed (the oracle makes the same generator calls in the same order, so
dropout masks / eps / scheduled-sampling draws coincide) and prints max abs differences.
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import acvae_oracle as O  # noqa: E402
import ref_shim  # noqa: E402


def load_state_into(model, state):
    sd = model.state_dict()
    assert list(sd.keys()) == list(state.keys()), (set(sd) ^ set(state))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(state[k].shape), (k, sd[k].shape, state[k].shape)
    model.load_state_dict({k: v.clone() for k, v in state.items()})


def ref_train_step(ref, model, feats, feat_lens, caps, cap_lens, vocab, ss_ratio, dis_ratio,
                   smoothing=0.1, kl_weight=0.5, alpha=1.0, max_grad_norm=1.0, lr=5e-4, optimizer=None):
    """runners/pytorch_runner_vae.py:76-98 (_forward) + :311-324, using the reference's own classes."""
    tu = ref.train_util
    crit = tu.LabelSmoothingLoss(vocab, smoothing=smoothing, device="cpu")
    klf = tu.Normal_kl_loss(device="cpu")
    cap_lens_t = torch.as_tensor(cap_lens)
    targets = torch.nn.utils.rnn.pack_padded_sequence(caps[:, 1:], cap_lens_t - 1, batch_first=True).data
    model.zero_grad()
    out = model(feats, feat_lens, caps, cap_lens, ss_ratio=ss_ratio, dis_ratio=dis_ratio)
    packed = torch.nn.utils.rnn.pack_padded_sequence(out["logits"], cap_lens_t - 1, batch_first=True).data
    ce = crit(packed, targets)
    kl = klf(out["q_means"], out["q_logs"], out["p_means"], out["p_logs"])
    loss = ce + kl_weight * kl
    mse = torch.nn.functional.mse_loss(out["q_means_utt"], out["p_means_utt"])
    loss = loss + alpha * mse
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    if optimizer is not None:
        optimizer.step()
    return out, dict(loss=loss.detach(), ce=ce.detach(), kl=kl.detach(), mse=mse.detach(), grad_norm=gn), grads


def main():
    ref = ref_shim.load()
    worst = 0.0
    for (B, T, V, E, L, ragged, ss, dis) in [(4, 64, 50, 64, 8, True, 1.0, 0), (3, 96, 40, 64, 6, True, 0.6, 0.5),
                                             (2, 64, 300, 512, 7, False, 1.0, 0)]:
        shapes = O.state_shapes(V, E, E, None, E, 512)
        state = O.closed_form_state(shapes)
        model = ref_shim.build_reference_model(ref, V, E, E)
        load_state_into(model, state)
        model.train()
        feats, caps, feat_lens, cap_lens = O.synthetic_batch(B, T, V, L, seed=3, ragged=ragged)
        opt = torch.optim.Adam(model.parameters(), lr=5e-4)
        if ss < 1.0:
            # The reference cannot back-propagate with ss_ratio < 1: the word fed to nn.Embedding is a
            # view of output["seqs"], which is written in place afterwards (models/vae_model.py:831,855)
            # -> autograd "modified by an inplace operation".  Forward-only comparison for that mode.
            torch.manual_seed(11); random.seed(11)
            with torch.no_grad():
                rout = model(feats, feat_lens.copy(), caps, cap_lens, ss_ratio=ss, dis_ratio=dis)
            ostate = {k: v.clone() for k, v in state.items()}
            torch.manual_seed(11); random.seed(11)
            with torch.no_grad():
                oout = O.hybrid_forward(ostate, feats, feat_lens.copy(), caps, cap_lens, ss_ratio=ss, dis_ratio=dis)
            print(f"--- B={B} T={T} V={V} E={E} ragged={ragged} ss={ss} dis={dis} (forward only)")
            for k in ("logits", "seqs", "p_means", "p_z", "attn_weights", "p_means_utt"):
                d = (rout[k].detach().double() - oout[k].detach().double()).abs().max().item()
                worst = max(worst, d)
                print(f"  {k:18s} maxabs {d:.3e}")
            continue
        torch.manual_seed(11); random.seed(11)
        rout, rl, rg = ref_train_step(ref, model, feats, feat_lens.copy(), caps, cap_lens, V, ss, dis, optimizer=opt)
        ostate = {k: v.clone() for k, v in state.items()}
        tr = O.OracleTrainer(ostate, V)
        torch.manual_seed(11); random.seed(11)
        res = tr.step(feats, feat_lens.copy(), caps, cap_lens, ss_ratio=ss, dis_ratio=dis)
        oout = res["out"]
        print(f"--- B={B} T={T} V={V} E={E} ragged={ragged} ss={ss} dis={dis}")
        for k in ("logits", "outputs", "seqs", "sampled_logprobs", "attn_weights", "p_means", "p_logs", "p_z",
                  "q_means", "q_logs", "q_z", "q_means_utt", "p_means_utt"):
            d = (rout[k].detach().double() - oout[k].detach().double()).abs().max().item()
            worst = max(worst, d)
            print(f"  {k:18s} maxabs {d:.3e}")
        for k in ("loss", "ce", "kl", "mse", "grad_norm"):
            d = abs(float(rl[k]) - float(res[k])) / max(1.0, abs(float(rl[k])))
            print(f"  {k:18s} ref {float(rl[k]):.7f} oracle {float(res[k]):.7f} diff {d:.2e}")
            worst = max(worst, d)
        gd = max((rg[k].double() - res["grads"][k].double()).abs().max().item() for k in rg)
        assert set(rg) == set(res["grads"]), set(rg) ^ set(res["grads"])
        print(f"  grads              maxabs {gd:.3e}")
        sd = model.state_dict()
        pd = max((sd[k].double() - ostate[k].detach().double()).abs().max().item() for k in sd)
        print(f"  state after Adam   maxabs {pd:.3e}")
        worst = max(worst, gd, pd)
        # inference twin (greedy, N=5 z-samples per clip: runner :101-104 replication)
        model.eval()
        est = {k: v.detach().clone() for k, v in ostate.items()}
        f5 = feats.repeat(5, 1, 1); l5 = [int(x) for x in feat_lens for _ in range(5)]
        torch.manual_seed(5)
        with torch.no_grad():
            ro = model(f5, list(l5), method="greedy", beam_size=5)
        torch.manual_seed(5)
        with torch.no_grad():
            oo = O.hybrid_forward(est, f5, list(l5), training=False)
        same = torch.equal(ro["seqs"], oo["seqs"])
        print(f"  greedy N=5 seqs equal: {same}  distinct rows {len(set(map(tuple, ro['seqs'].tolist())))}")
        assert same
    print("worst", worst)
    assert worst < 2e-4


if __name__ == "__main__":
    main()
