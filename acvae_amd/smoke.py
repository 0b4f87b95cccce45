"""One tiny training step of the hot path on cuda:0, checked against the oracle (used by
__graft_entry__.smoke()).  The oracle is imported here only as the checker."""
import random

import numpy as np
import torch


def run():
    import acvae_oracle as O
    from acvae_amd.decoder import VAERNNBahdanauAttnDecoder
    from acvae_amd.encoder import Cnn10
    from acvae_amd.trainer import TrainStep
    from acvae_amd.vae_model import Hybrid_VAEModel

    V, E, B, T, L = 52, 64, 3, 64, 7
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    feats, caps, feat_lens, cap_lens = O.synthetic_batch(B, T, V, L, seed=4, ragged=True)
    ostate = {k: v.clone() for k, v in state.items()}
    rec = {}
    torch.manual_seed(4); random.seed(4)
    tr = O.OracleTrainer(ostate, V)
    ores = tr.step(feats, feat_lens.copy(), caps, cap_lens, 1.0, 0, record=rec)

    dec = VAERNNBahdanauAttnDecoder(vocab_size=V, enc_mem_size=E, embed_size=E, hidden_size=E, attn_size=E)
    model = Hybrid_VAEModel(Cnn10(64, 512), dec, posterior_model="PosteriorRNN_hybrid",
                            posterior_args={"hidden_size": E}, prior_model="PriorRNN", prior_args={"hidden_size": E})
    model.load_state_dict({k: v.clone() for k, v in state.items()})
    model = model.cuda().train()
    model.encoder.dropout_masks = rec["dropout"]
    model.noise = dict(eps_q=rec["eps_q"], eps_p=rec["eps_p"])
    ts = TrainStep(model, V)
    random.seed(4)
    parts = ts.step(feats.cuda(), feat_lens.copy(), caps, cap_lens, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5)
    torch.cuda.synchronize()
    dl = abs(float(parts["loss"]) - float(ores["loss"]))
    dg = abs(float(parts["grad_norm"]) - float(ores["grad_norm"])) / float(ores["grad_norm"])
    sd = model.state_dict()
    diffs = torch.cat([(sd[k].cpu().double() - ostate[k].detach().double()).abs().reshape(-1) for k in sd])
    dp, dmean = float(diffs.max()), float(diffs.mean())
    print(f"smoke: loss hip {float(parts['loss']):.6f} oracle {float(ores['loss']):.6f} |d|={dl:.2e}; "
          f"grad-norm rel d={dg:.2e}; |param - oracle| after Adam: max {dp:.2e} mean {dmean:.2e}")
    assert dl <= 1e-4 * max(1.0, abs(float(ores["loss"]))), "loss differs from the oracle"
    assert dg <= 1e-3, "gradient norm differs from the oracle"
    # Adam's first step moves every weight by ~lr*sign(g): where g is at rounding-noise level the sign can differ,
    # so the max is bounded by 2*lr while the mean must stay tiny
    assert dp <= 2.2 * ts.lr and dmean <= 1e-5, "parameters after one Adam step differ from the oracle"
    return True


if __name__ == "__main__":
    run()
