"""GPU: the persistent decode loop (csrc/decode_persist.hip: the Tc steps of the prior and decoder chains of a teacher-forced
training forward as ONE launch with in-launch hand-offs) against the per-step path of csrc/decoder.hip, which the golden
tests pin to the reference (models/vae_model.py:700-730,792-816).  Both run the same arithmetic in the same order, so every
forward output and every tensor saved for the backward must be BIT-identical - any stale hand-off, any missed wait shows
up as a difference; the backward launch (other summation order) must agree to rounding in every gradient.  Shapes: BASELINE configs[1] (N=32, S=62), configs[3] (N=16, S=187), ragged small
batches, repeated launches (the arrival counters are re-zeroed per launch)."""
import random

import numpy as np
import pytest
import torch

from acvae_amd import _lib
from acvae_amd.decoder import VAERNNBahdanauAttnDecoder
from acvae_amd.encoder import Cnn10
from acvae_amd.train_util import LabelSmoothingLoss, MSELoss, Normal_kl_loss
from acvae_amd.vae_model import Hybrid_VAEModel

pytestmark = pytest.mark.gpu


def build(V, E, seed=3):
    torch.manual_seed(seed)
    dec = VAERNNBahdanauAttnDecoder(vocab_size=V, enc_mem_size=E, embed_size=E, hidden_size=E, attn_size=E)
    m = Hybrid_VAEModel(Cnn10(64, 512), dec, posterior_model="PosteriorRNN_hybrid", posterior_args={"hidden_size": E},
                        prior_model="PriorRNN", prior_args={"hidden_size": E})
    return m.cuda().train()


def batch(B, T, V, L, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 64, generator=g)
    lens = np.sort(np.random.RandomState(seed).randint(max(3, L // 3), L + 1, B))[::-1].copy(); lens[0] = L
    caps = torch.zeros(B, L)
    for b in range(B):
        n = int(lens[b])
        caps[b, 0] = 1; caps[b, 1:n - 1] = torch.randint(4, V, (n - 2,), generator=g).float(); caps[b, n - 1] = 2
    fl = np.random.RandomState(seed + 1).randint(T // 2, T + 1, B); fl[0] = T
    return feats, caps, fl, lens


def run(model, V, E, feats, caps, fl, cl, eps_q, eps_p, persist):
    prev = _lib.lib().acvae_set_decode_persist(1 if persist else 0)
    # the per-step reference with one workgroup per attention row: the persistent kernel restates THAT arithmetic bit for bit
    # (the split-over-frames form of acvae_attn_fwd combines the softmax in another order)
    prev_split = _lib.lib().acvae_set_attn_split(0)
    try:
        for p in model.parameters():
            p.grad = None
        model.encoder.dropout_masks = None
        model.encoder._seed_base, model.encoder._calls = 77, 0      # the same dropout masks in every run
        model.noise = dict(eps_q=eps_q, eps_p=eps_p)
        random.seed(5)
        out = model(feats, fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0)
        lens1 = np.asarray(cl) - 1
        ce = LabelSmoothingLoss(V, 0.1).masked(out["logits"], caps[:, 1:].to(torch.long), lens1)
        kl = Normal_kl_loss()(out["q_means"], out["q_logs"], out["p_means"], out["p_logs"])
        mse = MSELoss()(out["q_means_utt"], out["p_means_utt"])
        (ce + 0.5 * kl + mse).backward()
        torch.cuda.synchronize()
        keep = {k: out[k].detach().clone() for k in ("logits", "outputs", "seqs", "sampled_logprobs", "attn_weights", "p_means",
                                                     "p_logs", "p_z", "p_means_utt", "state", "last_z")}
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        return keep, grads
    finally:
        _lib.lib().acvae_set_decode_persist(1 if prev != 0 else 0)
        _lib.lib().acvae_set_attn_split(prev_split)


@pytest.mark.parametrize("B,T,V,E,L", [(32, 1000, 5000, 512, 22), (16, 3000, 5000, 512, 22), (5, 200, 300, 512, 9),
                                        (3, 64, 50, 64, 7), (17, 333, 200, 128, 12)])
def test_persistent_decode_is_bit_identical_to_the_per_step_path(B, T, V, E, L):
    model = build(V, E)
    feats, caps, fl, cl = batch(B, T, V, L, seed=B + T)
    g = torch.Generator().manual_seed(1)
    eps_q = torch.randn(B, L - 1, E, generator=g); eps_p = torch.randn(L - 1, B, E, generator=g)
    f = feats.cuda()
    ref_out, ref_grads = run(model, V, E, f, caps, fl, cl, eps_q, eps_p, persist=False)
    for rep in range(3):                      # repeated launches: counters zeroed each time, no state carried over
        out, grads = run(model, V, E, f, caps, fl, cl, eps_q, eps_p, persist=True)
        for k in ref_out:
            a, b = ref_out[k], out[k]
            if isinstance(a, (tuple, list)):
                a, b = torch.cat([x.reshape(-1) for x in a]), torch.cat([x.reshape(-1) for x in b])
            assert torch.equal(a, b), (k, rep, float((a.double() - b.double()).abs().max()))
        assert set(grads) == set(ref_grads)
        # the persistent BPTT launch sums its products over eight wave shares in one pass (the per-step path: split-K slabs)
        # and the attention's frames in one sweep: equal to rounding, not bit for bit (measured <= 3e-6 of the tensor's max)
        for k in ref_grads:
            a, b = ref_grads[k].double(), grads[k].double()
            tol = 2e-5 * float(a.abs().max()) + 1e-9
            assert float((a - b).abs().max()) <= tol, (k, rep, float((a - b).abs().max()), float(a.abs().max()))
    assert bool(torch.isfinite(ref_out["logits"]).all())


def test_persistent_decode_under_a_busy_gpu():
    """Uneven load: a long convolution-heavy stream runs beside the decode (the posterior and the encoder of the NEXT batch
    on another stream in a real step); the hand-offs must not depend on timing or placement."""
    V, E, B, T, L = 500, 512, 32, 400, 22
    model = build(V, E)
    feats, caps, fl, cl = batch(B, T, V, L, seed=9)
    g = torch.Generator().manual_seed(2)
    eps_q = torch.randn(B, L - 1, E, generator=g); eps_p = torch.randn(L - 1, B, E, generator=g)
    f = feats.cuda()
    ref_out, _ = run(model, V, E, f, caps, fl, cl, eps_q, eps_p, persist=False)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(10 + 10 * rep):
                a = torch.tanh(a @ a) * 0.01 + 1.0       # keeps many CUs busy while the decode runs
        out, _ = run(model, V, E, f, caps, fl, cl, eps_q, eps_p, persist=True)
        side.synchronize()
        for k in ("logits", "outputs", "p_z", "attn_weights"):
            assert torch.equal(ref_out[k], out[k]), (k, rep)
