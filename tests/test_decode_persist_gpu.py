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
    # the per-step reference with one workgroup per attention row: the persistent kernel restates THAT arithmetic bit for bit
    # (the split-over-frames form of acvae_attn_fwd combines the softmax in another order)
    with _lib.override(persist=persist, attn_split=False):
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


def _step_tensors(B=32, T=1000, V=5000, E=512, L=22, seed=3):
    model = build(V, E, seed)
    feats, caps, fl, cl = batch(B, T, V, L, seed=B + T + seed)
    g = torch.Generator().manual_seed(seed)
    eps_q = torch.randn(B, L - 1, E, generator=g); eps_p = torch.randn(L - 1, B, E, generator=g)
    return model, feats.cuda(), caps, fl, cl, eps_q, eps_p


def test_two_persistent_steps_from_two_streams_are_serialised():
    """Two spin-wait grids must never share the chip: at the BASELINE configs[1] shape a persistent decode launch is 224
    workgroups of up to 150 KB of LDS (one per CU), so two of them queued from two host threads on two streams cannot both be
    resident - each one's resident roles would spin on roles that are not scheduled until the bounded waits give up.  The
    library chains a device's persistent launches behind each other (decode_persist.hip, persist_launch): both steps must
    complete, bit-identical to each one run alone, with no abort reported."""
    import threading
    V, E = 5000, 512
    jobs = [_step_tensors(seed=3), _step_tensors(seed=4)]
    alone = [run(m, V, E, f, c, fl, cl, eq, ep, persist=True) for (m, f, c, fl, cl, eq, ep) in jobs]
    torch.cuda.synchronize()
    res, errs = [None, None], []
    barrier = threading.Barrier(2)

    def worker(i):
        try:
            m, f, c, fl, cl, eq, ep = jobs[i]
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(st):
                barrier.wait()
                for _ in range(3):
                    res[i] = run(m, V, E, f, c, fl, cl, eq, ep, persist=True)
        except Exception as e:          # noqa: BLE001
            errs.append(e)

    # random.seed / model.noise are per call inside run(); the two models share nothing but the device
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errs, errs
    _lib.check_persist_status("cuda")
    for i in range(2):
        for k in ("logits", "outputs", "p_z", "attn_weights", "p_means_utt"):
            assert torch.equal(alone[i][0][k], res[i][0][k]), (i, k)
        for k in alone[i][1]:
            assert torch.equal(alone[i][1][k], res[i][1][k]), (i, k)


@pytest.mark.parametrize("which", ["decode_fwd", "decode_bwd", "posterior_fwd", "posterior_bwd"])
def test_a_persistent_launch_that_cannot_finish_is_reported_and_poisoned(which):
    """ACVAE_FLAG_TEST_STALL queues a persistent launch one workgroup short with a short spin bound: the roles that wait for
    the missing workgroup give up exactly as they would if part of the grid were not resident.  The call still returns
    ACVAE_OK (nothing synchronises), but the tail kernel behind the launch must overwrite its outputs with NaN and raise the
    device's status word, which check_persist_status() turns into a RuntimeError - once; a clean step afterwards passes."""
    V, E, B, T, L = 300, 512, 5, 200, 9
    model, f, caps, fl, cl, eps_q, eps_p = _step_tensors(B, T, V, E, L, seed=6)
    status = _lib.persist_status("cuda")
    status.zero_()
    lens1 = np.asarray(cl) - 1

    def step(stall_fwd, stall_bwd):
        for p in model.parameters():
            p.grad = None
        model.noise = dict(eps_q=eps_q, eps_p=eps_p)
        random.seed(5)
        # the flag travels with each call (`flags` is the last argument): only the chosen entry point gets it
        real_call = _lib.call
        def call(name, *a):
            tag = {"acvae_decode_fwd_sampled": "decode_fwd", "acvae_decode_bwd": "decode_bwd",
                   "acvae_posterior_fwd": "posterior_fwd", "acvae_posterior_bwd": "posterior_bwd"}.get(name)
            if tag == which and (stall_fwd if tag.endswith("fwd") else stall_bwd):
                a = a[:-1] + (a[-1] | _lib.FLAG_TEST_STALL,)
            return real_call(name, *a)
        _lib.call = call
        try:
            out = model(f, fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0)
            ce = LabelSmoothingLoss(V, 0.1).masked(out["logits"], caps[:, 1:].to(torch.long), lens1)
            kl = Normal_kl_loss()(out["q_means"], out["q_logs"], out["p_means"], out["p_logs"])
            (ce + 0.5 * kl).backward()
            torch.cuda.synchronize()
        finally:
            _lib.call = real_call
        return out, float(ce + 0.5 * kl)

    out, loss = step(True, True)
    k = {"decode_fwd": 0, "decode_bwd": 1, "posterior_fwd": 2, "posterior_bwd": 3}[which]
    assert int(status[k]) == 1 and int(status[4]) == 1, status
    if which.endswith("fwd"):
        assert not np.isfinite(loss)                       # NaN outputs -> NaN loss: cannot be trained on
        key = "logits" if which == "decode_fwd" else "q_means"
        assert bool(torch.isnan(out[key]).any())
    else:
        names = ("decoder.model.weight_hh_l0", "pnet.network.weight_hh_l0") if which == "decode_bwd" else \
                ("qnet.network.weight_hh_l0", "qnet.network.weight_hh_l0_reverse")
        g = dict(model.named_parameters())
        assert any(bool(torch.isnan(g[n].grad).any()) for n in names)
    with pytest.raises(RuntimeError, match="persistent launch could not complete"):
        _lib.check_persist_status("cuda")
    _lib.check_persist_status("cuda")                      # cleared by the raise
    out, loss = step(False, False)
    assert np.isfinite(loss) and int(status[4]) == 0
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)


def test_persistent_path_is_not_taken_when_the_grid_cannot_be_resident():
    """The launchers ask the occupancy calculator whether the WHOLE grid fits the device: at E = H = A = 2048 the
    decoder chain alone is 256 + N + 128 workgroups with one workgroup per CU when the attention keeps its memory in LDS -
    more than the 256 CUs.  The forward must then either run the variant that streams the memory (small LDS, several
    workgroups per CU) or fall back to the per-step path - and in both cases agree with the per-step path bit for bit."""
    V, E, B, T, L = 60, 2048, 4, 64, 6
    torch.manual_seed(1)
    dec = VAERNNBahdanauAttnDecoder(vocab_size=V, enc_mem_size=E, embed_size=E, hidden_size=E, attn_size=E)
    from acvae_amd.encoder import Cnn14_16k
    m = Hybrid_VAEModel(Cnn14_16k(64, 2048), dec, posterior_model="PosteriorRNN_hybrid", posterior_args={"hidden_size": 256},
                        prior_model="PriorRNN", prior_args={"hidden_size": E}).cuda().train()
    feats, caps, fl, cl = batch(B, T, V, L, seed=2)
    g = torch.Generator().manual_seed(1)
    eps_q = torch.randn(B, L - 1, E, generator=g); eps_p = torch.randn(L - 1, B, E, generator=g)
    f = feats.cuda()
    ref_out, _ = run(m, V, E, f, caps, fl, cl, eps_q, eps_p, persist=False)
    out, _ = run(m, V, E, f, caps, fl, cl, eps_q, eps_p, persist=True)
    _lib.check_persist_status("cuda")
    for k in ("logits", "outputs", "p_z", "attn_weights"):
        assert torch.equal(ref_out[k], out[k]), k
