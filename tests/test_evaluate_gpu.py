"""GPU: the evaluation wrapper (N2) and the runner-style batch forward (A0) end to end on the HIP path, checked against
the oracle: same captions for greedy / N-samples-per-clip / beam decoding given the same CPU-generator state."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import acvae_oracle as O  # noqa: E402
import host_oracle as HO  # noqa: E402

from acvae_amd import batch as B  # noqa: E402
from acvae_amd import evaluate as EV  # noqa: E402
from test_model_gpu import build_model  # noqa: E402

pytestmark = pytest.mark.gpu
V, E = 40, 64


def vocab():
    v = EV.Vocabulary()
    for w in ["<pad>", "<start>", "<end>", "<unk>"] + [f"w{i}" for i in range(V - 4)]:
        v.add_word(w)
    return v


def clips(n, seed=5):
    g = torch.Generator().manual_seed(seed)
    return [(f"clip{i}", torch.randn(int(t), 64, generator=g)) for i, t in enumerate([96, 64, 80, 112][:n])]


def oracle_captions(state, items, n_samples, max_length):
    """one clip per batch, as the reference's evaluation DataLoader (batch_size = 1)"""
    keys_pb, seqs_pb = [], []
    for key, feat in items:
        keys, feats, lens = HO.eval_replicate_reference([key], feat[None], [feat.shape[0]], n_samples)
        eps = torch.stack([torch.randn(n_samples, E) for _ in range(max_length)], 0)
        with torch.no_grad():
            out = O.hybrid_forward({k: v.clone() for k, v in state.items()}, feats, np.array(lens), training=False,
                                   max_length=max_length, noise=dict(eps_p=eps))
        keys_pb.append(keys); seqs_pb.append(out["seqs"].numpy())
    return keys_pb, seqs_pb


@pytest.mark.parametrize("n_samples", [1, 3])
def test_evaluate_greedy_matches_oracle(tmp_path, n_samples):
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = build_model(V, E, state)
    items, voc = clips(3), vocab()
    torch.manual_seed(21)
    path = tmp_path / "eval_output.json"
    got = EV.evaluate(model, items, voc, caption_output=path, method="greedy", beam_size=n_samples, max_length=9)
    torch.manual_seed(21)
    keys_pb, seqs_pb = oracle_captions(state, items, n_samples, 9)
    want = HO.predictions(keys_pb, seqs_pb, voc.idx2word)
    assert got == want
    assert json.load(open(path)) == want
    entry = got["predictions"][0]
    assert entry["filename"] == "clip0" and (("captions" in entry) == (n_samples > 1))


def test_evaluate_batched_keeps_keys_with_their_clips():
    """batch_size > 1 (the reference mis-pairs keys and features there): captions must equal the
    one-clip-per-batch result when every clip gets the noise it got there."""
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = build_model(V, E, state)
    voc = vocab()
    items = [(k, f[:64]) for k, f in clips(4)]                  # equal lengths: batch mates do not change padding
    n, ml = 2, 7
    torch.manual_seed(3)
    eps = torch.randn(ml, 4 * n, E)
    single = {}
    for i, it in enumerate(items):
        model.noise = dict(eps_p=eps[:, i * n:(i + 1) * n])
        single.update({p["filename"]: p for p in EV.evaluate(model, [it], voc, method="greedy", beam_size=n,
                                                              max_length=ml)["predictions"]})
    model.noise = dict(eps_p=eps)
    both = EV.evaluate(model, items, voc, method="greedy", beam_size=n, max_length=ml, batch_size=4)["predictions"]
    assert [p["filename"] for p in both] == [k for k, _ in items]
    for p in both:
        assert p == single[p["filename"]]


def test_evaluate_beam_matches_oracle():
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = build_model(V, E, state)
    items, voc = clips(2), vocab()
    torch.manual_seed(8)
    got = EV.evaluate(model, items, voc, method="beam", beam_size=3, max_length=8)
    torch.manual_seed(8)
    keys_pb, seqs_pb = [], []
    for key, feat in items:      # the runner replicates the clip beam_size times for every method but "dbs" (:101-104)
        keys, feats, lens = HO.eval_replicate_reference([key], feat[None], [feat.shape[0]], 3)
        with torch.no_grad():
            seqs = O.beam_search({k: v.clone() for k, v in state.items()}, feats, np.array(lens), 3, 8)
        keys_pb.append(keys); seqs_pb.append(seqs.numpy())
    want = HO.predictions(keys_pb, seqs_pb, voc.idx2word)
    assert got == want and len(got["predictions"][0]["captions"]) == 3


def test_forward_batch_train_mode_packs_like_the_runner():
    feats, caps, feat_lens, cap_lens = O.synthetic_batch(4, 64, V, 8, seed=2, ragged=True)
    items = [(feats[i, :int(feat_lens[i])], caps[i, :int(cap_lens[i])].long(), f"c{i}") for i in range(4)]
    random.shuffle(items)
    batch = B.collate_fn([0, 1], 1)(items)
    assert list(batch[-1]) == sorted(batch[-1], reverse=True)
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = build_model(V, E, state).train()
    out = B.forward_batch(model, batch, "train", ss_ratio=1.0, dis_ratio=0)
    lens1 = np.asarray(batch[-1]) - 1
    assert out["packed_logits"].shape == (int(lens1.sum()), V)
    assert torch.equal(out["packed_logits"].cpu(), HO.packed(out["logits"].detach().cpu(), lens1))
    assert torch.equal(out["targets"].cpu(), HO.packed(batch[1][:, 1:], lens1))
    out["packed_logits"].sum().backward()                       # the gather is autograd-connected to the decode
    assert model.decoder.classifier.weight.grad is not None


def test_evaluate_dbs_matches_oracle():
    """method="dbs": no replication (:101), the search returns [clips, beams, len] and every beam becomes a caption."""
    state = O.closed_form_state(O.state_shapes(V, E, E, None, E, 512))
    model = build_model(V, E, state)
    items, voc = clips(2), vocab()
    torch.manual_seed(13)
    got = EV.evaluate(model, items, voc, method="dbs", beam_size=4, group_size=2, diversity_lambda=0.7, max_length=6)
    torch.manual_seed(13)
    keys_pb, seqs_pb = [], []
    for key, feat in items:
        with torch.no_grad():
            seqs = O.diverse_beam_search({k: v.clone() for k, v in state.items()}, feat[None], np.array([feat.shape[0]]),
                                         beam_size=4, group_size=2, diversity_lambda=0.7, max_length=6)
        keys_pb.append([key]); seqs_pb.append(seqs.numpy())
    want = HO.predictions(keys_pb, seqs_pb, voc.idx2word)
    assert got == want and len(got["predictions"][1]["captions"]) == 4
