"""CPU: the host-side batch / evaluation contract (A0, N2).  The golden g10_host was produced by the reference's own
collate_fn, Vocabulary and BaseRunner._convert_idx2sentence (oracle/make_golden.py:g10_host); the oracle restatement
(oracle/host_oracle.py) and the product (acvae_amd/batch.py, acvae_amd/evaluate.py) must both reproduce it."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import host_oracle as HO  # noqa: E402

from acvae_amd import batch as B  # noqa: E402
from acvae_amd import evaluate as EV  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "g10_host.npz"))


def items():
    out = []
    for i in range(4):
        t, l = int(G["in_T"][i]), int(G["in_L"][i])
        out.append((torch.from_numpy(G["in_feats"][i, :t].copy()), torch.from_numpy(G["in_caps"][i, :l].copy()), f"clip{i}"))
    return out


def check_train_batch(out):
    assert torch.equal(out[0], torch.from_numpy(G["tr_feats"])) and out[0].dtype == torch.float32
    assert torch.equal(out[1], torch.from_numpy(G["tr_caps"])) and out[1].dtype == torch.float32     # F10: float caps
    assert list(out[2]) == list(G["tr_keys"])
    assert np.array_equal(out[3], G["tr_feat_lens"]) and np.array_equal(out[4], G["tr_cap_lens"])
    assert len(out) == 5


def check_eval_batch(out):
    assert list(out[0]) == list(G["ev_keys"]) and torch.equal(out[1], torch.from_numpy(G["ev_feats"]))
    assert np.array_equal(out[2], G["ev_feat_lens"]) and len(out) == 3


def test_oracle_collate_matches_reference():
    check_train_batch(HO.collate(items(), [0, 1], 1))
    check_eval_batch(HO.collate([(k, f) for f, _, k in items()[:3]], [1]))


def test_product_collate_matches_reference():
    check_train_batch(B.collate_fn([0, 1], 1)(items()))
    check_eval_batch(B.collate_fn([1, ])([(k, f) for f, _, k in items()[:3]]))


def test_collate_ties_keep_dataset_order_and_scalar_fields():
    """equal caption lengths keep their order (stable sort, as list.sort); 0-d tensor fields are stacked."""
    its = [(torch.ones(3, 2) * i, torch.arange(4), torch.tensor(float(i))) for i in range(3)]
    out = B.collate_fn([0], 1)(list(its))
    ref = HO.collate(its, [0], 1)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2]) and torch.equal(out[2], torch.tensor([0., 1., 2.]))
    assert np.array_equal(out[3], ref[3])


def test_idx2sentence_and_vocabulary_pickle():
    vocab = EV.load_vocabulary(G["vocab_pickle"].tobytes())           # a pickle of the reference's own class
    assert isinstance(vocab, EV.Vocabulary) and len(vocab) == len(G["words"])
    assert [vocab.idx2word[i] for i in range(len(vocab))] == list(G["words"])
    assert vocab("w3") == 7 and vocab("never-seen") == vocab("<unk>") == 3
    idx2word = {i: w for i, w in enumerate(G["words"])}
    for row, want, want_zh in zip(G["rows"], G["sentences"], G["sentences_zh"]):
        assert HO.idx2sentence(row, idx2word) == want
        assert EV.convert_idx2sentence(row, vocab) == want
        assert " ".join(EV.convert_idx2sentence(row, vocab, zh=True)) == want_zh


def test_pack_rows_is_pack_padded_sequence():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 7, 11, generator=g)
    lens = np.array([7, 6, 6, 3, 1])
    assert torch.equal(B.pack_rows(x, lens), HO.packed(x, lens))
    caps = torch.randint(0, 9, (5, 8), generator=g).float()
    assert torch.equal(B.pack_rows(caps[:, 1:], lens), HO.packed(caps[:, 1:], lens))
    with pytest.raises(RuntimeError):
        B.pack_rows(x, np.array([3, 7, 6, 3, 1]))


def test_replication_equals_reference_for_one_clip_per_batch():
    feats = torch.randn(1, 9, 4)
    k, f, l = B.replicate_for_sampling(["a"], feats, [9], 5)
    rk, rf, rl = HO.eval_replicate_reference(["a"], feats, [9], 5)
    assert k == rk and torch.equal(f, rf) and l == rl
    feats = torch.arange(3).float().reshape(3, 1, 1).expand(3, 2, 4)
    k, f, l = B.replicate_for_sampling(["a", "b", "c"], feats, [2, 2, 1], 2)
    assert k == ["a", "a", "b", "b", "c", "c"] and l == [2, 2, 2, 2, 1, 1]
    assert f[:, 0, 0].tolist() == [0, 0, 1, 1, 2, 2]            # every key sits beside its own clip's features


def test_predictions_payload_structure():
    vocab = EV.load_vocabulary(G["vocab_pickle"].tobytes())
    idx2word = {i: w for i, w in enumerate(G["words"])}
    rows = G["rows"]
    # one caption per clip
    k2p = EV.collect_predictions(["x", "y"], rows[:2], vocab, False, {})
    got = EV.predictions_payload(k2p)
    assert got == HO.predictions([["x", "y"]], [rows[:2]], idx2word)
    assert got == {"predictions": [{"filename": "x", "caption": "w1 w2 w3", "tokens": "w1 w2 w3"},
                                   {"filename": "y", "caption": "w5 w5 w0 w6 w7 w8 w9", "tokens": "w5 w5 w0 w6 w7 w8 w9"}]}
    # N captions per clip: replicated keys, and a [rows, k, len] search output
    k2p = EV.collect_predictions(["x", "x", "y", "y"], rows, vocab, False, {})
    got = EV.predictions_payload(k2p)
    assert got == HO.predictions([["x", "x", "y", "y"]], [rows], idx2word)
    assert [c["cap_id"] for c in got["predictions"][0]["captions"]] == [0, 1] and "caption" not in got["predictions"][0]
    nested = rows.reshape(2, 2, -1)
    assert EV.predictions_payload(EV.collect_predictions(["x", "y"], nested, vocab, False, {})) == \
        HO.predictions([["x", "y"]], [nested], idx2word)
    # zh: characters joined without / with spaces
    got = EV.predictions_payload(EV.collect_predictions(["x"], rows[:1], vocab, True, {}), zh=True)
    assert got == HO.predictions([["x"]], [rows[:1]], idx2word, zh=True)
    assert got["predictions"][0] == {"filename": "x", "caption": "w1w2w3", "tokens": "w1 w2 w3"}
    json.dumps(got)


# ------------------------------------------------------------------------------------------------ batch source (N4)
def annotations():
    words = [f"w{i}" for i in range(16)]
    rng = np.random.RandomState(3)
    info = []
    for a in range(5):
        caps = [{"tokens": " ".join(rng.choice(words + ["zzz"], size=rng.randint(2, 7)))} for _ in range(1 + a % 3)]
        info.append({"audio_id": f"clip{a}", "captions": caps})
    feats = {f"clip{a}": rng.randn(1, 6 + a, 4).astype(np.float32) for a in range(5)}     # squeezed on read
    return info, feats


def test_caption_dataset_items_and_sampler_order():
    from acvae_amd import dataset as DS
    info, feats = annotations()
    vocab = EV.load_vocabulary(G["vocab_pickle"].tobytes())
    ds = DS.CaptionDataset(feats, info, vocab)
    assert len(ds) == sum(len(i["captions"]) for i in info) == 9
    feat, cap, key = ds[(3, 0)]
    toks = info[3]["captions"][0]["tokens"].split()
    assert key == "clip3" and tuple(feat.shape) == (9, 4) and torch.equal(feat, torch.from_numpy(feats["clip3"][0]))
    assert cap.tolist() == [1] + [vocab(t) for t in toks] + [2] and vocab("zzz") == 3        # <start> ... <end>, <unk>
    ev = DS.CaptionEvalDataset(lambda k: feats[k], [i["audio_id"] for i in info])
    assert ev[1][0] == "clip1" and tuple(ev[1][1].shape) == (7, 4) and len(ev) == 5
    # sampler: clip-major pairs; shuffled order = random.shuffle of that list under the caller's seed
    import random
    plain = list(DS.CaptionSampler(ds))
    assert plain == [(a, c) for a in range(5) for c in range(len(info[a]["captions"]))] and len(DS.CaptionSampler(ds)) == 9
    random.seed(5); got = list(DS.CaptionSampler(ds, shuffle=True))
    want = list(plain); random.seed(5); random.shuffle(want)
    assert got == want
    assert list(DS.CaptionSampler(ds, audio_subset_indices=[4, 1])) == [(4, 0), (4, 1), (1, 0), (1, 1)]
    # a training batch straight through the collate function
    batch = B.collate_fn([0, 1], 1)([ds[p] for p in plain[:4]])
    assert batch[0].shape[0] == 4 and list(batch[-1]) == sorted(batch[-1], reverse=True)


def test_distributed_sampler_partitions_every_pair():
    from acvae_amd import dataset as DS
    info, feats = annotations()
    ds = DS.CaptionDataset(feats, info, EV.load_vocabulary(G["vocab_pickle"].tobytes()))
    world = 4
    parts = [list(DS.CaptionDistributedSampler(ds, num_replicas=world, rank=r, seed=2)) for r in range(world)]
    assert all(len(p) == 3 for p in parts)                       # ceil(9 / 4), padded by wrapping
    flat = [x for p in parts for x in p]
    assert set(flat) == set(DS.caption_pairs(info)) and len(flat) == 12
    # every rank shuffles the same list the same way (random.seed(seed + epoch)), so the shards interleave one order
    import random
    order = DS.caption_pairs(info); random.seed(2); random.shuffle(order)
    order += order[:3]
    assert [parts[r] for r in range(world)] == [order[r::world] for r in range(world)]


def test_sampler_matches_reference_fixture():
    """pair order of the reference's own CaptionSampler (golden g10: plain, shuffled under random.seed(77), subset)"""
    import random
    import types
    from acvae_amd import dataset as DS
    info = [{"audio_id": f"clip{a}", "captions": [{"tokens": "x"}] * int(n)} for a, n in enumerate(G["sampler_ncaps"])]
    src = types.SimpleNamespace(_caption_info=info)
    assert [list(p) for p in DS.CaptionSampler(src)] == G["sampler_plain"].tolist()
    random.seed(77)
    assert [list(p) for p in DS.CaptionSampler(src, shuffle=True)] == G["sampler_shuffled77"].tolist()
    assert [list(p) for p in DS.CaptionSampler(src, audio_subset_indices=[4, 2])] == G["sampler_subset42"].tolist()


def test_overlap_report_on_a_synthetic_two_rank_trace(tmp_path):
    """tools/overlap_report.py (what tools/scale_check.sh runs on the kernel trace of `bench.py --gpus N`): per RCCL kernel the
    part no compute kernel of the rank covers, the tail behind the last backward kernel, and the assertion that the collectives
    appear on exactly N ranks - on a hand-made trace whose answers are known."""
    import csv
    import importlib.util
    import io
    import pathlib
    spec = importlib.util.spec_from_file_location("overlap_report", pathlib.Path(__file__).resolve().parents[1] / "tools" / "overlap_report.py")
    orp = importlib.util.module_from_spec(spec); spec.loader.exec_module(orp)
    cols = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id",
            "Start_Timestamp", "End_Timestamp"]
    for rank in range(2):
        d = tmp_path / f"rank{rank}"
        d.mkdir()
        with open(d / "t_kernel_trace.csv", "w", newline="") as fh:
            w = csv.writer(fh); w.writerow(cols)
            t = 1000
            for step in range(4):
                rows = [("conv_wino_wgrad_kernel_s4(WinoWgradParams)", t, t + 500_000),
                        ("ncclDevKernel_Generic(ncclDevKernelArgs)", t + 100_000, t + 300_000),          # fully covered
                        ("conv_wino_kernel(WinoParams)", t + 500_000, t + 900_000),
                        ("ncclDevKernel_Generic(ncclDevKernelArgs)", t + 800_000, t + 1_000_000),        # 100 us behind the backward
                        ("adam_kernel(float*)", t + 1_000_000, t + 1_100_000)]
                for name, s, e in rows:
                    w.writerow(["KERNEL_DISPATCH", "Agent 2", 1, 0, 1, 0, 0, name, 0, s, e])
                t += 2_000_000
    ranks = orp.load(str(tmp_path))
    assert len(ranks) == 2
    buf = io.StringIO()
    summ = orp.report(ranks, n_gpus=2, steps_from=1, out=buf)
    for s in summ.values():
        assert abs(s["comm_us"] - 400.0) < 1e-6 and abs(s["exposed_us"] - 100.0) < 1e-6 and abs(s["tail_us"] - 100.0) < 1e-6
    assert "beside conv_wino_wgrad_kernel_s4" in buf.getvalue()
    import pytest
    with pytest.raises(AssertionError, match="expected 4"):
        orp.report(ranks, n_gpus=4, out=io.StringIO())
