"""CPU restatement of the host-side batch / evaluation contract.  TEST INFRASTRUCTURE (only tests/, smoke() and
bench.py's cpu_baseline leg may import anything under oracle/).

  collate            datasets/caption_dataset.py:278-318   (pinned: tests/golden/g10_host.npz, made by running the
                                                             reference's own collate_fn — oracle/make_golden.py)
  idx2sentence       runners/base_runner.py:146-157         (pinned the same way through BaseRunner._convert_idx2sentence)
  eval_replicate     runners/pytorch_runner_vae.py:101-104  (restated; the runner module needs nni/tensorboard/ignite/
  predictions        runners/base_runner.py:247-292          pycocoevalcap and its evaluate() is file-bound: not run)
  runner_forward_train  runners/pytorch_runner_vae.py:76-98
"""
import numpy as np
import torch


def collate(items, length_idxs, sort_idx=None):
    """caption_dataset.py:278-318 on a list of tuples; tensors padded with float32 zeros along dim 0."""
    items = list(items)
    if sort_idx:
        items = sorted(items, key=lambda it: -len(it[sort_idx]))        # list.sort(reverse=True) is stable as well
    out, lens, last = [], [], None
    for i in range(len(items[0])):
        col = [it[i] for it in items]
        if torch.is_tensor(col[0]):
            if col[0].dim() == 0:
                last = torch.stack(col)
            elif col[0].shape[0] > 1:
                n0 = [c.shape[0] for c in col]
                tail = tuple(max(c.shape[d] for c in col) for d in range(1, col[0].dim()))
                pad = torch.zeros((len(col), max(n0)) + tail, dtype=torch.float32)
                for r, c in enumerate(col):
                    pad[r, :n0[r]] = c.to(torch.float32)
                last = pad
                if i in length_idxs:
                    lens.append(np.array(n0))
        else:
            last = tuple(col)
        out.append(last)
    return out + lens


def idx2sentence(word_ids, idx2word, zh=False):
    """base_runner.py:146-157."""
    words = []
    for w in word_ids:
        token = idx2word[int(w)]
        if token == "<end>":
            break
        if token == "<start>":
            continue
        words.append(token)
    return words if zh else " ".join(words)


def eval_replicate_reference(keys, feats, feat_lens, n):
    """pytorch_runner_vae.py:101-104 exactly as written: keys clip-major, features tiled batch-major."""
    return [k for k in keys for _ in range(n)], feats.repeat(n, 1, 1), [l for l in feat_lens for _ in range(n)]


def predictions(keys_per_batch, seqs_per_batch, idx2word, zh=False):
    """base_runner.py:247-292: sentences per key in arrival order -> the prediction document."""
    key2pred = {}
    for keys, seqs in zip(keys_per_batch, seqs_per_batch):
        for idx, seq in enumerate(np.asarray(seqs)):
            if seq.ndim > 1:
                for i in range(seq.shape[0]):
                    key2pred.setdefault(keys[idx], []).append(idx2sentence(seq[i], idx2word, zh))
            else:
                key2pred.setdefault(keys[idx], []).append(idx2sentence(seq, idx2word, zh))
    data = []
    for key, pred in key2pred.items():
        if len(pred) > 1:
            data.append({"filename": key, "captions": [
                {"caption": "".join(p) if zh else p, "cap_id": i, "tokens": " ".join(p) if zh else p}
                for i, p in enumerate(pred)]})
        else:
            data.append({"filename": key, "caption": "".join(pred[0]) if zh else pred[0],
                         "tokens": " ".join(pred[0]) if zh else pred[0]})
    return {"predictions": data}


def packed(x, lens):
    """torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True).data (pytorch_runner_vae.py:89-96)."""
    return torch.nn.utils.rnn.pack_padded_sequence(x, torch.as_tensor(np.asarray(lens)), batch_first=True).data
