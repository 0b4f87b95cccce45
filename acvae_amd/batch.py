"""A0 — the batch contract on either side of the hot path (host logic, no device arithmetic).

``collate_fn`` mirrors ``datasets/caption_dataset.py:278-318`` (the DataLoader collate: optional sort by the length of
one field, zero padding into float32, lengths of the requested fields appended at the end) and ``forward_batch``
mirrors ``Runner._forward`` (``runners/pytorch_runner_vae.py:76-108``): which batch slots feed the model in each mode,
the packed logits / targets of the training loss and the N-samples-per-clip replication of the evaluation mode.
"""
from typing import List

import numpy as np
import torch


def _pad_stack(seqs):
    """Zero-pad tensors that differ in their first dimension into one float32 [n, max...] tensor (captions therefore
    arrive as floats, F10) and return it with the first-dimension sizes as a numpy array (caption_dataset.py:286-298)."""
    shapes = np.array([tuple(t.shape) for t in seqs])
    # numpy row copies: torch's copy_/zeros fan each of these small operations out over the whole intra-op thread pool,
    # which costs tens of milliseconds per batch on a host with fewer cores than pool threads
    out = np.empty((len(seqs),) + tuple(shapes.max(axis=0)), dtype=np.float32)
    sizes = shapes[:, 0]
    for row, t in enumerate(seqs):
        n = sizes[row]
        np.copyto(out[row, :n], t.detach().numpy()[:n], casting="unsafe")
        out[row, n:] = 0
    return torch.from_numpy(out), sizes


def collate_fn(length_idxs: List = [], sort_idx=None):
    """caption_dataset.py:278-318.  Returns the collate callable the reference hands to its DataLoaders:
    ``collate_fn([0, 1], 1)`` for training items ``(feature [T,F], caption [L], audio_id)`` and ``collate_fn([1])``
    for evaluation items ``(audio_id, feature)``.  Output = the collated fields in item order, followed by the length
    arrays of the fields named in ``length_idxs``."""

    def collate(items):
        if sort_idx:                       # sort_idx = 0 does not sort, as in the reference (:283)
            items.sort(key=lambda item: len(item[sort_idx]), reverse=True)
        fields, lengths = [], []
        current = None
        for pos, column in enumerate(zip(*items)):
            head = column[0]
            if not isinstance(head, torch.Tensor):
                current = column                                   # e.g. the tuple of audio ids
            elif head.dim() == 0:
                current = torch.as_tensor(column)
            elif head.size(0) > 1:
                current, sizes = _pad_stack(column)
                if pos in length_idxs:
                    lengths.append(sizes)
            # a tensor column whose first dimension is 1 is not collated: the slot repeats the previous column (:303-311)
            fields.append(current)
        return fields + lengths

    return collate


def pack_rows(x, lens):
    """``pack_padded_sequence(x, lens, batch_first=True).data`` for lens sorted descending: rows in time-major order
    (all clips that still run at t = 0, then t = 1, ...).  One device gather; x is [N, T, ...]."""
    lens = np.asarray(lens).astype(np.int64)
    if len(lens) > 1 and np.any(np.diff(lens) > 0):
        raise RuntimeError("`lengths` array must be sorted in decreasing order")      # torch's own message
    idx = [n * x.shape[1] + t for t in range(int(lens.max())) for n in range(len(lens)) if t < lens[n]]
    idx = torch.as_tensor(idx, dtype=torch.long)
    if x.device.type == "cuda":
        from . import _lib
        idx = _lib.h2d(idx, x.device)
    return x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:])).index_select(0, idx)


def replicate_for_sampling(keys, feats, feat_lens, n):
    """pytorch_runner_vae.py:101-104 — N z-samples per clip: every clip appears n times in the batch.

    The reference repeats the keys clip by clip ([k0]*n + [k1]*n ...) but tiles the features batch by batch
    (``feats.repeat(n, 1, 1)``), which only agrees for its default evaluation batch size of 1.  Here both
    are clip-major, which equals the reference for one clip per batch and keeps captions with their clip otherwise."""
    keys = [k for k in keys for _ in range(n)]
    feats = feats.repeat_interleave(n, dim=0)
    feat_lens = [l for l in feat_lens for _ in range(n)]
    return keys, feats, feat_lens


def forward_batch_shared_encoder(model, batch, device=None, **kwargs):
    """Evaluation forward with N z-samples per clip that runs the encoder ONCE per clip: in evaluation mode (running
    BatchNorm statistics, no dropout) the encoder output of a replica equals that of its clip bit for bit, so the
    replicas share it and only the decode loop sees N rows per clip.  Same outputs as ``forward_batch(mode="eval")``
    with the clip-major replication; ``batch[0]`` is replaced by the replicated keys likewise."""
    n = kwargs["beam_size"]
    assert not model.training and n > 1 and kwargs["method"] != "dbs"
    device = device if device is not None else next(model.parameters()).device
    encoded = model.encoder(batch[1].to(device), batch[-1])
    batch[0] = [k for k in batch[0] for _ in range(n)]
    lens = torch.as_tensor(encoded["audio_embeds_lens"]).repeat_interleave(n, dim=0)
    rep = {"audio_embeds": encoded["audio_embeds"].repeat_interleave(n, dim=0),
           "audio_embeds_pooled": encoded["audio_embeds_pooled"].repeat_interleave(n, dim=0),
           "audio_embeds_lens": lens, "state": None}
    return model.inference_forward(rep, **kwargs)


def forward_batch(model, batch, mode, device=None, **kwargs):
    """Runner._forward (pytorch_runner_vae.py:76-108).  ``batch`` is what ``collate_fn`` returned.  In evaluation mode
    with beam_size > 1 and a method other than "dbs", ``batch[0]`` (the keys) is replaced by the replicated keys, as
    the reference does in place."""
    assert mode in ("train", "validation", "eval")
    device = device if device is not None else next(model.parameters()).device
    if mode == "train":
        feats = batch[0].to(device)
        caps, feat_lens, cap_lens = batch[1], batch[-2], batch[-1]
        lens1 = np.asarray(cap_lens) - 1
        output = model(feats, feat_lens, caps, cap_lens, **kwargs)
        output["packed_logits"] = pack_rows(output["logits"], lens1)           # :94-96
        output["targets"] = pack_rows(caps[:, 1:], lens1)                      # :89-90
        return output
    feats = batch[1].to(device)
    feat_lens = batch[-1]
    if mode == "eval" and kwargs["beam_size"] > 1 and kwargs["method"] != "dbs":
        batch[0], feats, feat_lens = replicate_for_sampling(batch[0], feats, feat_lens, kwargs["beam_size"])
    return model(feats, feat_lens, **kwargs)
