"""Batch source on the host side of A0 (the part that needs no HDF5): caption annotations + vocabulary
-> training / evaluation items, and the order in which (clip, caption) pairs are visited.

Mirrors ``datasets/caption_dataset.py``: ``CaptionEvalDataset`` (:20-52), ``CaptionDataset`` (:66-112),
``CaptionSampler`` (:199-224), ``CaptionDistributedSampler`` (:226-276).  The reference reads the log-mel features
``[T, 64]`` of an ``audio_id`` from HDF5 files (h5py is not available here); these classes take any mapping or callable
``audio_id -> array`` instead (a dict of arrays, ``numpy.load`` on per-clip files, an HDF5 group, ...).  The annotation
format is the reference's: ``{"audios": [{"audio_id", "captions": [{"tokens": "a b c", ...}, ...]}, ...]}``.
"""
import math
import random
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

Features = Union[Dict[str, np.ndarray], Callable[[str], np.ndarray]]


def _fetch(features: Features, audio_id):
    feat = features(audio_id) if callable(features) else features[audio_id]
    return np.asarray(feat).squeeze()


class CaptionEvalDataset(torch.utils.data.Dataset):
    """Items ``(audio_id, feature [T, F])`` in the order of ``audio_ids`` (caption_dataset.py:39-52)."""

    def __init__(self, features: Features, audio_ids: Sequence[str], transform: Optional[List] = None):
        self._features, self._audio_ids, self._transform = features, list(audio_ids), transform

    def _feature(self, audio_id):
        feature = _fetch(self._features, audio_id)
        for transform in self._transform or []:
            feature = transform(feature)
        return torch.as_tensor(feature)

    def __getitem__(self, index):
        audio_id = self._audio_ids[index]
        return audio_id, self._feature(audio_id)

    def __len__(self):
        return len(self._audio_ids)


class CaptionDataset(CaptionEvalDataset):
    """Items ``(feature, caption ids with <start>/<end>, audio_id)`` addressed by ``(audio_idx, cap_idx)``
    (caption_dataset.py:89-112); the length is the number of captions."""

    def __init__(self, features: Features, caption_info: List, vocabulary, transform: Optional[List] = None):
        super().__init__(features, [info["audio_id"] for info in caption_info], transform)
        self._caption_info, self._vocabulary = caption_info, vocabulary

    def __getitem__(self, index: Tuple[int, int]):
        audio_idx, cap_idx = index
        audio_id = self._audio_ids[audio_idx]
        tokens = self._caption_info[audio_idx]["captions"][cap_idx]["tokens"].split()
        voc = self._vocabulary
        caption = torch.as_tensor([voc("<start>")] + [voc(token) for token in tokens] + [voc("<end>")])
        return self._feature(audio_id), caption, audio_id

    def __len__(self):
        return sum(len(item["captions"]) for item in self._caption_info)


def caption_pairs(caption_info: List, audio_subset_indices: Optional[Sequence[int]] = None) -> List[Tuple[int, int]]:
    """All (audio_idx, cap_idx) pairs, clip-major (the element list both samplers build)."""
    audio_idxs = audio_subset_indices if audio_subset_indices is not None else range(len(caption_info))
    return [(a, c) for a in audio_idxs for c in range(len(caption_info[a]["captions"]))]


class CaptionSampler(torch.utils.data.Sampler):
    """caption_dataset.py:199-224: every pair once, shuffled with Python's ``random`` when asked."""

    def __init__(self, data_source: CaptionDataset, audio_subset_indices: Optional[Sequence[int]] = None,
                 shuffle: bool = False):
        self._caption_info = data_source._caption_info
        self._audio_subset_indices, self._shuffle = audio_subset_indices, shuffle
        self._num_sample = None

    def __iter__(self):
        elems = caption_pairs(self._caption_info, self._audio_subset_indices)
        self._num_sample = len(elems)
        if self._shuffle:
            random.shuffle(elems)
        return iter(elems)

    def __len__(self):
        if self._num_sample is None:
            self.__iter__()
        return self._num_sample


class CaptionDistributedSampler(torch.utils.data.Sampler):
    """caption_dataset.py:226-276 (drop_last False as there): the pair list is shuffled with ``random.seed(seed +
    epoch)``, padded by wrapping to a multiple of the world size and dealt round-robin (rank, rank + world, ...).
    One process per GPU: rank / world default to the initialised process group, or may be given."""

    def __init__(self, dataset: CaptionDataset, audio_subset_indices: Optional[Sequence[int]] = None,
                 shuffle: bool = True, num_replicas: Optional[int] = None, rank: Optional[int] = None, seed: int = 0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.num_replicas, self.rank, self.shuffle, self.seed, self.epoch = num_replicas, rank, shuffle, seed, 0
        self.indices = caption_pairs(dataset._caption_info, audio_subset_indices)
        self.num_samples = math.ceil(len(self.indices) / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __iter__(self):
        if self.shuffle:
            random.seed(self.seed + self.epoch)
            random.shuffle(self.indices)          # in place, cumulative over epochs, as in the reference
        indices = list(self.indices)
        padding = self.total_size - len(indices)
        if padding <= len(indices):
            indices += indices[:padding]
        else:
            indices += (indices * math.ceil(padding / len(indices)))[:padding]
        return iter(indices[self.rank:self.total_size:self.num_replicas])

    def __len__(self):
        return self.num_samples
