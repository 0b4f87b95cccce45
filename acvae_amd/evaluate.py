"""N2 — the evaluation wrapper around the inference twin (host logic).

Mirrors ``BaseRunner.evaluate`` / ``_convert_idx2sentence`` (``runners/base_runner.py:146-157, 199-293``): decode every
clip of an evaluation set (greedy, N z-samples per clip, beam or diverse beam search), turn the token ids into
sentences with the training vocabulary and write the prediction file the reference's scoring scripts read:

    {"predictions": [{"filename": id, "caption": str, "tokens": str}, ...]}                       one caption per clip
    {"predictions": [{"filename": id, "captions": [{"caption", "cap_id", "tokens"}, ...]}, ...]}  N captions per clip

Scoring itself (pycocoevalcap BLEU/ROUGE/CIDEr/METEOR/SPICE, base_runner.py:295-320) is outside the path.
"""
import io
import json
import pickle
from pathlib import Path

import torch

from .batch import collate_fn, forward_batch, forward_batch_shared_encoder


class Vocabulary(object):
    """``utils/build_vocab.py:9-28``: word <-> index maps; unknown words map to ``<unk>``."""

    def __init__(self):
        self.word2idx = {}
        self.idx2word = {}
        self.idx = 0

    def add_word(self, word):
        if word not in self.word2idx:
            self.word2idx[word] = self.idx
            self.idx2word[self.idx] = word
            self.idx += 1

    def __call__(self, word):
        return self.word2idx[word] if word in self.word2idx else self.word2idx["<unk>"]

    def __len__(self):
        return len(self.word2idx)


class _VocabUnpickler(pickle.Unpickler):
    """The reference pickles ``utils.build_vocab.Vocabulary`` instances (``config["vocab_file"]``); resolve that
    class name to the one above so its vocabulary files load without the reference tree on the path."""

    def find_class(self, module, name):
        if name == "Vocabulary" and module.split(".")[-1] == "build_vocab":
            return Vocabulary
        return super().find_class(module, name)


def load_vocabulary(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return _VocabUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, "rb") as fh:
        return _VocabUnpickler(fh).load()


def convert_idx2sentence(word_ids, vocabulary, zh=False):
    """base_runner.py:146-157: words up to (not including) the first ``<end>``, ``<start>`` dropped; a space-joined
    string, or the word list itself when ``zh``."""
    words = []
    for word_id in word_ids:
        word = vocabulary.idx2word[int(word_id)]
        if word == "<end>":
            break
        if word != "<start>":
            words.append(word)
    return words if zh else " ".join(words)


def collect_predictions(keys, seqs, vocabulary, zh, key2pred):
    """base_runner.py:247-266: append the sentence(s) of every row of ``seqs`` ([rows, len], or [rows, k, len] when a
    search returns k hypotheses per row) to ``key2pred[key of that row]``."""
    for key, seq in zip(keys, seqs):
        rows = seq if getattr(seq, "ndim", 1) > 1 else [seq]
        for row in rows:
            key2pred.setdefault(key, []).append(convert_idx2sentence(row, vocabulary, zh))
    return key2pred


def predictions_payload(key2pred, zh=False):
    """base_runner.py:272-292: the JSON document, clips in first-seen order."""
    def entry(pred):
        return {"caption": "".join(pred) if zh else pred, "tokens": " ".join(pred) if zh else pred}

    out = []
    for key, preds in key2pred.items():
        if len(preds) > 1:
            caps = []
            for i, pred in enumerate(preds):
                e = entry(pred)
                caps.append({"caption": e["caption"], "cap_id": i, "tokens": e["tokens"]})
            out.append({"filename": key, "captions": caps})
        else:
            e = entry(preds[0])
            out.append({"filename": key, "caption": e["caption"], "tokens": e["tokens"]})
    return {"predictions": out}


def evaluate(model, items, vocabulary, caption_output=None, zh=False, batch_size=1, device=None, **kwargs):
    """Decode an evaluation set and (optionally) write the prediction file.

    ``items``: iterable of ``(audio_id, feature [T, F] tensor)`` in the order of the reference's ``CaptionEvalDataset``;
    they are batched with ``collate_fn([1])`` exactly as its DataLoader does.  ``kwargs`` go to the model as in
    ``evaluate(**kwargs)`` there: ``method`` ("greedy" | "beam" | "dbs"), ``beam_size`` (with "greedy": z-samples per
    clip), ``max_length``.  Returns the payload dict."""
    kwargs.setdefault("method", "greedy")
    kwargs.setdefault("beam_size", 1)
    collate = collate_fn([1, ])
    model.eval()
    key2pred = {}
    pending = []

    def flush():
        if not pending:
            return
        batch = collate(list(pending))
        pending.clear()
        with torch.no_grad():
            if kwargs["beam_size"] > 1 and kwargs["method"] != "dbs":     # N samples per clip: one encoder pass per clip
                output = forward_batch_shared_encoder(model, batch, device=device, **kwargs)
            else:
                output = forward_batch(model, batch, "eval", device=device, **kwargs)
        collect_predictions(batch[0], output["seqs"].cpu().numpy(), vocabulary, zh, key2pred)

    for item in items:
        pending.append(item)
        if len(pending) == batch_size:
            flush()
    flush()
    payload = predictions_payload(key2pred, zh)
    if caption_output is not None:
        with open(Path(caption_output), "w") as fh:
            json.dump(payload, fh, indent=4)
    return payload
