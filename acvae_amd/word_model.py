"""Base of the caption models, after ``models/word_model.py`` ``CaptionModel`` (:14-44): the vocabulary index
constants the rest of the path relies on and the (encoder, decoder) pair.  The ``forward`` arity contract of :46-65
(4 inputs = training, 2 inputs = inference) is implemented by ``Hybrid_VAEModel.forward``, the only model of this path."""
import torch.nn as nn


class CaptionModel(nn.Module):
    pad_idx = 0
    start_idx = 1
    end_idx = 2
    max_length = 20

    def __init__(self, encoder: nn.Module, decoder: nn.Module, **kwargs):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.vocab_size = decoder.vocab_size
        if "freeze_encoder" in kwargs and kwargs["freeze_encoder"]:
            for param in self.encoder.parameters():
                param.requires_grad = False

    @classmethod
    def set_index(cls, start_idx, end_idx):
        cls.start_idx = start_idx
        cls.end_idx = end_idx
