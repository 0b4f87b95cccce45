"""Mirror of ``models/word_model.py`` ``CaptionModel`` (:14-65): class constants, constructor and the
``forward`` arity contract (4 inputs = training, 2 inputs = inference)."""
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

    def forward(self, *input, **kwargs):
        """models/word_model.py:46-65"""
        if len(input) == 4:
            feats, feat_lens, caps, cap_lens = input
            encoded = self.encoder(feats, feat_lens)
            output = self.train_forward(encoded, caps, cap_lens, **kwargs)
        elif len(input) == 2:
            feats, feat_lens = input
            encoded = self.encoder(feats, feat_lens)
            output = self.inference_forward(encoded, **kwargs)
        else:
            raise Exception("Number of input should be either 4 (feats, feat_lens, caps, cap_lens) or 2 (feats, feat_lens)")
        return output
