"""Import the reference's hot-path modules in the BUILD CONTAINER only.  TEST INFRASTRUCTURE.

/root/reference is read-only, absent on the GPU box, and two of its package files are broken as
released.  The shim is the two in-memory registrations SURVEY §8(c) describes;
nothing in the reference tree is edited, copied or compiled:

  1. an empty package object ``models`` whose ``__path__`` points at /root/reference/models, so the
     broken ``models/__init__.py`` (imports absent transformer modules) is skipped;
  2. a stub ``turtle`` module exposing ``forward`` (``models/text_encoder.py:4`` imports it; tkinter
     is absent here).

Used only by ``oracle/make_golden.py`` and ``oracle/check_oracle_vs_reference.py``.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ACVAE_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def load():
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if "models" not in sys.modules:
        pkg = types.ModuleType("models")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "models")]
        sys.modules["models"] = pkg
    if "turtle" not in sys.modules:
        t = types.ModuleType("turtle")
        t.forward = lambda *a, **k: None
        sys.modules["turtle"] = t
    import models.encoder as enc
    import models.decoder as dec
    import models.attn_model as attn
    import models.text_encoder as tenc
    import models.vae_model as vae
    import models.word_model as wm
    import utils.train_util as tu
    return types.SimpleNamespace(encoder=enc, decoder=dec, attn_model=attn, text_encoder=tenc,
                                 vae_model=vae, word_model=wm, train_util=tu)


def load_host_side():
    """The host-side modules of the batch / evaluation contract: ``datasets.caption_dataset`` (collate_fn),
    ``utils.build_vocab`` (Vocabulary) and ``runners.base_runner`` (BaseRunner._convert_idx2sentence).  They import
    packages that are absent here and that the functions under test never touch (h5py, fire, ignite): empty stand-in
    modules are registered for those names only."""
    load()
    for name in ("h5py", "fire"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "ignite" not in sys.modules:
        for name in ("ignite", "ignite.engine", "ignite.engine.engine", "ignite.contrib", "ignite.contrib.handlers"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["ignite.engine.engine"].Engine = type("Engine", (), {})
        sys.modules["ignite.contrib.handlers"].ProgressBar = type("ProgressBar", (), {})
    import datasets.caption_dataset as cd
    import utils.build_vocab as bv
    import runners.base_runner as br
    return types.SimpleNamespace(caption_dataset=cd, build_vocab=bv, base_runner=br)


def load_losses():
    """``losses/loss.py`` (the masked CrossEntropyLoss :12-37 and LabelSmoothingLoss :39-70).  It imports
    ``ignite.metrics`` and ``ignite.engine.engine.Engine`` at module level for a metric class the two loss classes
    never touch; ignite is absent here, so empty stand-in modules are registered for those two names only (as
    ``load_host_side`` does for the runner).  ``utils.score_util`` and ``models.utils`` are the reference's own files
    and import for real."""
    load()
    for name in ("ignite", "ignite.metrics", "ignite.engine", "ignite.engine.engine"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["ignite.engine.engine"], "Engine"):
        sys.modules["ignite.engine.engine"].Engine = type("Engine", (), {})
    sys.modules["ignite"].metrics = sys.modules["ignite.metrics"]
    for cls in ("Metric", "Loss"):                 # base classes of the metric wrappers at the end of the file
        if not hasattr(sys.modules["ignite.metrics"], cls):
            setattr(sys.modules["ignite.metrics"], cls, type(cls, (), {}))
    import losses.loss as loss
    return loss


def build_reference_model(ref, vocab, embed=512, hidden=512, q_hidden=None, encoder="Cnn10", dec_dropout=0.0,
                          proj_embed=None):
    """Hybrid_VAEModel(Cnn10, VAERNNBahdanauAttnDecoder, PosteriorRNN_hybrid, PriorRNN): the
    self-consistent combination of SURVEY F6, built the way runners/pytorch_runner_vae.py:33-73 does."""
    encoder = ref.encoder.Cnn10(64, 512) if encoder == "Cnn10" else ref.encoder.Cnn14_16k(64, 2048)
    decoder = ref.decoder.VAERNNBahdanauAttnDecoder(
        vocab_size=vocab, enc_mem_size=embed, embed_size=embed, hidden_size=hidden, dropout=dec_dropout,
        num_layers=1, rnn_type="GRU", attn_size=hidden)
    if proj_embed is not None:          # runners/pytorch_runner_vae.py:51-56: pretrained vectors of another width
        import numpy as np
        decoder.load_word_embeddings(np.zeros((vocab, proj_embed), dtype=np.float32), tune=True, projection=True)
    model = ref.vae_model.Hybrid_VAEModel(
        encoder, decoder, posterior_model="PosteriorRNN_hybrid",
        posterior_args={"hidden_size": q_hidden or embed, "dropout": 0.0},
        prior_model="PriorRNN", prior_args={"hidden_size": embed, "dropout": 0.0})
    return model
