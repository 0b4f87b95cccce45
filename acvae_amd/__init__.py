"""acvae_amd — MI355X-native (gfx950) implementation of AC-VAE's training hot path.

Host side: Python mirrors of the reference's module classes (same names, constructor arguments,
forward signatures, output-dict keys and state-dict names).  Device side: hand-written HIP kernels in
libacvae_hip.so behind the C ABI of include/acvae_hip.h.  No CPU fallback exists by design.
"""
from . import _lib  # noqa: F401
