"""acvae_amd — MI355X-native (gfx950) implementation of AC-VAE's training hot path.

Host side: Python mirrors of the reference's module classes (same names, constructor arguments,
forward signatures, output-dict keys and state-dict names).  Device side: hand-written HIP kernels in
libacvae_hip.so behind the C ABI of include/acvae_hip.h.  No CPU fallback exists by design.
"""
import os as _os

# A training step enqueues ~1000 small kernels; the HIP runtime stages every launch's kernel-argument segment in a
# 1 MiB ring by default and the launching thread stalls when the ring wraps onto launches that have not run yet, which
# keeps the host barely one step ahead of the GPU (measured: 37 us instead of 4.5 us per skinny-GEMM launch behind
# 100 ms of queued work).  A larger ring lets the host run ahead.  Read by the runtime when it initialises, so it only
# takes effect if this package (or the launcher's environment) sets it before the first HIP call of the process.
_os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))

from . import _lib  # noqa: E402,F401
