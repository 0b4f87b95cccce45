"""CPU: the C-ABI library builds, loads and exports every symbol include/acvae_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os

import __graft_entry__ as ge
from acvae_amd import _lib


def test_build_and_symbols():
    ge.build()
    assert os.path.exists(_lib.LIB_PATH)
    protos, _ = _lib.parse_header()
    assert len(protos) >= 10
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(so, name), f"{name} declared in include/acvae_hip.h but not exported"
    assert _lib.lib().acvae_abi_version() == 1


def test_bad_arguments_are_reported_not_thrown():
    lib = _lib.lib()
    # null pointers / bad dims -> negative code, no launch attempted (safe without a GPU)
    assert lib.acvae_reparam_fwd(None, 0, None, 0, None, None, None, 0, None, 0, 4, 4, None) == -1
    assert lib.acvae_gemm_nt(None, 0, None, 0, None, None, 0, 4, 4, 4, 0, None) == -1
    assert lib.acvae_attn_fwd(None, 0, 0, None, None, None, None, None, 0, 0, None, 0, 0, 1, 1, 1, 1, 1, None) == -1
