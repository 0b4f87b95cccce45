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
    assert _lib.lib().acvae_abi_version() == 3


def test_bad_arguments_are_reported_not_thrown():
    lib = _lib.lib()
    # null pointers / bad dims -> negative code, no launch attempted (safe without a GPU)
    assert lib.acvae_reparam_fwd(None, 0, None, 0, None, None, None, 0, None, 0, 4, 4, None) == -1
    assert lib.acvae_gemm_nt(None, 0, None, 0, None, None, 0, 4, 4, 4, 0, None) == -1
    assert lib.acvae_attn_fwd(None, 0, 0, None, None, None, None, None, 0, 0, None, 0, 0, 1, 1, 1, 1, 1, None, 0, None, 0) == -1
    assert lib.acvae_persist_status_register(-1, None) == -1
    # workspace sizes are host arithmetic: 0 where the split-over-frames form never runs, counters + partials where it does
    assert lib.acvae_attn_fwd_workspace_bytes(32, 21, 62, 512, 512) == 0
    assert lib.acvae_attn_fwd_workspace_bytes(16, 1, 187, 512, 512) == 1024 + 16 * 12 * 516 * 4


def test_library_owns_no_device_memory_and_no_behaviour_switches():
    """SURVEY 8(b) Ownership: the caller owns every buffer, workspaces included - no hipMalloc / hipFree (nor the managed /
    async / host-allocating variants) anywhere in csrc/, and no process-global acvae_set_* switches in the ABI."""
    import pathlib
    import re
    root = pathlib.Path(__file__).resolve().parents[1]
    banned = re.compile(r"\bhip(Malloc\w*|Free\w*|HostMalloc|HostAlloc|MallocManaged|MemPool\w*)\s*\(")
    for path in sorted((root / "acvae_amd" / "csrc").glob("*")):
        if path.suffix in (".hip", ".h"):
            code = re.sub(r"//[^\n]*", "", path.read_text())
            assert not banned.search(code), f"{path.name} allocates or frees memory"
    protos, _ = _lib.parse_header()
    assert not [n for n in protos if n.startswith("acvae_set_")]
    assert "ACVAE_DEV_LIB" not in (root / "acvae_amd" / "_lib.py").read_text()


def test_product_never_reaches_into_the_oracle():
    """oracle/ is test infrastructure: nothing under acvae_amd/ may import, load or execute it (only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg may), and the product has no CPU code path to fall back to."""
    import ast
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    banned = {"acvae_oracle", "host_oracle", "ref_shim", "make_golden", "oracle"}
    for path in sorted((root / "acvae_amd").rglob("*.py")):
        src = path.read_text()
        for node in ast.walk(ast.parse(src)):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] in banned for n in names), f"{path} imports {names}"
        assert "oracle" not in src.lower().replace("the oracle", "").replace("oracle's", ""), f"{path} mentions oracle/"
    for path in sorted((root / "acvae_amd" / "csrc").glob("*")):
        if path.suffix in (".hip", ".h"):
            assert "oracle" not in path.read_text().lower(), f"{path} mentions the oracle"


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the host-side mirrors fail loudly when handed CPU tensors instead of computing on the host."""
    import numpy as np
    import pytest
    import torch
    from acvae_amd.encoder import Cnn10
    enc = Cnn10(64, 512)
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X|cuda|GPU"):
        enc(torch.zeros(1, 64, 64), np.array([64]))


def test_hot_kernels_keep_everything_in_registers():
    """The persistent decode / posterior kernels and the Winograd kernels run one workgroup per CU at (or near) the 256-register
    limit: a register the compiler cannot place goes to scratch memory, i.e. a global-memory round trip inside the step loop
    (round 3 shipped decode_persist_bwd_kernel with 28 B per lane of it).  The build keeps hipcc's per-kernel resource remarks
    (-Rpass-analysis=kernel-resource-usage) beside every object file; any scratch in these kernels fails here."""
    from acvae_amd import build as b
    ge.build()
    usage = b.resource_usage()
    hot = ("decode_persist_kernel", "decode_persist_bwd_kernel", "posterior_persist_fwd_kernel", "posterior_persist_bwd_kernel",
           "conv_wino_kernel", "conv_wino_bnred_kernel", "conv_wino_stats_kernel", "conv_wino_act_kernel", "conv_wino_wgrad_kernel_s1", "conv_wino_wgrad_kernel_s2",
           "conv_wino_wgrad_kernel_s3", "conv_wino_wgrad_kernel_s4")
    for k in hot:
        hits = {n: u for n, u in usage.items() if k + "E" in n or n.endswith(k)}
        assert hits, f"{k}: no resource record (rebuild with python -m acvae_amd.build --force)"
        for n, u in hits.items():
            assert u.get("scratch", -1) == 0, f"{n}: {u.get('scratch')} bytes per lane of scratch"
            assert u.get("vgprs", 999) <= 256, (n, u)
