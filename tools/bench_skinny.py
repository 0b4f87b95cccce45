"""Time the skinny (M=32) GEMM shapes of the decode / BPTT steps in isolation, cache-cold-ish (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acvae_amd import _lib
S = _lib.current_stream()
flush = torch.empty(64 << 20, device="cuda")           # 256 MB: evict L2 + most of the Infinity Cache between calls
for (M, N, K) in [(32, 512, 512), (32, 1536, 512), (32, 2048, 1024), (32, 512, 1536), (32, 512, 2048), (32, 1024, 512), (32, 5000, 512)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    ts = []
    for it in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.call("acvae_gemm_nt", a, K, b, K, None, c, N, M, N, K, 0, S); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    warm = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.call("acvae_gemm_nt", a, K, b, K, None, c, N, M, N, K, 0, S); e1.record()
        torch.cuda.synchronize(); warm.append(e0.elapsed_time(e1) * 1e3)
    warm.sort()
    print(f"{M}x{N}x{K}: cold median {ts[6]:.1f} us, warm median {warm[6]:.1f} us, weights {N*K*4/1e6:.1f} MB")
