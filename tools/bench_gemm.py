"""Time acvae_gemm_nt / acvae_gemm_tn at a few shapes (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acvae_amd import _lib
def bench(fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9
S = _lib.current_stream()
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 4096), (524288, 128, 1152), (2048000, 64, 576), (32768, 512, 4608)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    ms, tf = bench(lambda: _lib.call("acvae_gemm_nt", a, K, b, K, None, c, N, M, N, K, 0, S), 2.0 * M * N * K)
    print(f"NT {M}x{N}x{K}: {ms:.3f} ms {tf:.1f} TF")
    del a, b, c
for (M, N, K) in [(4096, 4096, 4096), (128, 1152, 524288), (512, 4608, 32768)]:
    a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda"); c = torch.empty(M, N, device="cuda")
    wsb = _lib.call("acvae_gemm_tn_workspace_bytes", M, N, K); ws = torch.empty(max(wsb, 4) // 4, device="cuda")
    ms, tf = bench(lambda: _lib.call("acvae_gemm_tn", a, M, b, N, c, N, M, N, K, 0, ws, wsb, S), 2.0 * M * N * K)
    print(f"TN {M}x{N}x{K}: {ms:.3f} ms {tf:.1f} TF")
    del a, b, c, ws
