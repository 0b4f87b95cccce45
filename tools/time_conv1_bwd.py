"""Times acvae_conv1_first_bwd (first convolution's weight gradient + bn0 gradients) at the configs[1] shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acvae_amd import _lib
N, T, F = 32, 1000, 64
x = torch.randn(N, T, F, device="cuda"); bn0 = torch.rand(4, 64, device="cuda") + 0.5
w1 = torch.randn(64, 1, 3, 3, device="cuda"); dy = torch.randn(N, T, F, 64, device="cuda")
dW1 = torch.empty(64, 1, 3, 3, device="cuda"); dg = torch.empty(64, device="cuda"); db = torch.empty(64, device="cuda")
wsb = int(_lib.call("acvae_conv3x3_workspace_bytes", N, T, F, 1, 64)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
fn = lambda: _lib.call("acvae_conv1_first_bwd", x, bn0, w1, dy, dW1, dg, db, ws, wsb, N, T, F, _lib.current_stream())
for _ in range(20): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): fn()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 20
print(f"conv1_first_bwd (kernel + 2 column sums): {us:.1f} us per call; dY stream {dy.numel() * 4 / us / 1e6:.2f} TB/s")
