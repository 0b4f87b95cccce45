"""Time the encoder forward+backward at a given shape (dev tool)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acvae_amd import _lib
if os.environ.get("ACVAE_DEV_LIB"):          # this TOOL's hook (tools/lab_wino.py, ablations): time another build of the library
    _lib.use_library(os.environ["ACVAE_DEV_LIB"])
from acvae_amd.encoder import Cnn10, Cnn14_16k
B, T = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
arch = sys.argv[4] if len(sys.argv) > 4 else "Cnn10"
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
enc = (Cnn10(64, 512, compute_dtype=dtype) if arch == "Cnn10" else Cnn14_16k(64, 2048, compute_dtype=dtype)).cuda().train()
x = torch.randn(B, T, 64, device="cuda")
R = torch.randn(B, T // enc.TIME_DIV, enc.OUT_CHANNELS, device="cuda")
def step():
    for p in enc.parameters(): p.grad = None
    o = enc(x, [T] * B)["audio_embeds"]
    o.backward(R)
for _ in range(2): step()
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
for _ in range(iters):
    for p in enc.parameters(): p.grad = None
    e0.record(); o = enc(x, [T] * B)["audio_embeds"]; e1.record(); o.backward(R); e2.record()
    torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
flops = (26.03e9 if arch == "Cnn10" else 40.1e9) * B * T / 1000   # conv MACs x2 per 1000-frame clip
print(f"B={B} T={T} fwd {tf/iters:.2f} ms ({flops/(tf/iters)/1e9:.1f} TF)  bwd {tb/iters:.2f} ms ({2*flops/(tb/iters)/1e9:.1f} TF)")
