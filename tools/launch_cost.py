"""Host cost per tiny launch: default stream, a second stream, alternating between the two; GPU idle or busy."""
import time, torch
a = torch.randn(8192, 8192, device="cuda")
x = torch.zeros(64, device="cuda"); y = torch.zeros(64, device="cuda")
side = torch.cuda.Stream()
for _ in range(3): b = a @ a
with torch.cuda.stream(side): y.add_(1.0)
torch.cuda.synchronize()
def run(label, fn, busy):
    torch.cuda.synchronize()
    if busy:
        for _ in range(10): b = a @ a
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-40s busy=%d  %6.2f us per launch" % (label, busy, dt / 2000 * 1e6))
def on_main():
    for _ in range(2000): x.add_(1.0)
def on_side():
    with torch.cuda.stream(side):
        for _ in range(2000): y.add_(1.0)
def alternate(k):
    def f():
        for i in range(2000 // (2 * k)):
            for _ in range(k): x.add_(1.0)
            with torch.cuda.stream(side):
                for _ in range(k): y.add_(1.0)
    return f
for busy in (0, 1):
    run("default stream", on_main, busy)
    run("second stream", on_side, busy)
    run("alternate every launch", alternate(1), busy)
    run("alternate every 5 launches", alternate(5), busy)
    run("alternate every 50 launches", alternate(50), busy)

# ---- the same through libacvae_hip.so (one launch per call)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acvae_amd import _lib
A = torch.randn(32, 512, device="cuda"); W = torch.randn(1536, 512, device="cuda"); C = torch.empty(32, 1536, device="cuda")
T_in = torch.randn(64, 64, device="cuda"); T_out = torch.empty(64, 64, device="cuda")
def lib_calls(stream, which):
    st = stream.cuda_stream
    def f():
        for _ in range(2000):
            if which == "transpose":
                _lib.call("acvae_transpose", T_in, 64, T_out, 64, 64, 64, st)
            else:
                _lib.call("acvae_gemm_nt", A, 512, W, 512, None, C, 1536, 32, 1536, 512, 0, st)
    return f
main = torch.cuda.current_stream()
for busy in (0, 1):
    for which in ("transpose", "gemm_nt"):
        run("lib %s on default stream" % which, lib_calls(main, which), busy)
        run("lib %s on second stream" % which, lib_calls(side, which), busy)
