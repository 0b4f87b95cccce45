"""Does an asynchronous host->device copy out of page-locked memory hold the host while the GPU is busy?"""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from acvae_amd import _lib
a = torch.randn(8192, 8192, device="cuda")
for _ in range(3): b = a @ a
torch.cuda.synchronize()
big = torch.randn(21, 32, 512).pin_memory()
small = torch.arange(32).pin_memory()
page = torch.randn(21, 32, 512)
side = torch.cuda.Stream()
def probe(label, fn):
    torch.cuda.synchronize()
    for _ in range(10): b = a @ a            # ~100 ms of queued work
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-46s host time %9.1f us" % (label, dt * 1e6))
probe("pinned 1.4 MB non_blocking", lambda: big.to("cuda", non_blocking=True))
probe("pinned 256 B non_blocking", lambda: small.to("cuda", non_blocking=True))
probe("pinned 1.4 MB non_blocking (again)", lambda: big.to("cuda", non_blocking=True))
probe("ring 1.4 MB", lambda: _lib.h2d(page, "cuda"))
probe("ring 1.4 MB (again)", lambda: _lib.h2d(page, "cuda"))
probe("ring 256 B", lambda: _lib.h2d(torch.arange(32), "cuda"))
def on_side():
    with torch.cuda.stream(side):
        return big.to("cuda", non_blocking=True)
probe("pinned 1.4 MB non_blocking on an idle stream", on_side)
probe("pageable 1.4 MB", lambda: page.to("cuda"))
probe("torch.zeros on device", lambda: torch.zeros(1024, device="cuda"))
probe("d2d copy_", lambda: b.copy_(a))
