"""How far ahead of the GPU can the host enqueue?  Queue ~200 ms of big GEMMs, then time the host side of N tiny
launches: the index where the per-launch host time jumps from microseconds to milliseconds is the in-flight limit."""
import sys, time, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
a = torch.randn(8192, 8192, device="cuda")
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
for _ in range(3): b = a @ a
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): b = a @ a
t_big = time.perf_counter() - t0
ts = []
for i in range(n):
    t1 = time.perf_counter(); x.add_(1.0); ts.append(time.perf_counter() - t1)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("enqueue 20 GEMMs %.2f ms; host done at %.1f ms; GPU done at %.1f ms" % (t_big * 1e3, t_host * 1e3, t_all * 1e3))
slow = [(i, t) for i, t in enumerate(ts) if t > 200e-6]
print("launches slower than 200 us:", len(slow), "first few:", [(i, round(t * 1e3, 2)) for i, t in slow[:12]])
import statistics
print("median launch %.1f us" % (statistics.median(ts) * 1e6))
