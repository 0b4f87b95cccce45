"""Soak: N training steps on one fixed synthetic batch — loss must fall, stay finite, memory must not grow."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd.trainer import TrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0)
feats, caps, fl, cl = bench.synthetic(1)
feats = feats.cuda()
import psutil
proc = psutil.Process()
losses, mem, rss = [], [], []
t0 = time.perf_counter()
for i in range(n):
    random.seed(i)
    p = ts.step(feats, fl.copy(), caps, cl, 1.0, 0, 0.5)
    if i % 25 == 0 or i == n - 1:
        losses.append(float(p["loss"])); mem.append(torch.cuda.memory_reserved() >> 20); rss.append(proc.memory_info().rss >> 20)
torch.cuda.synchronize()
print("steps %d in %.1f s (%.1f ms/step)" % (n, time.perf_counter() - t0, (time.perf_counter() - t0) / n * 1e3))
print("loss every 25 steps:", [round(x, 3) for x in losses])
print("reserved MiB:", mem)
print("host RSS MiB:", rss)
assert all(x == x and abs(x) < 1e4 for x in losses) and losses[-1] < losses[0]
assert mem[-1] <= mem[2] * 1.02
assert rss[-1] <= rss[3] + 64, rss
