"""Training-step rate with the features handed over as HOST buffers every step (PCIe-inclusive), beside the headline
figure where they are already resident in HBM: pageable host tensor -> .cuda() each step, and page-locked host tensor ->
non_blocking copy each step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd.trainer import TrainStep

model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0)
feats, caps, fl, cl = bench.synthetic(1)
resident = feats.cuda()
pinned = feats.pin_memory()


def run(get, n=30, warm=8):
    for _ in range(warm):
        ts.step(get(), fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        ts.step(get(), fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, get in (("resident in HBM", lambda: resident), ("pageable host -> .cuda()", lambda: feats.cuda()),
                  ("page-locked host, non_blocking", lambda: pinned.cuda(non_blocking=True))):
    ms = run(get)
    print("%-34s %7.2f ms/step  %7.1f captions/s" % (name, ms, bench.B / ms * 1e3))
