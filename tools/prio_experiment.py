"""A/B: the training step issued on a HIGH-priority HIP stream (the posterior's side stream stays at normal priority), so
that the dispatcher prefers the MFMA-bound conv workgroups over the small text-side kernels that share the GPU with them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd.trainer import TrainStep

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0, precision=dtype)
feats, caps, fl, cl = bench.synthetic(1)
feats = feats.cuda()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")


def run(stream, n=30, warm=8):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(warm):
            ts.step(feats, fl.copy(), caps, cl, 1.0, 0, 0.5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            ts.step(feats, fl.copy(), caps, cl, 1.0, 0, 0.5)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


hp = torch.cuda.Stream(priority=-1)
for rep in range(2):
    print("default stream      %.3f ms/step" % run(None))
    print("high-priority main  %.3f ms/step" % run(hp))
model.use_side_stream = False
print("no side stream      %.3f ms/step" % run(None))
