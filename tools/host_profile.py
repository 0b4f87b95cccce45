"""cProfile of the host side of the training step (no synchronisation): which Python/torch call holds the host."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd.trainer import TrainStep
model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0)
feats, caps, feat_lens, cap_lens = bench.synthetic(1)
feats = feats.cuda()
def run(n):
    for _ in range(n):
        ts.step(feats, feat_lens.copy(), caps, cap_lens, 1.0, 0, 0.5)
run(3); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(int(sys.argv[1]) if len(sys.argv) > 1 else 20); pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
