import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch, random
import bench
from acvae_amd.trainer import TrainStep
model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0)
feats, caps, fl, cl = bench.synthetic(1)
feats = feats.cuda()
evs = []
for i in range(40):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    ts.step(feats, fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
print(" ".join("%.1f" % a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:])))
