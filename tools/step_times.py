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
evs, host = [], []
for i in range(40):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    h0 = time.perf_counter()
    ts.step(feats, fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5)
    host.append((time.perf_counter() - h0) * 1e3)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
print("gpu :", " ".join("%.1f" % a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:])))
print("host:", " ".join("%.1f" % h for h in host))
import gc
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())
# after an idle pause: do the first steps run slow again (clock / power-state ramp) or not (one-off warm-up)?
time.sleep(3.0 if os.environ.get('STEP_TIMES_IDLE') else 0.0)
evs = []
for i in range(12):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    ts.step(feats, fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
print("after 3 s idle:", " ".join("%.1f" % a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:])))
