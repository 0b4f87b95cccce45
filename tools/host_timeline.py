"""Host-side wall time of the phases of TrainStep.step with no synchronisation in between: a phase whose host time
tracks the GPU time is where the host blocks (pageable copies, allocator, queue depth)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd import _lib
from acvae_amd.trainer import TrainStep

model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0)
feats, caps, feat_lens, cap_lens = bench.synthetic(1)
feats = feats.cuda()
for _ in range(3):
    ts.step(feats, feat_lens.copy(), caps, cap_lens, 1.0, 0, 0.5)
torch.cuda.synchronize()
acc = {}
def mark(k, t): acc[k] = acc.get(k, 0.0) + (time.perf_counter() - t)
import acvae_amd.vae_model as vm
_call, _h2d = _lib.call, _lib.h2d
def call(name, *a):
    t = time.perf_counter(); r = _call(name, *a); mark("    C " + name, t); return r
def h2d(*a, **k):
    t = time.perf_counter(); r = _h2d(*a, **k); mark("    h2d", t); return r
_lib.call, _lib.h2d = call, h2d
orig_prep = model._host_prepare
def prep(*a, **k):
    t = time.perf_counter(); r = orig_prep(*a, **k); mark("  host_prepare", t); return r
model._host_prepare = prep
orig_enc = model.encoder.forward
def encf(*a, **k):
    t = time.perf_counter(); r = orig_enc(*a, **k); mark("  encoder.forward", t); return r
model.encoder.forward = encf
orig_q = model.qnet.forward
def qf(*a, **k):
    t = time.perf_counter(); r = orig_q(*a, **k); mark("  qnet.forward", t); return r
model.qnet.forward = qf
orig_sw = model.stepwise_forward
def sw(*a, **k):
    t = time.perf_counter(); r = orig_sw(*a, **k); mark("  stepwise_forward", t); return r
model.stepwise_forward = sw
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t00 = time.perf_counter()
for _ in range(n):
    t = time.perf_counter()
    for p in ts.order: p.grad = None
    mark("zero_grad", t); t = time.perf_counter()
    loss, parts, _ = ts.forward_loss(feats, feat_lens.copy(), caps, cap_lens, 1.0, 0, 0.5)
    mark("forward_loss", t); t = time.perf_counter()
    ts.exchange.begin(); loss.backward(); g = ts.exchange.finish()
    mark("backward", t); t = time.perf_counter()
    ts._check_grad_aliasing()
    mark("check_alias", t); t = time.perf_counter()
    st = _lib.current_stream()
    _lib.call("acvae_grad_norm", ts.flat_g, ts.n_active, g, ts.norm_partials, ts.total_norm, st)
    ts.step_count += 1
    _lib.call("acvae_adam_step", ts.flat_p, ts.flat_g, ts.exp_avg, ts.exp_avg_sq, ts.n_active, ts.lr, ts.betas[0], ts.betas[1],
              ts.eps, ts.weight_decay, ts.step_count, g, 1.0, ts.total_norm, st)
    mark("clip_adam", t)
host = time.perf_counter() - t00
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print("host loop %.2f ms/step, with final sync %.2f ms/step" % (host / n * 1e3, tot / n * 1e3))
for k, v in acc.items():
    print("%-22s %.3f ms/step" % (k, v / n * 1e3))
