"""GPU time of the phases of a training step measured with events on the main stream in an un-profiled run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd import _lib
from acvae_amd.trainer import TrainStep

DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
model = bench.build_model().cuda().train()
ts = TrainStep(model, bench.V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0, precision=DT)
feats, caps, feat_lens, cap_lens = bench.synthetic(1)
feats = feats.cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
orig_enc = model.encoder.forward
def encf(*a, **k):
    r = orig_enc(*a, **k); ev("encoder fwd")
    r["audio_embeds"].register_hook(lambda g: ev("decode bwd"))
    return r
model.encoder.forward = encf
orig_sw = model.stepwise_forward
def sw(*a, **k):
    r = orig_sw(*a, **k); ev("decode fwd"); return r
model.stepwise_forward = sw
for _ in range(3):
    ts.step(feats, feat_lens.copy(), caps, cap_lens, 1.0, 0, 0.5)
torch.cuda.synchronize(); marks.clear()
for _ in range(n):
    ev("start")
    for p in ts.order: p.grad = None
    loss, parts, _ = ts.forward_loss(feats, feat_lens.copy(), caps, cap_lens, 1.0, 0, 0.5); ev("loss fwd")
    ts.exchange.begin(); loss.backward(); g = ts.exchange.finish(); ev("encoder bwd")
    st = _lib.current_stream()
    _lib.call("acvae_grad_norm", ts.flat_g, ts.n_active, g, ts.norm_partials, ts.total_norm, st)
    ts.step_count += 1
    _lib.call("acvae_adam_step", ts.flat_p, ts.flat_g, ts.exp_avg, ts.exp_avg_sq, ts.n_active, ts.lr, ts.betas[0], ts.betas[1],
              ts.eps, ts.weight_decay, ts.step_count, g, 1.0, ts.total_norm, st); ev("clip+adam")
ev("start")
torch.cuda.synchronize()
acc = {}
for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
    acc.setdefault(t1, []).append(e0.elapsed_time(e1))
tot = 0
for k in ("encoder fwd", "decode fwd", "loss fwd", "decode bwd", "encoder bwd", "clip+adam", "start"):
    v = sorted(acc[k]); m = v[len(v) // 2]; tot += m
    print("%-14s median %7.3f ms" % (k if k != "start" else "(step gap)", m))
print("sum %.3f ms" % tot)
