"""Is the HIP train step bit-reproducible?  Runs forward + loss + backward of the same batch several times on fresh
gradients and compares every parameter gradient bit for bit (also: the loss).  usage: python tools/determinism_check.py [B] [T] [reps] [f32|bf16]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import bench
import acvae_oracle as O
from acvae_amd.train_util import LabelSmoothingLoss, MSELoss, Normal_kl_loss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T = int(sys.argv[2]) if len(sys.argv) > 2 else 999
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
V, L = bench.V, 22
torch.manual_seed(5)
model = bench.build_model().cuda().train()
model.encoder.p_block = model.encoder.p_fc = 0.0          # no dropout: every run sees the same network
if len(sys.argv) > 4:
    model.encoder.compute_dtype = sys.argv[4]
feats, caps, fl, cl = O.synthetic_batch(B, T, V, L, seed=4, ragged=True)
g = torch.Generator().manual_seed(3)
E = model.decoder.embed_size
noise = dict(eps_q=torch.randn(B, L - 1, E, generator=g), eps_p=torch.randn(L - 1, B, E, generator=g))
ref = None
for r in range(reps):
    for p in model.parameters():
        p.grad = None
    model.noise = noise
    random.seed(9)
    torch.manual_seed(11)                      # dropout masks (Philox seed drawn from the torch generator)
    out = model(feats.cuda(), fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0)
    lens1 = np.asarray(cl) - 1
    ce = LabelSmoothingLoss(V, 0.1).masked(out["logits"], caps[:, 1:].to(torch.long), lens1)
    kl = Normal_kl_loss()(out["q_means"], out["q_logs"], out["p_means"], out["p_logs"])
    mse = MSELoss()(out["q_means_utt"], out["p_means_utt"])
    loss = ce + 0.5 * kl + 1.0 * mse
    loss.backward()
    torch.cuda.synchronize()
    cur = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    cur["__loss__"] = loss.detach().clone()
    if ref is None:
        ref = cur
        print(f"run 0: loss {float(loss):.7f}, {len(cur) - 1} gradients")
        continue
    bad = [(n, float((cur[n] - ref[n]).abs().max())) for n in ref if not torch.equal(cur[n], ref[n])]
    print(f"run {r}: " + ("bit-identical to run 0" if not bad else f"{len(bad)} tensors differ, e.g. {bad[:6]}"))
