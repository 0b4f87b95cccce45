"""Where an evaluation batch spends its time: collate / upload / encoder / decode / readback, per method."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd import batch as BT

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1000
model = bench.build_model().cuda().eval()
g = torch.Generator().manual_seed(1)
items = [(f"clip{i}", torch.randn(T, 64, generator=g)) for i in range(B)]
collate = BT.collate_fn([1, ])


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms, batch = timed(lambda: collate(list(items)))
print("collate            %7.2f ms" % ms)
feats, lens = batch[1], batch[2]
ms, fd = timed(lambda: feats.cuda())
print("upload (pageable)  %7.2f ms" % ms)
with torch.no_grad():
    ms, enc = timed(lambda: model.encoder(fd, torch.as_tensor(lens).clone()))
    print("encoder (eval)     %7.2f ms" % ms)
    for method, bs in (("greedy", 1), ("beam", 3)):
        kw = dict(method=method, beam_size=bs, max_length=20)
        ms, out = timed(lambda: model(fd, torch.as_tensor(lens).clone(), **kw))
        print("model(%s,%d)   %7.2f ms" % (method, bs, ms))
        ms, _ = timed(lambda: out["seqs"].cpu())
        print("  readback         %7.2f ms" % ms)
        ms, _ = timed(lambda: BT.forward_batch(model, batch, "eval", **kw))
        print("  forward_batch    %7.2f ms" % ms)

if os.environ.get("INFER_PROFILE"):
    import cProfile, pstats
    from acvae_amd import evaluate as EV
    voc = EV.Vocabulary()
    for w in ["<pad>", "<start>", "<end>", "<unk>"] + [f"w{i}" for i in range(bench.V - 4)]:
        voc.add_word(w)
    many = items * 4
    kw = dict(method="greedy", beam_size=1, max_length=20, batch_size=B)
    EV.evaluate(model, many[:B], voc, **kw)
    pr = cProfile.Profile(); pr.enable(); EV.evaluate(model, many, voc, **kw); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
