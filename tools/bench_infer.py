"""Inference twin (BASELINE configs[4]: greedy decode, N = 5 z-samples per clip): captions/s through evaluate()."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
import torch
import bench
from acvae_amd import evaluate as EV
B, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 5, 1000
model = bench.build_model().cuda().eval()
voc = EV.Vocabulary()
for w in ["<pad>", "<start>", "<end>", "<unk>"] + [f"w{i}" for i in range(bench.V - 4)]:
    voc.add_word(w)
g = torch.Generator().manual_seed(1)
items = [(f"clip{i}", torch.randn(T, 64, generator=g)) for i in range(B * 4)]
import json
records = []
for method, bs, rng in (("greedy", N, None), ("greedy", 1, None), ("sample", N, "host"), ("sample", N, "device"),
                        ("gumbel", N, "host"), ("gumbel", N, "device"), ("beam", 3, None), ("dbs", 5, None)):
    kw = dict(method=method, beam_size=bs, max_length=20, batch_size=B)
    if rng:
        kw.update(rng=rng, temp=1.0)
    if method == "dbs":
        kw["group_size"] = 5
    n_items = items[:64] if method == "dbs" else items
    EV.evaluate(model, n_items[:B], voc, **kw)          # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = EV.evaluate(model, n_items, voc, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    caps = sum(len(p.get("captions", [0])) for p in out["predictions"])
    print("%-7s beam_size=%d batch=%d rng=%s: %d clips, %d captions in %.3f s = %.0f clips/s, %.0f captions/s" % (
        method, bs, B, rng, len(n_items), caps, dt, len(n_items) / dt, caps / dt), flush=True)
    records.append({"method": method, "beam_size": bs, "batch_size": B, "rng": rng, "clips": len(n_items), "captions": caps,
                    "seconds": dt, "clips_per_s": len(n_items) / dt, "captions_per_s": caps / dt})
if os.environ.get("INFER_JSON"):
    with open(os.environ["INFER_JSON"], "w") as fh:
        json.dump({"workload": "configs[4] twin: Cnn10 + GRU/attention decoder, T=%d, V=%d, max_length 20, evaluate() end to end "
                   "(collate, upload, encoder, decode, ids -> sentences)" % (T, bench.V), "records": records}, fh, indent=1)
if os.environ.get("INFER_PROFILE"):
    import cProfile, pstats
    kw = dict(method="greedy", beam_size=5, max_length=20, batch_size=B)
    pr = cProfile.Profile(); pr.enable(); EV.evaluate(model, items, voc, **kw); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
