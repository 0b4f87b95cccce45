"""Folds the three rocprofv3 passes of tools/profile_bench.sh into the files kept under profiles/:
<tag>_trainstep_kernel_stats.csv (the --stats table as rocprofv3 wrote it), <tag>_bench.json,
<tag>_bench_under_rocprof.json and traffic_conv_igemm.json (per-launch HBM-side bytes of the dominant kernels,
FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950, WRITE_SIZE taken as is)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
bf16 = "bf16" in sys.argv[3:]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"no {pat} under {out}/{sub}")
    return hits[0]


shutil.copy(find("stats", "*kernel_stats.csv"), os.path.join(dst, f"{tag}_trainstep_kernel_stats.csv"))
for f in ("bench.json", "bench_under_rocprof.json"):
    shutil.copy(os.path.join(out, f), os.path.join(dst, f"{tag}_{f}"))


def counter(sub, name):
    per = defaultdict(list)
    with open(find(sub, "*counter_collection.csv")) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            short = ("conv_wgrad" if ("conv_wgrad" in k or "conv_wino_wgrad" in k) else "conv_igemm" if ("conv_igemm" in k or "conv_wino_kernel" in k or "conv_wino_act_kernel" in k
                                                                               or "conv_wino_stats_kernel" in k or "conv_wino_bnred_kernel" in k) else None)
            if short:
                per[short].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
res = {"command": "tools/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) "
                  "-- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
       "correction": "FETCH_SIZE (KB) x2 on gfx950 for 16-B/lane coalesced reads (MI355X_MICROARCH.md HBM section); "
                     "WRITE_SIZE (KB) exact; Infinity-Cache hits are counted",
       "tag": tag}
for k in ("conv_igemm", "conv_wgrad"):
    f, nf = fetch[k]
    w, _ = write[k]
    res[k] = {"dispatches_averaged": nf, "FETCH_SIZE_KB_avg_raw": f, "WRITE_SIZE_KB_avg": w,
              "fetch_bytes_per_launch": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
              "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
wino = not bf16 and os.environ.get("ACVAE_CONV_WINO", "1") != "0"
res["kernel"] = ("conv_igemm_bf16_kernel<128|64> (bf16 storage, v_mfma_f32_32x32x16_bf16)" if bf16 else
                 "conv_wino_kernel / _bnred_kernel / _stats_kernel / _act_kernel (conv3x3 as Winograd F(2x2,3x3): data gradient / data gradient + the next "
                 "BatchNorm backward's sums / forward / forward with BatchNorm + ReLU on the operand; all 14 launches of a step averaged)" if wino else
                 "conv_igemm3_kernel<128|64> (conv3x3 implicit GEMM with horizontal-tap reuse, forward + data gradient)")
res["hbm_bytes_per_launch"] = res["conv_igemm"]["hbm_bytes_per_launch"]
# one read of the conv input + one write of its output + the weights, over the 14 launches of a step at B=32, T=1000
# (7 forward convolutions + 7 data gradients, every layer but the Cin = 1 one): 5.346 GB fp32, 2.673 GB bf16
# Round 4 (fp32 Winograd): four of the data gradients also reduce the BatchNorm backward that consumes them
# (conv_wino_bnred_kernel) and read the normalised tensor for it - 0.983 GB per step that bn_bwd_reduce_kernel no longer reads
conv_bytes = 2.672934912e9 if bf16 else 5.345869824e9
fused_reads = 0.98304e9 if wino else 0.0
res["algorithmic_bytes_per_launch"] = (conv_bytes + fused_reads) / 14
res["algorithmic_reads_per_launch"] = (conv_bytes / 2 + fused_reads) / 14
res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
res["fetch_over_algorithmic_reads"] = res["conv_igemm"]["fetch_bytes_per_launch"] / res["algorithmic_reads_per_launch"]
json.dump(res, open(os.path.join(dst, f"{tag}_traffic_conv_igemm.json"), "w"), indent=1)
if not bf16:
    json.dump(res, open(os.path.join(dst, "traffic_conv_igemm.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
