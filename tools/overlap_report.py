#!/usr/bin/env python3
"""How much of the gradient exchange hides behind the backward pass: reads the kernel trace(s) of ONE
`rocprofv3 --kernel-trace` run of `bench.py --gpus N` (one `*kernel_trace.csv` per rank process, or one file with several
agents) and prints, per rank and per optimiser step,

  * every RCCL kernel (the pieces of the four gradient buckets of acvae_amd/trainer.py: FlatGradExchange) with its interval,
  * the compute kernels of the step that ran while it was on the GPU (convolution / BatchNorm backward, text-side products),
  * the EXPOSED part of the exchange: time an RCCL kernel was running and no compute kernel of the same rank was, and in
    particular the tail between the end of the step's last backward kernel and the end of its last RCCL kernel -
    the only part of the all-reduce of runners/pytorch_runner_vae.py:204-207 (torch DDP) the step should still pay for.

It asserts that RCCL kernels appear on exactly N ranks (an N-rank run whose ranks never met on the wire measured nothing).
Not part of the product; nothing under acvae_amd/ imports it.

usage: python tools/overlap_report.py <dir-or-csv> [--gpus N] [--steps-from K]
       (e.g. rocprofv3 --kernel-trace -d gpurun_out/overlap -o t --output-format csv -- python3 bench.py --gpus 8 ...)
"""
import argparse
import csv
import glob
import os
import re
import sys
from collections import defaultdict

RCCL = re.compile(r"nccl|rccl", re.I)
STEP_END = "adam_kernel"                     # one launch per optimiser step (csrc/optim.hip)
BACKWARD = re.compile(r"conv_wino|conv_igemm|conv_wgrad|bn_bwd|conv1_first_bwd|decode_persist_bwd|posterior_persist_bwd|gemm_tn|"
                      r"slab_reduce|colsum|wino_wgrad_reduce|embed_scatter|attn_bwd", re.I)


def load(path):
    """-> {rank key: [(start_ns, end_ns, kernel name)]} from one CSV or every *kernel_trace.csv below a directory."""
    files = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        raise SystemExit(f"overlap_report: no *kernel_trace.csv under {path}")
    ranks = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (os.path.basename(os.path.dirname(f)) if len(files) > 1 else "", r.get("Agent_Id", ""))
                ranks[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for v in ranks.values():
        v.sort()
    return dict(ranks)


def union_len(intervals):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)


def subtract(interval, covers):
    """Length of `interval` not covered by any of `covers`."""
    s, e = interval
    clipped = [(max(s, a), min(e, b)) for a, b in covers if b > s and a < e]
    return (e - s) - union_len(clipped)


def steps_of(kernels):
    """Split a rank's kernels into optimiser steps (a step ends with its adam_kernel)."""
    out, cur = [], []
    for k in kernels:
        cur.append(k)
        if STEP_END in k[2]:
            out.append(cur)
            cur = []
    return out


def analyse_step(step):
    comm = [k for k in step if RCCL.search(k[2])]
    comp = [k for k in step if not RCCL.search(k[2])]
    bwd = [k for k in comp if BACKWARD.search(k[2])]
    if not comm:
        return None
    covers = [(s, e) for s, e, _ in comp]
    rows = []
    for s, e, name in comm:
        over = [n for (a, b, n) in comp if b > s and a < e]
        rows.append({"start": s, "end": e, "name": name, "exposed_ns": subtract((s, e), covers),
                     "overlapped": sorted(set(re.sub(r"\(.*", "", n).split("::")[-1][:40] for n in over))})
    last_bwd = max((e for _, e, n in bwd), default=comm[0][0])
    tail = max(0, max(e for _, e, _ in comm) - last_bwd)
    return {"t0": step[0][0], "comm": rows, "comm_total_ns": union_len([(s, e) for s, e, _ in comm]),
            "exposed_total_ns": sum(r["exposed_ns"] for r in rows), "tail_ns": tail,
            "step_ns": step[-1][1] - step[0][0]}


def report(ranks, n_gpus=None, steps_from=2, out=sys.stdout):
    with_comm = {k: v for k, v in ranks.items() if any(RCCL.search(n) for _, _, n in v)}
    print(f"overlap_report: {len(ranks)} rank trace(s), RCCL kernels on {len(with_comm)}", file=out)
    if n_gpus is not None and n_gpus > 1:
        assert len(with_comm) == n_gpus, f"RCCL kernels appear on {len(with_comm)} rank(s), expected {n_gpus}: the ranks never exchanged gradients"
    summary = {}
    for key, kernels in sorted(with_comm.items()):
        steps = [a for a in (analyse_step(s) for s in steps_of(kernels)[steps_from:]) if a]
        if not steps:
            continue
        mid = steps[len(steps) // 2]
        print(f"\nrank {key}: {len(steps)} steps with collectives; a middle step ({mid['step_ns'] / 1e6:.3f} ms):", file=out)
        for r in mid["comm"]:
            print(f"  {(r['start'] - mid['t0']) / 1e3:9.1f} us +{(r['end'] - r['start']) / 1e3:8.1f} us  exposed {r['exposed_ns'] / 1e3:7.1f} us  "
                  f"{re.sub(r'<.*', '', r['name'])[:48]:48s} beside {', '.join(r['overlapped'][:6]) or '-'}", file=out)
        n = len(steps)
        avg = lambda k: sum(s[k] for s in steps) / n / 1e3
        summary[key] = {"comm_us": avg("comm_total_ns"), "exposed_us": avg("exposed_total_ns"), "tail_us": avg("tail_ns"),
                        "step_us": avg("step_ns")}
        s = summary[key]
        print(f"  mean over {n} steps: collectives on the GPU {s['comm_us']:.1f} us per step, exposed (no compute kernel of this rank "
              f"running) {s['exposed_us']:.1f} us, tail behind the last backward kernel {s['tail_us']:.1f} us, step {s['step_us'] / 1e3:.3f} ms", file=out)
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--gpus", type=int, default=None, help="assert that RCCL kernels appear on exactly this many ranks")
    ap.add_argument("--steps-from", type=int, default=2, help="skip this many leading (warm-up) steps")
    a = ap.parse_args()
    report(load(a.path), a.gpus, a.steps_from)


if __name__ == "__main__":
    main()
