#!/usr/bin/env python3
"""bench.py — training-step throughput of the AC-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" is one full optimiser step (Runner.train inner loop, runners/pytorch_runner_vae.py:311-324):
forward (Cnn10 encoder -> posterior -> prior/decoder loop) -> CE + 0.5*KL + 1.0*MSE -> backward ->
global-norm clip -> Adam, on BASELINE.json configs[1]: B=32 clips per GPU, T=1000 frames, F=64 mel bins,
22-token captions, vocab 5000, E=H=512, fp32, synthetic seeded data, random-init weights.  Weak scaling:
every rank processes its own 32-clip batch; gradients are averaged over RCCL.  Every step's 8.2 MB feature batch is uploaded
from page-locked host memory inside the timed region (prefetched during the previous step).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the HIP runtime's kernel-argument ring (see acvae_amd/__init__.py); must be in the environment before HIP initialises
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))

B, T, F, V, L, E = 32, 1000, 64, 5000, 22, 512
HOP_S = 0.010  # assumed log-mel hop (PANNs/Cnn10 convention; the reference never states it: SURVEY §8(d))
ENC_FWD_GFLOP_PER_CLIP = 26.03   # conv MACs x2 of Cnn10 at T=1000
FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md, Peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md, Peak BF16 MFMA, dense


def build_model():
    import torch
    from acvae_amd.decoder import VAERNNBahdanauAttnDecoder
    from acvae_amd.encoder import Cnn10
    from acvae_amd.vae_model import Hybrid_VAEModel
    torch.manual_seed(1)
    dec = VAERNNBahdanauAttnDecoder(vocab_size=V, enc_mem_size=E, embed_size=E, hidden_size=E, dropout=0.0,
                                    num_layers=1, rnn_type="GRU", attn_size=E)
    return Hybrid_VAEModel(Cnn10(F, 512), dec, posterior_model="PosteriorRNN_hybrid",
                           posterior_args={"hidden_size": E}, prior_model="PriorRNN", prior_args={"hidden_size": E})


def synthetic(seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, F, generator=g)
    caps = torch.zeros(B, L)
    caps[:, 0] = 1; caps[:, -1] = 2
    caps[:, 1:-1] = torch.randint(4, V, (B, L - 2), generator=g).float()
    import numpy as np
    return feats, caps, np.full(B, T, dtype=np.int64), np.full(B, L, dtype=np.int64)


def host_cores():
    """CPU cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (the GPU
    box reports 256 logical CPUs but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline():
    """The oracle (CPU restatement of the reference, validated against it: oracle/) timed on this box's host cores as
    SURVEY 8(d) defines the CPU figure: the same synthetic workload (B=32, T=1000), one warm-up step ON the workload, then
    the mean of 3 full optimiser steps (fwd + loss + bwd + clip + Adam) - at all host cores this process may use and again
    at 8 threads (the build container's count, for comparability with BASELINE.md).  About a minute of CPU work."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import acvae_oracle as O
    cores = host_cores()
    torch.manual_seed(1); random.seed(1)
    state = {k: (torch.randn(s) * 0.05 if len(s) > 1 else torch.zeros(s)) if "num_batches" not in k else torch.zeros((), dtype=torch.long)
             for k, s in O.state_shapes(V, E, E, None, E, 512).items()}
    for k in state:
        if k.endswith("running_var") or (".bn" in k and k.endswith("weight")):
            state[k] = torch.ones_like(state[k])
    tr = O.OracleTrainer(state, V)
    feats, caps, feat_lens, cap_lens = synthetic(1)
    STEPS = 3

    def timed(threads, warm):
        torch.set_num_threads(threads)
        for _ in range(warm):
            tr.step(feats, feat_lens.copy(), caps, cap_lens)
        t0 = time.perf_counter()
        for _ in range(STEPS):
            tr.step(feats, feat_lens.copy(), caps, cap_lens)
        return (time.perf_counter() - t0) / STEPS, torch.get_num_threads()

    dt_all, used = timed(cores, 1)
    out = {"value": B / dt_all, "unit": "captions/s", "cores": used, "kind": "port",
           "sample": f"mean of {STEPS} full optimiser steps of the same workload (B={B}, T={T}, V={V}) after one warm-up step on it; "
                     f"{dt_all:.2f} s per step on {used} host threads (torch-CPU ops)"}
    if cores != 8:
        dt8, used8 = timed(min(8, cores), 1 if cores < 8 else 0)
        out["at_8_threads"] = {"value": B / dt8, "unit": "captions/s", "cores": used8, "s_per_step": dt8}
    else:
        out["at_8_threads"] = {"value": B / dt_all, "unit": "captions/s", "cores": used, "s_per_step": dt_all}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)     # steps 1-7 of a fresh process run 31-34 ms, later ones 28.7
                                                         # (tools/step_times.py): allocator, event and clock warm-up
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = BASELINE configs[1] (the headline, default); bf16 = configs[2]: bf16 forward / fp32 loss "
                         "(conv stack on the bf16 MFMA pipe, activations stored in bf16; text side, loss, parameters fp32)")
    ap.add_argument("--lib", default=None, help="dev: time another build of the library (tools/ab_build.py) instead of the "
                                                "product's; recorded in config.lib")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as CHILD processes (one per GPU, torch.distributed.run
        # over 127.0.0.1) before this process has made any GPU call, and leave with their exit code.  Nothing is
        # re-executed in place: a process that has initialised the GPU must never exec.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
               "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        cmd += ["--dtype", args.dtype]
        if args.lib:
            cmd += ["--lib", args.lib]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) "
                         "or run `python bench.py --gpus N` without a launcher; refusing to report a mislabelled number")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # test hooks (one-GPU rehearsal of the multi-rank path): every rank on device 0 over gloo
    rehearsal = os.environ.get("ACVAE_BENCH_REHEARSAL") == "1"
    torch.cuda.set_device(0 if rehearsal else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl")   # "nccl" is RCCL on ROCm
    import __graft_entry__ as ge
    from acvae_amd import _lib
    if args.lib:
        _lib.use_library(args.lib)
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from acvae_amd.trainer import TrainStep, max_over_ranks

    model = build_model().cuda().train()
    ts = TrainStep(model, V, lr=5e-4, max_grad_norm=1.0, smoothing=0.1, alpha=1.0, precision=args.dtype)
    feats_host, caps, feat_lens, cap_lens = synthetic(1 + rank)
    feats = feats_host.cuda()
    # The feature batch of EVERY step arrives as a page-locked host buffer, as a DataLoader(pin_memory=True) hands it over, and
    # is uploaded inside the timed region (Runner._forward: `feats = batch[0].to(device)`, runners/pytorch_runner_vae.py:80):
    # TrainStep.prefetch queues the NEXT step's copy on a copy stream while the current step runs.  The first batch is
    # resident before the timed region starts.  (config.resident_ms_per_step: the same loop on one resident tensor.)
    pinned = feats_host.pin_memory()

    def step(x=None):
        random.seed(0)   # scheduled-sampling draws (ss_ratio = 1: always teacher forcing, as in epoch 1 of the reference)
        return ts.step(feats if x is None else x, feat_lens.copy(), caps, cap_lens, ss_ratio=1.0, dis_ratio=0,
                       kl_weight=0.5)

    nxt = ts.prefetch(pinned)
    for _ in range(args.warmup):
        cur, nxt = nxt, ts.prefetch(pinned)
        parts = step(cur)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # live roofline measurement: HIP events around the conv launches of every PROF_EVERY-th timed step (an event pair
    # costs a few microseconds of queue bubble; 42 pairs in every step would take ~1 % off the headline)
    PROF_EVERY = int(os.environ.get("ACVAE_BENCH_PROF_EVERY", "4"))
    _lib.lib().acvae_prof_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampled = 0
    for i in range(args.steps):
        on = i % PROF_EVERY == 0
        _lib.lib().acvae_prof_pause(0 if on else 1)
        sampled += int(on)
        cur, nxt = nxt, ts.prefetch(pinned)          # this step's batch was uploaded during the previous step
        parts = step(cur)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, device="cuda")
    # live per-kernel timing of the dominant kernel family (HIP events on the launch stream)
    import ctypes
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    _lib.lib().acvae_prof_read(0, ctypes.byref(ms), ctypes.byref(cnt))
    igemm_ms, igemm_n = ms.value, cnt.value
    _lib.lib().acvae_prof_read(1, ctypes.byref(ms), ctypes.byref(cnt))
    wgrad_ms, wgrad_n = ms.value, cnt.value
    _lib.lib().acvae_prof_enable(0)
    loss = float(parts["loss"])
    ts.synchronize()                                  # raises if a persistent launch of the run gave up
    # Beside it: the same loop on ONE resident tensor (no upload at all), and with the upload queued in front of each step on
    # the compute stream (no prefetch: what `feats.to(device)` inside the step costs when nothing hides it).
    EXTRA = 8
    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(EXTRA):
            fn()
        torch.cuda.synchronize()
        return max_over_ranks((time.perf_counter() - t1) / EXTRA * 1e3, device="cuda")
    resident_ms = timed(lambda: step())
    inline_ms = timed(lambda: step(pinned.cuda(non_blocking=True)))

    if rank == 0:
        n_gpus = world
        caps_per_s = n_gpus * B * args.steps / dt
        frames_per_s = caps_per_s * T
        # dominant kernel: conv3x3 implicit GEMM on fp32 MFMA.  Per step it is launched 14 times: 7 forward convs
        # (all but the Cin=1 first conv) + 7 data gradients (all but the one into the Cin=1 input).  Algorithmic
        # flops of those launches: 2 x (26.03 - 0.074) GFLOP per clip (forward + data-gradient share the shape).
        flops_per_step = 2 * (ENC_FWD_GFLOP_PER_CLIP - 0.074) * 1e9 * B * (T / 1000.0)
        launches_per_step = igemm_n / max(1, sampled)
        avg_ms = igemm_ms / max(1, igemm_n)
        achieved = flops_per_step * sampled / (igemm_ms * 1e-3) / 1e12 if igemm_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_conv_igemm.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        bf16 = args.dtype == "bf16"
        peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
        wino = not bf16 and os.environ.get("ACVAE_CONV_WINO", "1") != "0"
        kernel = ("conv_igemm_bf16_kernel (conv3x3 implicit GEMM fwd+dgrad, v_mfma_f32_32x32x16_bf16, bf16 activations)"
                  if bf16 else
                  "conv_wino_kernel + conv_wino_bnred_kernel + conv_wino_stats_kernel + conv_wino_act_kernel (conv3x3 data gradient / data gradient "
                  "with the next BatchNorm backward's sums / forward / forward with BatchNorm+ReLU operand as Winograd F(2x2,3x3) on "
                  "v_mfma_f32_32x32x2_f32: 3 + 4 + 3 + 4 launches per step)" if wino else
                  "conv_igemm3_kernel (conv3x3 implicit GEMM fwd+dgrad with horizontal-tap reuse, fp32 MFMA)")
        which = ("BASELINE configs[2] per-GPU shape (bf16 forward / fp32 loss)" if bf16 else "BASELINE configs[1]")
        # `achieved` / `frac` = flops the kernel actually ISSUES to the matrix pipe over its measured time (<= 1 by
        # construction): Winograd F(2x2,3x3) issues 16/36 of the direct convolution's multiplies.  The direct-convolution
        # (algorithmic) figure is reported beside it as effective_* and may exceed the peak.
        exec_ratio = 16.0 / 36.0 if wino else 1.0
        executed = achieved * exec_ratio
        ms_per_step = dt / args.steps * 1e3
        # whole step: executed conv flops (forward + data gradient + weight gradient of the 7 MFMA layers) over ms_per_step
        step_exec_flops = 3 * (ENC_FWD_GFLOP_PER_CLIP - 0.074) * 1e9 * B * (T / 1000.0) * exec_ratio
        out = {
            "metric": "train-step captions/s (and audio-sec/s) at B=32, 1/2/4/8 MI355X",
            "value": caps_per_s, "unit": "captions/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{which}: per-GPU batch B={B}, T={T} frames, F={F} mel, {L}-token captions "
                                   f"(Tc={L - 1} decode steps), vocab {V}, E=H=A={E}, {'bf16 conv stack, fp32 text side / loss / parameters' if bf16 else 'fp32'}; full optimiser step "
                                   "(fwd + CE/KL/MSE loss + bwd + global-norm clip + Adam), ss_ratio=1, dis_ratio=0",
                       "global_batch": n_gpus * B, "parallelism": f"dp{n_gpus}" if n_gpus > 1 else "single",
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                       "frames_per_s": frames_per_s, "audio_s_per_s": frames_per_s * HOP_S, "hop_s_assumed": HOP_S,
                       "loss_last_step": loss, **({"lib": args.lib} if args.lib else {}),
                       "input": "every step's feature batch (8.2 MB) is uploaded from page-locked host memory INSIDE the timed "
                                "region, prefetched on a copy stream during the previous step (TrainStep.prefetch)",
                       "pcie_inclusive_ms_per_step": ms_per_step,
                       "resident_ms_per_step": resident_ms, "inline_upload_ms_per_step": inline_ms,
                       "pcie_inclusive_note": f"ms_per_step IS the upload-inclusive figure; resident_ms_per_step = {EXTRA} extra "
                                              "steps on one resident tensor (no upload), inline_upload_ms_per_step = the upload "
                                              "queued in front of each step on the compute stream (no prefetch)"},
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": executed, "peak": peak, "unit": "TFLOP/s",
                         "frac": executed / peak, "traffic": None if bf16 else traffic,
                         "effective_tflops": achieved, "effective_frac": achieved / peak,
                         "step_frac": step_exec_flops / (ms_per_step * 1e-3) / 1e12 / peak,
                         "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                         "executed_gflop_per_launch": flops_per_step * exec_ratio / 1e9 / max(1.0, launches_per_step),
                         "algorithmic_gflop_per_launch": flops_per_step / 1e9 / max(1.0, launches_per_step),
                         "wgrad_avg_launch_ms": wgrad_ms / max(1, wgrad_n), "wgrad_launches_per_step": wgrad_n / max(1, sampled), "steps_sampled": sampled,
                         "note": "achieved/frac = flops issued to the MFMA pipe by the dominant kernel over its measured launch time; "
                                 "effective_* = direct-convolution flops of SURVEY 8(d) over the same time (F(2x2,3x3) needs 2.25x fewer "
                                 "multiplies, so it can exceed the peak); step_frac = executed conv flops of the whole step (fwd + dgrad + "
                                 "wgrad) over ms_per_step"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
