"""Train-step harness: the body of ``Runner.train``'s inner loop (``runners/pytorch_runner_vae.py:286,
311-324``) around the HIP model — zero_grad, ``_forward`` (:76-98), loss = CE + kl_weight*KL (+ alpha*MSE),
backward, ``clip_grad_norm_(max_grad_norm)``, ``Adam.step`` — with the MI355X-specific plumbing:

  * all parameters live in ONE flat fp32 buffer and all gradients in another (the backward kernels write
    straight into it), so the global-norm clip is one reduction and Adam is one fused pass
    (acvae_grad_norm / acvae_adam_step), and the data-parallel exchange is three large RCCL all-reduces
    (text-side bucket and the encoder's last ConvBlock, each launched from inside the backward as soon as its
    gradients are queued and overlapped with the rest of the encoder backward, then the remaining 5 MB of the
    encoder) instead of ~60 per-tensor buckets;
  * one process per GPU over ``torch.distributed`` (backend "nccl" = RCCL over xGMI); gradients are
    averaged like torch DDP does (reference :204-207); BatchNorm statistics stay local per rank (the
    reference uses plain BatchNorm2d) and rank 0's running buffers are broadcast before each forward
    (DDP's default broadcast_buffers=True).
"""
import contextlib
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .train_util import LabelSmoothingLoss, MSELoss, Normal_kl_loss, combine_losses


class FlatGradExchange:
    """Data-parallel gradient exchange over one flat buffer split into ordered buckets.

    ``ready(i, after=...)`` launches the asynchronous SUM all-reduce of bucket i (called from the backward as soon as
    the bucket's gradients are queued, so it overlaps with the rest of the backward); ``finish()`` waits for all of
    them and returns the factor 1/world that the consumer folds into its next kernel (norm / Adam), which makes it an
    average like torch DDP's.  Device-agnostic (RCCL on MI355X, gloo in the CPU tests).

    Stream contract (RCCL): a collective is ordered behind the stream that is CURRENT when it is issued, nothing else.
    Nothing here relies on which stream that happens to be: every bucket is issued from a dedicated exchange stream
    that first waits for (a) an event recorded on the announcing stream behind the last kernel queued there - the
    producer of the bucket - and (b) every event in ``after`` (recorded behind the last kernel that writes into the
    bucket on any OTHER stream).  ``finish()`` orders the collectives in front of the caller's current stream.
    A bucket larger than ``max_chunk`` elements goes on the wire as several collectives (first byte sooner; the ring
    pipelines them).  ``issued`` records the (bucket, chunk) sequence: it must be identical on every rank."""

    def __init__(self, flat, bucket_sizes, group=None, max_chunk=8 << 20):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.max_chunk = int(max_chunk)
        self.slices, off = [], 0
        for n in bucket_sizes:
            self.slices.append((off, off + n))
            off += n
        assert off <= flat.numel()
        self._works, self._done, self.issued = [], set(), []
        self._xstream = None

    def begin(self):
        self._works, self._done, self.issued = [], set(), []

    def ready(self, i, after=()):
        if self.world == 1 or i in self._done:
            return
        self._done.add(i)
        a, b = self.slices[i]
        if b <= a:
            return
        ctx = contextlib.nullcontext()
        if self.flat.is_cuda:
            dev = self.flat.device
            produced = torch.cuda.Event()
            produced.record(torch.cuda.current_stream(dev))       # behind the kernels the announcing node just queued
            if self._xstream is None:
                self._xstream = torch.cuda.Stream(dev)
            self._xstream.wait_event(produced)
            for ev in after:
                if ev is not None:
                    self._xstream.wait_event(ev)
            ctx = torch.cuda.stream(self._xstream)
        k = 0
        with ctx:
            while a < b:
                e = min(b, a + self.max_chunk)
                self._works.append(dist.all_reduce(self.flat[a:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.issued.append((i, k))
                a, k = e, k + 1

    def finish(self):
        for i in range(len(self.slices)):
            self.ready(i)                      # any bucket nobody announced (e.g. frozen sub-model)
        for w in self._works:
            w.wait()
        self._works = []
        return 1.0 / self.world


def max_over_ranks(value, device="cpu", group=None):
    """bench.py timing contract: the step time of the job is the MAX over ranks."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def kl_weight_for(epoch, epochs, beta):
    """runners/pytorch_runner_vae.py:286: max(0.5, epoch/epochs*beta)"""
    return max(0.5, float(epoch) / epochs * beta)


class TrainStep:
    def __init__(self, model, vocab_size, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 label_smoothing=True, smoothing=0.1, alpha=1.0, global_loss="MSE", process_group=None,
                 broadcast_buffers=True, data_parallel=True, precision=None):
        self.model = model
        if precision is not None:                   # "f32" | "bf16" (bf16 forward / fp32 loss: BASELINE configs[2])
            model.encoder.compute_dtype = precision
        self.vocab = vocab_size
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.smoothing = smoothing if label_smoothing else 0.0
        self.alpha, self.global_loss = alpha, global_loss
        if alpha is not None and global_loss != "MSE":
            raise NotImplementedError("global_loss other than 'MSE' (the shipped q_logs_utt/p_logs_utt are None)")
        self.criterion = LabelSmoothingLoss(vocab_size, self.smoothing)
        self.kl_loss = Normal_kl_loss()
        self.mse_loss = MSELoss()
        self.pg = process_group
        self.data_parallel = data_parallel          # False: a purely local step inside an initialised process group
        self.world = dist.get_world_size(process_group) if self._dist() else 1
        self.broadcast_buffers = broadcast_buffers
        self.step_count = 0
        self.max_steps_in_flight = int(os.environ.get("ACVAE_STEPS_IN_FLIGHT", "2"))
        self._in_flight = []
        self._copy_stream = None
        self._flatten()
        # Buckets in flat order, each announced from inside the backward as soon as its gradients are queued:
        #   0 decoder + prior + heads (everything acvae_decode_bwd writes, 76 MB at V=5000): behind the decode backward,
        #     i.e. before the encoder backward has even started;
        #   1 posterior (20 MB): behind the posterior backward, which runs on the side stream beside the encoder's;
        #   2 encoder up to the last ConvBlock (5 MB): at the end of the encoder backward - the only exposed part;
        #   3 the last ConvBlock (72 % of Cnn10's encoder gradients, 14 MB): from acvae_encoder_bwd_hooked's callback
        #     right after that block's kernels are queued, while the shallower blocks still run.
        # Buckets over 32 MB go out in 32 MB pieces.
        self.exchange = FlatGradExchange(self.flat_g, [self.n_text_dec, self.n_text - self.n_text_dec,
                                                       self.n_enc - self.n_enc_deep, self.n_enc_deep], process_group)
        if self.world == 1:
            self.exchange.world = 1
        if self.world > 1:                         # single process: no exchange, no callbacks out of the backward
            model._grad_ready_cb = self._on_grads_ready
            model.encoder._grad_ready_cb = self._on_grads_ready
        self._buf_work = None
        self._decode_event = self._decode_aux_event = self._text_event = None
        self._decode_deferred = self._projemb_seen = False
        self._proj_emb = isinstance(model.decoder.word_embeddings, torch.nn.Sequential)
        # The BatchNorm-buffer broadcast issued at the end of step() is joined wherever the buffers are read outside
        # step(): any forward of the model or of its encoder (validation right after the last training step, reference
        # runners/pytorch_runner_vae.py:339-345) and state_dict() (checkpointing).
        if self.world > 1:
            join = lambda *_a, **_k: self.sync_buffers()
            model.register_forward_pre_hook(join)
            model.encoder.register_forward_pre_hook(join)
            model.register_state_dict_pre_hook(join)
        if self._dist():
            dist.broadcast(self.flat_p, src=0, group=self.pg)
            if self.flat_buf is not None:
                dist.broadcast(self.flat_buf, src=0, group=self.pg)

    def _dist(self):
        return self.data_parallel and dist.is_available() and dist.is_initialized() and \
            (self.pg is not None or dist.get_world_size() > 1)

    # ------------------------------------------------------------------ flat parameter / gradient storage
    def _flatten(self):
        model = self.model
        enc_params = set(model.encoder.parameters())
        head = model.encoder._head()                                # embed_pooled / fc1: no gradient on this path
        never = {head.weight, head.bias}
        params = [p for p in model.parameters() if p.requires_grad]
        qset = set(model.qnet.parameters())
        text = [p for p in params if p not in enc_params]
        text = [p for p in text if p not in qset] + [p for p in text if p in qset]   # decode-written first, posterior last
        n_q = sum(1 for p in text if p in qset)
        enc = [p for p in params if p in enc_params and p not in never]
        deep = set(getattr(model.encoder, f"conv_block{model.encoder.N_BLOCKS}").parameters())
        enc = [p for p in enc if p not in deep] + [p for p in enc if p in deep]
        n_deep = sum(1 for p in enc if p in deep)
        tail = [p for p in params if p in never]
        order = text + enc + tail
        dev = order[0].device
        _lib.require_cuda(order[0])
        sizes = [(p.numel() + 3) // 4 * 4 for p in order]          # keep every view 16-B aligned
        total = sum(sizes)
        self.flat_p = torch.zeros(total, device=dev)
        self.flat_g = torch.zeros(total, device=dev)
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        views, off = {}, 0
        for p, sz in zip(order, sizes):
            pv = self.flat_p[off:off + p.numel()].view_as(p)
            pv.copy_(p.data)
            p.data = pv
            views[p] = self.flat_g[off:off + p.numel()].view_as(p)
            off += sz
        self.n_text = sum(sizes[:len(text)])
        self.n_text_dec = sum(sizes[:len(text) - n_q])
        self.n_enc = sum(sizes[len(text):len(text) + len(enc)])
        self.n_enc_deep = sum(sizes[len(text) + len(enc) - n_deep:len(text) + len(enc)])
        self.n_active = self.n_text + self.n_enc
        self.order, self.views, self.never = order, views, never
        model._set_grad_views(views)
        # BatchNorm running statistics in one buffer (for the DDP-style broadcast)
        bufs = [b for n, b in model.named_buffers() if b.dtype.is_floating_point]
        self.flat_buf = None
        if bufs:
            self.flat_buf = torch.zeros(sum(b.numel() for b in bufs), device=dev)
            off = 0
            for b in bufs:
                v = self.flat_buf[off:off + b.numel()].view_as(b)
                v.copy_(b.data); b.data = v
                off += b.numel()
        self.norm_partials = torch.empty(_lib.call("acvae_grad_norm_partials"), device=dev)
        self.total_norm = torch.zeros(1, device=dev)

    # ------------------------------------------------------------------ gradient exchange
    def _on_grads_ready(self, tag, event=None):
        """Called from inside the backward, on the thread and with the stream of the autograd node that just queued
        its kernels: ("decode", event) after acvae_decode_bwd unless it left gradients trailing on the side stream
        (`event` = recorded behind its last kernel), "text" after the posterior backward (every text-side gradient is
        queued now), ("encoder_block", b) from acvae_encoder_bwd_hooked after ConvBlock b, "encoder" when the encoder
        is through.  The order of these calls is fixed by the autograd graph, hence identical on every rank."""
        if tag == "decode":
            self._decode_event = event
            if not self._proj_emb:               # with projected embeddings their gradients are still to come ("projemb")
                self.exchange.ready(0)
        elif tag == "decode_deferred":
            # trailing-gradient mode: the decode backward's parameter gradients are still running on the decode calls' SECOND
            # stream: `event` = (event on the main stream, event behind the trailing work on the second stream)
            self._decode_event, self._decode_aux_event = event
            self._decode_deferred = True
        elif tag == "projemb":
            # _ProjTableFn's backward: the last three decode-side gradients.  Autograd runs it after the decode backward
            # and - the node being older than the posterior's - normally after "text" as well; bucket 0 goes out behind
            # the decode backward (event), the side stream's trailing work and the posterior backward (text event) and
            # the kernels just queued on the current stream.  Should it ever fire before "text" in trailing-gradient
            # mode, nothing on this stream covers the side stream's work yet: leave the bucket to "text".
            self._projemb_seen = True
            if self._text_event is not None or not self._decode_deferred:
                self.exchange.ready(0, after=(self._decode_event, self._decode_aux_event, self._text_event))
        elif tag == "text":
            if torch.cuda.is_available() and self.flat_g.is_cuda:
                self._text_event = torch.cuda.Event()
                self._text_event.record(torch.cuda.current_stream())
            # decode-written gradients not announced yet (trailing-gradient mode): they were queued on the decode calls' second
            # stream (`_decode_aux_event`) or on the main stream behind `_decode_event`; with projected embeddings bucket 0
            # waits for "projemb"
            if not self._proj_emb or self._projemb_seen:
                self.exchange.ready(0, after=(self._decode_event, self._decode_aux_event))
            self.exchange.ready(1)
        elif isinstance(tag, tuple):
            if tag[1] == self.model.encoder.N_BLOCKS:
                self.exchange.ready(3)
        else:
            self.exchange.ready(3)
            self.exchange.ready(2)

    # ------------------------------------------------------------------ one optimiser step
    def forward_loss(self, feats, feat_lens, caps, cap_lens, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5):
        """Runner._forward(mode='train') + the loss line (:315-318).  Returns (loss, parts, output)."""
        out = self.model(feats, feat_lens, caps, cap_lens, ss_ratio=ss_ratio, dis_ratio=dis_ratio)
        staged = getattr(self.model, "staged", None) or {}
        lens1 = staged.get("lens1_d")
        if lens1 is None:
            lens1 = np.asarray(cap_lens) - 1
        caps_src = staged.get("caps_d")
        targets = (caps if caps_src is None else caps_src)[:, 1:1 + out["logits"].shape[1]].to(torch.long)
        ce = self.criterion.masked(out["logits"], targets, lens1)     # == criterion(packed_logits, packed targets)
        kl = self.kl_loss(out["q_means"], out["q_logs"], out["p_means"], out["p_logs"])
        mse = None
        if self.alpha is not None:
            mse = self.mse_loss(out["q_means_utt"], out["p_means_utt"])
        loss = combine_losses(ce, kl, mse, kl_weight, self.alpha if self.alpha is not None else 0.0)    # ce + w kl + alpha mse
        return loss, {"ce": ce.detach(), "kl": kl.detach(), "mse": None if mse is None else mse.detach()}, out

    def prefetch(self, feats_host):
        """Queue the upload of a LATER step's feature batch now, on a copy stream of its own: the 8 MB host-to-device copy
        that Runner._forward makes in front of every step (`feats = batch[0].to(device)`, runners/pytorch_runner_vae.py:80)
        then travels beside the step that is running instead of in front of the next encoder.  Returns the device tensor to
        hand to step(); step() makes its stream wait for the copy.  Page-locked input is copied from where it lies;
        pageable input is staged through the package's page-locked ring first (a host memcpy).  Typical loop:
            nxt = ts.prefetch(batch0)
            for batch in batches[1:] + [None]:
                cur, nxt = nxt, (ts.prefetch(batch) if batch is not None else None)
                ts.step(cur, ...)"""
        dev = self.flat_p.device
        t = torch.as_tensor(feats_host)
        if t.is_cuda:
            return t
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self._copy_stream):
            d = t.to(dev, non_blocking=True) if t.is_pinned() else _lib.h2d(t.float(), dev)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        d._acvae_ready = ev
        return d

    def step(self, feats, feat_lens, caps, cap_lens, ss_ratio=1.0, dis_ratio=0, kl_weight=0.5):
        self.sync_buffers()
        ready = getattr(feats, "_acvae_ready", None)
        if ready is not None:                       # a batch uploaded by prefetch(): order this step behind its copy
            cur = torch.cuda.current_stream()
            cur.wait_event(ready)
            feats.record_stream(cur)
        self._decode_event = self._decode_aux_event = self._text_event = None
        self._decode_deferred = self._projemb_seen = False
        for p in self.order:
            p.grad = None                                             # optimizer.zero_grad(set_to_none=True)
        loss, parts, _ = self.forward_loss(feats, feat_lens, caps, cap_lens, ss_ratio, dis_ratio, kl_weight)
        self.exchange.begin()
        loss.backward()
        gscale = self.exchange.finish()
        skipped = self._check_grad_aliasing()
        st = _lib.current_stream()
        n = self.n_active
        tn = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            _lib.call("acvae_grad_norm", self.flat_g, n, gscale, self.norm_partials, self.total_norm, st)
            tn = self.total_norm
        self.step_count += 1
        # torch.optim.Adam leaves a parameter whose .grad is None alone (no moment decay, no update): the fused pass
        # then runs over the segments between such parameters (their gradient slices are zero, so the norm is unaffected).
        # On this path every parameter gets a gradient every step and the loop runs once over the whole buffer.
        for a, b in self._segments(skipped, n):
            _lib.call("acvae_adam_step", self.flat_p[a:b], self.flat_g[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], b - a,
                      self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, gscale,
                      float(self.max_grad_norm or 0.0), tn, st)
        parts["loss"] = loss.detach()
        parts["grad_norm"] = self.total_norm
        # DDP's broadcast_buffers: rank 0's BatchNorm running statistics reach the other ranks before the next forward
        # touches them.  Issued here, behind this step's kernels and off the next step's critical path; joined by
        # sync_buffers() at the start of the next step (or by the caller before an evaluation pass).
        if self.world > 1 and self.broadcast_buffers and self.flat_buf is not None:
            self._buf_work = dist.broadcast(self.flat_buf, src=0, group=self.pg, async_op=True)
        # Bound how far the host may run ahead of the GPU.  Unbounded, the first steps of a run queue several steps'
        # worth of launches, the runtime grows its queues / kernel-argument / signal pools while the GPU is working and
        # those steps run 31-36 ms instead of 28.7 (tools/step_times.py); two steps of slack hide every host hiccup.
        if self.max_steps_in_flight > 0:
            ev = torch.cuda.Event()
            ev.record()
            self._in_flight.append(ev)
            if len(self._in_flight) > self.max_steps_in_flight:
                self._in_flight.pop(0).synchronize()
                # a persistent launch of that step that could not complete has poisoned the step with NaN and raised the
                # device's status word: turn it into an error at the first synchronisation point the loop has anyway
                _lib.check_persist_status(self.flat_p.device)
        return parts

    def synchronize(self):
        """Wait for every queued step and raise if one of their persistent launches gave up (acvae_persist_status_register)."""
        torch.cuda.synchronize(self.flat_p.device)
        self._in_flight.clear()
        _lib.check_persist_status(self.flat_p.device)

    def sync_buffers(self):
        """Join the asynchronous BatchNorm-buffer broadcast of the previous step (a stream dependency under RCCL)."""
        if self._buf_work is not None:
            self._buf_work.wait()
            self._buf_work = None

    def _check_grad_aliasing(self):
        """The backward kernels write into the flat gradient buffer and autograd is expected to adopt those
        views as .grad; if it cloned one instead, repair the alias; if a parameter got no gradient at all, zero its
        slice and report it: returns the list of (offset, end) ranges of active parameters without a gradient."""
        skipped, off = [], 0
        for p in self.order:
            v = self.views[p]
            sz = (p.numel() + 3) // 4 * 4
            if p.grad is None:
                if off < self.n_active:
                    v.zero_()
                    skipped.append((off, off + sz))
            elif p.grad.data_ptr() != v.data_ptr():
                p.grad = v            # autograd cloned the view; the flat buffer (already all-reduced) is authoritative
            off += sz
        return skipped

    @staticmethod
    def _segments(skipped, n):
        if not skipped:
            return [(0, n)]
        segs, cur = [], 0
        for a, b in skipped:
            if a > cur:
                segs.append((cur, a))
            cur = max(cur, b)
        if cur < n:
            segs.append((cur, n))
        return segs

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    # ------------------------------------------------------------------ checkpoint in the reference's format
    def _offsets(self):
        off, table = 0, {}
        for p in self.order:
            table[p] = off
            off += (p.numel() + 3) // 4 * 4
        return table

    def optimizer_state_dict(self):
        """The moments in ``torch.optim.Adam.state_dict()`` layout, parameters indexed in ``model.parameters()`` order
        (the optimiser of the reference is built from it, ``runners/pytorch_runner_vae.py:233-236``), so that
        ``{"model": model.state_dict(), "optimizer": ts.optimizer_state_dict()}`` is the checkpoint the reference writes
        (``:380-388``) and a ``torch.optim.Adam`` over a reference model loads it."""
        table = self._offsets()
        params = list(self.model.parameters())     # the reference builds Adam over ALL model.parameters() (:233-236)
        state = {}
        for i, p in enumerate(params):
            # torch.optim.Adam holds no state for a parameter that never had a gradient (frozen ones, and the pooled
            # head embed_pooled / fc1 whose output this path does not consume): no entry, like a reference checkpoint
            if self.step_count == 0 or p not in table or p in self.never:
                continue
            o = table[p]
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + p.numel()].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """Inverse of ``optimizer_state_dict`` (also accepts the state dict of a ``torch.optim.Adam`` built over the
        same model): resume training where the checkpoint left off."""
        table = self._offsets()
        params = list(self.model.parameters())
        group = sd["param_groups"][0]
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        self.weight_decay = group.get("weight_decay", 0.0)
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for i, st in sd["state"].items():
            p = params[int(i)]
            if p not in table:
                continue                                   # frozen here: nothing to resume
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} does not match parameter "
                                 f"{tuple(p.shape)} (state indexed over a different parameter list?)")
            o = table[p]
            self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        # one step counter for the flat buffer: per-parameter counts that differ (legal under DDP's
        # find_unused_parameters) resume at the largest, which is exact for every parameter that was never skipped
        self.step_count = max(steps) if steps else 0
