"""Mirror of ``models/vae_model.py`` ``Hybrid_VAEModel`` (:674-894): the autoregressive-prior +
global-constraint AC-VAE.  Same constructor ``Hybrid_VAEModel(Audioencoder, Textdecoder,
posterior_model=, posterior_args=, prior_model=, prior_args=)``, same forward contract

    forward(feats, feat_lens, caps, cap_lens, ss_ratio=, dis_ratio=)   -> training dict
    forward(feats, feat_lens, method="greedy", max_length=, ...)       -> inference dict ("seqs", ...)

and the same state-dict names.  Host code here only draws the random decisions in the reference's
order (python ``random.random()`` per step for scheduled sampling :826, CPU ``torch.randn`` for both
reparameterisations — SURVEY F9 —, ``torch.rand(1)`` per step when dis_ratio != 0 :805), allocates
outputs and wires three autograd nodes (encoder, posterior, decode loop), each ONE call into
libacvae_hip.so for forward and one for backward.
"""
import os
import random
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib, text_encoder
from .encoder import ptr_table, scratch_buffer
from .word_model import CaptionModel


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, mem, mem_lens_d, caps_d, lens1_d, q_z, eps_p, ss_flags, dis_flags, Tc, sampling, *weights):
        N, S, Eenc = mem.shape
        dec = model.decoder
        E, H, A, V = dec.embed_size, dec.model.hidden_size, dec.attn.attn_size, dec.vocab_size
        dev = mem.device
        train = caps_d is not None
        params = model._text_table()
        dims = (N, Tc, S, E, H, A, V, Eenc)
        saved_b = _lib.call("acvae_decode_saved_bytes", *dims)
        scratch_b = _lib.call("acvae_decode_scratch_bytes", *dims)
        if saved_b < 0:
            raise RuntimeError(f"decode: unsupported dims {dims}")
        saved = torch.empty(saved_b, dtype=torch.uint8, device=dev)
        scratch = scratch_buffer(scratch_b, dev)
        f = lambda *s: torch.empty(*s, device=dev)
        logits, outputs = f(N, Tc, V), f(N, Tc, H)
        seqs = torch.empty(N, Tc, dtype=torch.long, device=dev)
        slp, attw = f(N, Tc), f(N, Tc, S)
        pm, pl, pz = f(N, Tc, E), f(N, Tc, E), f(N, Tc, E)
        putt = f(N, 2 * E) if train else None
        hfin, hp, cp = f(N, H), f(N, E), f(N, E)
        IntArr = _lib.ctypes.c_int * Tc
        ss_arr = IntArr(*[int(bool(x)) for x in ss_flags]) if train else None
        dis_arr = IntArr(*[int(bool(x)) for x in dis_flags]) if train else IntArr(*([1] * Tc))
        mem = mem.contiguous()
        method, temp, noise = sampling["sample"] if sampling and sampling.get("sample") else (0, 1.0, None)
        keep, drop_p = sampling["emb_keep"] if sampling and sampling.get("emb_keep") else (None, 0.0)
        _lib.persist_status(dev)                 # the device's status words are registered before the first persistent launch
        _lib.call("acvae_decode_fwd_sampled", ptr_table(params), mem, mem_lens_d, caps_d,
                  caps_d.stride(0) if train else 0, lens1_d, q_z, eps_p, ss_arr, dis_arr, logits, outputs, seqs, slp,
                  attw, pm, pl, pz, putt, hfin, hp, cp, saved, saved_b, scratch, scratch_b, *dims, model.start_idx,
                  model.end_idx, _lib.current_stream(), model._aux_stream(), int(method), float(temp), noise, keep,
                  float(drop_p), _lib.call_flags())
        ctx.set_materialize_grads(False)         # outputs the loss does not use (outputs, p_z, ..) arrive as None, not as zero tensors
        ctx.model, ctx.saved, ctx.dims, ctx.dis_arr = model, saved, dims, dis_arr
        ctx.emb_keep, ctx.emb_p = keep, float(drop_p)
        # outputs kept as plain ctx attributes would form tensor -> grad_fn -> ctx -> tensor cycles that are never collected
        ctx.save_for_backward(mem, mem_lens_d, lens1_d, eps_p, outputs, attw, pl)
        ctx.mark_non_differentiable(seqs, slp, attw, hfin, hp, cp)
        if not train:
            putt = torch.zeros(0, device=dev)
            ctx.mark_non_differentiable(putt)
        return logits, outputs, seqs, slp, attw, pm, pl, pz, putt, hfin, hp, cp

    @staticmethod
    def backward(ctx, d_logits, d_outputs, _s, _l, _a, d_pm, d_pl, d_pz, d_putt, *_rest):
        model = ctx.model
        N, Tc, S, E, H, A, V, Eenc = ctx.dims
        mem, mem_lens_d, lens1_d, eps_p, outputs, attw, pl = ctx.saved_tensors
        dev = mem.device
        params = model._text_table()
        grads = [None] * len(params)
        mine = set(range(0, 10)) | set(range(21, 35))
        for i, p in enumerate(params):
            if p is not None and p.requires_grad and i in mine:
                grads[i] = model._grad_buffer(p)
        c = lambda t: None if t is None else t.contiguous().float()
        d_mem = torch.empty(N, S, Eenc, device=dev)
        d_qz = torch.empty(N, Tc, E, device=dev)
        scratch_b = _lib.call("acvae_decode_scratch_bytes", *ctx.dims)
        scratch = scratch_buffer(scratch_b, dev, tag="decode")
        ups = [c(t) for t in (d_logits, d_outputs, d_pm, d_pl, d_pz, d_putt)]
        main, aux = _lib.current_stream(), model._aux_stream()
        flags = _lib.call_flags(defer=model.defer_param_grads)
        _lib.call("acvae_decode_bwd", ptr_table(params), ptr_table(grads), mem, mem_lens_d, lens1_d, eps_p, ctx.dis_arr,
                  outputs, attw, pl, *ups, d_mem, d_qz, ctx.saved, ctx.saved.numel(), scratch, scratch_b, *ctx.dims, main,
                  aux, ctx.emb_keep, ctx.emb_p, flags)
        defers = bool(_lib.lib().acvae_decode_bwd_defers(ctx.dis_arr, Tc, main, aux, flags))   # a yes / no answer, not a status
        if model._grad_ready_cb is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            if not defers:                       # every gradient this call writes is ordered on the current stream
                model._grad_ready_cb("decode", ev)
            else:                                # ... or on the decode calls' second stream: a second event behind its trailing work
                ev_aux = torch.cuda.Event()
                ev_aux.record(model._side_stream(torch.cuda.current_stream()))
                model._grad_ready_cb("decode_deferred", (ev, ev_aux))
        if defers:
            # The parameter gradients are still being computed on the decode calls' second stream (beside the posterior's and
            # the encoder's backward, which start now: d_mem and d_q_z are complete on the current stream).  Their consumers:
            # the gradient exchange (the event above) and whoever reads .grad after backward() - joined here at the end of
            # the pass.
            side, cur = model._side_stream(torch.cuda.current_stream()), torch.cuda.current_stream()
            # (mem too: without an ln projection the trailing products read the encoder memory itself, not a copy in `saved`)
            for t in [ctx.saved, outputs, d_qz, mem] + [u for u in ups if u is not None]:
                t.record_stream(side)                      # freed by autograd while the side stream still reads them
            if any(p is not None and p.is_leaf and p.grad is not None for p in params):   # (a projected embedding table is not a leaf)
                cur.wait_stream(side)                      # autograd will accumulate into .grad on this stream right away
            else:
                torch.autograd.Variable._execution_engine.queue_callback(lambda: cur.wait_stream(side))
        ctx.saved = None
        outs = [next((g for p, g in zip(params, grads) if p is w), None) for w in model._decode_weights()]
        return (None, d_mem, None, None, None, d_qz, None, None, None, None, None, *outs)


class Hybrid_VAEModel(CaptionModel):
    def __init__(self, Audioencoder: nn.Module, Textdecoder: nn.Module, **kwargs):
        super().__init__(Audioencoder, Textdecoder, **kwargs)
        E = Textdecoder.embed_size
        self.qnet = getattr(text_encoder, kwargs["posterior_model"])(
            word_dim=E, embed_size=E, vocab_size=Textdecoder.vocab_size, **kwargs["posterior_args"])
        self.pnet = getattr(text_encoder, kwargs["prior_model"])(
            word_dim=E, audiofeats_size=E, embed_size=E, vocab_size=Textdecoder.vocab_size, **kwargs["prior_args"])
        self.mean_log_out = nn.Linear(E, 2 * E)
        if E != Audioencoder.embed_size:
            self.ln = nn.Linear(Audioencoder.embed_size, E)
            nn.init.xavier_uniform_(self.ln.weight)
        nn.init.xavier_uniform_(self.mean_log_out.weight)
        self.qnet._owner = weakref.ref(self)
        self.pnet._owner = weakref.ref(self)
        self.decoder._owner = weakref.ref(self)
        self._encproj_cache = {}
        self.use_side_stream = os.environ.get("ACVAE_SIDE_STREAM", "1") != "0"
        # the decode backward leaves its parameter gradients trailing on the second stream beside the encoder backward
        # (_DecodeFn.backward joins it; ACVAE_FLAG_DEFER_PARAM_GRADS per call); ACVAE_DECODE_DEFER=0 or
        # model.defer_param_grads = False keeps everything on the main stream
        self.defer_param_grads = True
        self.staged = None         # device copies of caps / cap_lens-1 made by the last training forward
        self.noise = None          # optional replay: dict(eps_q=[N,Tc,E], eps_p=[Tc,N,E]) consumed by the next forward
        self._grad_views = None    # {param: flat-gradient view}, set by the train-step harness
        self._grad_ready_cb = None # called with "text" once every text-side gradient has been written

    # ---- plumbing: the text-side parameter table in state-dict order (include/acvae_hip.h)
    def _text_table(self):
        d, q, p = self.decoder, self.qnet, self.pnet
        g, qn, pn = d.model, q.network, p.network
        t = [d.embedding_table(), g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
             d.classifier.weight, d.classifier.bias, d.attn.v, d.attn.h2attn.weight, d.attn.h2attn.bias,
             q.word_embedding.weight, qn.weight_ih_l0, qn.weight_hh_l0, qn.bias_ih_l0, qn.bias_hh_l0,
             qn.weight_ih_l0_reverse, qn.weight_hh_l0_reverse, qn.bias_ih_l0_reverse, qn.bias_hh_l0_reverse,
             q.token_mean_log.weight, q.token_mean_log.bias,
             p.word_embedding.weight, p.word_attn.v, p.word_attn.h2attn.weight, p.word_attn.h2attn.bias,
             pn.weight_ih_l0, pn.weight_hh_l0, pn.bias_ih_l0, pn.bias_hh_l0, p.mean_log_out.weight, p.mean_log_out.bias,
             self.mean_log_out.weight, self.mean_log_out.bias]
        t += [self.ln.weight, self.ln.bias] if hasattr(self, "ln") else [None, None]
        assert len(t) == _lib.ENUMS_TEXT_N
        return t

    def _decode_weights(self):
        t = self._text_table()
        return [p for i, p in enumerate(t) if p is not None and (i < 10 or i >= 21)]

    def _grad_buffer(self, p):
        if self._grad_views is not None and p in self._grad_views:
            return self._grad_views[p].detach()
        return torch.empty_like(p)

    def _set_grad_views(self, views):
        self._grad_views = views
        self.encoder._grad_views = views

    def _encproj(self, which, enc_mem):
        """Hoisted encoder half of an attention (0: decoder.attn, 1: pnet.word_attn) for single-step calls; cached per
        memory tensor so a beam-search loop projects the clip once instead of once per step."""
        key = (which, enc_mem.data_ptr(), tuple(enc_mem.shape), enc_mem._version)
        hit = self._encproj_cache.get(which)
        if hit is not None and hit[0] == key:
            return hit[1]
        N, S, E = enc_mem.shape
        H, A = self.decoder.model.hidden_size, self.decoder.attn.attn_size
        out = torch.empty(N, S, A if which == 0 else E, device=enc_mem.device)
        _lib.call("acvae_attn_precompute", ptr_table(self._text_table()), which, enc_mem, out, N, S, E, H, A,
                  _lib.current_stream())
        self._encproj_cache[which] = (key, out, enc_mem)      # keep enc_mem alive so its address cannot be reused
        return out

    @torch.no_grad()
    def beam_search(self, encoded, max_length, beam_size):
        """Validation beam search, models/vae_model.py:896-995: beams expanded over the flat beam*V log-probabilities of
        a clip, states re-gathered by prev_word_inds; returns beam 0 (the reference never fills done_beams, :986-995).
        All clips advance together; no host synchronisation inside the loop."""
        mem_all = encoded["audio_embeds"].contiguous()
        dev = mem_all.device
        if hasattr(self, "ln"):                              # vae_model.py:754-755
            Nn, Ss, Ee = mem_all.shape
            proj = torch.empty(Nn, Ss, self.decoder.embed_size, device=dev)
            _lib.call("acvae_gemm_nt", mem_all, Ee, self.ln.weight, Ee, self.ln.bias, proj, self.decoder.embed_size,
                      Nn * Ss, self.decoder.embed_size, Ee, 0, _lib.current_stream())
            mem_all = proj
        lens_all = torch.as_tensor(encoded["audio_embeds_lens"]).to(device=dev, dtype=torch.long).contiguous()
        N, S, E = mem_all.shape
        V = self.vocab_size
        H, A = self.decoder.model.hidden_size, self.decoder.attn.attn_size
        replay = self.noise.get("eps_beam") if self.noise is not None else None
        self.noise = None
        # The reference walks the clips one after the other; their searches are independent, so all N x beam rows advance
        # together inside one library call.  Its noise order (clip-major: text_encoder.py:259 inside the clip loop) is
        # kept by drawing eps[clip][t] in that order on the host; the library reads it step-major.
        if replay is None:
            eps_all = _lib.h2d_fill((max_length, N, beam_size, E), torch.float32, dev,
                                    lambda buf: [torch.randn(beam_size, E, out=buf[t, i]) for i in range(N)
                                                 for t in range(max_length)])
        else:
            eps_all = _lib.h2d(torch.as_tensor(replay).reshape(N, max_length, beam_size, E).transpose(0, 1).contiguous(),
                               dev, torch.float32)
        seqs = torch.empty(N, max_length, dtype=torch.long, device=dev)
        attw = torch.empty(N, S, max_length, device=dev)
        sb = _lib.call("acvae_beam_search_scratch_bytes", N, beam_size, max_length, S, E, H, A, V)
        scratch = scratch_buffer(sb, dev)
        _lib.call("acvae_beam_search", ptr_table(self._text_table()), mem_all, lens_all, eps_all, int(self.start_idx), seqs,
                  attw, scratch, sb, N, beam_size, max_length, S, E, H, A, V, _lib.current_stream())
        return {"seqs": seqs, "attn_weights": attw}

    @torch.no_grad()
    def diverse_beam_search(self, encoded, max_length, beam_size, group_size, diversity_lambda, temperature, group_nbest):
        """CaptionModel.diverse_beam_search (models/word_model.py:297-394) with this model's hooks (vae_model.py:
        997-1040): per clip, ``group_size`` groups of ``bdash = beam_size // group_size`` beams; group g runs one
        step behind group g-1 and its log-probabilities at a local step are lowered by ``diversity_lambda`` x the
        number of times the earlier groups chose each word at that step; finished beams score logprob / length.
        Returns {"seqs": i64 [N, beam_size or group_size, max_length]}.  Host bookkeeping (sequence tables, finished
        beams) as in the reference, which also reads the chosen words back every step; prior / decoder step, the score
        transform and the flat top-k are library calls."""
        mem_all = encoded["audio_embeds"].contiguous()
        dev = mem_all.device
        if hasattr(self, "ln"):
            Nn, Ss, Ee = mem_all.shape
            proj = torch.empty(Nn, Ss, self.decoder.embed_size, device=dev)
            _lib.call("acvae_gemm_nt", mem_all, Ee, self.ln.weight, Ee, self.ln.bias, proj, self.decoder.embed_size,
                      Nn * Ss, self.decoder.embed_size, Ee, 0, _lib.current_stream())
            mem_all = proj
        lens_all = torch.as_tensor(encoded["audio_embeds_lens"]).to(torch.long)
        N, S, E = mem_all.shape
        V = self.vocab_size
        bdash = beam_size // group_size
        if bdash < 1 or bdash > 16:
            raise ValueError("diverse_beam_search: beam_size // group_size must be in 1..16")
        R = N * bdash
        st = _lib.current_stream
        # The clips are independent, so they advance together: rows = clips x bdash per (t, group) step; one host read-back
        # per step for the whole batch (the reference reads back per clip and step).  The reference's noise order — clip,
        # then t, then group (text_encoder.py:259 inside those loops) — is kept by drawing every eps up front in it.
        steps = [(t, g) for t in range(max_length + group_size - 1) for g in range(group_size)
                 if 0 <= t - g <= max_length - 1]
        eps_all = _lib.h2d_fill((N, len(steps), bdash, E), torch.float32, dev,
                                lambda buf: [torch.randn(bdash, E, out=buf[i, k]) for i in range(N)
                                             for k in range(len(steps))])
        mem = mem_all.repeat_interleave(bdash, dim=0).contiguous()
        lens = lens_all.repeat_interleave(bdash)
        scores = torch.empty(R, V, device=dev)
        vals = torch.empty(R, device=dev)
        idx, prev_d, nxt_d = (torch.empty(R, dtype=torch.long, device=dev) for _ in range(3))
        seq = [[np.zeros((bdash, 0), np.int64) for _ in range(group_size)] for _ in range(N)]
        score = [np.zeros((N, bdash), np.float32) for _ in range(group_size)]
        done = [[[] for _ in range(group_size)] for _ in range(N)]
        carry = [None] * group_size
        base = (np.arange(N) * bdash)[:, None]
        for k, (t, g) in enumerate(steps):
            lt = t - g
            if lt == 0:
                w = torch.full((R,), self.start_idx, dtype=torch.long, device=dev)
                state = self.decoder.init_hidden(R).to(dev)
                hid = self.pnet.init_hidden(R, dev)
                last_z = torch.zeros(R, E, device=dev)
            else:
                state0, hid0, z0, w, parent = carry[g]
                state = state0[:, parent].contiguous()
                hid = (hid0[0][:, parent].contiguous(), hid0[1][:, parent].contiguous())
                last_z = z0[parent].contiguous()
            pn = self.pnet(w.unsqueeze(1), mem, hid, last_z, lens, eps=eps_all[:, k].reshape(R, E))
            dn = self.decoder(word=w.unsqueeze(1), state=state, enc_mem=mem, enc_mem_lens=lens, z=pn["z"])
            logits = dn["logits"].squeeze(1)
            counts = None
            if g > 0:                                        # add_diversity (:298-312), one count vector per clip
                c = np.zeros((N, V), np.float32)
                for i in range(N):
                    for earlier in range(g):
                        np.add.at(c[i], seq[i][earlier][:, lt], 1.0)
                counts = _lib.h2d(c, dev)
            _lib.call("acvae_dbs_scores", logits, V, float(temperature), counts, float(diversity_lambda),
                      _lib.h2d(score[g].reshape(-1), dev), scores, R, V, bdash if g > 0 else 0, st())
            _lib.call("acvae_topk_flat_batched", scores, V if lt == 0 else bdash * V, bdash * V, bdash, V, vals, idx, prev_d,
                      nxt_d, N, bdash, st())
            top = vals.cpu().numpy().reshape(N, bdash).copy()              # one read-back per step for all clips
            parent_h = prev_d.cpu().numpy().reshape(N, bdash) - base        # beam index within the clip
            nxt_h = nxt_d.cpu().numpy().reshape(N, bdash)
            last = t == max_length + g - 1
            for i in range(N):
                sq = np.concatenate([seq[i][g][parent_h[i]] if lt > 0 else seq[i][g], nxt_h[i][:, None]], axis=1)
                seq[i][g] = sq
                ended = sq[:, lt] == self.end_idx
                if last:
                    ended[:] = True
                for b_ in range(bdash):
                    if ended[b_]:
                        done[i][g].append({"seq": sq[b_].copy(), "score": float(top[i, b_]) / (lt + 1)})
                top[i][ended] -= np.float32(1000)
            score[g] = top
            carry[g] = (dn["state"], pn["hiddens_state"], pn["z"], nxt_d.clone(), prev_d.clone())
        out = torch.full((N, beam_size if group_nbest else group_size, max_length), self.end_idx, dtype=torch.long)
        for i in range(N):
            ranked = [sorted(d, key=lambda x: -x["score"])[:bdash] for d in done[i]]
            chosen = sum(ranked, []) if group_nbest else [d[0] for d in ranked]
            for r, beam in enumerate(chosen):
                out[i, r, :len(beam["seq"])] = torch.from_numpy(beam["seq"])
        return {"seqs": out.to(dev)}

    def check_persistent_launches(self, device=None):
        """Raise if a persistent decode / posterior launch on the model's device gave up (its outputs are NaN then).  Call
        behind a synchronisation point; TrainStep does at its in-flight event."""
        dev = device if device is not None else next(self.parameters()).device
        _lib.check_persist_status(dev)

    def _side_stream(self, main):
        """Second stream of the decode calls (prior chain forward, trailing parameter gradients backward)."""
        if getattr(self, "_side", None) is None or self._side.device != main.device:
            self._side = torch.cuda.Stream(device=main.device, priority=int(os.environ.get("ACVAE_SIDE_PRIO", "0")))
        return self._side

    def _post_stream(self, main):
        """The posterior's own stream (forward beside the encoder, backward beside the decode calls' trailing parameter
        gradients: on the SAME stream its backward - which the encoder's backward waits for - sat behind 0.6 ms of them)."""
        if os.environ.get("ACVAE_POST_STREAM", "1") == "0":       # A/B: the posterior on the decode calls' second stream, as until round 4
            return self._side_stream(main)
        if getattr(self, "_side_q", None) is None or self._side_q.device != main.device:
            self._side_q = torch.cuda.Stream(device=main.device, priority=int(os.environ.get("ACVAE_POST_PRIO", "0")))
        return self._side_q

    def _aux_stream(self):
        """Second HIP stream handle for the decode calls (prior chain beside the decoder chain), or None."""
        if not self.use_side_stream or os.environ.get("ACVAE_DECODE_AUX", "1") == "0":
            return None
        return self._side_stream(torch.cuda.current_stream()).cuda_stream

    # ---- reference API
    def train_forward(self, encoded, caps, cap_lens, **kwargs):
        return self.stepwise_forward(encoded, caps, cap_lens, **kwargs)

    def inference_forward(self, encoded, **kwargs):
        method = kwargs.get("method", "greedy")
        max_length = kwargs.get("max_length", self.max_length)
        if method == "beam":                                              # vae_model.py:884-886
            return self.beam_search(encoded, max_length, kwargs.get("beam_size", 3))
        if method == "dbs":                                               # vae_model.py:887-893
            return self.diverse_beam_search(encoded, max_length, kwargs.get("beam_size", 5), kwargs.get("group_size", 5),
                                            kwargs.get("diversity_lambda", 0.5), kwargs.get("temperature", 1.0),
                                            kwargs.get("group_nbest", True))
        return self.stepwise_forward(encoded, None, None, **kwargs)     # greedy / "gumbel" / anything else = multinomial

    def forward(self, *input, **kwargs):
        """models/vae_model.py:732-760"""
        self._forward_token = getattr(self, "_forward_token", 0) + 1     # per-forward caches (decoder.embedding_table)
        if len(input) == 4:
            feats, feat_lens, caps, cap_lens = input
            # The posterior (42 serial BiGRU steps of tiny kernels) does not depend on the encoder: run it on a side
            # HIP stream beside the MFMA-bound encoder; autograd replays its backward on that stream too, where it
            # overlaps with the encoder backward.
            main = torch.cuda.current_stream()
            side = self._post_stream(main) if self.use_side_stream else main
            eps_q = None if self.noise is None else self.noise.get("eps_q")
            if eps_q is None:                                      # same generator order as the reference: the
                lens1 = np.asarray(cap_lens) - 1                   # posterior's randn precedes the per-step draws
                eps_q = torch.randn(feats.shape[0], int(lens1.max()), self.decoder.embed_size)
            # Host-side draws and the small H2D copies of the decode loop go in front of the encoder launch: a
            # pageable-memory copy waits for the stream to drain, which behind the encoder would stall the host.
            prep = self._host_prepare(feats.shape[0], feats.device, caps, cap_lens, kwargs)
            if side is not main:
                side.wait_stream(main)
                prep["caps_d"].record_stream(side)
            # The encoder is queued FIRST and the posterior second (it still starts at once: the side stream only waits
            # for what main held before this point).  Autograd runs the later-created node first, so in the backward
            # the posterior's kernels and its gradient bucket are queued before the long encoder backward: under data
            # parallelism the 20 MB posterior bucket then travels beside the encoder backward instead of behind it.
            encoded = self.encoder(feats, feat_lens)
            with torch.cuda.stream(side):
                qnetout = self.qnet(prep["caps_d"], cap_lens, eps=eps_q)
            if side is not main:
                main.wait_stream(side)
                for v in qnetout.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(main)
            encoded["_prep"] = prep
            encoded.update(qnetout)
            return self.train_forward(encoded, caps, cap_lens, **kwargs)
        if len(input) == 2:
            feats, feat_lens = input
            encoded = self.encoder(feats, feat_lens)
            return self.inference_forward(encoded, **kwargs)
        raise Exception("Number of input should be either 4 (feats, feat_lens, caps, cap_lens) or 2 (feats, feat_lens)")

    def _host_prepare(self, N, dev, caps, cap_lens, kwargs):
        """The decode loop's host-side random decisions, in the reference's per-step order (scheduled-sampling coin
        :826, prior noise text_encoder.py:259 on the CPU generator (F9), disentangle coin :802-806), and the device
        copies of the caption ids / lengths / noise."""
        E = self.decoder.embed_size
        train = caps is not None
        if train:
            lens1 = np.asarray(cap_lens) - 1
            Tc = int(max(cap_lens)) - 1
            ss_ratio, dis_ratio = kwargs["ss_ratio"], kwargs["dis_ratio"]
        else:
            Tc = kwargs.get("max_length", self.max_length)
        replay = self.noise
        self.noise = None
        ss_flags, dis_flags = [], []
        draw = replay is None or replay.get("eps_p") is None
        # sample_next_word's method (models/word_model.py:173-207): "greedy", "gumbel", anything else = multinomial
        # sampling with `temp`.  The non-greedy branches draw one [N,V] tensor per step on the CPU generator right after
        # the step's prior noise (and dis_ratio coin): torch.rand for the Gumbel noise (:189-191), and - inside
        # torch.multinomial(., 1) - empty(N,V).exponential_(1).  Both are drawn here in that order and uploaded once.
        method = kwargs.get("method", "greedy")
        temp = float(kwargs.get("temp", 1))
        V = self.vocab_size
        code = 0 if method == "greedy" else (1 if method == "gumbel" else 2)
        # rng="device" (or model.sample_rng = "device") - opt-in, not the reference's stream: ONE draw on the CPU generator
        # seeds a counter-based generator on the device that fills the [Tc,N,V] noise (acvae_sample_noise); same
        # distributions, so the captions are samples of the same model, but not the words the reference would draw from
        # this torch seed.  The default ("host") reproduces the reference's draws one for one.
        rng = kwargs.get("rng", getattr(self, "sample_rng", "host"))
        if rng not in ("host", "device"):
            raise ValueError(f"rng must be 'host' or 'device', got {rng!r}")
        sample_noise = None
        device_noise = bool(code) and rng == "device" and (replay is None or replay.get("sample_noise") is None)
        if code and not device_noise and (replay is None or replay.get("sample_noise") is None):
            sample_noise = torch.empty(Tc, N, V)
        # the decoder's word-embedding nn.Dropout (models/decoder.py:33,184): one [N,1,E] Bernoulli draw per step, made
        # by decoder.forward, i.e. after the step's prior noise and disentangle coin and before sample_next_word
        drop_p = float(self.decoder.dropoutlayer.p) if self.decoder.training else 0.0   # nn.Dropout acts on decoder.training alone,
        # also in the 2-input forward of a model left in train() (models/decoder.py:184)
        dec_keep = None
        if drop_p > 0.0 and (replay is None or replay.get("dec_keep") is None):
            dec_keep = torch.empty(Tc, N, E, dtype=torch.bool)

        def host_draws(eps):            # eps: [Tc, N, E] staging slot (None when the noise is replayed)
            for t in range(Tc):
                if train:
                    ss_flags.append(random.random() < ss_ratio)                    # :826
                if eps is not None:
                    torch.randn(N, E, out=eps[t])                                    # text_encoder.py:259 (CPU, F9)
                if train:
                    dis_flags.append(bool(dis_ratio != 0 and torch.rand(1) <= dis_ratio))   # :802-806
                if dec_keep is not None:
                    dec_keep[t].bernoulli_(1 - drop_p)
                if sample_noise is not None:
                    if code == 1:
                        U = torch.rand(N, V)                                          # word_model.py:189-191
                        torch.neg(torch.log(-torch.log(U + 1e-20) + 1e-20), out=sample_noise[t])
                    else:
                        sample_noise[t].exponential_(1)                               # torch.multinomial(prob, 1)

        if draw:
            eps_p = _lib.h2d_fill((Tc, N, E), torch.float32, dev, host_draws)
        else:
            host_draws(None)
            eps_p = _lib.h2d(replay["eps_p"][:Tc], dev, torch.float32).contiguous()
        caps_d = lens1_d = None
        if train:
            caps_d = _lib.h2d(caps, dev, torch.long).contiguous()
            lens1_d = _lib.h2d(lens1, dev, torch.long)
        sampling = {}
        if device_noise:
            seed = int(torch.randint(0, 2 ** 62, (1,)))             # after the step draws: torch.manual_seed fixes it
            noise_d = torch.empty(Tc, N, V, device=dev, dtype=torch.float32)
            _lib.call("acvae_sample_noise", noise_d, noise_d.numel(), code, seed, _lib.current_stream())
            sampling["sample"] = (code, temp, noise_d)
        elif code:
            if sample_noise is None:
                sample_noise = torch.as_tensor(replay["sample_noise"])[:Tc]
            sampling["sample"] = (code, temp, _lib.h2d(sample_noise, dev, torch.float32).contiguous())
        if drop_p > 0.0:
            if dec_keep is None:
                dec_keep = torch.as_tensor(replay["dec_keep"])[:Tc]
            sampling["emb_keep"] = (_lib.h2d(dec_keep.to(torch.uint8), dev).contiguous(), drop_p)
        return {"Tc": Tc, "ss_flags": ss_flags, "dis_flags": dis_flags, "eps_p": eps_p, "caps_d": caps_d,
                "lens1_d": lens1_d, "sampling": sampling}

    def stepwise_forward(self, encoded, caps, cap_lens, **kwargs):
        """models/vae_model.py:700-730 with decode_step (:792-816), prepare_decoder_input (:818-848) and
        stepwise_process_step (:850-869) fused into one device-side loop."""
        mem = encoded["audio_embeds"]
        dev = mem.device
        N = mem.shape[0]
        E = self.decoder.embed_size
        train = caps is not None
        mem_lens_d = encoded.get("audio_embeds_lens_dev")
        if mem_lens_d is None:
            mem_lens_d = _lib.h2d(encoded["audio_embeds_lens"], dev, torch.long)
        prep = encoded.pop("_prep", None) or self._host_prepare(N, dev, caps, cap_lens, kwargs)
        Tc, ss_flags, dis_flags, eps_p = prep["Tc"], prep["ss_flags"], prep["dis_flags"], prep["eps_p"]
        caps_d, lens1_d = prep["caps_d"], prep["lens1_d"]
        q_z = encoded["q_z"] if train else None
        self.staged = {"caps_d": caps_d, "lens1_d": lens1_d}           # device copies the loss can reuse
        outs = _DecodeFn.apply(self, mem, mem_lens_d, caps_d, lens1_d, q_z, eps_p, ss_flags, dis_flags, Tc,
                               prep.get("sampling"), *self._decode_weights())
        logits, outputs, seqs, slp, attw, pm, pl, pz, putt, hfin, hp, cp = outs
        output = {"seqs": seqs, "logits": logits, "outputs": outputs, "sampled_logprobs": slp,
                  "attn_weights": attw.transpose(1, 2), "p_means": pm, "p_logs": pl, "p_z": pz,
                  "state": hfin.unsqueeze(0), "hiddens_state": (hp.unsqueeze(0), cp.unsqueeze(0)), "last_z": pz[:, -1]}
        if train:
            for k in ("q_means", "q_logs", "q_z", "q_means_utt", "q_logs_utt"):
                output[k] = encoded[k]
            output["p_means_utt"] = putt
            output["p_logs_utt"] = None
        return output
