/* acvae_hip.h — C ABI of libacvae_hip.so: the MI355X (gfx950) kernels behind AC-VAE's training hot path.
 *
 * The reference (XinMing0411/AC-VAE) is pure Python on torch.nn and has NO FFI / operator API of its own
 * (SURVEY.md F1, §8(b)); this ABI is therefore build-defined.  Each entry point names the reference
 * computation it replaces (file:line relative to the reference root).  The Python host side
 * (the acvae_amd package) mirrors the reference's module classes and calls these through ctypes; see
 * INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless its name ends in _host.
 *   - the caller owns every buffer, workspaces included; the library never allocates or frees device
 *     memory (no hipMalloc / hipFree anywhere in csrc/; tests/test_abi_cpu.py greps for it).
 *   - Mutable state inside the library, ALL of it:
 *       (1) a per-THREAD, per-device pool of 64 HIP events used for the fork / join edges of the two-stream calls
 *           (thread_local: concurrent calls from different host threads share nothing, so every entry point is
 *           re-entrant across threads and streams);
 *       (2) the opt-in timing ring (acvae_prof_*, mutex-guarded, empty unless enabled);
 *       (3) one slot per DEVICE for the persistent launches (csrc/decode_persist.hip), mutex-guarded: the HIP event that
 *           chains a device's persistent launches behind each other, the device's CU count, which kernels had their
 *           dynamic-LDS limit raised there, and the status pointer registered with acvae_persist_status_register.
 *     There are no process-global behaviour switches: what used to be acvae_set_decode_persist / _decode_defer /
 *     _attn_split are per-call `flags` (ACVAE_FLAG_*).  The only environment variables the library reads are six A/B /
 *     tuning switches, each read ONCE per process into a static constant (never written afterwards): ACVAE_CONV_WINO
 *     (encoder.hip: 0 = implicit-GEMM convolutions), ACVAE_SKINNY_PAIR, ACVAE_HEADS_AUX (decoder.hip: launch pairing / which
 *     stream the heads' gradients use), ACVAE_SKINNY_SPLITK (gemm.hip), ACVAE_BF16_BDMA, ACVAE_WGB_SLABCOST (conv_bf16.hip).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing synchronises.
 *   - return 0 on success, a negative ACVAE_E* code for bad arguments, or a positive hipError_t.
 *   - fp32 everywhere ("dtype f32"), token ids / lengths int64 (as torch.long).
 *   - "ld*" are leading dimensions in ELEMENTS; rows are addressed as base + row*ld.
 */
#ifndef ACVAE_HIP_H
#define ACVAE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ACVAE_ABI_VERSION 3
int acvae_abi_version(void);

/* Per-call option bits (`flags`, the last argument of the entry points that take them; 0 = defaults). */
#define ACVAE_FLAG_NO_PERSIST 1          /* decode / posterior: the per-step launches instead of the persistent kernels */
#define ACVAE_FLAG_DEFER_PARAM_GRADS 2   /* acvae_decode_bwd: parameter gradients trail on aux_stream (see there) */
#define ACVAE_FLAG_NO_ATTN_SPLIT 4       /* acvae_attn_fwd: never the split-over-frames form */
#define ACVAE_FLAG_TEST_STALL 8          /* test aid: a persistent launch is queued one workgroup short with a short spin
                                            bound, so that its roles give up as they would if part of the grid were not
                                            resident; exercises the abort -> NaN -> status path below */

/* Persistent launches (the teacher-forced decode loop and the posterior BiGRU, forward and backward: one launch each whose
 * workgroups hand results to each other and therefore must ALL be resident).  The library (1) takes that path only when the
 * occupancy calculator says the whole grid fits the current device, (2) chains a device's persistent launches behind each
 * other so that two never share the chip, and (3) bounds every wait: a launch that still cannot finish (another PROCESS
 * holds the CUs) gives up, a one-workgroup tail kernel behind it overwrites its outputs with NaN and sets status words
 * the caller registered for the device:
 *   status_words_host: 8 uint32 in page-locked, device-visible host memory owned by the caller (hipHostMalloc /
 *   torch pin_memory), zeroed by the caller; word k (0 decode fwd, 1 decode bwd, 2 posterior fwd, 3 posterior bwd) and word 4
 *   ("any") become 1.  The caller reads them at a point where the stream has drained (acvae_amd: TrainStep's in-flight event,
 *   Hybrid_VAEModel.check_persistent_launches) and raises.  NULL unregisters.  Without a registered pointer only the NaNs tell. */
int acvae_persist_status_register(int device, void* status_words_host);

/* ---------------------------------------------------------------------------------------------
 * Generic dense products on fp32 MFMA (v_mfma_f32_32x32x2_f32).  Replace torch.nn.Linear /
 * F.linear call sites of the path (models/attn_model.py:32, models/decoder.py:198,
 * models/text_encoder.py:192,255, models/vae_model.py:726) and their autograd backward.
 *   NT: C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ C if accumulate)
 *   TN: C[M,N] = sum_k A[K,M]^T . B[K,N]   (weight gradients: dW = dY^T . X)
 * ------------------------------------------------------------------------------------------- */
int acvae_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                  int64_t ldc, int M, int N, int K, int accumulate, void* stream);
int acvae_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N,
                  int K, int accumulate, float* slab_ws, int64_t slab_ws_bytes, void* stream);
int64_t acvae_gemm_tn_workspace_bytes(int M, int N, int K);
/* out[c] = sum_r x[r,c] of a contiguous [rows, cols] matrix (bias gradients); deterministic (fixed-order fp64 combine) */
int64_t acvae_colsum_workspace_bytes(int cols);
int acvae_colsum(const float* x, int rows, int cols, float* out, void* ws, int64_t ws_bytes, void* stream);
/* out[c,r] = in[r,c] */
int acvae_transpose(const float* in, int64_t ld_in, float* out, int64_t ld_out, int rows, int cols, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A3  Seq2SeqAttention.forward  models/attn_model.py:20-46, with the loop-invariant half of
 * h2attn hoisted:  encproj[n,s,:] = W[:, hs_dec:] . h_enc[n,s] + b  (once per batch),
 * qproj[row,:] = W[:, :hs_dec] . h_dec[row]  (per query).  Query row (n,j), n<N, j<Tq lives at
 * base + n*stride_n + j*stride_j for qproj / ctx / weights (lets a decode step address column t of
 * batch-major [N,Tc,.] buffers).  score = v . tanh(qproj + encproj), positions s >= lens[n] get
 * -1e10 before the softmax (attn_model.py:41), ctx = sum_s w_s h_enc[n,s].
 * ------------------------------------------------------------------------------------------- */
/* acvae_attn_fwd with N * Tq <= 128 query rows and S > 16 splits the frames of a row over workgroups (one decode step of the
 * step API / beam search / sampled decode; results equal to the one-workgroup form up to the summation order of the softmax
 * denominator and the context) - when the caller hands over `ws` of acvae_attn_fwd_workspace_bytes (0: the shape never
 * splits).  Its first 1024 bytes are arrival counters: ZERO them once (hipMemsetAsync) before the first call; every call
 * leaves them zero, so stream-ordered calls may reuse the workspace; calls that can run side by side need one each.
 * ws == NULL or ACVAE_FLAG_NO_ATTN_SPLIT: the one-workgroup form. */
int64_t acvae_attn_fwd_workspace_bytes(int N, int Tq, int S, int A, int E);
int acvae_attn_fwd(const float* qproj, int64_t q_sn, int64_t q_sj, const float* encproj, const float* enc,
                   const int64_t* lens, const float* v, float* ctx, int64_t c_sn, int64_t c_sj, float* weights,
                   int64_t w_sn, int64_t w_sj, int N, int Tq, int S, int A, int E, void* ws, int64_t ws_bytes, void* stream,
                   int flags);
/* Test aid: y[i] = the tanh the attention kernels evaluate (hardware exp2 / rcp form, absolute error <= 2.5e-7 (measured 2.1e-7); a library built
 * with -DACVAE_EXACT_TANH uses tanhf instead, for parity debugging). */
int acvae_tanh_att(const float* x, float* y, int64_t n, void* stream);
/* Backward of the above for upstream dctx (attention weights carry no gradient on this path).
 * dencproj [N,S,A] and denc [N,S,E] are ACCUMULATED into (+=); dv_part [N,A] is accumulated into;
 * dqproj rows are written.  `ws`: scratch of acvae_attn_bwd_workspace_bytes(N,Tq,S,A). */
int64_t acvae_attn_bwd_workspace_bytes(int N, int Tq, int S, int A);
int acvae_attn_bwd(const float* dctx, int64_t dc_sn, int64_t dc_sj, const float* qproj, int64_t q_sn, int64_t q_sj,
                   const float* encproj, const float* enc, const int64_t* lens, const float* v,
                   const float* weights, int64_t w_sn, int64_t w_sj, float* dqproj, int64_t dq_sn, int64_t dq_sj,
                   float* dencproj, float* denc, float* dv_part, float* ws, int64_t ws_bytes, int N, int Tq, int S,
                   int A, int E, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reparameterisation  z = eps * exp(.5*logvar) + mu   models/text_encoder.py:196-197,259-262.
 * `ml` holds [rows, 2E] = [mu | logvar] (the Linear output, split at half: text_encoder.py:257-258).
 * Writes mean/log/z (each row stride ld_out) and, if z2 != NULL, a second copy of z (row stride ld_z2).
 * ------------------------------------------------------------------------------------------- */
int acvae_reparam_fwd(const float* ml, int64_t ld_ml, const float* eps, int64_t ld_eps, float* mean, float* logv,
                      float* z, int64_t ld_out, float* z2, int64_t ld_z2, int rows, int E, void* stream);
/* dml[:, :E] = dz + dmean_ext ; dml[:, E:] = dz*eps*.5*exp(.5*logvar) + dlog_ext  (ext terms may be NULL) */
int acvae_reparam_bwd(const float* dz, int64_t ld_dz, const float* dmean_ext, const float* dlog_ext, int64_t ld_ext,
                      const float* logv, int64_t ld_lv, const float* eps, int64_t ld_eps, float* dml, int64_t ld_dml,
                      int rows, int E, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A9  Normal_kl_loss.forward  utils/train_util.py:259-266:  sum over E, mean over ALL rows (unmasked, F8).
 * Inputs are [rows,E] contiguous.  `partials` is a scratch of acvae_kl_partials(rows,E) floats.
 * ------------------------------------------------------------------------------------------- */
int64_t acvae_kl_partials(int64_t n_elem);
int acvae_gauss_kl_fwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2, float* partials,
                       float* out_scalar, int64_t rows, int E, void* stream);
/* grad_out: device scalar (dLoss/dKL).  Any of the four outputs may be NULL. */
int acvae_gauss_kl_bwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2, const float* grad_out,
                       float* dmu1, float* dlv1, float* dmu2, float* dlv2, int64_t rows, int E, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A6  CaptionModel.sample_next_word (greedy)  models/word_model.py:173-207  and
 * A8  LabelSmoothingLoss.forward  utils/train_util.py:243-251 / torch CrossEntropyLoss (runner :222-227)
 * A13 masked CrossEntropyLoss / LabelSmoothingLoss  losses/loss.py:18-70
 * One pass over logits [N,T,V] (row (n,t) at logits + n*ld_n + t*ld_t) producing per-row
 * argmax (first maximum), max log-prob and log-sum-exp; the CE kernels reuse the row stats.
 * A row (n,t) is valid iff t < lens1[n] (lens1 = cap_lens-1; NULL = all valid).
 * ------------------------------------------------------------------------------------------- */
int acvae_row_logsoftmax_argmax(const float* logits, int64_t ld_n, int64_t ld_t, int64_t* argmax, float* max_logprob,
                                float* lse, int64_t o_sn, int64_t o_st, int N, int T, int V, void* stream);
/* The non-greedy branches of sample_next_word (models/word_model.py:188-203) for rows (n,t) of logits as above.
 *   ACVAE_SAMPLE_GUMBEL      w = argmax_c (log_softmax(logits)_c + g_c) / temp, g = -log(-log(U+1e-20)+1e-20) with
 *                            U = torch.rand(N,V) drawn by the caller on the CPU generator (:189-191);
 *   ACVAE_SAMPLE_MULTINOMIAL w = torch.multinomial(exp(log_softmax(logits) / temp), 1) = argmax_c p_c / q_c with
 *                            q = empty(N,V).exponential_(1), which is how ATen draws one sample per row; the caller
 *                            draws q the same way.
 * noise row (n,t) at noise + n*nz_sn + t*nz_st (V floats: g or q).  Writes w (int64) and log_softmax(logits)[w]
 * (the reference's "probs") at w_out / logprob_out + n*o_sn + t*o_st.  First maximum wins (torch.max). */
#define ACVAE_SAMPLE_GREEDY 0
#define ACVAE_SAMPLE_GUMBEL 1
#define ACVAE_SAMPLE_MULTINOMIAL 2
/* n floats of sampling noise generated on the device (counter-based Philox, element i from (seed, i)): Gumbel noise for
 * ACVAE_SAMPLE_GUMBEL, Exp(1) draws for ACVAE_SAMPLE_MULTINOMIAL - what the caller otherwise draws on the CPU generator
 * (models/word_model.py:188-203).  Same distributions, another random stream: the opt-in fast mode of method="sample" /
 * "gumbel"; the CPU-generator mode stays the parity default. */
int acvae_sample_noise(float* noise, int64_t n, int method, uint64_t seed, void* stream);
int acvae_sample_next_word(const float* logits, int64_t ld_n, int64_t ld_t, const float* noise, int64_t nz_sn,
                           int64_t nz_st, int method, float temp, int64_t* w_out, float* logprob_out, int64_t o_sn,
                           int64_t o_st, int N, int T, int V, void* stream);
/* reduction: 0 none (writes loss_rows only), 1 mean over valid rows, 2 sum.  loss_rows [N,T] (0 at invalid rows). */
int acvae_ls_ce_fwd(const float* logits, int64_t ld_n, int64_t ld_t, const int64_t* targets, int64_t tg_sn,
                    const int64_t* lens1, const float* lse, float smoothing, int reduction, float* loss_rows,
                    float* out_scalar, int N, int T, int V, void* stream);
/* dlogits [N,T,V] contiguous rows (ld_n, ld_t as logits); grad_out device scalar; grad_rows optional [N,T]
 * (reduction none).  Invalid rows get zeros. */
int acvae_ls_ce_bwd(const float* logits, int64_t ld_n, int64_t ld_t, const int64_t* targets, int64_t tg_sn,
                    const int64_t* lens1, const float* lse, float smoothing, int reduction, const float* grad_out,
                    const float* grad_rows, float* dlogits, int N, int T, int V, void* stream);
/* mean((a-b)^2) over n elements and its backward (runner :317, nn.MSELoss). */
int acvae_mse_fwd(const float* a, const float* b, float* partials, float* out_scalar, int64_t n, void* stream);
int acvae_mse_bwd(const float* a, const float* b, const float* grad_out, float* da, float* db, int64_t n, void* stream);
/* A10 loss assembly, runners/pytorch_runner_vae.py:315-320: loss = ce + w_kl * kl (+ w_mse * mse; mse may be NULL) on device
 * scalars, and its gradient (g, g * w_kl, g * w_mse) - one launch each instead of a chain of scalar tensor kernels. */
int acvae_loss_combine_fwd(const float* ce, const float* kl, const float* mse, float w_kl, float w_mse, float* out, void* stream);
int acvae_loss_combine_bwd(const float* grad_out, float w_kl, float w_mse, float* g_ce, float* g_kl, float* g_mse, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A1  Cnn10.forward  models/encoder.py:672-707 (ConvBlock :606-649) and its backward.
 * feats f32 [N,T,F=64] -> audio_embeds f32 [N,S,512] (S = T/16), audio_embeds_pooled f32 [N,512].
 * `params` / `grads`: pointer tables in the reference's state-dict order for the encoder:
 *   bn0.{weight,bias,running_mean,running_var,num_batches_tracked},
 *   conv_block{1..4}.{conv1.weight, conv2.weight, bn1.{w,b,rm,rv,nbt}, bn2.{w,b,rm,rv,nbt}},
 *   embed_pooled.{weight,bias}                                   (55 entries; weights in OIHW).
 * training != 0: BatchNorm uses batch statistics and updates the running buffers (momentum 0.1,
 * unbiased variance) and dropout (p_block after every block, p_fc around embed_pooled) is applied,
 * drawn from Philox(seed) unless `masks` supplies the 6 (Cnn14_16k: 8) keep-masks explicitly (uint8, the reference's
 * NCHW / [N,512] order: parity tests).  `saved` (acvae_encoder_saved_bytes) carries activations from
 * fwd to bwd; `scratch` (acvae_encoder_scratch_bytes) is free between calls.  Gradients are WRITTEN
 * (not accumulated) for every conv / bn weight and bias; embed_pooled receives none (its output is
 * not consumed on this path, models/vae_model.py:821).
 * `arch` selects the PANNs encoder: ACVAE_ARCH_CNN10 (above) or ACVAE_ARCH_CNN14_16K (SURVEY §8(f) N4,
 * models/encoder.py:871-964): conv_block{1..6} up to 2048 channels, the sixth block pooled (1,1), S = T/32,
 * audio_embeds [N,S,2048], the pooled head is fc1 (2048x2048); its table has 79 entries (…, conv_block6.*, fc1.{w,b})
 * and 8 dropout sites.  acvae_encoder_nparams / acvae_encoder_out_dims report the table length and (S, C).
 * ------------------------------------------------------------------------------------------- */
#define ACVAE_ENC_NPARAMS 55          /* Cnn10 */
#define ACVAE_ARCH_CNN10 0
#define ACVAE_ARCH_CNN14_16K 1
/* OR-ed into `arch`: BASELINE configs[2] "bf16 forward / fp32 loss".  The conv stack's activations (raw conv outputs,
 * pooled tensors, their gradients) and the repacked conv weights are STORED in bf16 and the 3x3 convolutions run on
 * v_mfma_f32_32x32x16_bf16; accumulation, BatchNorm statistics (taken from the rounded tensor that is stored), parameter
 * gradients, parameters and every output of these calls (audio_embeds, pooled) stay fp32.  Same tables, same dropout
 * streams; saved / scratch sizes roughly halve (ask acvae_encoder_saved_bytes / _scratch_bytes with the flag set). */
#define ACVAE_ENC_BF16 256
int acvae_encoder_nparams(int arch);
int acvae_encoder_out_dims(int arch, int T, int* S, int* C);
int64_t acvae_encoder_saved_bytes(int arch, int N, int T, int F);
int64_t acvae_encoder_scratch_bytes(int arch, int N, int T, int F);
int acvae_encoder_fwd(const void* const* params, const float* feats, float* audio_embeds, float* pooled,
                      void* saved, int64_t saved_bytes, void* scratch, int64_t scratch_bytes, int arch, int N, int T,
                      int F, int training, float p_block, float p_fc, uint64_t seed, const uint8_t* const* masks,
                      void* stream);
/* `training` must be the value the forward that filled `saved` was called with: 0 = evaluation-mode BatchNorm
 * (running statistics, no dropout; dY = scale * g, fine-tuning with frozen statistics and the data-parallel
 * equivalence test), != 0 = batch statistics. */
int acvae_encoder_bwd(const void* const* params, void* const* grads, const float* feats,
                      const float* d_audio_embeds, void* saved, int64_t saved_bytes, void* scratch,
                      int64_t scratch_bytes, int arch, int N, int T, int F, int training, float p_block, uint64_t seed,
                      const uint8_t* const* masks, void* stream);
/* Test aid: the ReLU decisions (y * scale + shift > 0, as the backward kernels evaluate them) of BN+ReLU site `site`
 * (0 .. 2*blocks-1: ConvBlock site/2 + 1, bn1 then bn2) read from the `saved` buffer of a forward call, written as uint8
 * [N,C,H,W] - the reference's layout.  A checker evaluates the reference under exactly these decisions instead of
 * tolerating mask bits that two fp32 summation orders put on different sides of zero. */
int acvae_encoder_relu_mask(const void* saved, int64_t saved_bytes, int arch, int N, int T, int F, int site,
                            uint8_t* mask_nchw, void* stream);
/* The same with a host callback: `block_done`, if not NULL, is a `void (*)(int block, void* user)` (passed as void*)
 * that is called on the calling thread right after the kernels producing ALL parameter gradients of ConvBlock `block`
 * (conv1 / conv2 / bn1 / bn2; blocks run nb .. 1) have been queued on `stream`: a data-parallel caller starts the
 * all-reduce of that block's gradient bucket there, behind `stream`, while the shallower blocks still run. */
int acvae_encoder_bwd_hooked(const void* const* params, void* const* grads, const float* feats,
                             const float* d_audio_embeds, void* saved, int64_t saved_bytes, void* scratch,
                             int64_t scratch_bytes, int arch, int N, int T, int F, int training, float p_block,
                             uint64_t seed, const uint8_t* const* masks, void* stream, void* block_done, void* user);

/* ---------------------------------------------------------------------------------------------
 * The encoder's kernels one by one (conv3x3[_bn], bn_mel, bn pieces): the same launchers
 * acvae_encoder_fwd / _bwd sequence, each callable - and tested - alone.  Activations are NHWC
 * [N][H][W][C] fp32 (the reference is NCHW: torch.nn.Conv2d(.., (3,3), (1,1), (1,1), bias=False),
 * models/encoder.py:612-622, 633-649), weights OIHW as in the state dict.  `ws`: scratch of
 * acvae_conv3x3_workspace_bytes / acvae_bn_workspace_bytes for the same dims, 16-byte aligned.
 *   acvae_conv3x3_fwd   Y = conv3x3(act(X), W), act = relu(x * in_scale[ci] + in_shift[ci]) (the PREVIOUS layer's fused
 *                       BatchNorm+ReLU) or the identity when in_scale == NULL.  If bn_out != NULL it also returns THIS
 *                       layer's BatchNorm as [4][Cout] = scale | shift | mean | invstd (y_bn = Y*scale + shift) from the
 *                       batch statistics (training != 0: running buffers updated, momentum 0.1, unbiased variance) or
 *                       from the running buffers (training == 0).  Cin == 1 is the stack's first convolution:
 *                       in_scale / in_shift are then bn0's per-MEL affine [W] (required), W = 64, Cout = 64.
 *   acvae_conv3x3_dgrad dX = conv3x3(dY, flipped / transposed W)           (Cin >= 32)
 *   acvae_conv3x3_wgrad dW[co][ci][tap] = sum_p dY[p][co] * act(X)[p + tap][ci]      (Cin >= 4; Cin == 1: below)
 *   acvae_conv1_first_bwd  first convolution: dW1 [64][1][3][3] and bn0's dgamma / dbeta [64] (x [N,T,64] features,
 *                       bn0 = [4][64] from acvae_bn_mel_fwd).
 *   acvae_bn_mel_fwd    bn0 (BatchNorm2d(64) over the MEL axis, models/encoder.py:655, 679-681) statistics of x [rows,64]
 *                       -> bn_out [4][64]; the normalised tensor is never written: the first convolution applies it.
 *   acvae_bn_relu_pool_fwd  P = dropout(avg_pool2x2(relu(Y*scale + shift)))  (pool != 0) or dropout(relu(..)) (pool == 0:
 *                       ConvBlock pool_size (1,1)); bn = [4][C] as above; dropout p_drop with Philox(seed, site) or an
 *                       explicit keep mask (NCHW order), p_drop = 0 disables.
 *   acvae_bn_relu_bwd   backward of relu(bn(Y)) for upstream dO: upstream 0 = dO [N,H,W,C] as is, 1 = dO [N,H/2,W/2,C]
 *                       through dropout + 2x2 average pool, 2 = through dropout only.  Writes dgamma, dbeta [C] and dY.
 *                       training == 0: evaluation-mode BatchNorm (dY = scale * g).
 * ------------------------------------------------------------------------------------------- */
int64_t acvae_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int acvae_conv3x3_fwd(const float* X, const float* W_oihw, const float* in_scale, const float* in_shift, float* Y,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, int training, float* bn_out, void* ws, int64_t ws_bytes, int N, int H,
                      int W, int Cin, int Cout, void* stream);
int acvae_conv3x3_dgrad(const float* dY, const float* W_oihw, float* dX, void* ws, int64_t ws_bytes, int N, int H, int W,
                        int Cin, int Cout, void* stream);
int acvae_conv3x3_wgrad(const float* dY, const float* X, const float* in_scale, const float* in_shift, float* dW_oihw,
                        void* ws, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream);
int acvae_conv1_first_bwd(const float* x, const float* bn0, const float* W1_oihw, const float* dY, float* dW1,
                          float* dgamma0, float* dbeta0, void* ws, int64_t ws_bytes, int N, int T, int F, void* stream);
/* Winograd F(2x2,3x3) forms of the forward convolution and the data gradient (conv_wino.hip): same contracts, fp32
 * throughout, 2.25x fewer matrix-pipe flops; results differ from the implicit-GEMM forms by fp32 rounding only.
 * W a power of two in 4..64, Cin % 16 == 0, Cout % 64 == 0 (dgrad: the roles of Cin / Cout swap), otherwise
 * ACVAE_EUNSUPPORTED.  Same workspace. */
int acvae_conv3x3_fwd_wino(const float* X, const float* W_oihw, const float* in_scale, const float* in_shift, float* Y,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, int training, float* bn_out, void* ws, int64_t ws_bytes, int N,
                           int H, int W, int Cin, int Cout, void* stream);
int acvae_conv3x3_dgrad_wino(const float* dY, const float* W_oihw, float* dX, void* ws, int64_t ws_bytes, int N, int H,
                             int W, int Cin, int Cout, void* stream);
/* The data gradient of a convolution whose INPUT was relu(batchnorm(Yprev)) - the second convolution of a ConvBlock
 * (models/encoder.py:631-641) - with the first pass of that BatchNorm + ReLU's backward inside: besides dX it returns
 * sum_g[Cin] = sum over pixels of g and sum_gy[Cin] = sum of g * (Yprev - mean) * invstd, g = dX where Yprev * scale + shift > 0.
 * bn_prev = [4][Cin]: scale | shift | mean | invstd (the bn_out of the layer that made Yprev).  The workspace must cover
 * acvae_conv3x3_workspace_bytes for (Cin, Cout) and for (Cout, Cin). */
int acvae_conv3x3_dgrad_bnred_wino(const float* dY, const float* W_oihw, float* dX, const float* Yprev, const float* bn_prev,
                                   float* sum_g, float* sum_gy, void* ws, int64_t ws_bytes, int N, int H, int W, int Cin,
                                   int Cout, void* stream);
/* weight gradient in the same form (Cin % 64 == 0, Cout % 64 == 0): 16 GEMMs over the tiles, split over workgroups into
 * fp32 slabs that are summed in fixed order (in double) and folded through G^T . G. */
int acvae_conv3x3_wgrad_wino(const float* dY, const float* X, const float* in_scale, const float* in_shift, float* dW_oihw,
                             void* ws, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream);
/* bf16-storage forms of the three convolutions (acvae_encoder_* with ACVAE_ENC_BF16): X / dY / dX / Y are bf16 NHWC
 * (uint16 bit patterns, round-to-nearest-even), weights OIHW fp32 (rounded to bf16 inside), dW fp32.  Cin % 64 == 0.
 * acvae_conv3x3_fwd_bf16 returns this layer's BatchNorm in bn_out exactly like the fp32 form, from the statistics of the
 * ROUNDED output.  Same workspace size as the fp32 forms. */
int acvae_conv3x3_fwd_bf16(const void* X, const float* W_oihw, const float* in_scale, const float* in_shift, void* Y,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, int training, float* bn_out, void* ws, int64_t ws_bytes, int N,
                           int H, int W, int Cin, int Cout, void* stream);
int acvae_conv3x3_dgrad_bf16(const void* dY, const float* W_oihw, void* dX, void* ws, int64_t ws_bytes, int N, int H, int W,
                             int Cin, int Cout, void* stream);
int acvae_conv3x3_wgrad_bf16(const void* dY, const void* X, const float* in_scale, const float* in_shift, float* dW_oihw,
                             void* ws, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream);
int64_t acvae_bn_workspace_bytes(int N, int H, int W, int C);
int acvae_bn_mel_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, int training, float* bn_out, void* ws, int64_t ws_bytes, int64_t rows,
                     int F, void* stream);
int acvae_bn_relu_pool_fwd(const float* Y, const float* bn, float* P, int N, int H, int W, int C, int pool, float p_drop,
                           uint64_t seed, int site, const uint8_t* keep_mask, void* stream);
int acvae_bn_relu_bwd(const float* Y, const float* dO, int upstream, const float* bn, float* dgamma, float* dbeta,
                      float* dY, void* ws, int64_t ws_bytes, int N, int H, int W, int C, int training, float p_drop,
                      uint64_t seed, int site, const uint8_t* keep_mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The recurrent cells one by one (gru_step, lstm_step, bigru_seq), torch.nn.GRU / LSTM
 * semantics and weight layout (gate order r|z|n and i|f|g|o):
 *   acvae_gru_step   one step of the decoder's GRU, models/decoder.py:39-44, 190: x [N,I], h [N,H] -> h_out [N,H]
 *   acvae_lstm_step  one step of the prior's LSTM, models/text_encoder.py:229-235, 254: (h, c) -> (h_out, c_out)
 *   acvae_bigru_seq  the posterior's packed bidirectional GRU, models/text_encoder.py:189-191: X [N,Tc,E] (batch-major),
 *                    lens [N] int64 (rows stop at their length; padded outputs are zero, pad_packed_sequence),
 *                    w = {w_ih, w_hh, b_ih, b_hh, then the same four of the reverse direction}, hidden [N,Tc,2H].
 * `ws`: scratch of acvae_rnn_workspace_bytes(N, Tc (1 for the single steps), I, H).
 * ------------------------------------------------------------------------------------------- */
int64_t acvae_rnn_workspace_bytes(int N, int Tc, int I, int H);
int acvae_gru_step(const float* x, const float* h, const float* w_ih, const float* w_hh, const float* b_ih,
                   const float* b_hh, float* h_out, void* ws, int64_t ws_bytes, int N, int I, int H, void* stream);
int acvae_lstm_step(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                    const float* b_ih, const float* b_hh, float* h_out, float* c_out, void* ws, int64_t ws_bytes, int N,
                    int I, int H, void* stream);
int acvae_bigru_seq(const float* X, const int64_t* lens, const void* const* w, float* hidden, void* ws, int64_t ws_bytes,
                    int N, int Tc, int E, int H, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text side of the path.  `params` / `grads` are pointer tables in the reference's state-dict order
 * after the encoder (ACVAE_TEXT_NPARAMS entries; ln.* may be NULL when absent):
 *   decoder.{word_embeddings.weight, model.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0},
 *            classifier.{weight,bias}, attn.{v, h2attn.weight, h2attn.bias}},
 *   qnet.{word_embedding.weight, network.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0,
 *         *_reverse x4}, token_mean_log.{weight,bias}},
 *   pnet.{word_embedding.weight, word_attn.{v, h2attn.weight, h2attn.bias},
 *         network.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0}, mean_log_out.{weight,bias}},
 *   mean_log_out.{weight,bias}, ln.{weight,bias}.
 * All [N,Tc,*] tensors are batch-major and contiguous like the reference's outputs.
 * Gradients are WRITTEN (not accumulated).
 * ------------------------------------------------------------------------------------------- */
#define ACVAE_TEXT_NPARAMS 35

/* A2  PosteriorRNN_hybrid.forward  models/text_encoder.py:182-216.  caps: int64 token ids [N, >=Tc]
 * (row stride ld_caps), lens1 = cap_lens-1 (device int64), Tc = max(lens1); eps_q [N,Tc,E] is the
 * torch.randn draw of :196 (made on the host, F9).  Packed-sequence semantics: a row stops at its
 * length, padded outputs are zero (so q_means/q_logs there equal the Linear bias). */
int64_t acvae_posterior_saved_bytes(int N, int Tc, int E, int Hq, int V);
int64_t acvae_posterior_scratch_bytes(int N, int Tc, int E, int Hq, int V);
int acvae_posterior_fwd(const void* const* params, const int64_t* caps, int64_t ld_caps, const int64_t* lens1,
                        const float* eps_q, float* q_means, float* q_logs, float* q_z, float* q_means_utt,
                        void* saved, int64_t saved_bytes, void* scratch, int64_t scratch_bytes, int N, int Tc, int E,
                        int Hq, int V, void* stream, int flags);
int acvae_posterior_bwd(const void* const* params, void* const* grads, const int64_t* lens1, const float* eps_q,
                        const float* q_logs, const float* d_q_means, const float* d_q_logs, const float* d_q_z,
                        const float* d_q_means_utt, void* saved, int64_t saved_bytes, void* scratch,
                        int64_t scratch_bytes, int N, int Tc, int E, int Hq, int V, void* stream, int flags);

/* The teacher-forced decode loop (all words known, no step feeds the prior's z to the decoder: acvae_decode_fwd with every
 * ss flag set and every dis flag clear) runs as ONE persistent launch (csrc/decode_persist.hip) when N <= 32, S <= 512,
 * E is a power of two in 32..2048, H, A are multiples of 32 and the whole grid fits the device at once; its results are
 * bit-identical to the per-step path.  The posterior's packed BiGRU (acvae_posterior_fwd / _bwd) runs the same way - one
 * launch per pass for both directions - when N <= 32 and Hq is a multiple of 32 up to 512.  ACVAE_FLAG_NO_PERSIST forces the
 * per-step paths (A/B, parity tests).  Replaces nothing in the reference (scheduling only). */

/* A4+A5+A6+A7 (+A12 when caps == NULL): the step-by-step decode of Hybrid_VAEModel
 * models/vae_model.py:700-730,792-869 with PriorRNN (text_encoder.py:247-268),
 * VAERNNBahdanauAttnDecoder (decoder.py:175-203) and greedy sample_next_word (word_model.py:173-207).
 *   mem_in [N,S,Eenc] audio_embeds (projected by ln when Eenc != E), mem_lens [N] (= feat_lens//16),
 *   caps/ld_caps/lens1 as above (NULL caps = inference: words come from the previous argmax, z from
 *   the prior, finished rows emit <end>),  q_z [N,Tc,E] posterior samples,
 *   eps_p [Tc,N,E] the per-step torch.randn draws of text_encoder.py:259,
 *   ss_flags_host[t] != 0: step t is teacher-forced (random.random() < ss_ratio, vae_model.py:826),
 *   dis_flags_host[t] != 0: step t feeds the PRIOR's z to the decoder (torch.rand(1) <= dis_ratio, :805).
 * Outputs: logits [N,Tc,V], outputs [N,Tc,H], seqs i64 [N,Tc], sampled_logprobs [N,Tc],
 *   attn_w [N,Tc,S] (the reference's attn_weights transposed), p_means/p_logs/p_z [N,Tc,E],
 *   p_means_utt [N,2E] (training only), final states h [N,H], (hp, cp) [N,E].
 * aux_stream (may be NULL): a second HIP stream of the same device.  With teacher forcing the prior's recurrence
 *   (embedding -> attention -> LSTM -> z_t) and the decoder's (attention -> GRU) are two independent chains of ~10 us
 *   kernels unless a step feeds the prior's z to the decoder; given a second stream the call forks after the hoisted
 *   GEMMs, runs the prior chain there and joins before it returns, so on return all work is ordered on `stream`. */
int64_t acvae_decode_saved_bytes(int N, int Tc, int S, int E, int H, int A, int V, int Eenc);
int64_t acvae_decode_scratch_bytes(int N, int Tc, int S, int E, int H, int A, int V, int Eenc);
int acvae_decode_fwd(const void* const* params, const float* mem_in, const int64_t* mem_lens, const int64_t* caps,
                     int64_t ld_caps, const int64_t* lens1, const float* q_z, const float* eps_p,
                     const int* ss_flags_host, const int* dis_flags_host, float* logits, float* outputs, int64_t* seqs,
                     float* sampled_logprobs, float* attn_w, float* p_means, float* p_logs, float* p_z,
                     float* p_means_utt, float* h_final, float* hp_final, float* cp_final, void* saved,
                     int64_t saved_bytes, void* scratch, int64_t scratch_bytes, int N, int Tc, int S, int E, int H,
                     int A, int V, int Eenc, int start_idx, int end_idx, void* stream, void* aux_stream, int flags);
/* The same with sample_next_word's method (models/word_model.py:173-207) chosen by the caller: ACVAE_SAMPLE_GREEDY
 * (= acvae_decode_fwd), or GUMBEL / MULTINOMIAL with `temp` and `sample_noise` [Tc,N,V] (see acvae_sample_next_word; the
 * per-step draws of the reference in step order).  seqs / sampled_logprobs then hold the sampled words; with
 * scheduled sampling or in inference the sampled word of step t-1 is the input of step t (vae_model.py:829-832).
 * emb_keep (may be NULL): keep mask uint8 [Tc,N,E] of the decoder's word-embedding nn.Dropout(emb_drop_p)
 * (models/decoder.py:33,184), drawn by the host on the CPU generator after each step's prior noise; kept rows are
 * scaled by 1/(1-p).  The same mask and p go to acvae_decode_bwd. */
int acvae_decode_fwd_sampled(const void* const* params, const float* mem_in, const int64_t* mem_lens,
                             const int64_t* caps, int64_t ld_caps, const int64_t* lens1, const float* q_z,
                             const float* eps_p, const int* ss_flags_host, const int* dis_flags_host, float* logits,
                             float* outputs, int64_t* seqs, float* sampled_logprobs, float* attn_w, float* p_means,
                             float* p_logs, float* p_z, float* p_means_utt, float* h_final, float* hp_final,
                             float* cp_final, void* saved, int64_t saved_bytes, void* scratch, int64_t scratch_bytes,
                             int N, int Tc, int S, int E, int H, int A, int V, int Eenc, int start_idx, int end_idx,
                             void* stream, void* aux_stream, int sample_method, float temp, const float* sample_noise,
                             const uint8_t* emb_keep, float emb_drop_p, int flags);
/* Backward for upstream gradients of logits / outputs / p_means / p_logs / p_z / p_means_utt (each may be
 * NULL).  Writes every decoder / pnet / mean_log_out / ln gradient, d_mem_in [N,S,Eenc] and d_q_z [N,Tc,E].
 * Stream contract: d_mem_in and d_q_z are ordered on `stream` when the call returns.  With ACVAE_FLAG_DEFER_PARAM_GRADS, when
 * acvae_decode_bwd_defers says 1 for the same flags / streams (a second stream is given and no step fed the prior's z to the
 * decoder), everything else - the parameter gradients - is queued on `aux_stream` behind
 * the call, so that it runs beside whatever `stream` does next (the posterior's and the encoder's backward): the caller joins
 * aux_stream before those gradients are read on another stream, and keeps saved / scratch / outputs / mem_in / the upstream
 * gradients untouched until aux_stream has drained (mem_in: without an ln projection the products read it in place).  Without the flag (0) everything is ordered on `stream` on return.  Hybrid_VAEModel, which joins
 * the second stream at the end of the backward pass, passes it unless ACVAE_DECODE_DEFER=0 (-0.07 ms per step on the
 * reference configuration). */
int acvae_decode_bwd_defers(const int* dis_flags_host, int Tc, void* stream, void* aux_stream, int flags);
int acvae_decode_bwd(const void* const* params, void* const* grads, const float* mem_in, const int64_t* mem_lens,
                     const int64_t* lens1, const float* eps_p, const int* dis_flags_host, const float* outputs,
                     const float* attn_w, const float* p_logs, const float* d_logits, const float* d_outputs_ext,
                     const float* d_p_means, const float* d_p_logs, const float* d_p_z, const float* d_p_means_utt,
                     float* d_mem_in, float* d_q_z, void* saved, int64_t saved_bytes, void* scratch,
                     int64_t scratch_bytes, int N, int Tc, int S, int E, int H, int A, int V, int Eenc, void* stream,
                     void* aux_stream, const uint8_t* emb_keep, float emb_drop_p, int flags);
/* float caption ids (collate pads with torch.zeros -> float32, caption_dataset.py:293) -> int64 */
int acvae_caps_to_long(const float* caps, int64_t* out, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A10  runners/pytorch_runner_vae.py:321-324: torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step over
 * one flat fp32 buffer.  acvae_grad_norm writes ||grad_scale * g||_2 to a device scalar (grad_scale = 1/world
 * after a SUM all-reduce); acvae_adam_step applies coef = min(1, max_grad_norm / (norm + 1e-6)) (skipped when
 * total_norm is NULL or max_grad_norm <= 0) and the bias-corrected Adam update in one pass.
 * ------------------------------------------------------------------------------------------- */
int64_t acvae_grad_norm_partials(void);
int acvae_grad_norm(const float* grads, int64_t n, float grad_scale, float* partials, float* out_norm, void* stream);
int acvae_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                    float max_grad_norm, const float* total_norm, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Opt-in kernel timing (bench.py's live roofline figure): while enabled, the conv launches are bracketed
 * by HIP events on their stream; acvae_prof_read waits for them and returns the summed duration and the
 * launch count of a tag, then clears it.  Tags: 0 = conv3x3 implicit GEMM (forward + data gradient),
 * 1 = conv3x3 weight gradient.  This ring is the only mutable state in the library and exists only
 * while enabled.  total_ms_host / launches_host are HOST pointers.
 * ------------------------------------------------------------------------------------------- */
int acvae_prof_enable(int enable);
int acvae_prof_pause(int paused);   /* stop / resume marking, keeping what was recorded (sampled measurement) */
int acvae_prof_read(int tag, double* total_ms_host, int64_t* launches_host);

/* ---------------------------------------------------------------------------------------------
 * Single decode steps (inference, no gradients): the per-call API of the reference's sub-modules,
 *   PriorRNN.forward(word, enc_mem, hiddens_state, last_z, lens)            models/text_encoder.py:247-268
 *   VAERNNBahdanauAttnDecoder.forward(word=, state=, enc_mem=, enc_mem_lens=, z=)   models/decoder.py:175-203
 * as used by the validation beam search (models/vae_model.py:896-995).  encproj_* may be NULL
 * (computed into scratch) or the result of acvae_attn_precompute (hoisted out of the step loop).  word int64 [N].
 * acvae_logprob_add / acvae_topk_flat are the beam bookkeeping of vae_model.py:909-916 (log_softmax + running beam
 * score; flat top-k over beam*V with idx / V and idx % V).
 * ------------------------------------------------------------------------------------------- */
int64_t acvae_step_scratch_bytes(int N, int S, int E, int H, int A, int V);
int acvae_attn_precompute(const void* const* params, int which, const float* mem, float* encproj, int N, int S, int E,
                          int H, int A, void* stream);
int acvae_prior_step_fwd(const void* const* params, const int64_t* word, const float* mem, const int64_t* mem_lens,
                         const float* encproj_p, const float* h_prev, const float* c_prev, const float* last_z,
                         const float* eps, float* mean, float* logv, float* z, float* h_out, float* c_out, float* attw,
                         void* scratch, int64_t scratch_bytes, int N, int S, int E, int V, void* stream);
int acvae_decoder_step_fwd(const void* const* params, const int64_t* word, const float* h_prev, const float* mem,
                           const int64_t* mem_lens, const float* encproj_d, const float* z, float* logits, float* h_out,
                           float* attw, float* rnn_input, void* scratch, int64_t scratch_bytes, int N, int S, int E,
                           int H, int A, int V, void* stream);
int acvae_logprob_add(const float* logits, int64_t ld, const float* lse, const float* prev, float* out, int N, int V,
                      void* stream);
int acvae_topk_flat(const float* x, int64_t n, int k, int V, float* vals, int64_t* idx, int64_t* row, int64_t* col,
                    void* stream);
/* All clips of a batch at once: group g searches x[g * group_stride .. + n) (n = beam*V scores of one clip, or V: only
 * its first beam row), outputs [groups][k]; `row` comes back as group * row_base + idx / V, i.e. as an index into the
 * batch's beam rows when row_base = beam. */
int acvae_topk_flat_batched(const float* x, int64_t n, int64_t group_stride, int k, int V, float* vals, int64_t* idx,
                            int64_t* row, int64_t* col, int groups, int row_base, void* stream);
/* Validation beam search as one call: Hybrid_VAEModel.beam_search, models/vae_model.py:896-995, for
 * all N clips at once.  mem [N,S,E] (after the optional `ln` projection), mem_lens [N], eps [max_length][N*beam][E]
 * (the N(0,1) draws of PriorRNN.forward, text_encoder.py:259, step-major), start_idx = vocabulary index of <start>.
 * Outputs, as the reference keeps them (beam 0 of each clip, :990-995): seqs int64 [N,max_length],
 * attn_weights [N,S,max_length].  No host synchronisation; scratch of acvae_beam_search_scratch_bytes(). */
int64_t acvae_beam_search_scratch_bytes(int N, int beam, int max_length, int S, int E, int H, int A, int V);
int acvae_beam_search(const void* const* params, const float* mem, const int64_t* mem_lens, const float* eps,
                      int64_t start_idx, int64_t* seqs, float* attn_weights, void* scratch, int64_t scratch_bytes, int N,
                      int beam, int max_length, int S, int E, int H, int A, int V, void* stream);
/* Diverse beam search, models/word_model.py:344-348 with add_diversity :298-312: per beam row
 *   out[n,c] = log_softmax(log_softmax(logits[n]) / temperature)[c] - diversity_lambda * counts[c] + prev[n]
 * counts (may be NULL: first group) = how often the earlier groups chose word c at this local step: one vector [V] for
 * all rows (rows_per_count = 0) or one per clip, [N / rows_per_count][V]; prev [N] (may be NULL) the running beam
 * log-probabilities. */
int acvae_dbs_scores(const float* logits, int64_t ld, float temperature, const float* counts, float diversity_lambda,
                     const float* prev, float* out, int N, int V, int rows_per_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACVAE_HIP_H */
