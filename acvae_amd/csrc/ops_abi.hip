// Per-op C ABI: the kernels the composite drivers (encoder.hip, decoder.hip) are made of, each behind an entry point of
// its own so that it can be called - and tested - alone (SURVEY §8(b)'s operator list: conv3x3[_bn], bn, gru_step,
// lstm_step, bigru_seq).  Thin wrappers: argument checks, workspace carving, then the same internal launchers the
// drivers use; nothing here is a second implementation.
#include "common.h"
#include "conv.h"
#include "rnn.h"
#include "../../include/acvae_hip.h"

namespace {
inline long al64(long x) { return (x + 63) & ~63L; }

struct ConvWs {
  long wp, partials, dpart, slab, total;   // float offsets
};
ConvWs conv_ws(int N, int H, int W, int Cin, int Cout) {
  ConvWs w;
  long o = 0;
  w.wp = o; o = al64(o + (long)Cout * 16 * Cin);     // 9: repacked taps; 16: Winograd positions
  long part = Cin == 1 ? (long)acvae::conv1_first_blocks(N, H) * 128 : (long)acvae::conv_partials_rows(N, H, W) * 2 * Cout;
  if (Cin > 1 && W >= 4 && W % 2 == 0 && (long)acvae::conv_wino_partials_rows(N, H, W) * 2 * Cout > part)
    part = (long)acvae::conv_wino_partials_rows(N, H, W) * 2 * Cout;
  w.partials = o; o = al64(o + part);
  w.dpart = o; o = al64(o + 2 * acvae::colsum_scratch_doubles(2 * Cout > 1024 ? 2 * Cout : 1024));
  long slab = Cin == 1 ? (long)acvae::conv1_first_blocks(N, H) * (576 + 128)
                       : acvae::conv3x3_wgrad_slab_floats(N, H, W, Cin, Cout);
  if (Cin % 64 == 0) {                              // the bf16 forms share this workspace
    const long sb = acvae::conv3x3_wgrad_bf16_slab_floats(N, H, W, Cin, Cout);
    if (sb > slab) slab = sb;
  }
  if (Cin > 1) {
    const long sw = acvae::conv3x3_wino_wgrad_slab_floats(N, H, W, Cin, Cout);
    if (sw > slab) slab = sw;
  }
  w.slab = o; o = al64(o + slab);
  w.total = o;
  return w;
}
bool conv_dims_ok(int N, int H, int W, int Cin, int Cout) {
  return N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (long)N * H * W < (1L << 31);
}
}  // namespace

// ------------------------------------------------------------------------------------------ conv3x3
extern "C" int64_t acvae_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  if (!conv_dims_ok(N, H, W, Cin, Cout)) return -1;
  return conv_ws(N, H, W, Cin, Cout).total * 4;
}

extern "C" int acvae_conv3x3_fwd(const float* X, const float* W_oihw, const float* in_scale, const float* in_shift,
                                 float* Y, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, int64_t* num_batches_tracked, int training, float* bn_out,
                                 void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !X || !W_oihw || !Y || !ws_v) return ACVAE_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return ACVAE_EINVAL;
  if (bn_out && (!gamma || !beta || !running_mean || !running_var)) return ACVAE_EINVAL;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  if (!aligned16(ws_v)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  float* partials = (bn_out && training) ? ws + L.partials : nullptr;
  int nparts;
  if (Cin == 1) {
    // the first convolution of the stack (models/encoder.py:656 conv_block1.conv1): direct kernel, its operand transform
    // is bn0's per-MEL affine (in_scale / in_shift are [W] here), F = W = 64 and 64 output channels
    if (W != 64 || Cout != 64 || !in_scale) return ACVAE_EUNSUPPORTED;
    ACVAE_TRY(acvae::conv1_first_fwd(X, in_scale, in_shift, W_oihw, Y, partials, N, H, W, st));
    nparts = acvae::conv1_first_blocks(N, H);
  } else {
    ACVAE_TRY(acvae::repack_weights<float>(W_oihw, ws + L.wp, nullptr, Cout, Cin, st));
    ACVAE_TRY(acvae::conv3x3_igemm(X, in_scale, in_shift, ws + L.wp, Y, partials, N, H, W, Cin, Cout, st));
    nparts = acvae::conv_partials_rows(N, H, W);
  }
  if (bn_out) ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + L.dpart), st));
  if (bn_out)
    ACVAE_TRY(acvae::bn_finalize(partials, nparts, Cout, (double)N * H * W, gamma, beta, running_mean, running_var,
                                 num_batches_tracked, training, bn_out, bn_out + Cout, bn_out + 2 * Cout,
                                 bn_out + 3 * Cout, (double*)(ws + L.dpart), st));
  return ACVAE_OK;
}

extern "C" int acvae_conv3x3_dgrad(const float* dY, const float* W_oihw, float* dX, void* ws_v, int64_t ws_bytes, int N,
                                   int H, int W, int Cin, int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !W_oihw || !dX || !ws_v) return ACVAE_EINVAL;
  if (Cin == 1) return ACVAE_EUNSUPPORTED;       // the path never needs d(features)
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  ACVAE_TRY(acvae::repack_weights<float>(W_oihw, nullptr, ws + L.wp, Cout, Cin, st));
  // the data gradient is the same implicit GEMM on dY with the flipped, transposed weights: Cout plays Cin
  return acvae::conv3x3_igemm(dY, nullptr, nullptr, ws + L.wp, dX, nullptr, N, H, W, Cout, Cin, st);
}

extern "C" int acvae_conv3x3_wgrad(const float* dY, const float* X, const float* in_scale, const float* in_shift,
                                   float* dW_oihw, void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout,
                                   void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !X || !dW_oihw || !ws_v) return ACVAE_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return ACVAE_EINVAL;
  if (Cin == 1) return ACVAE_EUNSUPPORTED;       // acvae_conv1_first_bwd
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  return acvae::conv3x3_wgrad(dY, X, in_scale, in_shift, dW_oihw, (float*)ws_v + L.slab, N, H, W, Cin, Cout,
                              (hipStream_t)stream);
}

// ---- Winograd F(2x2,3x3)
extern "C" int acvae_conv3x3_fwd_wino(const float* X, const float* W_oihw, const float* in_scale, const float* in_shift,
                                      float* Y, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, int training, float* bn_out,
                                      void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !X || !W_oihw || !Y || !ws_v) return ACVAE_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return ACVAE_EINVAL;
  if (bn_out && (!gamma || !beta || !running_mean || !running_var)) return ACVAE_EINVAL;
  if (!acvae::conv3x3_wino_ok(H, W, Cin, Cout)) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  if (!aligned16(ws_v)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  float* partials = (bn_out && training) ? ws + L.partials : nullptr;
  ACVAE_TRY(acvae::conv3x3_wino_weights(W_oihw, ws + L.wp, Cout, Cin, false, st));
  ACVAE_TRY(acvae::conv3x3_wino(X, in_scale, in_shift, ws + L.wp, Y, partials, N, H, W, Cin, Cout, st));
  if (bn_out) ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + L.dpart), st));
  if (bn_out)
    ACVAE_TRY(acvae::bn_finalize(partials, acvae::conv_wino_partials_rows(N, H, W), Cout, (double)N * H * W, gamma, beta,
                                 running_mean, running_var, num_batches_tracked, training, bn_out, bn_out + Cout,
                                 bn_out + 2 * Cout, bn_out + 3 * Cout, (double*)(ws + L.dpart), st));
  return ACVAE_OK;
}

extern "C" int acvae_conv3x3_dgrad_wino(const float* dY, const float* W_oihw, float* dX, void* ws_v, int64_t ws_bytes, int N,
                                        int H, int W, int Cin, int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !W_oihw || !dX || !ws_v) return ACVAE_EINVAL;
  if (!acvae::conv3x3_wino_ok(H, W, Cout, Cin)) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  if (!aligned16(ws_v)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  ACVAE_TRY(acvae::conv3x3_wino_weights(W_oihw, ws + L.wp, Cout, Cin, true, st));
  return acvae::conv3x3_wino(dY, nullptr, nullptr, ws + L.wp, dX, nullptr, N, H, W, Cout, Cin, st);
}

// the data gradient that also reduces the BatchNorm + ReLU backward its output feeds (conv_wino_bnred_kernel): sum_g = sum of the
// masked gradient (= d beta), sum_gy = sum of masked gradient x normalised activation (= d gamma), as acvae_bn_relu_bwd's first pass
extern "C" int acvae_conv3x3_dgrad_bnred_wino(const float* dY, const float* W_oihw, float* dX, const float* Yprev, const float* bn_prev,
                                              float* sum_g, float* sum_gy, void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin,
                                              int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !W_oihw || !dX || !Yprev || !bn_prev || !sum_g || !sum_gy || !ws_v) return ACVAE_EINVAL;
  if (!acvae::conv3x3_wino_ok(H, W, Cout, Cin)) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  const ConvWs Lr = conv_ws(N, H, W, Cout, Cin);          // the partial sums have Cin columns (the data gradient's outputs)
  if (ws_bytes < L.total * 4 || ws_bytes < Lr.total * 4) return ACVAE_EWORKSPACE;
  if (!aligned16(ws_v)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + Lr.dpart), st));
  ACVAE_TRY(acvae::conv3x3_wino_weights(W_oihw, ws + L.wp, Cout, Cin, true, st));
  const acvae::WinoBnReduce red{Yprev, bn_prev, bn_prev + Cin, bn_prev + 2 * Cin, bn_prev + 3 * Cin};     // scale | shift | mean | invstd
  ACVAE_TRY(acvae::conv3x3_wino(dY, nullptr, nullptr, ws + L.wp, dX, ws + Lr.partials, N, H, W, Cout, Cin, st, &red));
  return acvae::colsum2(ws + Lr.partials, acvae::conv_wino_partials_rows(N, H, W), 2 * Cin, (double*)(ws + Lr.dpart), sum_g, sum_gy, Cin, st);
}

extern "C" int acvae_conv3x3_wgrad_wino(const float* dY, const float* X, const float* in_scale, const float* in_shift,
                                        float* dW_oihw, void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout,
                                        void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !X || !dW_oihw || !ws_v) return ACVAE_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return ACVAE_EINVAL;
  if (!acvae::conv3x3_wino_wgrad_ok(H, W, Cin, Cout)) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  if (!aligned16(ws_v)) return ACVAE_EALIGN;
  return acvae::conv3x3_wino_wgrad(dY, X, in_scale, in_shift, dW_oihw, (float*)ws_v + L.slab, N, H, W, Cin, Cout,
                                   (hipStream_t)stream);
}

// ---- bf16 storage
extern "C" int acvae_conv3x3_fwd_bf16(const void* X, const float* W_oihw, const float* in_scale, const float* in_shift,
                                      void* Y, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, int training, float* bn_out,
                                      void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !X || !W_oihw || !Y || !ws_v) return ACVAE_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return ACVAE_EINVAL;
  if (bn_out && (!gamma || !beta || !running_mean || !running_var)) return ACVAE_EINVAL;
  if (Cin % 64 != 0) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  if (!aligned16(ws_v)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  float* partials = (bn_out && training) ? ws + L.partials : nullptr;
  bf16_t* wp = (bf16_t*)(ws + L.wp);
  ACVAE_TRY(acvae::repack_weights<bf16_t>(W_oihw, wp, nullptr, Cout, Cin, st));
  ACVAE_TRY(acvae::conv3x3_igemm_bf16((const bf16_t*)X, in_scale, in_shift, wp, (bf16_t*)Y, partials, N, H, W, Cin, Cout, st));
  if (bn_out) ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + L.dpart), st));
  if (bn_out)
    ACVAE_TRY(acvae::bn_finalize(partials, acvae::conv_partials_rows(N, H, W), Cout, (double)N * H * W, gamma, beta,
                                 running_mean, running_var, num_batches_tracked, training, bn_out, bn_out + Cout,
                                 bn_out + 2 * Cout, bn_out + 3 * Cout, (double*)(ws + L.dpart), st));
  return ACVAE_OK;
}

extern "C" int acvae_conv3x3_dgrad_bf16(const void* dY, const float* W_oihw, void* dX, void* ws_v, int64_t ws_bytes, int N,
                                        int H, int W, int Cin, int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !W_oihw || !dX || !ws_v) return ACVAE_EINVAL;
  if (Cin % 64 != 0 || Cout % 64 != 0) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wd = (bf16_t*)((float*)ws_v + L.wp);
  ACVAE_TRY(acvae::repack_weights<bf16_t>(W_oihw, nullptr, wd, Cout, Cin, st));
  return acvae::conv3x3_igemm_bf16((const bf16_t*)dY, nullptr, nullptr, wd, (bf16_t*)dX, nullptr, N, H, W, Cout, Cin, st);
}

extern "C" int acvae_conv3x3_wgrad_bf16(const void* dY, const void* X, const float* in_scale, const float* in_shift,
                                        float* dW_oihw, void* ws_v, int64_t ws_bytes, int N, int H, int W, int Cin,
                                        int Cout, void* stream) {
  if (!conv_dims_ok(N, H, W, Cin, Cout) || !dY || !X || !dW_oihw || !ws_v) return ACVAE_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return ACVAE_EINVAL;
  if (Cin % 64 != 0) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, H, W, Cin, Cout);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  return acvae::conv3x3_wgrad((const bf16_t*)dY, (const bf16_t*)X, in_scale, in_shift, dW_oihw, (float*)ws_v + L.slab, N, H,
                              W, Cin, Cout, (hipStream_t)stream);
}

extern "C" int acvae_conv1_first_bwd(const float* x, const float* bn0, const float* W1_oihw, const float* dY, float* dW1,
                                     float* dgamma0, float* dbeta0, void* ws_v, int64_t ws_bytes, int N, int T, int F,
                                     void* stream) {
  if (N <= 0 || T <= 0 || !x || !bn0 || !W1_oihw || !dY || !dW1 || !dgamma0 || !dbeta0 || !ws_v) return ACVAE_EINVAL;
  if (F != 64) return ACVAE_EUNSUPPORTED;
  const ConvWs L = conv_ws(N, T, F, 1, 64);
  if (ws_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  float* ws = (float*)ws_v;
  const long nb = acvae::conv1_first_blocks(N, T);
  ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + L.dpart), (hipStream_t)stream));
  return acvae::conv1_first_bwd(x, bn0, bn0 + 64, bn0 + 128, bn0 + 192, W1_oihw, dY, ws + L.slab, ws + L.slab + nb * 576,
                                dW1, dgamma0, dbeta0, (double*)(ws + L.dpart), N, T, F, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ BatchNorm pieces
extern "C" int64_t acvae_bn_workspace_bytes(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return -1;
  const long part = (long)acvae::bn_bwd_blocks(N, H, W, C) * 2 * C;
  const long p0 = (long)acvae::bn0_partials_rows((long)N * H) * 128;
  return (al64(part > p0 ? part : p0) + al64(2 * acvae::colsum_scratch_doubles(2 * C > 1024 ? 2 * C : 1024))) * 4;
}

extern "C" int acvae_bn_mel_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int64_t* num_batches_tracked, int training, float* bn_out,
                                void* ws_v, int64_t ws_bytes, int64_t rows, int F, void* stream) {
  if (rows <= 0 || !x || !gamma || !beta || !running_mean || !running_var || !bn_out || !ws_v) return ACVAE_EINVAL;
  if (F != 64) return ACVAE_EUNSUPPORTED;
  const long part = al64((long)acvae::bn0_partials_rows(rows) * 128);
  if (ws_bytes < (part + al64(2 * acvae::colsum_scratch_doubles(1024))) * 4) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  int nparts = 0;
  if (training) ACVAE_TRY(acvae::bn0_stats(x, ws, rows, F, &nparts, st));
  ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + part), st));
  return acvae::bn_finalize(ws, nparts, F, (double)rows, gamma, beta, running_mean, running_var, num_batches_tracked,
                            training, bn_out, bn_out + F, bn_out + 2 * F, bn_out + 3 * F, (double*)(ws + part), st);
}

extern "C" int acvae_bn_relu_pool_fwd(const float* Y, const float* bn, float* P, int N, int H, int W, int C, int pool,
                                      float p_drop, uint64_t seed, int site, const uint8_t* keep_mask, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !Y || !bn || !P) return ACVAE_EINVAL;
  DropoutSpec d{p_drop, keep_mask, seed, (uint32_t)site};
  return acvae::bn_relu_pool(Y, bn, bn + C, P, N, H, W, C, d, (hipStream_t)stream, pool != 0);
}

extern "C" int acvae_bn_relu_bwd(const float* Y, const float* dO, int upstream, const float* bn, float* dgamma,
                                 float* dbeta, float* dY, void* ws_v, int64_t ws_bytes, int N, int H, int W, int C,
                                 int training, float p_drop, uint64_t seed, int site, const uint8_t* keep_mask,
                                 void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !Y || !dO || !bn || !dgamma || !dbeta || !dY || !ws_v) return ACVAE_EINVAL;
  if (upstream != UP_PLAIN && upstream != UP_POOL && upstream != UP_DROP) return ACVAE_EINVAL;
  if (ws_bytes < acvae_bn_workspace_bytes(N, H, W, C)) return ACVAE_EWORKSPACE;
  float* ws = (float*)ws_v;
  const long part = al64((long)acvae::bn_bwd_blocks(N, H, W, C) * 2 * C);
  const long p0 = al64((long)acvae::bn0_partials_rows((long)N * H) * 128);
  DropoutSpec d{p_drop, keep_mask, seed, (uint32_t)site};
  ACVAE_TRY(acvae::colsum_tickets_reset((double*)(ws + (part > p0 ? part : p0)), (hipStream_t)stream));
  // sum_g (= dbeta) and sum_gy (= dgamma) are also inputs of the apply pass: written first, then read
  return acvae::bn_bwd(Y, dO, upstream, bn, bn + C, bn + 2 * C, bn + 3 * C, ws, dbeta, dgamma, dY,
                       (double*)(ws + (part > p0 ? part : p0)), N, H, W, C, d, (hipStream_t)stream, training != 0);
}

// ------------------------------------------------------------------------------------------ recurrent cells
extern "C" int64_t acvae_rnn_workspace_bytes(int N, int Tc, int I, int H) {
  if (N <= 0 || Tc <= 0 || I <= 0 || H <= 0) return -1;
  const long R = (long)N * Tc;
  // skinny split-K workspace | gi [R,4H] (both directions of a BiGRU: 2 x 3H <= 8H) | gh [N,4H] | h [N,H] | scratch rows
  return (al64(acvae_skinny_ws_floats()) + al64(R * 8 * H) + al64((long)N * 4 * H) + al64((long)N * H) +
          al64(R * 5 * H) + al64(R * H)) * 4;
}

extern "C" int acvae_gru_step(const float* x, const float* h, const float* w_ih, const float* w_hh, const float* b_ih,
                              const float* b_hh, float* h_out, void* ws_v, int64_t ws_bytes, int N, int I, int H,
                              void* stream) {
  if (N <= 0 || I <= 0 || H <= 0 || !x || !h || !w_ih || !w_hh || !h_out || !ws_v) return ACVAE_EINVAL;
  if (ws_bytes < acvae_rnn_workspace_bytes(N, 1, I, H)) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  float* sk = ws;
  float* gi = ws + al64(acvae_skinny_ws_floats());
  float* gh = gi + al64((long)N * 8 * H);
  ACVAE_TRY(acvae_skinny_ws_reset(sk, st));
  ACVAE_TRY(acvae_gemm_nt_dual(x, I, w_ih, I, I, nullptr, 0, nullptr, 0, 0, b_ih, gi, 3 * H, N, 3 * H, 0, st, sk));
  ACVAE_TRY(acvae_gemm_nt_dual(h, H, w_hh, H, H, nullptr, 0, nullptr, 0, 0, b_hh, gh, 3 * H, N, 3 * H, 0, st, sk));
  return acvae::gru_fwd(gi, 3 * H, gh, 3 * H, h, H, h_out, H, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, N, H, st);
}

extern "C" int acvae_lstm_step(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                               const float* b_ih, const float* b_hh, float* h_out, float* c_out, void* ws_v,
                               int64_t ws_bytes, int N, int I, int H, void* stream) {
  if (N <= 0 || I <= 0 || H <= 0 || !x || !h || !c || !w_ih || !w_hh || !h_out || !c_out || !ws_v) return ACVAE_EINVAL;
  if (ws_bytes < acvae_rnn_workspace_bytes(N, 1, I, H)) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)ws_v;
  float* sk = ws;
  float* gates = ws + al64(acvae_skinny_ws_floats());
  ACVAE_TRY(acvae_skinny_ws_reset(sk, st));
  // gates = x . W_ih^T + b_ih + h . W_hh^T + b_hh   (torch.nn.LSTM; the decode loop hoists the first product)
  ACVAE_TRY(acvae_gemm_nt_dual(x, I, w_ih, I, I, nullptr, 0, nullptr, 0, 0, b_ih, gates, 4 * H, N, 4 * H, 0, st, sk));
  ACVAE_TRY(acvae_gemm_nt_dual(h, H, w_hh, H, H, nullptr, 0, nullptr, 0, 0, b_hh, gates, 4 * H, N, 4 * H, 1, st, sk));
  return acvae::lstm_fwd(gates, 4 * H, c, H, h_out, H, c_out, H, nullptr, 0, N, H, st);
}

extern "C" int acvae_bigru_seq(const float* X, const int64_t* lens, const void* const* w, float* hidden, void* ws_v,
                               int64_t ws_bytes, int N, int Tc, int E, int H, void* stream) {
  if (N <= 0 || Tc <= 0 || E <= 0 || H <= 0 || !X || !lens || !w || !hidden || !ws_v) return ACVAE_EINVAL;
  for (int i = 0; i < 8; ++i)
    if (!w[i]) return ACVAE_EINVAL;
  if (ws_bytes < acvae_rnn_workspace_bytes(N, Tc, E, H)) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const long R = (long)N * Tc;
  float* ws = (float*)ws_v;
  float* sk = ws;
  float* gi = ws + al64(acvae_skinny_ws_floats());
  float* gh = gi + al64(R * 8 * H);
  float* hcur = gh + al64((long)N * 4 * H);
  float* save = hcur + al64((long)N * H);
  float* hps = save + al64(R * 5 * H);
  ACVAE_TRY(acvae_skinny_ws_reset(sk, st));
  for (int dir = 0; dir < 2; ++dir) {
    const float* wih = (const float*)w[dir * 4 + 0];
    const float* whh = (const float*)w[dir * 4 + 1];
    const float* bih = (const float*)w[dir * 4 + 2];
    const float* bhh = (const float*)w[dir * 4 + 3];
    ACVAE_TRY(acvae_gemm_nt_dual(X, E, wih, E, E, nullptr, 0, nullptr, 0, 0, bih, gi, 3 * H, (int)R, 3 * H, 0, st, sk));
    ACVAE_TRY(acvae::copy_rows(hcur, H, nullptr, 0, N, H, st));
    for (int k = 0; k < Tc; ++k) {
      const int t = dir ? Tc - 1 - k : k;
      ACVAE_TRY(acvae_gemm_nt_dual(hcur, H, whh, H, H, nullptr, 0, nullptr, 0, 0, bhh, gh, 3 * H, N, 3 * H, 0, st, sk));
      ACVAE_TRY(acvae::gru_fwd(gi + (long)t * 3 * H, (long)Tc * 3 * H, gh, 3 * H, hcur, H, hcur, H,
                               hidden + (long)t * 2 * H + dir * H, (long)Tc * 2 * H, save + (long)t * 4 * H,
                               (long)Tc * 4 * H, hps + (long)t * H, (long)Tc * H, lens, t, N, H, st));
    }
  }
  return ACVAE_OK;
}

// ------------------------------------------------------------------------------------------ column sums
// out[c] = sum_r x[r][c] for a contiguous [rows, cols] fp32 matrix (bias gradients), fixed-order fp64 combine.
extern "C" int64_t acvae_colsum_workspace_bytes(int cols) {
  return cols > 0 ? al64(2 * acvae::colsum_scratch_doubles(cols)) * 4 : -1;
}
extern "C" int acvae_colsum(const float* x, int rows, int cols, float* out, void* ws, int64_t ws_bytes, void* stream) {
  if (!x || !out || !ws || rows <= 0 || cols <= 0) return ACVAE_EINVAL;
  if (ws_bytes < acvae_colsum_workspace_bytes(cols)) return ACVAE_EWORKSPACE;
  ACVAE_TRY(acvae::colsum_tickets_reset((double*)ws, (hipStream_t)stream));
  return acvae::colsum2(x, rows, cols, (double*)ws, out, nullptr, 0, (hipStream_t)stream);
}
