// Internal C++ launcher API of conv.hip (used by the composite encoder driver and the per-op C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

struct DropoutSpec {
  float p;              // drop probability; 0 disables
  const uint8_t* mask;  // optional explicit keep-mask in the reference's NCHW / [N,C] order (parity tests)
  uint64_t seed;        // Philox key when mask == nullptr
  uint32_t site;        // dropout call-site id in forward order (0..5 for Cnn10, 0..7 for Cnn14_16k)
};

// upstream of a BN+ReLU output in the backward: UP_PLAIN = dO as is; UP_POOL = through dropout + 2x2 average pool;
// UP_DROP = through dropout only (pool_size (1,1))
enum { UP_PLAIN = 0, UP_POOL = 1, UP_DROP = 2 };

namespace acvae {
int conv3x3_igemm(const float* X, const float* scale, const float* shift, const float* Wp, float* Y, float* partials,
                  int N, int H, int W, int Cin, int Cout, hipStream_t st);
int conv_partials_rows(int N, int H, int W);
// bf16 storage (conv_bf16.hip): same contracts, X / Wp / Y in bf16, fp32 accumulation, statistics of the rounded output
int conv3x3_igemm_bf16(const bf16_t* X, const float* scale, const float* shift, const bf16_t* Wp, bf16_t* Y,
                       float* partials, int N, int H, int W, int Cin, int Cout, hipStream_t st);
long conv3x3_wgrad_bf16_slab_floats(int N, int H, int W, int Cin, int Cout);
int conv3x3_wgrad_bf16_splits(int N, int H, int W, int Cin, int Cout);
int conv3x3_wgrad_bf16_launch(const bf16_t* dY, const bf16_t* X, const float* scale, const float* shift, float* slab, int N,
                              int H, int W, int Cin, int Cout, hipStream_t st);
// dW_oihw[co][ci][tap] = sum_z slab[z][co][tap*Cin + ci]  (fixed order)
int wgrad_reduce(const float* slab, int nsplit, float* dW_oihw, int Cout, int Cin, hipStream_t st);
long conv3x3_wgrad_slab_floats(int N, int H, int W, int Cin, int Cout);
int conv3x3_wgrad(const float* dY, const float* X, const float* scale, const float* shift, float* dW_oihw, float* slab,
                  int N, int H, int W, int Cin, int Cout, hipStream_t st);
// Winograd F(2x2,3x3) forward / data gradient (conv_wino.hip), fp32.  U = conv3x3_wino_weights image (16*Cin*Cout floats)
bool conv3x3_wino_ok(int H, int W, int Cin, int Cout);
long conv3x3_wino_weight_floats(int Cin, int Cout);
int conv_wino_partials_rows(int N, int H, int W);
int conv3x3_wino_weights(const float* W_oihw, float* U, int Cout, int Cin, bool dgrad, hipStream_t st);
// several images in one launch (fill W, U, Cout, Cin, dgrad and n; start is computed)
struct WinoWeightsBatch {
  static constexpr int MAXL = 32;
  const float* W[MAXL]; float* U[MAXL]; int Cout[MAXL], Cin[MAXL], dgrad[MAXL]; long start[MAXL + 1]; int n = 0;
  void add(const float* w, float* u, int cout, int cin, bool dg) { W[n] = w; U[n] = u; Cout[n] = cout; Cin[n] = cin; dgrad[n] = dg ? 1 : 0; ++n; }
};
int conv3x3_wino_weights_batch(WinoWeightsBatch& b, hipStream_t st);
// red (data gradient only; partials = [conv_wino_partials_rows][2][Cout] then receives bn_bwd_reduce_kernel's sums): the launch's
// output is the upstream gradient of the BatchNorm + ReLU that normalised red->Y (same shape as the output)
struct WinoBnReduce { const float* Y; const float* scale; const float* shift; const float* mean; const float* invstd; };
int conv3x3_wino(const float* X, const float* scale, const float* shift, const float* U, float* Y, float* partials, int N,
                 int H, int W, int Cin, int Cout, hipStream_t st, const WinoBnReduce* red = nullptr);
bool conv3x3_wino_wgrad_ok(int H, int W, int Cin, int Cout);
long conv3x3_wino_wgrad_slab_floats(int N, int H, int W, int Cin, int Cout);     // 0 when the shape is not taken
int conv3x3_wino_wgrad(const float* dY, const float* X, const float* scale, const float* shift, float* dW_oihw, float* slab,
                       int N, int H, int W, int Cin, int Cout, hipStream_t st);
// T = float or bf16_t (storage type of activations / repacked weights; arithmetic is fp32 either way)
template <class T>
int repack_weights(const float* W_oihw, T* Wf, T* Wd, int Cout, int Cin, hipStream_t st);
// several repacks (implicit-GEMM weight layouts: forward [cout][tap][cin], data gradient [cin][8 - tap][cout]) in one launch
struct RepackBatch {
  static constexpr int MAXL = 32;
  const float* W[MAXL]; void* dst[MAXL]; int Cout[MAXL], Cin[MAXL], dgrad[MAXL]; long start[MAXL + 1]; int n = 0;
  void add(const float* w, void* d, int cout, int cin, bool dg) { W[n] = w; dst[n] = d; Cout[n] = cout; Cin[n] = cin; dgrad[n] = dg ? 1 : 0; ++n; }
};
template <class T>
int repack_weights_batch(RepackBatch& b, hipStream_t st);
// storage-type overloads so that the encoder driver is one template
inline int conv3x3_igemm(const bf16_t* X, const float* scale, const float* shift, const bf16_t* Wp, bf16_t* Y,
                         float* partials, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  return conv3x3_igemm_bf16(X, scale, shift, Wp, Y, partials, N, H, W, Cin, Cout, st);
}
inline int conv3x3_wgrad(const bf16_t* dY, const bf16_t* X, const float* scale, const float* shift, float* dW_oihw,
                         float* slab, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  int r = conv3x3_wgrad_bf16_launch(dY, X, scale, shift, slab, N, H, W, Cin, Cout, st);
  if (r != 0) return r;
  return wgrad_reduce(slab, conv3x3_wgrad_bf16_splits(N, H, W, Cin, Cout), dW_oihw, Cout, Cin, st);
}
template <class T>
inline long conv3x3_wgrad_slab_floats_t(int N, int H, int W, int Cin, int Cout) {
  return sizeof(T) == 2 ? conv3x3_wgrad_bf16_slab_floats(N, H, W, Cin, Cout) : conv3x3_wgrad_slab_floats(N, H, W, Cin, Cout);
}
int bn0_stats(const float* x, float* partials, long rows, int F, int* nparts, hipStream_t st);
int bn0_partials_rows(long rows);
int bn_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* beta,
                float* running_mean, float* running_var, int64_t* nbt, int training, float* scale, float* shift,
                float* mean, float* invstd, double* dpart, hipStream_t st);
long colsum_scratch_doubles(int width);
// zero the tickets at the head of a dpart scratch: once per composite call, in front of its first column sum / bn_finalize
int colsum_tickets_reset(double* dpart, hipStream_t st);
long colsum_ticket_words();             // ... the same tickets as a region for a ZeroBatch (rnn.h)
// deterministic column sums of x[P][width] (ld == width): out[i] = sum_p x[p][i]; entries >= split (if > 0) go to out2
int colsum2(const float* x, int P, int width, double* dpart, float* out, float* out2, int split, hipStream_t st);
// several column sums in ONE launch (the bias gradients of a group of linear layers: up to six 6-14 us launches in a row on the
// text side); job j: out[i] (and out_b[i], if given: the two LSTM biases share a gradient) = sum_p x[p][i].  Each job keeps the
// row groups and the order of additions of its own colsum2 launch (bit-identical); they share the tickets and the scratch of
// one dpart: falls back to separate launches when those do not hold them all.
struct ColsumBatch {
  static constexpr int MAXJ = 6;
  const float* x[MAXJ]; int P[MAXJ], width[MAXJ], R[MAXJ], blk0[MAXJ + 1]; long d0[MAXJ]; float* out[MAXJ]; float* out_b[MAXJ]; int n = 0;
  void add(const float* xs, int p, int w, float* o, float* ob = nullptr) {       // n > MAXJ: too many jobs (colsum_batch refuses)
    if (n < MAXJ) { x[n] = xs; P[n] = p; width[n] = w; out[n] = o; out_b[n] = ob; }
    ++n;
  }
};
int colsum_batch(ColsumBatch& b, double* dpart, long dpart_doubles, hipStream_t st);
int conv1_first_blocks(int N, int T);
template <class TY>
int conv1_first_fwd(const float* x, const float* scale0, const float* shift0, const float* W1, TY* Y,
                    float* partials, int N, int T, int F, hipStream_t st);
template <class TY>
int conv1_first_bwd(const float* x, const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                    const float* W1, const TY* dY, float* dw_part, float* bn_part, float* dW1, float* dgamma0,
                    float* dbeta0, double* dpart, int N, int T, int F, hipStream_t st);
// pool: 2x2 average pool after BN+ReLU (ConvBlock pool_size (2,2)); !pool: pool_size (1,1) (Cnn14's last block)
template <class T>
int bn_relu_pool(const T* Y, const float* scale, const float* shift, T* P, int N, int H, int W, int C,
                 DropoutSpec drop, hipStream_t st, bool pool = true);
int bn_bwd_blocks(int N, int H, int W, int C);
template <class T>
// ready_rows > 0 (plain upstream only): `partials` already holds that many rows of the reduction's sums (written by the data
// gradient that produced dO, conv3x3_wino with a WinoBnReduce): the reduction pass is skipped
int bn_bwd(const T* Y, const T* dO, int upstream, const float* scale, const float* shift, const float* mean,
           const float* invstd, float* partials, float* sum_g, float* sum_gy, T* dY, double* dpart, int N, int H,
           int W, int C, DropoutSpec drop, hipStream_t st, bool batch_stats = true, int ready_rows = 0);
template <class T>
int relu_mask(const T* Y, const float* scale, const float* shift, uint8_t* out, int N, int H, int W, int C, hipStream_t st);
template <class T>
int freq_mean(const T* P, float* out, long rows, int Fp, int C, hipStream_t st);
template <class T>
int freq_mean_bwd(const float* dae, T* dP, long rows, int Fp, int C, hipStream_t st);
int time_pool(const float* ae, float* out, int N, int S, int C, DropoutSpec drop, hipStream_t st);
int relu_dropout(float* x, int total, DropoutSpec drop, hipStream_t st);
}  // namespace acvae

// gemm.hip
int acvae_gemm_nt_dual(const float* A1, int64_t lda1, const float* B1, int64_t ldb1, int K1, const float* A2,
                       int64_t lda2, const float* B2, int64_t ldb2, int K2, const float* bias, float* C, int64_t ldc,
                       int M, int N, int accumulate, hipStream_t st, float* skws = nullptr);
// two independent products C0 = A0 . B0^T (+ bias0), C1 = A1 . B1^T (+ bias1) of M <= 64 rows in one launch
int acvae_gemm_nt_pair(const float* A0, int64_t lda0, const float* B0, int64_t ldb0, int K0, const float* bias0, float* C0,
                       int64_t ldc0, int N0, int acc0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, int K1,
                       const float* bias1, float* C1, int64_t ldc1, int N1, int acc1, int M, hipStream_t st);
// gemm_tn with its slab sum in the same launch (gemm.hip); ws = [TN_TICKETS words, zeroed once per composite call | slabs]
constexpr int TN_TICKETS = 256;
int acvae_gemm_tn_fused(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                        int accumulate, float* ws, int64_t ws_bytes, hipStream_t st);
long acvae_skinny_ws_floats();
long acvae_skinny_ticket_words();       // words at the head of a skinny workspace that a composite call zeroes once
int acvae_skinny_ws_reset(float* ws, hipStream_t st);
#include "transpose_batch.h"
