// Internal C++ launcher API of conv.hip (used by the composite encoder driver and the per-op C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
long conv3x3_wgrad_slab_floats(int N, int H, int W, int Cin, int Cout);
int conv3x3_wgrad(const float* dY, const float* X, const float* scale, const float* shift, float* dW_oihw, float* slab,
                  int N, int H, int W, int Cin, int Cout, hipStream_t st);
int repack_weights(const float* W_oihw, float* Wf, float* Wd, int Cout, int Cin, hipStream_t st);
int bn0_stats(const float* x, float* partials, long rows, int F, int* nparts, hipStream_t st);
int bn0_partials_rows(long rows);
int bn_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* beta,
                float* running_mean, float* running_var, int64_t* nbt, int training, float* scale, float* shift,
                float* mean, float* invstd, double* dpart, hipStream_t st);
long colsum_scratch_doubles(int width);
// deterministic column sums of x[P][width] (ld == width): out[i] = sum_p x[p][i]; entries >= split (if > 0) go to out2
int colsum2(const float* x, int P, int width, double* dpart, float* out, float* out2, int split, hipStream_t st);
int conv1_first_blocks(int N, int T);
int conv1_first_fwd(const float* x, const float* scale0, const float* shift0, const float* W1, float* Y,
                    float* partials, int N, int T, int F, hipStream_t st);
int conv1_first_bwd(const float* x, const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                    const float* W1, const float* dY, float* dw_part, float* bn_part, float* dW1, float* dgamma0,
                    float* dbeta0, double* dpart, int N, int T, int F, hipStream_t st);
// pool: 2x2 average pool after BN+ReLU (ConvBlock pool_size (2,2)); !pool: pool_size (1,1) (Cnn14's last block)
int bn_relu_pool(const float* Y, const float* scale, const float* shift, float* P, int N, int H, int W, int C,
                 DropoutSpec drop, hipStream_t st, bool pool = true);
int bn_bwd_blocks(int N, int H, int W);
int bn_bwd(const float* Y, const float* dO, int upstream, const float* scale, const float* shift, const float* mean,
           const float* invstd, float* partials, float* sum_g, float* sum_gy, float* dY, double* dpart, int N, int H,
           int W, int C, DropoutSpec drop, hipStream_t st, bool batch_stats = true);
int freq_mean(const float* P, float* out, long rows, int Fp, int C, hipStream_t st);
int freq_mean_bwd(const float* dae, float* dP, long rows, int Fp, int C, hipStream_t st);
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
long acvae_skinny_ws_floats();
int acvae_skinny_ws_reset(float* ws, hipStream_t st);
