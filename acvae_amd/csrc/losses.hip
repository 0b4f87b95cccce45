// Reparameterisation, Gaussian KL, label-smoothed cross entropy, greedy argmax, MSE.
// References: models/text_encoder.py:196-197,257-262 (reparam), utils/train_util.py:243-266 (CE, KL),
// models/word_model.py:173-207 (log_softmax + max), losses/loss.py:18-70 (masked CE),
// runners/pytorch_runner_vae.py:315-318 (loss assembly).
// All HBM-bound: vectorised streaming + wave64 shuffle reductions; every reduction is a fixed-shape
// two-level tree (per-block partials, then one block) so results are run-to-run deterministic.
#include "common.h"
#include "../../include/acvae_hip.h"

namespace {

constexpr int EW_THREADS = 256;
constexpr int RED_BLOCKS_MAX = 1024;

// ------------------------------------------------------------------ reparam
__global__ void reparam_fwd_kernel(const float* __restrict__ ml, long ld_ml, const float* __restrict__ eps,
                                   long ld_eps, float* __restrict__ mean, float* __restrict__ logv,
                                   float* __restrict__ z, long ld_out, float* __restrict__ z2, long ld_z2, int rows,
                                   int E) {
  const long total = (long)rows * E;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / E), e = (int)(i % E);
    const float mu = ml[r * ld_ml + e], lv = ml[r * ld_ml + E + e];
    const float zz = eps[r * ld_eps + e] * expf(.5f * lv) + mu;
    mean[r * ld_out + e] = mu;
    logv[r * ld_out + e] = lv;
    z[r * ld_out + e] = zz;
    if (z2) z2[r * ld_z2 + e] = zz;
  }
}

__global__ void reparam_bwd_kernel(const float* __restrict__ dz, long ld_dz, const float* __restrict__ dmean_ext,
                                   const float* __restrict__ dlog_ext, long ld_ext, const float* __restrict__ logv,
                                   long ld_lv, const float* __restrict__ eps, long ld_eps, float* __restrict__ dml,
                                   long ld_dml, int rows, int E) {
  const long total = (long)rows * E;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / E), e = (int)(i % E);
    const float g = dz ? dz[r * ld_dz + e] : 0.f;
    float dm = g, dl = g * eps[r * ld_eps + e] * .5f * expf(.5f * logv[r * ld_lv + e]);
    if (dmean_ext) dm += dmean_ext[r * ld_ext + e];
    if (dlog_ext) dl += dlog_ext[r * ld_ext + e];
    dml[r * ld_dml + e] = dm;
    dml[r * ld_dml + E + e] = dl;
  }
}

// ------------------------------------------------------------------ generic two-level sum
__global__ void final_sum_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)partials[i];
  // fixed-order tree in double: deterministic and accurate for <= 1024 partials
  acc = wave_sum_d(acc);
  __shared__ double redd[16];
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += redd[i];
    out[0] = (float)(t * (double)scale);
  }
  (void)red;
}

// ------------------------------------------------------------------ KL
__device__ __forceinline__ float kl_elem(float mu1, float lv1, float mu2, float lv2) {
  const float d = mu1 - mu2;
  return lv2 * .5f - lv1 * .5f + (expf(lv1) + d * d) / (2.f * expf(lv2)) - .5f;
}

__global__ __launch_bounds__(EW_THREADS) void kl_fwd_kernel(const float* __restrict__ mu1,
                                                            const float* __restrict__ lv1,
                                                            const float* __restrict__ mu2,
                                                            const float* __restrict__ lv2, float* __restrict__ partials,
                                                            long n) {
  __shared__ float red[16];
  float acc = 0.f;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)EW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * EW_THREADS) {
    const float4 a = reinterpret_cast<const float4*>(mu1)[i], b = reinterpret_cast<const float4*>(lv1)[i];
    const float4 c = reinterpret_cast<const float4*>(mu2)[i], d = reinterpret_cast<const float4*>(lv2)[i];
    acc += kl_elem(a.x, b.x, c.x, d.x) + kl_elem(a.y, b.y, c.y, d.y) + kl_elem(a.z, b.z, c.z, d.z) +
           kl_elem(a.w, b.w, c.w, d.w);
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += EW_THREADS) acc += kl_elem(mu1[i], lv1[i], mu2[i], lv2[i]);
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ void kl_bwd_kernel(const float* __restrict__ mu1, const float* __restrict__ lv1,
                              const float* __restrict__ mu2, const float* __restrict__ lv2,
                              const float* __restrict__ grad_out, float inv_rows, float* __restrict__ dmu1,
                              float* __restrict__ dlv1, float* __restrict__ dmu2, float* __restrict__ dlv2, long n) {
  const float g = grad_out[0] * inv_rows;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = mu1[i] - mu2[i];
    const float v1 = expf(lv1[i]), iv2 = 1.f / expf(lv2[i]);
    if (dmu1) dmu1[i] = g * d * iv2;
    if (dmu2) dmu2[i] = -g * d * iv2;
    if (dlv1) dlv1[i] = g * (-.5f + .5f * v1 * iv2);
    if (dlv2) dlv2[i] = g * (.5f - .5f * (v1 + d * d) * iv2);
  }
}

// ------------------------------------------------------------------ MSE
__global__ __launch_bounds__(EW_THREADS) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ partials, long n) {
  __shared__ float red[16];
  float acc = 0.f;
  for (long i = blockIdx.x * (long)EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) {
    const float d = a[i] - b[i];
    acc += d * d;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ grad_out, float scale, float* __restrict__ da,
                               float* __restrict__ db, long n) {
  const float g = grad_out[0] * scale;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = g * (a[i] - b[i]);
    if (da) da[i] = d;
    if (db) db[i] = -d;
  }
}

// ------------------------------------------------------------------ row log-softmax stats + argmax
// One workgroup per row; the row (V floats, 20 KB at V=5000) is read twice, the second time from L1/L2.
__global__ __launch_bounds__(EW_THREADS) void row_stats_kernel(const float* __restrict__ logits, long ld_n, long ld_t,
                                                               int64_t* __restrict__ argmax,
                                                               float* __restrict__ max_logprob,
                                                               float* __restrict__ lse, long o_sn, long o_st, int T,
                                                               int V) {
  __shared__ float red[16];
  __shared__ int redi[16];
  const int n = blockIdx.x / T, t = blockIdx.x % T;
  const float* x = logits + n * ld_n + t * ld_t;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += EW_THREADS) {
    const float v = x[c];
    if (v > m) { m = v; mi = c; }  // strict '>' keeps the first maximum within a thread's stride
  }
  // wave argmax with first-index tie-break
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = m; redi[w] = mi; }
  __syncthreads();
  m = red[0]; mi = redi[0];
  for (int i = 1; i < EW_THREADS / 64; ++i)
    if (red[i] > m || (red[i] == m && redi[i] < mi)) { m = red[i]; mi = redi[i]; }
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += EW_THREADS) s += expf(x[c] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float ls = logf(s);
    const long o = n * o_sn + t * o_st;
    if (argmax) argmax[o] = mi;
    if (max_logprob) max_logprob[o] = -ls;  // (m - m) - log(sum exp(x - m))
    if (lse) lse[o] = m + ls;
  }
}

// ------------------------------------------------------------------ sample_next_word, non-greedy branches
// models/word_model.py:188-203.  logprobs = log_softmax(logits) as torch evaluates it, (x - max) - log(sum exp(x - max));
//   method 1 "gumbel"  (:188-197): w = argmax_c (logprobs_c + g_c) / temp,  g = -log(-log(U + 1e-20) + 1e-20) drawn by the
//                      caller (U = torch.rand on the CPU generator); the reference takes the argmax of
//                      log_softmax((logprobs + g) / temp), a monotone map of the same scores;
//   method 2 "sample"  (:198-203): w = torch.multinomial(exp(logprobs / temp), 1), which ATen evaluates as
//                      argmax_c exp(logprobs_c / temp) / q_c with q ~ Exp(1) drawn as one [N,V] tensor
//                      (aten/src/ATen/native/Distributions.cpp, multinomial_out, n_sample == 1): the caller draws q.
// First maximum wins, as torch.max / argmax do.  Writes w and logprobs[w] (the reference's sampled_logprobs).
__global__ __launch_bounds__(EW_THREADS) void sample_rows_kernel(const float* __restrict__ logits, long ld_n, long ld_t,
                                                                 const float* __restrict__ noise, long nz_sn, long nz_st,
                                                                 int method, float temp, int64_t* __restrict__ w_out,
                                                                 float* __restrict__ lp_out, long o_sn, long o_st, int T,
                                                                 int V) {
  __shared__ float red[16];
  __shared__ int redi[16];
  const int n = blockIdx.x / T, t = blockIdx.x % T;
  const float* x = logits + n * ld_n + t * ld_t;
  const float* z = noise + n * nz_sn + t * nz_st;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < V; c += EW_THREADS) m = fmaxf(m, x[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += EW_THREADS) s += expf(x[c] - m);
  s = block_sum(s, red);
  const float ls = logf(s);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += EW_THREADS) {
    const float lp = (x[c] - m) - ls;
    const float sc = method == 1 ? (lp + z[c]) / temp : expf(lp / temp) / z[c];
    if (sc > best || bi == 0x7fffffff) { best = sc; bi = c; }   // strict '>': first maximum within the stride
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __syncthreads();
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = best; redi[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    best = red[0]; bi = redi[0];
    for (int i = 1; i < EW_THREADS / 64; ++i)
      if (red[i] > best || (red[i] == best && redi[i] < bi)) { best = red[i]; bi = redi[i]; }
    if (bi < 0 || bi >= V) bi = 0;                      // every score NaN: torch returns index 0 as well
    const long o = n * o_sn + t * o_st;
    w_out[o] = bi;
    if (lp_out) lp_out[o] = (x[bi] - m) - ls;
  }
}

// ------------------------------------------------------------------ label-smoothed CE
// loss_row = -[(1-s) * lp_tgt + s/(V-1) * (sum_c lp_c - lp_tgt)],  lp_c = x_c - lse.
__global__ __launch_bounds__(EW_THREADS) void ce_rows_kernel(const float* __restrict__ logits, long ld_n, long ld_t,
                                                             const int64_t* __restrict__ targets, long tg_sn,
                                                             const int64_t* __restrict__ lens1,
                                                             const float* __restrict__ lse, float smoothing,
                                                             float* __restrict__ loss_rows, int T, int V) {
  __shared__ float red[16];
  const int n = blockIdx.x / T, t = blockIdx.x % T;
  const long o = (long)n * T + t;
  if (lens1 && t >= (int)lens1[n]) {
    if (threadIdx.x == 0) loss_rows[o] = 0.f;
    return;
  }
  const float* x = logits + n * ld_n + t * ld_t;
  const float l = lse[o];
  float sumlp = 0.f;
  if (smoothing != 0.f) {
    for (int c = threadIdx.x; c < V; c += EW_THREADS) sumlp += x[c] - l;
    sumlp = block_sum(sumlp, red);
  }
  if (threadIdx.x == 0) {
    const float lpt = x[targets[n * tg_sn + t]] - l;
    loss_rows[o] = -((1.f - smoothing) * lpt + (smoothing != 0.f ? smoothing / (float)(V - 1) * (sumlp - lpt) : 0.f));
  }
}

__global__ void ce_reduce_kernel(const float* __restrict__ loss_rows, const int64_t* __restrict__ lens1, int N, int T,
                                 int mean, float* __restrict__ out) {
  __shared__ double redd[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < N * T; i += blockDim.x) acc += (double)loss_rows[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += redd[i];
    if (mean) {
      long cnt = 0;
      for (int n = 0; n < N; ++n) cnt += lens1 ? (lens1[n] < T ? lens1[n] : T) : T;
      t /= (double)cnt;
    }
    out[0] = (float)t;
  }
}

__global__ __launch_bounds__(EW_THREADS) void ce_bwd_kernel(const float* __restrict__ logits, long ld_n, long ld_t,
                                                            const int64_t* __restrict__ targets, long tg_sn,
                                                            const int64_t* __restrict__ lens1,
                                                            const float* __restrict__ lse, float smoothing,
                                                            int reduction, const float* __restrict__ grad_out,
                                                            const float* __restrict__ grad_rows,
                                                            float* __restrict__ dlogits, int N, int T, int V) {
  const int n = blockIdx.x / T, t = blockIdx.x % T;
  float* d = dlogits + n * ld_n + t * ld_t;
  if (lens1 && t >= (int)lens1[n]) {
    for (int c = threadIdx.x; c < V; c += EW_THREADS) d[c] = 0.f;
    return;
  }
  float g;
  if (reduction == 0) {
    g = grad_rows[(long)n * T + t];
  } else {
    g = grad_out[0];
    if (reduction == 1) {
      long cnt = 0;
      for (int i = 0; i < N; ++i) cnt += lens1 ? (lens1[i] < T ? lens1[i] : T) : T;
      g /= (float)cnt;
    }
  }
  const float* x = logits + n * ld_n + t * ld_t;
  const float l = lse[(long)n * T + t];
  const int tg = (int)targets[n * tg_sn + t];
  const float off = smoothing / (float)(V - 1), on = 1.f - smoothing;
  for (int c = threadIdx.x; c < V; c += EW_THREADS) d[c] = g * (expf(x[c] - l) - (c == tg ? on : off));
}


// ------------------------------------------------------------------ beam-search helpers (N1)
// out[n,c] = logits[n,c] - lse[n] + prev[n]      (log_softmax + accumulated beam log-prob, vae_model.py:909-912)
__global__ void logprob_add_kernel(const float* __restrict__ logits, long ld, const float* __restrict__ lse,
                                   const float* __restrict__ prev, float* __restrict__ out, int N, int V) {
  const long total = (long)N * V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / V), c = (int)(i % V);
    out[i] = logits[n * ld + c] - lse[n] + (prev ? prev[n] : 0.f);
  }
}
// Diverse beam search scores (N3, word_model.py:344-348): per row
//   out[n,c] = log_softmax(log_softmax(logits[n]) / T)[c] - lambda * counts[c] + prev[n]
// counts (nullable) = how often the earlier groups chose word c at this step.  One workgroup per row.
__global__ __launch_bounds__(256) void dbs_scores_kernel(const float* __restrict__ logits, long ld, float temperature,
                                                         const float* __restrict__ counts, float lambda,
                                                         const float* __restrict__ prev, float* __restrict__ out,
                                                         int V, int rows_per_count) {
  __shared__ float red[16];
  const int n = blockIdx.x;
  if (counts && rows_per_count > 0) counts += (long)(n / rows_per_count) * V;   // one count vector per clip
  const float* x = logits + (long)n * ld;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < V; c += blockDim.x) m = fmaxf(m, x[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) s += expf(x[c] - m);
  s = block_sum(s, red);
  const float lse1 = m + logf(s);
  float m2 = -INFINITY;
  for (int c = threadIdx.x; c < V; c += blockDim.x) m2 = fmaxf(m2, (x[c] - lse1) / temperature);
  m2 = block_max(m2, red);
  float s2 = 0.f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) s2 += expf((x[c] - lse1) / temperature - m2);
  s2 = block_sum(s2, red);
  const float lse2 = m2 + logf(s2);
  const float pv = prev ? prev[n] : 0.f;
  float* o = out + (long)n * V;
  for (int c = threadIdx.x; c < V; c += blockDim.x) {
    float v = (x[c] - lse1) / temperature - lse2;
    if (counts) v -= counts[c] * lambda;
    o[c] = pv + v;
  }
}
// flat top-k (k <= 16) of x[0..n), sorted descending, ties -> lower index first (torch.topk(sorted=True) order for
// distinct values); one workgroup, k selection passes.  Also emits idx / V and idx % V.
__global__ __launch_bounds__(1024) void topk_flat_kernel(const float* __restrict__ x, long n, int k, int V,
                                                         float* __restrict__ vals, int64_t* __restrict__ idx,
                                                         int64_t* __restrict__ row, int64_t* __restrict__ col,
                                                         int row_base, long group_stride) {
  __shared__ float rv[16];
  __shared__ long ri[16];
  __shared__ float sel_v[16];
  __shared__ long sel_i[16];
  // one workgroup per group (batched beam search: group = clip, x[g][n], outputs [g][k]); row_base offsets `row` so
  // that it indexes the rows of the whole batch (g * row_base + idx / V)
  x += (long)blockIdx.x * group_stride;
  vals += (long)blockIdx.x * k; idx += (long)blockIdx.x * k;
  if (row) row += (long)blockIdx.x * k;
  if (col) col += (long)blockIdx.x * k;
  const long rbase = (long)blockIdx.x * row_base;
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    long bi = 0x7fffffffffffffffL;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      const float v = x[i];
      bool taken = false;
      for (int q = 0; q < j; ++q) taken = taken || (sel_i[q] == i);
      if (!taken && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const long oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = bv; ri[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
      sel_v[j] = bv; sel_i[j] = bi;
      vals[j] = bv; idx[j] = bi;
      if (row) row[j] = rbase + bi / V;
      if (col) col[j] = bi % V;
    }
    __syncthreads();
  }
}

inline int grid_for(long n, int per_block) {
  long b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > RED_BLOCKS_MAX ? RED_BLOCKS_MAX : b));
}

}  // namespace

extern "C" int acvae_abi_version(void) { return ACVAE_ABI_VERSION; }

extern "C" int acvae_reparam_fwd(const float* ml, int64_t ld_ml, const float* eps, int64_t ld_eps, float* mean,
                                 float* logv, float* z, int64_t ld_out, float* z2, int64_t ld_z2, int rows, int E,
                                 void* stream) {
  if (!ml || !eps || !mean || !logv || !z || rows <= 0 || E <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for((long)rows * E, EW_THREADS)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, ml, ld_ml, eps, ld_eps, mean, logv, z, ld_out, z2, ld_z2, rows, E);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_reparam_bwd(const float* dz, int64_t ld_dz, const float* dmean_ext, const float* dlog_ext,
                                 int64_t ld_ext, const float* logv, int64_t ld_lv, const float* eps, int64_t ld_eps,
                                 float* dml, int64_t ld_dml, int rows, int E, void* stream) {
  if (!logv || !eps || !dml || rows <= 0 || E <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for((long)rows * E, EW_THREADS)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, dz, ld_dz, dmean_ext, dlog_ext, ld_ext, logv, ld_lv, eps, ld_eps, dml,
                     ld_dml, rows, E);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int64_t acvae_kl_partials(int64_t n_elem) { return grid_for(n_elem, EW_THREADS * 4); }

extern "C" int acvae_gauss_kl_fwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2,
                                  float* partials, float* out_scalar, int64_t rows, int E, void* stream) {
  if (!mu1 || !lv1 || !mu2 || !lv2 || !partials || !out_scalar || rows <= 0 || E <= 0) return ACVAE_EINVAL;
  if (!aligned16(mu1) || !aligned16(lv1) || !aligned16(mu2) || !aligned16(lv2)) return ACVAE_EALIGN;
  const long n = rows * E;
  const int nb = (int)acvae_kl_partials(n);
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(nb), dim3(EW_THREADS), 0, (hipStream_t)stream, mu1, lv1, mu2, lv2, partials,
                     n);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb,
                     (float)(1.0 / (double)rows), out_scalar);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_gauss_kl_bwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2,
                                  const float* grad_out, float* dmu1, float* dlv1, float* dmu2, float* dlv2,
                                  int64_t rows, int E, void* stream) {
  if (!mu1 || !lv1 || !mu2 || !lv2 || !grad_out || rows <= 0 || E <= 0) return ACVAE_EINVAL;
  const long n = rows * E;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, (hipStream_t)stream, mu1, lv1,
                     mu2, lv2, grad_out, (float)(1.0 / (double)rows), dmu1, dlv1, dmu2, dlv2, n);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_mse_fwd(const float* a, const float* b, float* partials, float* out_scalar, int64_t n,
                             void* stream) {
  if (!a || !b || !partials || !out_scalar || n <= 0) return ACVAE_EINVAL;
  const int nb = (int)acvae_kl_partials(n);
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(nb), dim3(EW_THREADS), 0, (hipStream_t)stream, a, b, partials, (long)n);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb,
                     (float)(1.0 / (double)n), out_scalar);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_mse_bwd(const float* a, const float* b, const float* grad_out, float* da, float* db, int64_t n,
                             void* stream) {
  if (!a || !b || !grad_out || n <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, (hipStream_t)stream, a, b,
                     grad_out, (float)(2.0 / (double)n), da, db, (long)n);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// loss = ce + w_kl * kl (+ w_mse * mse): the runner's loss assembly (runners/pytorch_runner_vae.py:315-320) as one launch each way
// instead of ~a dozen scalar torch kernels in front of the decode backward.  Same arithmetic and order as the torch expression.
__global__ void loss_combine_fwd_kernel(const float* ce, const float* kl, const float* mse, float w_kl, float w_mse, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float v = ce[0] + w_kl * kl[0];
    if (mse) v = v + w_mse * mse[0];
    out[0] = v;
  }
}
__global__ void loss_combine_bwd_kernel(const float* g, float w_kl, float w_mse, float* g_ce, float* g_kl, float* g_mse) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float v = g[0];
    g_ce[0] = v;
    g_kl[0] = v * w_kl;
    if (g_mse) g_mse[0] = v * w_mse;
  }
}
extern "C" int acvae_loss_combine_fwd(const float* ce, const float* kl, const float* mse, float w_kl, float w_mse, float* out,
                                      void* stream) {
  if (!ce || !kl || !out) return ACVAE_EINVAL;
  hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ce, kl, mse, w_kl, w_mse, out);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
extern "C" int acvae_loss_combine_bwd(const float* grad_out, float w_kl, float w_mse, float* g_ce, float* g_kl, float* g_mse,
                                      void* stream) {
  if (!grad_out || !g_ce || !g_kl) return ACVAE_EINVAL;
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grad_out, w_kl, w_mse, g_ce, g_kl, g_mse);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_row_logsoftmax_argmax(const float* logits, int64_t ld_n, int64_t ld_t, int64_t* argmax,
                                           float* max_logprob, float* lse, int64_t o_sn, int64_t o_st, int N, int T,
                                           int V, void* stream) {
  if (!logits || N <= 0 || T <= 0 || V <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(row_stats_kernel, dim3(N * T), dim3(EW_THREADS), 0, (hipStream_t)stream, logits, ld_n, ld_t,
                     argmax, max_logprob, lse, o_sn, o_st, T, V);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// Device-side noise for the non-greedy branches (opt-in: `rng="device"`): element i of the buffer from the counter-based
// Philox generator (seed, i) - Gumbel noise g = -log(-log(U + 1e-20) + 1e-20) or an Exp(1) draw q = -log(1 - U), the two
// quantities the host draws on the CPU generator in the reference's order (parity mode).  Same distribution, not the same
// stream: 16 M draws per batch take the host ~0.2 s and this kernel ~30 us.
__global__ void sample_noise_kernel(float* __restrict__ out, long n, int method, uint64_t seed) {
  for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 * 4 < n; i4 += (long)gridDim.x * blockDim.x) {
    const uint4 r = philox4x32(seed, (uint64_t)i4, 0x5a3d1u);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long i = i4 * 4 + k;
      if (i < n) {
        const float u = (float)(w[k] >> 8) * (1.0f / 16777216.0f);            // [0, 1)
        out[i] = method == ACVAE_SAMPLE_GUMBEL ? -logf(-logf(u + 1e-20f) + 1e-20f) : fmaxf(-log1pf(-u), 1e-30f);
      }
    }
  }
}

extern "C" int acvae_sample_noise(float* noise, int64_t n, int method, uint64_t seed, void* stream) {
  if (!noise || n <= 0 || (method != ACVAE_SAMPLE_GUMBEL && method != ACVAE_SAMPLE_MULTINOMIAL)) return ACVAE_EINVAL;
  const long blocks = (n / 4 + EW_THREADS - 1) / EW_THREADS;
  hipLaunchKernelGGL(sample_noise_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : (blocks < 1 ? 1 : blocks))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, noise, (long)n, method, seed);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_sample_next_word(const float* logits, int64_t ld_n, int64_t ld_t, const float* noise,
                                      int64_t nz_sn, int64_t nz_st, int method, float temp, int64_t* w_out,
                                      float* logprob_out, int64_t o_sn, int64_t o_st, int N, int T, int V, void* stream) {
  if (!logits || !noise || !w_out || N <= 0 || T <= 0 || V <= 0) return ACVAE_EINVAL;
  if ((method != ACVAE_SAMPLE_GUMBEL && method != ACVAE_SAMPLE_MULTINOMIAL) || !(temp > 0.f)) return ACVAE_EINVAL;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(N * T), dim3(EW_THREADS), 0, (hipStream_t)stream, logits, ld_n, ld_t, noise,
                     nz_sn, nz_st, method, temp, w_out, logprob_out, o_sn, o_st, T, V);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_ls_ce_fwd(const float* logits, int64_t ld_n, int64_t ld_t, const int64_t* targets, int64_t tg_sn,
                               const int64_t* lens1, const float* lse, float smoothing, int reduction,
                               float* loss_rows, float* out_scalar, int N, int T, int V, void* stream) {
  if (!logits || !targets || !lse || !loss_rows || N <= 0 || T <= 0 || V <= 1) return ACVAE_EINVAL;
  if (reduction < 0 || reduction > 2 || (reduction != 0 && !out_scalar)) return ACVAE_EINVAL;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(N * T), dim3(EW_THREADS), 0, (hipStream_t)stream, logits, ld_n, ld_t,
                     targets, tg_sn, lens1, lse, smoothing, loss_rows, T, V);
  if (reduction != 0)
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, lens1, N, T,
                       reduction == 1, out_scalar);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_ls_ce_bwd(const float* logits, int64_t ld_n, int64_t ld_t, const int64_t* targets, int64_t tg_sn,
                               const int64_t* lens1, const float* lse, float smoothing, int reduction,
                               const float* grad_out, const float* grad_rows, float* dlogits, int N, int T, int V,
                               void* stream) {
  if (!logits || !targets || !lse || !dlogits || N <= 0 || T <= 0 || V <= 1) return ACVAE_EINVAL;
  if (reduction < 0 || reduction > 2) return ACVAE_EINVAL;
  if ((reduction == 0 && !grad_rows) || (reduction != 0 && !grad_out)) return ACVAE_EINVAL;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(N * T), dim3(EW_THREADS), 0, (hipStream_t)stream, logits, ld_n, ld_t, targets,
                     tg_sn, lens1, lse, smoothing, reduction, grad_out, grad_rows, dlogits, N, T, V);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_logprob_add(const float* logits, int64_t ld, const float* lse, const float* prev, float* out, int N,
                                 int V, void* stream) {
  if (!logits || !lse || !out || N <= 0 || V <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(logprob_add_kernel, dim3(grid_for((long)N * V, EW_THREADS)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, logits, ld, lse, prev, out, N, V);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_dbs_scores(const float* logits, int64_t ld, float temperature, const float* counts,
                                float diversity_lambda, const float* prev, float* out, int N, int V,
                                int rows_per_count, void* stream) {
  if (!logits || !out || N <= 0 || V <= 0 || !(temperature > 0.f) || rows_per_count < 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(dbs_scores_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, ld, temperature, counts,
                     diversity_lambda, prev, out, V, rows_per_count);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_topk_flat(const float* x, int64_t n, int k, int V, float* vals, int64_t* idx, int64_t* row,
                               int64_t* col, void* stream) {
  if (!x || !vals || !idx || n <= 0 || k <= 0 || k > 16 || k > n || V <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(topk_flat_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long)n, k, V, vals, idx, row,
                     col, 0, 0L);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_topk_flat_batched(const float* x, int64_t n, int64_t group_stride, int k, int V, float* vals,
                                       int64_t* idx, int64_t* row, int64_t* col, int groups, int row_base,
                                       void* stream) {
  if (!x || !vals || !idx || n <= 0 || k <= 0 || k > 16 || k > n || V <= 0 || groups <= 0 || group_stride < n)
    return ACVAE_EINVAL;
  hipLaunchKernelGGL(topk_flat_kernel, dim3(groups), dim3(1024), 0, (hipStream_t)stream, x, (long)n, k, V, vals, idx,
                     row, col, row_base, (long)group_stride);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
