// A3: additive (Bahdanau) attention step, forward and backward.  Reference: models/attn_model.py:20-46.
// The reference re-projects all S encoder frames through h2attn at every decode step; here the
// encoder half (encproj) is computed once per batch and the query half (qproj) per step by the GEMM
// kernels, and this file does score -> mask -> softmax -> context.
//
// HBM/L2-bound: per query row the kernel streams encproj[n] (S*A floats) and enc[n] (S*E floats).
// One 256-thread workgroup (4 wavefronts) per query row; one wavefront per encoder frame for the
// tanh-dot (wave64 shuffle reduction), scores/weights staged in LDS, context by E-strided lanes so
// every enc row is read as full 128-B lines.
#include "common.h"
#include "../../include/acvae_hip.h"

namespace {

constexpr int ATT_THREADS = 256;

__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(
    const float* __restrict__ qproj, long q_sn, long q_sj, const float* __restrict__ encproj,
    const float* __restrict__ enc, const int64_t* __restrict__ lens, const float* __restrict__ v,
    float* __restrict__ ctx, long c_sn, long c_sj, float* __restrict__ weights, long w_sn, long w_sj, int Tq, int S,
    int A, int E) {
  extern __shared__ float smem[];  // [S] scores/weights, then [16] reduction scratch
  float* sc = smem;
  float* red = smem + S;
  const int n = blockIdx.x / Tq, j = blockIdx.x % Tq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ATT_THREADS / 64;
  const float* q = qproj + n * q_sn + j * q_sj;
  const float* P = encproj + (long)n * S * A;
  const int len = (int)lens[n];
  // ---- scores: wave per frame, lanes over A (float4 when A % 4 == 0)
  for (int s = wave; s < S; s += nw) {
    const float* p = P + (long)s * A;
    float acc = 0.f;
    if ((A & 3) == 0) {
      for (int a = lane * 4; a < A; a += 256) {
        const float4 pv = *reinterpret_cast<const float4*>(p + a);
        const float4 qv = *reinterpret_cast<const float4*>(q + a);
        const float4 vv = *reinterpret_cast<const float4*>(v + a);
        acc += vv.x * tanhf(qv.x + pv.x) + vv.y * tanhf(qv.y + pv.y) + vv.z * tanhf(qv.z + pv.z) +
               vv.w * tanhf(qv.w + pv.w);
      }
    } else {
      for (int a = lane; a < A; a += 64) acc += v[a] * tanhf(q[a] + p[a]);
    }
    acc = wave_sum(acc);
    if (lane == 0) sc[s] = (s < len) ? acc : -1e10f;  // masked_fill(mask == 0, -1e10)
  }
  __syncthreads();
  // ---- softmax over S
  float m = -INFINITY;
  for (int s = threadIdx.x; s < S; s += ATT_THREADS) m = fmaxf(m, sc[s]);
  m = block_max(m, red);
  float sum = 0.f;
  for (int s = threadIdx.x; s < S; s += ATT_THREADS) {
    const float e = expf(sc[s] - m);
    sc[s] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  const float inv = 1.f / sum;
  float* wout = weights + n * w_sn + j * w_sj;
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += ATT_THREADS) {
    const float w = sc[s] * inv;
    sc[s] = w;
    wout[s] = w;
  }
  __syncthreads();
  // ---- context
  const float* Hn = enc + (long)n * S * E;
  float* c = ctx + n * c_sn + j * c_sj;
  for (int e = threadIdx.x; e < E; e += ATT_THREADS) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += sc[s] * Hn[(long)s * E + e];
    c[e] = acc;
  }
}

// Backward: one workgroup per clip n, looping over that clip's Tq queries so that the += into
// dencproj[n] / denc[n] is race-free and deterministic.  tanh is recomputed (cheaper than saving
// [R,S,A]).
constexpr int ATB_THREADS = 512;

__global__ __launch_bounds__(ATB_THREADS) void attn_bwd_kernel(
    const float* __restrict__ dctx, long dc_sn, long dc_sj, const float* __restrict__ qproj, long q_sn, long q_sj,
    const float* __restrict__ encproj, const float* __restrict__ enc, const int64_t* __restrict__ lens,
    const float* __restrict__ v, const float* __restrict__ weights, long w_sn, long w_sj, float* __restrict__ dqproj,
    long dq_sn, long dq_sj, float* __restrict__ dencproj, float* __restrict__ denc, float* __restrict__ dv_part,
    int Tq, int S, int A, int E) {
  extern __shared__ float smem[];  // [S] dscore, [S] w, [16] red
  float* ds = smem;
  float* ws = smem + S;
  float* red = smem + 2 * S;
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ATB_THREADS / 64;
  const float* P = encproj + (long)n * S * A;
  const float* Hn = enc + (long)n * S * E;
  float* dP = dencproj + (long)n * S * A;
  float* dH = denc + (long)n * S * E;
  const int len = (int)lens[n];
  // per-thread dv accumulators for a = threadIdx.x + k*ATB_THREADS (A <= 4*ATB_THREADS)
  float dvacc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < Tq; ++j) {
    const float* dc = dctx + n * dc_sn + j * dc_sj;
    const float* w = weights + n * w_sn + j * w_sj;
    const float* q = qproj + n * q_sn + j * q_sj;
    float* dq = dqproj + n * dq_sn + j * dq_sj;
    __syncthreads();
    // dw_s = dctx . enc_s  (wave per frame)
    for (int s = wave; s < S; s += nw) {
      float acc = 0.f;
      for (int e = lane; e < E; e += 64) acc += dc[e] * Hn[(long)s * E + e];
      acc = wave_sum(acc);
      if (lane == 0) { ds[s] = acc; ws[s] = w[s]; }
    }
    __syncthreads();
    float dot = 0.f;
    for (int s = threadIdx.x; s < S; s += ATB_THREADS) dot += ws[s] * ds[s];
    dot = block_sum(dot, red);
    __syncthreads();
    // dscore_s; masked_fill blocks the gradient at s >= len (matters only when len == 0)
    for (int s = threadIdx.x; s < S; s += ATB_THREADS) ds[s] = (s < len) ? ws[s] * (ds[s] - dot) : 0.f;
    __syncthreads();
    // denc[n,s,e] += w_s * dctx_e
    for (int e = threadIdx.x; e < E; e += ATB_THREADS) {
      const float d = dc[e];
      for (int s = 0; s < S; ++s) dH[(long)s * E + e] += ws[s] * d;
    }
    // du = dscore_s * v_a * (1 - tanh^2) ; dP += du ; dq = sum_s du ; dv += sum_s dscore_s * tanh
    int k = 0;
    for (int a = threadIdx.x; a < A; a += ATB_THREADS, ++k) {
      const float qa = q[a], va = v[a];
      float dqa = 0.f, dva = 0.f;
      for (int s = 0; s < S; ++s) {
        const float th = tanhf(qa + P[(long)s * A + a]);
        const float g = ds[s];
        const float du = g * va * (1.f - th * th);
        dP[(long)s * A + a] += du;
        dqa += du;
        dva += g * th;
      }
      dq[a] = dqa;
      dvacc[k] += dva;
    }
  }
  int k = 0;
  for (int a = threadIdx.x; a < A; a += ATB_THREADS, ++k) dv_part[(long)n * A + a] += dvacc[k];
}

}  // namespace

extern "C" int acvae_attn_fwd(const float* qproj, int64_t q_sn, int64_t q_sj, const float* encproj, const float* enc,
                              const int64_t* lens, const float* v, float* ctx, int64_t c_sn, int64_t c_sj,
                              float* weights, int64_t w_sn, int64_t w_sj, int N, int Tq, int S, int A, int E,
                              void* stream) {
  if (!qproj || !encproj || !enc || !lens || !v || !ctx || !weights) return ACVAE_EINVAL;
  if (N <= 0 || Tq <= 0 || S <= 0 || A <= 0 || E <= 0) return ACVAE_EINVAL;
  if (S > 8192) return ACVAE_EUNSUPPORTED;
  if ((A & 3) == 0 && (!aligned16(qproj) || !aligned16(encproj) || !aligned16(v) || (q_sn & 3) || (q_sj & 3)))
    return ACVAE_EALIGN;
  const size_t shm = (size_t)(S + 16) * sizeof(float);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(N * Tq), dim3(ATT_THREADS), shm, (hipStream_t)stream, qproj, q_sn, q_sj,
                     encproj, enc, lens, v, ctx, c_sn, c_sj, weights, w_sn, w_sj, Tq, S, A, E);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_attn_bwd(const float* dctx, int64_t dc_sn, int64_t dc_sj, const float* qproj, int64_t q_sn,
                              int64_t q_sj, const float* encproj, const float* enc, const int64_t* lens,
                              const float* v, const float* weights, int64_t w_sn, int64_t w_sj, float* dqproj,
                              int64_t dq_sn, int64_t dq_sj, float* dencproj, float* denc, float* dv_part, int N,
                              int Tq, int S, int A, int E, void* stream) {
  if (!dctx || !qproj || !encproj || !enc || !lens || !v || !weights || !dqproj || !dencproj || !denc || !dv_part)
    return ACVAE_EINVAL;
  if (N <= 0 || Tq <= 0 || S <= 0 || A <= 0 || E <= 0) return ACVAE_EINVAL;
  if (S > 8192 || A > 4 * ATB_THREADS) return ACVAE_EUNSUPPORTED;
  const size_t shm = (size_t)(2 * S + 16) * sizeof(float);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(N), dim3(ATB_THREADS), shm, (hipStream_t)stream, dctx, dc_sn, dc_sj, qproj,
                     q_sn, q_sj, encproj, enc, lens, v, weights, w_sn, w_sj, dqproj, dq_sn, dq_sj, dencproj, denc,
                     dv_part, Tq, S, A, E);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
