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

constexpr int ATT_THREADS = 256;      // many query rows per launch (batched teacher-forced prior attention)
constexpr int ATT_THREADS_BIG = 1024;  // few query rows (one decode step): 16 wavefronts per row hide the L2 latency

// VEC: A % 4 == 0, E % 4 == 0, E / 4 <= blockDim.x and every row 16-B aligned (checked by the launcher).
template <bool VEC>
__global__ __launch_bounds__(ATT_THREADS_BIG) void attn_fwd_kernel(
    const float* __restrict__ qproj, long q_sn, long q_sj, const float* __restrict__ encproj,
    const float* __restrict__ enc, const int64_t* __restrict__ lens, const float* __restrict__ v,
    float* __restrict__ ctx, long c_sn, long c_sj, float* __restrict__ weights, long w_sn, long w_sj, int Tq, int S,
    int A, int E) {
  extern __shared__ float smem[];  // [S] scores/weights, [16] reduction scratch, VEC: [groups][E] context partials
  float* sc = smem;
  float* red = smem + S;
  const int n = blockIdx.x / Tq, j = blockIdx.x % Tq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* q = qproj + n * q_sn + j * q_sj;
  const float* P = encproj + (long)n * S * A;
  const int len = (int)lens[n];
  // ---- scores: wave per frame, lanes over A
  for (int s = wave; s < S; s += nw) {
    const float* p = P + (long)s * A;
    float acc = 0.f;
    if (VEC) {
      for (int a = lane * 4; a < A; a += 256) {
        const float4 pv = *reinterpret_cast<const float4*>(p + a);
        const float4 qv = *reinterpret_cast<const float4*>(q + a);
        const float4 vv = *reinterpret_cast<const float4*>(v + a);
        acc += vv.x * tanh_att(qv.x + pv.x) + vv.y * tanh_att(qv.y + pv.y) + vv.z * tanh_att(qv.z + pv.z) +
               vv.w * tanh_att(qv.w + pv.w);
      }
    } else {
      for (int a = lane; a < A; a += 64) acc += v[a] * tanh_att(q[a] + p[a]);
    }
    acc = wave_sum(acc);
    if (lane == 0) sc[s] = (s < len) ? acc : -1e10f;  // masked_fill(mask == 0, -1e10)
  }
  __syncthreads();
  // ---- softmax over S
  float m = -INFINITY;
  for (int s = threadIdx.x; s < S; s += blockDim.x) m = fmaxf(m, sc[s]);
  m = block_max(m, red);
  float sum = 0.f;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const float e = expf(sc[s] - m);
    sc[s] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  const float inv = 1.f / sum;
  float* wout = weights + n * w_sn + j * w_sj;
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const float w = sc[s] * inv;
    sc[s] = w;
    wout[s] = w;
  }
  __syncthreads();
  // ---- context
  const float* Hn = enc + (long)n * S * E;
  float* c = ctx + n * c_sn + j * c_sj;
  if (VEC) {
    // E/4 threads cover one enc row as float4; the thread groups take frames g, g+G, ... and meet in LDS (fixed order)
    float* part = smem + ((S + 16 + 3) & ~3);
    const int ev = E >> 2, G = blockDim.x / ev;
    const int g = threadIdx.x / ev, e4 = (threadIdx.x - g * ev) * 4;
    if (g < G) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s0 = g; s0 < S; s0 += G) {
        const float4 h = *reinterpret_cast<const float4*>(Hn + (long)s0 * E + e4);
        const float w = sc[s0];
        acc.x += w * h.x; acc.y += w * h.y; acc.z += w * h.z; acc.w += w * h.w;
      }
      *reinterpret_cast<float4*>(part + (long)g * E + e4) = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
      float acc = 0.f;
      for (int k = 0; k < G; ++k) acc += part[(long)k * E + e];
      c[e] = acc;
    }
  } else {
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += sc[s] * Hn[(long)s * E + e];
      c[e] = acc;
    }
  }
}

// Backward in three launches (all deterministic, no atomics):
//   score : one workgroup per query row -> dscore[r][s] = w_s * (dctx.enc_s - sum_j w_j dctx.enc_j)   (0 at masked s)
//   accum : grid (clip, chunk of ATB_CH frames): the workgroup OWNS dencproj/denc rows of its frames, keeps the
//           encproj tile and the running sums in registers while it walks the clip's Tq queries (tanh is
//           recomputed: cheaper than saving [R,S,A]), then does ONE += per element; writes per-chunk partials of
//           dq and dv
//   reduce: dq[r] = sum_chunk dq_part, dv_part[n] += sum_chunk dv_chunk   (fixed order)
constexpr int ATB_THREADS = 512;
constexpr int ATB_CH = 8;     // frames per chunk
constexpr int ATB_SLOTS = 4;  // A, E <= ATB_SLOTS * ATB_THREADS

template <bool VEC>
__global__ __launch_bounds__(ATT_THREADS_BIG) void attn_bwd_score_kernel(
    const float* __restrict__ dctx, long dc_sn, long dc_sj, const float* __restrict__ enc,
    const int64_t* __restrict__ lens, const float* __restrict__ weights, long w_sn, long w_sj,
    float* __restrict__ dscore, int Tq, int S, int E) {
  extern __shared__ float smem[];  // [S] dw, [16] red
  float* dw = smem;
  float* red = smem + S;
  const int n = blockIdx.x / Tq, j = blockIdx.x % Tq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* dc = dctx + n * dc_sn + j * dc_sj;
  const float* w = weights + n * w_sn + j * w_sj;
  const float* Hn = enc + (long)n * S * E;
  const int len = (int)lens[n];
  for (int s = wave; s < S; s += nw) {
    float acc = 0.f;
    if (VEC) {
      for (int e = lane * 4; e < E; e += 256) {
        const float4 d = *reinterpret_cast<const float4*>(dc + e);
        const float4 h = *reinterpret_cast<const float4*>(Hn + (long)s * E + e);
        acc += d.x * h.x + d.y * h.y + d.z * h.z + d.w * h.w;
      }
    } else {
      for (int e = lane; e < E; e += 64) acc += dc[e] * Hn[(long)s * E + e];
    }
    acc = wave_sum(acc);
    if (lane == 0) dw[s] = acc;
  }
  __syncthreads();
  float dot = 0.f;
  for (int s = threadIdx.x; s < S; s += blockDim.x) dot += w[s] * dw[s];
  dot = block_sum(dot, red);
  float* out = dscore + (long)blockIdx.x * S;
  // masked_fill blocks the gradient at s >= len (matters only when len == 0)
  for (int s = threadIdx.x; s < S; s += blockDim.x) out[s] = (s < len) ? w[s] * (dw[s] - dot) : 0.f;
}

__global__ __launch_bounds__(ATB_THREADS) void attn_bwd_accum_kernel(
    const float* __restrict__ dctx, long dc_sn, long dc_sj, const float* __restrict__ qproj, long q_sn, long q_sj,
    const float* __restrict__ encproj, const float* __restrict__ v, const float* __restrict__ weights, long w_sn,
    long w_sj, const float* __restrict__ dscore, float* __restrict__ dq_part, float* __restrict__ dv_chunk,
    float* __restrict__ dencproj, float* __restrict__ denc, int Tq, int S, int A, int E, int nchunk) {
  __shared__ float sds[ATB_CH], sw[ATB_CH];
  const int n = blockIdx.x, ch = blockIdx.y;
  const int s0 = ch * ATB_CH;
  const int cnt = min(ATB_CH, S - s0);
  const float* P = encproj + ((long)n * S + s0) * A;
  float pt[ATB_CH][ATB_SLOTS], dPa[ATB_CH][ATB_SLOTS], dHa[ATB_CH][ATB_SLOTS], dva[ATB_SLOTS], va[ATB_SLOTS];
#pragma unroll
  for (int k = 0; k < ATB_SLOTS; ++k) {
    const int a = threadIdx.x + k * ATB_THREADS;
    dva[k] = 0.f;
    va[k] = a < A ? v[a] : 0.f;
#pragma unroll
    for (int c = 0; c < ATB_CH; ++c) {
      pt[c][k] = (a < A && c < cnt) ? P[(long)c * A + a] : 0.f;
      dPa[c][k] = 0.f; dHa[c][k] = 0.f;
    }
  }
  for (int j = 0; j < Tq; ++j) {
    const long r = (long)n * Tq + j;
    __syncthreads();
    if (threadIdx.x < ATB_CH) {
      const bool ok = (int)threadIdx.x < cnt;
      sds[threadIdx.x] = ok ? dscore[r * S + s0 + threadIdx.x] : 0.f;
      sw[threadIdx.x] = ok ? weights[n * w_sn + j * w_sj + s0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    const float* q = qproj + n * q_sn + j * q_sj;
    const float* dc = dctx + n * dc_sn + j * dc_sj;
#pragma unroll
    for (int k = 0; k < ATB_SLOTS; ++k) {
      const int a = threadIdx.x + k * ATB_THREADS;
      if (a < A) {
        const float qa = q[a];
        float dqa = 0.f;
#pragma unroll
        for (int c = 0; c < ATB_CH; ++c) {
          const float th = tanh_att(qa + pt[c][k]);
          const float g = sds[c];
          const float du = g * va[k] * (1.f - th * th);
          dPa[c][k] += du;
          dqa += du;
          dva[k] += g * th;
        }
        dq_part[(r * nchunk + ch) * A + a] = dqa;
      }
      const int e = threadIdx.x + k * ATB_THREADS;
      if (e < E) {
        const float d = dc[e];
#pragma unroll
        for (int c = 0; c < ATB_CH; ++c) dHa[c][k] += sw[c] * d;
      }
    }
  }
  float* dP = dencproj + ((long)n * S + s0) * A;
  float* dH = denc + ((long)n * S + s0) * E;
#pragma unroll
  for (int k = 0; k < ATB_SLOTS; ++k) {
    const int a = threadIdx.x + k * ATB_THREADS;
    if (a < A) {
      dv_chunk[((long)n * nchunk + ch) * A + a] = dva[k];
#pragma unroll
      for (int c = 0; c < ATB_CH; ++c)
        if (c < cnt) dP[(long)c * A + a] += dPa[c][k];
    }
    if (a < E) {
#pragma unroll
      for (int c = 0; c < ATB_CH; ++c)
        if (c < cnt) dH[(long)c * E + a] += dHa[c][k];
    }
  }
}

__global__ void attn_bwd_reduce_kernel(const float* __restrict__ dq_part, const float* __restrict__ dv_chunk,
                                       float* __restrict__ dqproj, long dq_sn, long dq_sj,
                                       float* __restrict__ dv_part, int Tq, int A, int nchunk) {
  const int n = blockIdx.x / Tq, j = blockIdx.x % Tq;
  const long r = blockIdx.x;
  float* dq = dqproj + n * dq_sn + j * dq_sj;
  // The partials are loaded eight at a time (predicated): the kernel is nothing but the latency of these loads, and a
  // loop with a run-time trip count takes them one round trip after the other (decode backward 2.39 -> 2.27 ms).  The
  // same treatment of the score kernels (rows of several frames in flight, context rows fetched before the scores
  // exist) measured SLOWER (decode forward 1.16 -> 1.24 ms) and was dropped.
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    float acc = 0.f, dv = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {
      float p[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u < nchunk ? c0 + u : c0;
        p[u] = dq_part[(r * nchunk + c) * A + a];
        q[u] = j == 0 ? dv_chunk[((long)n * nchunk + c) * A + a] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u < nchunk) { acc += p[u]; dv += q[u]; }
    }
    dq[a] = acc;
    if (j == 0) dv_part[(long)n * A + a] += dv;
  }
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
  const int threads = (long)N * Tq < 256 ? ATT_THREADS_BIG : ATT_THREADS;
  const bool vec = (A & 3) == 0 && (E & 3) == 0 && E / 4 <= threads && aligned16(enc);
  const int groups = vec ? threads / (E / 4) : 0;
  const size_t shm = (size_t)(((S + 16 + 3) & ~3) + (long)groups * E) * sizeof(float);
  if (shm > 64 * 1024) return ACVAE_EUNSUPPORTED;
  if (vec)
    hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(N * Tq), dim3(threads), shm, (hipStream_t)stream, qproj, q_sn, q_sj,
                       encproj, enc, lens, v, ctx, c_sn, c_sj, weights, w_sn, w_sj, Tq, S, A, E);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(N * Tq), dim3(threads), shm, (hipStream_t)stream, qproj, q_sn,
                       q_sj, encproj, enc, lens, v, ctx, c_sn, c_sj, weights, w_sn, w_sj, Tq, S, A, E);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int64_t acvae_attn_bwd_workspace_bytes(int N, int Tq, int S, int A) {
  if (N <= 0 || Tq <= 0 || S <= 0 || A <= 0) return -1;
  const long nchunk = (S + ATB_CH - 1) / ATB_CH;
  return ((long)N * Tq * S + (long)N * Tq * nchunk * A + (long)N * nchunk * A + 64) * (int64_t)sizeof(float);
}

extern "C" int acvae_attn_bwd(const float* dctx, int64_t dc_sn, int64_t dc_sj, const float* qproj, int64_t q_sn,
                              int64_t q_sj, const float* encproj, const float* enc, const int64_t* lens,
                              const float* v, const float* weights, int64_t w_sn, int64_t w_sj, float* dqproj,
                              int64_t dq_sn, int64_t dq_sj, float* dencproj, float* denc, float* dv_part, float* ws,
                              int64_t ws_bytes, int N, int Tq, int S, int A, int E, void* stream) {
  if (!dctx || !qproj || !encproj || !enc || !lens || !v || !weights || !dqproj || !dencproj || !denc || !dv_part ||
      !ws)
    return ACVAE_EINVAL;
  if (N <= 0 || Tq <= 0 || S <= 0 || A <= 0 || E <= 0) return ACVAE_EINVAL;
  if (S > 8192 || A > ATB_SLOTS * ATB_THREADS || E > ATB_SLOTS * ATB_THREADS) return ACVAE_EUNSUPPORTED;
  if (ws_bytes < acvae_attn_bwd_workspace_bytes(N, Tq, S, A)) return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = (S + ATB_CH - 1) / ATB_CH;
  float* dscore = ws;
  float* dq_part = dscore + (long)N * Tq * S;
  float* dv_chunk = dq_part + (long)N * Tq * nchunk * A;
  const int threads = (long)N * Tq < 256 ? ATT_THREADS_BIG : ATT_THREADS;
  const bool vec = (E & 3) == 0 && aligned16(dctx) && aligned16(enc) && !(dc_sn & 3) && !(dc_sj & 3);
  if (vec)
    hipLaunchKernelGGL(attn_bwd_score_kernel<true>, dim3(N * Tq), dim3(threads), (size_t)(S + 16) * sizeof(float), st,
                       dctx, dc_sn, dc_sj, enc, lens, weights, w_sn, w_sj, dscore, Tq, S, E);
  else
    hipLaunchKernelGGL(attn_bwd_score_kernel<false>, dim3(N * Tq), dim3(threads), (size_t)(S + 16) * sizeof(float), st,
                       dctx, dc_sn, dc_sj, enc, lens, weights, w_sn, w_sj, dscore, Tq, S, E);
  hipLaunchKernelGGL(attn_bwd_accum_kernel, dim3(N, nchunk), dim3(ATB_THREADS), 0, st, dctx, dc_sn, dc_sj, qproj, q_sn,
                     q_sj, encproj, v, weights, w_sn, w_sj, dscore, dq_part, dv_chunk, dencproj, denc, Tq, S, A, E,
                     nchunk);
  hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3(N * Tq), dim3(256), 0, st, dq_part, dv_chunk, dqproj, dq_sn, dq_sj,
                     dv_part, Tq, A, nchunk);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
