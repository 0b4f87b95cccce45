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
#include <type_traits>
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
  // ---- scores: wave per frame, lanes over A.  The loop is a chain of L2 round trips (one decode step: 12 frames per wavefront),
  // so with A = 256 / 512 the encproj rows of four frames are fetched together; the arithmetic and its order are unchanged.
  auto scores4 = [&](auto na) {
    constexpr int NA = decltype(na)::value;            // A = 256 NA: every lane has NA float4 of a row, no tail conditions
    float4 qr[NA], vr[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      qr[i] = *reinterpret_cast<const float4*>(q + lane * 4 + 256 * i);
      vr[i] = *reinterpret_cast<const float4*>(v + lane * 4 + 256 * i);
    }
    for (int sb = wave; sb < S; sb += 4 * nw) {
      float4 pr[4][NA];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = sb + u * nw;
        const float* p = P + (long)(s < S ? s : sb) * A + lane * 4;       // past the end: a row that is there
#pragma unroll
        for (int i = 0; i < NA; ++i) pr[u][i] = *reinterpret_cast<const float4*>(p + 256 * i);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = sb + u * nw;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i)
          acc += vr[i].x * tanh_att(qr[i].x + pr[u][i].x) + vr[i].y * tanh_att(qr[i].y + pr[u][i].y) +
                 vr[i].z * tanh_att(qr[i].z + pr[u][i].z) + vr[i].w * tanh_att(qr[i].w + pr[u][i].w);
        acc = wave_sum(acc);
        if (lane == 0 && s < S) sc[s] = (s < len) ? acc : -1e10f;  // masked_fill(mask == 0, -1e10)
      }
    }
  };
  if (VEC && A == 512) {
    scores4(std::integral_constant<int, 2>());
  } else if (VEC && A == 256) {
    scores4(std::integral_constant<int, 1>());
  } else
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
      for (int sb = g; sb < S; sb += 8 * G) {            // eight enc rows in flight; same order of accumulation
        float4 h[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s0 = sb + u * G;
          h[u] = *reinterpret_cast<const float4*>(Hn + (long)(s0 < S ? s0 : sb) * E + e4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s0 = sb + u * G;
          if (s0 < S) {
            const float w = sc[s0];
            acc.x += w * h[u].x; acc.y += w * h[u].y; acc.z += w * h[u].z; acc.w += w * h[u].w;
          }
        }
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

// ---------------------------------------------------------------- few query rows: the frames split over workgroups
// One decode step (step API, beam search, sampled decode) has N * Tq = 16..160 query rows: with a workgroup per row most of
// the 256 CUs idle and each wavefront walks a dozen frames one load round trip after the other (27 us at N = 16, S = 187,
// profiles/r02_c4_t3000.json).  Here a row's S frames are split over `nsplit` workgroups of ATS_CHUNK frames or fewer (all loads of a
// workgroup are in flight at once) and the softmax is combined from per-split (max, sum, unnormalised context):
//   split k: m_k = max_s score, e_s = exp(score - m_k), l_k = sum e_s, c_k = sum e_s enc_s
//   row    : M = max m_k, L = sum_k l_k exp(m_k - M), weights_s = e_s exp(m_k - M) / L, ctx = sum_k c_k exp(m_k - M) / L
// The workgroup that arrives last at the row's counter does the combine, reading the splits in index order: the result does
// not depend on which one that is.  The partials travel as write-through stores / L1-bypassing loads (relaxed agent-scope
// atomics: sc1), ordered by s_waitcnt vmcnt(0) + barrier + the counter's atomic - no L2 write-back / invalidate.
constexpr int ATS_THREADS = 256;
constexpr int ATS_CHUNK = 16;           // frames per split: four per wavefront
#define ATS_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void ats_st(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), ATS_RLX_AGENT);
}
__device__ __forceinline__ float ats_ld(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), ATS_RLX_AGENT));
}

__global__ __launch_bounds__(ATS_THREADS) void attn_fwd_split_kernel(
    const float* __restrict__ qproj, long q_sn, long q_sj, const float* __restrict__ encproj,
    const float* __restrict__ enc, const int64_t* __restrict__ lens, const float* __restrict__ v,
    float* __restrict__ ctx, long c_sn, long c_sj, float* __restrict__ weights, long w_sn, long w_sj, int Tq, int S,
    int A, int E, int nsplit, float* __restrict__ part, unsigned* __restrict__ cnt) {
  extern __shared__ float smem[];            // [ATS_CHUNK] e_s | [16] reduction scratch | [nsplit] factors | [G][E] context partials
  float* sc = smem;
  float* red = smem + ATS_CHUNK;
  float* fac = red + 16;
  float* cpart = fac + ((nsplit + 3) & ~3);
  __shared__ int s_last;
  const int row = blockIdx.x / nsplit, sp = blockIdx.x - row * nsplit;
  const int n = row / Tq, j = row - n * Tq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s0 = sp * ATS_CHUNK, ns = min(ATS_CHUNK, S - s0);
  const float* q = qproj + n * q_sn + j * q_sj;
  const float* P = encproj + ((long)n * S + s0) * A;
  const float* Hn = enc + ((long)n * S + s0) * E;
  const int len = (int)lens[n];
  // context operands first: they depend on nothing (E/4 threads per frame, G frames at a time), at most ATS_CHUNK / G in flight
  const int ev = E >> 2, G = ATS_THREADS / ev;
  const int g = tid / ev, e4 = (tid - g * ev) * 4;
  constexpr int HMAX = 8;                    // frames per thread group kept in registers (G >= 2: E <= 512)
  float4 hreg[HMAX];
  const bool hregs = g < G && ATS_CHUNK <= HMAX * G;
  if (hregs) {
#pragma unroll
    for (int i = 0; i < HMAX; ++i) {
      const int s = g + i * G;
      hreg[i] = s < ns ? *reinterpret_cast<const float4*>(Hn + (long)s * E + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // ---- scores of this split: wave per frame, two frames per pass
  for (int s = wave; s < ns; s += 8) {
    const bool two = s + 4 < ns;
    const float* p0 = P + (long)s * A;
    const float* p1 = P + (long)(two ? s + 4 : s) * A;
    float a0 = 0.f, a1 = 0.f;
    for (int a = lane * 4; a < A; a += 256) {
      const float4 x0 = *reinterpret_cast<const float4*>(p0 + a);
      const float4 x1 = *reinterpret_cast<const float4*>(p1 + a);
      const float4 qv = *reinterpret_cast<const float4*>(q + a);
      const float4 vv = *reinterpret_cast<const float4*>(v + a);
      a0 += vv.x * tanh_att(qv.x + x0.x) + vv.y * tanh_att(qv.y + x0.y) + vv.z * tanh_att(qv.z + x0.z) +
            vv.w * tanh_att(qv.w + x0.w);
      a1 += vv.x * tanh_att(qv.x + x1.x) + vv.y * tanh_att(qv.y + x1.y) + vv.z * tanh_att(qv.z + x1.z) +
            vv.w * tanh_att(qv.w + x1.w);
    }
    a0 = row16_sum(a0);                                  // four DPP adds, then two LDS-pipe shuffles across the rows
    a1 = row16_sum(a1);
    a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64);
    a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64);
    if (lane == 0) {
      sc[s] = (s0 + s < len) ? a0 : -1e10f;            // masked_fill(mask == 0, -1e10)
      if (two) sc[s + 4] = (s0 + s + 4 < len) ? a1 : -1e10f;
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int s = 0; s < ns; ++s) m = fmaxf(m, sc[s]);     // ns <= 16 LDS broadcasts: cheaper than a block reduction
  float l = 0.f;
  for (int s = 0; s < ns; ++s) l += expf(sc[s] - m);
  __syncthreads();
  if (tid < ns) sc[tid] = expf(sc[tid] - m);
  __syncthreads();
  float* wout = weights + n * w_sn + j * w_sj + s0;
  if (tid < ns) ats_st(wout + tid, sc[tid]);
  // ---- unnormalised context of this split
  if (g < G) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hregs) {
#pragma unroll
      for (int i = 0; i < HMAX; ++i) {
        const int s = g + i * G;
        const float w = s < ns ? sc[s] : 0.f;
        acc.x += w * hreg[i].x; acc.y += w * hreg[i].y; acc.z += w * hreg[i].z; acc.w += w * hreg[i].w;
      }
    } else {
      for (int s = g; s < ns; s += G) {
        const float4 h = *reinterpret_cast<const float4*>(Hn + (long)s * E + e4);
        const float w = sc[s];
        acc.x += w * h.x; acc.y += w * h.y; acc.z += w * h.z; acc.w += w * h.w;
      }
    }
    *reinterpret_cast<float4*>(cpart + (long)g * E + e4) = acc;
  }
  __syncthreads();
  float* mine = part + ((long)row * nsplit + sp) * (E + 4);
  for (int e = tid; e < E; e += ATS_THREADS) {
    float acc = 0.f;
    for (int k = 0; k < G; ++k) acc += cpart[(long)k * E + e];
    ats_st(mine + e, acc);
  }
  if (tid == 0) { ats_st(mine + E, m); ats_st(mine + E + 1, l); }
  // ---- arrive; the last workgroup of the row combines
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = __hip_atomic_fetch_add(cnt + row, 1u, ATS_RLX_AGENT) == (unsigned)(nsplit - 1);
  __syncthreads();
  if (!s_last) return;
  // Everything the combine reads is fetched in ONE round trip (the stores above were write-through, so these loads go to the
  // memory side: ~1-2 us each way): thread k < nsplit takes (m_k, l_k), the first E/4 threads the context partials of up to
  // ATS_KB splits as float4 (registers), every thread its share of the e_s.
  const float* rowpart = part + (long)row * nsplit * (E + 4);
  float* wrow = weights + n * w_sn + j * w_sj;
  float* mk = cpart;                 // [nsplit] maxima, then [nsplit] sums (cpart is free again)
  float* lk = cpart + nsplit;
  constexpr int ATS_KB = 12;
  float4 pk[ATS_KB];
  const bool cth = tid < ev;
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowpart), 0, 0x7fffffff, 0x00020000);
  auto ld4 = [&](int k) {
    typedef unsigned ats_v4u __attribute__((ext_vector_type(4)));
    const ats_v4u r = __builtin_amdgcn_raw_buffer_load_b128(prs, (k * (E + 4) + tid * 4) * 4, 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
  };
  if (cth) {
#pragma unroll
    for (int kk = 0; kk < ATS_KB; ++kk) pk[kk] = kk < nsplit ? ld4(kk) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float es[2];
  es[0] = tid < S ? ats_ld(wrow + tid) : 0.f;
  es[1] = tid + ATS_THREADS < S ? ats_ld(wrow + tid + ATS_THREADS) : 0.f;
  for (int k = tid; k < nsplit; k += ATS_THREADS) {
    mk[k] = ats_ld(rowpart + (long)k * (E + 4) + E);
    lk[k] = ats_ld(rowpart + (long)k * (E + 4) + E + 1);
  }
  __syncthreads();
  if (tid == 0) {
    float M = -INFINITY;
    for (int k = 0; k < nsplit; ++k) M = fmaxf(M, mk[k]);
    float L = 0.f;
    for (int k = 0; k < nsplit; ++k) {
      const float f = expf(mk[k] - M);
      fac[k] = f;
      L += lk[k] * f;
    }
    const float inv = 1.f / L;
    for (int k = 0; k < nsplit; ++k) fac[k] *= inv;
    __hip_atomic_store(cnt + row, 0u, ATS_RLX_AGENT);        // ready for the next launch on this stream
  }
  __syncthreads();
  if (cth) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kk = 0; kk < ATS_KB; ++kk) {
      const float f = kk < nsplit ? fac[kk] : 0.f;
      acc.x += f * pk[kk].x; acc.y += f * pk[kk].y; acc.z += f * pk[kk].z; acc.w += f * pk[kk].w;
    }
    for (int k0 = ATS_KB; k0 < nsplit; k0 += ATS_KB) {        // S > 192: further batches, one round trip each
#pragma unroll
      for (int kk = 0; kk < ATS_KB; ++kk) pk[kk] = k0 + kk < nsplit ? ld4(k0 + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int kk = 0; kk < ATS_KB; ++kk) {
        const float f = k0 + kk < nsplit ? fac[k0 + kk] : 0.f;
        acc.x += f * pk[kk].x; acc.y += f * pk[kk].y; acc.z += f * pk[kk].z; acc.w += f * pk[kk].w;
      }
    }
    *reinterpret_cast<float4*>(ctx + n * c_sn + j * c_sj + tid * 4) = acc;
  }
  if (tid < S) wrow[tid] = es[0] * fac[tid / ATS_CHUNK];
  if (tid + ATS_THREADS < S) wrow[tid + ATS_THREADS] = es[1] * fac[(tid + ATS_THREADS) / ATS_CHUNK];
  for (int s2 = tid + 2 * ATS_THREADS; s2 < S; s2 += ATS_THREADS) wrow[s2] = ats_ld(wrow + s2) * fac[s2 / ATS_CHUNK];
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

// Workspace of the split kernel (caller-owned, acvae_attn_fwd_workspace_bytes): ATS_CNT_WORDS arrival counters, one per
// query row, then [rows][nsplit][E + 4] partials (context, max, sum, pad).  The counters must be zero when a call starts
// and are zero again when it has run (the combining workgroup resets its row's word), so a caller zeroes the first
// ATS_CNT_WORDS * 4 bytes ONCE and may then reuse the workspace for any number of stream-ordered calls; two calls that may
// run side by side (two streams) need a workspace each.
constexpr int ATS_CNT_WORDS = 256;
inline bool attn_split_shape(long rows, int S, int A, int E) {
  const int nsplit = (S + ATS_CHUNK - 1) / ATS_CHUNK;
  const int G = (E & 3) == 0 && E >= 4 && E / 4 <= ATS_THREADS ? ATS_THREADS / (E / 4) : 0;
  return (A & 3) == 0 && (E & 3) == 0 && G > 0 && rows * 2 <= ATS_CNT_WORDS && nsplit >= 2 && nsplit <= 512 &&
         (size_t)(ATS_CHUNK + 16 + ((nsplit + 3) & ~3) + (long)G * E) * sizeof(float) <= 64 * 1024;
}

}  // namespace

extern "C" int64_t acvae_attn_fwd_workspace_bytes(int N, int Tq, int S, int A, int E) {
  if (N <= 0 || Tq <= 0 || S <= 0 || A <= 0 || E <= 0) return -1;
  const long rows = (long)N * Tq;
  if (!attn_split_shape(rows, S, A, E)) return 0;          // the one-workgroup form needs none
  const long nsplit = (S + ATS_CHUNK - 1) / ATS_CHUNK;
  return (int64_t)ATS_CNT_WORDS * 4 + rows * nsplit * (E + 4) * (int64_t)sizeof(float);
}

extern "C" int acvae_attn_fwd(const float* qproj, int64_t q_sn, int64_t q_sj, const float* encproj, const float* enc,
                              const int64_t* lens, const float* v, float* ctx, int64_t c_sn, int64_t c_sj,
                              float* weights, int64_t w_sn, int64_t w_sj, int N, int Tq, int S, int A, int E, void* ws,
                              int64_t ws_bytes, void* stream, int flags) {
  if (!qproj || !encproj || !enc || !lens || !v || !ctx || !weights) return ACVAE_EINVAL;
  if (N <= 0 || Tq <= 0 || S <= 0 || A <= 0 || E <= 0) return ACVAE_EINVAL;
  if (S > 8192) return ACVAE_EUNSUPPORTED;
  if ((A & 3) == 0 && (!aligned16(qproj) || !aligned16(encproj) || !aligned16(v) || (q_sn & 3) || (q_sj & 3)))
    return ACVAE_EALIGN;
  const int threads = (long)N * Tq < 256 ? ATT_THREADS_BIG : ATT_THREADS;
  const bool vec = (A & 3) == 0 && (E & 3) == 0 && E / 4 <= threads && aligned16(enc);
  // few rows and more than one chunk of frames: split S over workgroups (see attn_fwd_split_kernel) - when the caller
  // handed over the workspace for it; without one (or with ACVAE_FLAG_NO_ATTN_SPLIT) the one-workgroup form runs
  const int nsplit = (S + ATS_CHUNK - 1) / ATS_CHUNK;
  if (ws && !(flags & ACVAE_FLAG_NO_ATTN_SPLIT) && vec && attn_split_shape((long)N * Tq, S, A, E) && aligned16(ctx) &&
      aligned16(ws) && (c_sn & 3) == 0 && (c_sj & 3) == 0) {
    if (ws_bytes < acvae_attn_fwd_workspace_bytes(N, Tq, S, A, E)) return ACVAE_EWORKSPACE;
    unsigned* cnt = reinterpret_cast<unsigned*>(ws);
    float* part = reinterpret_cast<float*>(ws) + ATS_CNT_WORDS;
    const int G = ATS_THREADS / (E / 4);
    const size_t shm = (size_t)(ATS_CHUNK + 16 + ((nsplit + 3) & ~3) + (long)G * E) * sizeof(float);
    hipLaunchKernelGGL(attn_fwd_split_kernel, dim3(N * Tq * nsplit), dim3(ATS_THREADS), shm, (hipStream_t)stream, qproj, q_sn,
                       q_sj, encproj, enc, lens, v, ctx, c_sn, c_sj, weights, w_sn, w_sj, Tq, S, A, E, nsplit, part, cnt);
    ACVAE_LAUNCH_CHECK();
    return ACVAE_OK;
  }
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

namespace {
__global__ void tanh_att_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = tanh_att(x[i]);
}
}  // namespace
extern "C" int acvae_tanh_att(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(tanh_att_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
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
