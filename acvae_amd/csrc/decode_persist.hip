// A4 / A5 / A7: the teacher-forced decode loop (models/vae_model.py:700-730,792-816 with PriorRNN.forward
// models/text_encoder.py:247-268 and VAERNNBahdanauAttnDecoder.forward models/decoder.py:175-203) as ONE persistent launch.
//
// The per-step form (decoder.hip) queues four dependent kernels per step and chain - 2 x 4 x 21 launches of 5-17 us each
// for microseconds of work: every kernel pays its own start, its own first-load latency and a boundary.  Here the Tc
// steps of both chains run inside one kernel whose workgroups keep FIXED roles and hand their results to the next role
// through global memory:
//
//   decoder chain                                               prior chain
//   D1 [A/32 + 3H/32 wgs]  qd = h.Watt_q^T, gh = h.Whh^T + b     P1 [Hp/8 wgs]  gates += z.Wih_z^T + hp.Whh^T + b, LSTM cell
//   D2 [N wgs]             attention of clip n -> ctx, weights   P2 [E/16 wgs]  [mu, logvar] = hp.Wml^T + b, z = eps.exp(lv/2) + mu
//   D3 [H/16 wgs]          gi += ctx.Wih_ctx^T, GRU cell -> h
//
// Every product is the arithmetic of gemm_skinny_kernel (one 32 x 32 tile, K dealt over 8 wavefronts in groups of 8,
// partial tiles summed in wave order), the attention is attn_fwd_kernel's, the cells are rnn.hip's: results are
// bit-identical to the per-step path, which stays for scheduled sampling, inference and shapes this kernel does not take
// and is its parity check (tests/test_decode_persist_gpu.py).
//
// Hand-offs (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md, Valid forms): every handed-off value is written by
// an agent-scope atomic store (write-through, `sc1`) and read by an agent-scope atomic load (bypasses the CU's L1, which
// no other CU's store ever refreshes); every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at
// a barrier, ONE lane adds to the step's arrival counter; the consumer's lane 0 polls that ONE word relaxed with s_sleep,
// then the workgroup's barrier.  No fence anywhere.  Counters are per (role, step), zeroed by the launcher's memset:
// nothing is ever reset or reused inside the launch.  All workgroups (160 + N + ...) fit the chip at once (512 threads,
// 34 KB of LDS: two per CU would fit); every spin is bounded and raises an abort word that ends all roles.
//
// Also in this file: the backward through time of the same loop (decode_persist_bwd_kernel: roles RA / RB / RC / PA / PB, K-splits
// with the combine on the reader's side) and the posterior's bidirectional GRU, forward and backward, as one launch per pass
// (posterior_persist_fwd / _bwd_kernel: one role, one hand-off per step and direction).
#include "mfma_tile.h"
#include "common.h"
#include "../../include/acvae_hip.h"
#include "decode_persist.h"
#include <atomic>
#include <mutex>

namespace {
using namespace mfma;

constexpr int PD_THREADS = 512;          // 8 wavefronts, as gemm_skinny_kernel
constexpr unsigned PD_SPIN_LIMIT = 1u << 24;   // polls x ~150 ns: seconds; never reached unless part of the grid is not resident

#define PD_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), PD_RLX_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), PD_RLX_AGENT));
}
// 16 bytes of handed-off data: buffer_load_dwordx4 ... sc1 (bypasses this CU's L1, like the atomic loads; one instruction
// instead of two 8-byte ones).  base: a wave-uniform pointer (kernel argument), idx: the lane's float index from it (< 2^29).
typedef unsigned pd_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_sc1_4(const float* base, long idx) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  const pd_v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 4), 0, 16 /* sc1 */);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Lane 0 waits until *cnt >= target (relaxed polls, s_sleep between them), then the workgroup's barrier.  Returns false
// when the launch is aborting (a bounded spin ran out somewhere): every role then leaves its loop.
__device__ __forceinline__ bool pd_wait(const unsigned* cnt, unsigned target, unsigned* abort_word, unsigned spin_limit, int* s_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    for (unsigned spins = 0; __hip_atomic_load(cnt, PD_RLX_AGENT) < target;) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0u && (spins > spin_limit || __hip_atomic_load(abort_word, PD_RLX_AGENT) != 0u)) {
        __hip_atomic_store(abort_word, 1u, PD_RLX_AGENT);
        ok = 0;
        break;
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  __syncthreads();          // s_flag may be rewritten by the next wait
  return ok;
}
// Every wave has drained its stores; one lane signals.
__device__ __forceinline__ void pd_arrive(unsigned* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, PD_RLX_AGENT);
}

// acc += A[32 rows][K] . B[32 rows][K]^T for this wave's share of the K-groups: the arithmetic of sk_accumulate<true>
// (mfma_tile.h: groups wave, wave + 8, .. in rising order, four MFMAs per group) with per-lane row pointers (ap / bp point at
// the lane's row, + 4 * (lane >> 5)).  A round trip to handed-off data costs 2-3 us (write-through stores drop the line from
// the L2s: the load goes to the memory side), so the loads are split by what they depend on: the weight fragments of the
// first PD_U groups are fetched BEFORE the role waits for its input (pd_fetch_b), the input fragments in ONE batch behind
// the wait (hipcc, left alone, waits for the first two groups, starts the MFMAs and fetches the rest behind a second wait).
constexpr int PD_U = 8;
struct PdFrag { float4 b[PD_U]; };
__device__ __forceinline__ void pd_fetch_b(PdFrag& f, const float* bp, int K, int wave) {
  const int Gfull = K / 8;
#pragma unroll
  for (int u = 0; u < PD_U; ++u) {
    const int gu = wave + u * SK_WAVES;
    f.b[u] = *reinterpret_cast<const float4*>(bp + ((gu < Gfull) ? (long)gu * 8 : 0));
  }
}
template <bool HANDED>
__device__ __forceinline__ void pd_accumulate(f32x16& acc, const float* abase, long aidx, const float* bp, int K, int wave) {
  const int Gfull = K / 8;
  for (int g = wave; g < Gfull; g += PD_U * SK_WAVES) {
    float4 a[PD_U], b[PD_U];
#pragma unroll
    for (int u = 0; u < PD_U; ++u) {
      const int gu = g + u * SK_WAVES;
      const long off = (gu < Gfull) ? (long)gu * 8 : 0;
      b[u] = *reinterpret_cast<const float4*>(bp + off);
      a[u] = HANDED ? ld_sc1_4(abase, aidx + off) : *reinterpret_cast<const float4*>(abase + aidx + off);
    }
    __builtin_amdgcn_sched_barrier(0);          // every load of the batch is in flight before the first MFMA waits
#pragma unroll
    for (int u = 0; u < PD_U; ++u) {
      const bool ok = (g + u * SK_WAVES) < Gfull;
      const float4 av = ok ? a[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[u].w, acc, 0, 0, 0);
    }
  }
}
// K <= 512: a wave's whole share is ONE batch of PD_U groups; the two halves of pd_accumulate separately, so that a role can
// issue the input loads of one product before it waits for the input of another (P1: hp before z)
template <bool HANDED>
__device__ __forceinline__ void pd_load_a(float4 (&a)[PD_U], const float* abase, long aidx, int K, int wave) {
  const int Gfull = K / 8;
#pragma unroll
  for (int u = 0; u < PD_U; ++u) {
    const int gu = wave + u * SK_WAVES;
    const long off = (gu < Gfull) ? (long)gu * 8 : 0;
    a[u] = HANDED ? ld_sc1_4(abase, aidx + off) : *reinterpret_cast<const float4*>(abase + aidx + off);
  }
}
__device__ __forceinline__ void pd_mfma(f32x16& acc, const float4 (&a)[PD_U], const PdFrag& f, int K, int wave) {
  const int Gfull = K / 8;
#pragma unroll
  for (int u = 0; u < PD_U; ++u) {
    const bool ok = (wave + u * SK_WAVES) < Gfull;
    const float4 av = ok ? a[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, f.b[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, f.b[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, f.b[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, f.b[u].w, acc, 0, 0, 0);
  }
}
// one product of a role: resident weight fragments when the wave's share is one batch, else the streaming loop
template <bool HANDED>
__device__ __forceinline__ void pd_product(f32x16& acc, const float* abase, long aidx, const float* bp, int K, int wave, const PdFrag& f) {
  if (K <= 64 * PD_U) {
    float4 a[PD_U];
    pd_load_a<HANDED>(a, abase, aidx, K, wave);
    __builtin_amdgcn_sched_barrier(0);
    pd_mfma(acc, a, f, K, wave);
  } else {
    pd_accumulate<HANDED>(acc, abase, aidx, bp, K, wave);
  }
}
// the wave's accumulator tile into red[wave][row][col] (gemm_skinny_kernel's layout); sum over waves in order by the caller
__device__ __forceinline__ void pd_stash(float (*red)[32][33], const f32x16& acc, int wave, int li, int lh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
}
__device__ __forceinline__ float pd_sum(float (*red)[32][33], int mm, int nn) {
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < SK_WAVES; ++w) v += red[w][mm][nn];
  return v;
}

struct PdSmem {
  float red[SK_WAVES][32][33];     // one reduction tile; D3 runs its two tiles through it one after the other
  float keep[32][33];              // D3: the first tile's sums while the second is reduced
  int flag;
};

// What a role can fetch without its input - its weight fragments (for K <= 512 a wave's whole share: 8 groups = 32 registers,
// fetched ONCE and kept for all Tc steps), biases, the hoisted projections of the step - is fetched before the role waits;
// recurrent state of the cells (h, c) stays in the registers of the thread that owns the element.

// ---------------------------------------------------------------- D1: qd = h . Watt[:, :H]^T;  gh = h . Whh^T + bhh
__device__ void role_d1(const PdParams& p, int tile, PdSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int nq = p.A / 32;
  const bool is_q = tile < nq;
  const int n0 = (is_q ? tile : tile - nq) * 32;
  const float* B = is_q ? p.w_att : p.w_hh;
  const long ldb = is_q ? (long)(p.E + p.H) : (long)p.H;
  const float* bp = B + (long)(n0 + li) * ldb + 4 * lh;
  const int arow = li < p.N ? li : 0;            // rows past N are never stored: they may read row 0
  PdFrag fb;
  pd_fetch_b(fb, bp, p.H, wave);
  const float bias = is_q ? 0.f : p.b_hh[n0 + (threadIdx.x & 31)];      // both outputs of a thread share the column
  for (int t = 0; t < p.Tc; ++t) {
    if (t > 0 && !pd_wait(p.cnt + PD_C_D3 * p.Tc + (t - 1), (unsigned)p.n_d3, p.abort_word, p.spin_limit, &sm.flag)) return;
    const float* hprev = t ? p.outputs + (long)(t - 1) * p.H : p.zeros;
    const long ldh = t ? (long)p.Tc * p.H : (long)p.H;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (t) pd_product<true>(acc, hprev, arow * ldh + 4 * lh, bp, p.H, wave, fb);
    else pd_product<false>(acc, hprev, arow * ldh + 4 * lh, bp, p.H, wave, fb);
    pd_stash(sm.red, acc, wave, li, lh);
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += PD_THREADS) {
      const int mm = e >> 5, nn = e & 31;
      if (mm < p.N) {
        const float v = pd_sum(sm.red, mm, nn);
        if (is_q) st_sc1(p.qd + (long)mm * p.Tc * p.A + (long)t * p.A + n0 + nn, v);
        else st_sc1(p.gh + (long)mm * 3 * p.H + n0 + nn, v + bias);
      }
    }
    pd_arrive(p.cnt + (is_q ? PD_C_D1Q : PD_C_D1H) * p.Tc + t);
  }
}

// ---------------------------------------------------------------- D2: attention of clip n (attn_fwd_kernel<true> as launched for
// one decode step: 1024 threads = GV context groups; here 512 threads play two groups each).  RES: the clip's projected
// memory (S x A floats) stays in LDS and its memory rows in registers for all Tc steps - per step only the 2 KB query
// crosses the chip; otherwise (long audio: S x A x 4 bytes beyond the LDS) both stream from L2 every step.
template <bool RES>
__device__ void role_d2(const PdParams& p, int n, float* smem, int* s_flag) {
  const int S = p.S, A = p.A, E = p.E;
  float* sc = smem;
  float* red = smem + S;
  float* part = smem + ((S + 16 + 3) & ~3);
  float* Pl = part + 4096;                                  // RES: [S][A]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Pg = p.encproj + (long)n * S * A;
  const float* Hn = p.mem + (long)n * S * E;
  const float* v = p.att_v;
  const int len = (int)p.mem_lens[n];
  const int ev = E >> 2, GR = PD_THREADS / ev;          // real context groups (E = 512: 4)
  const int g0 = threadIdx.x / ev, e4 = (threadIdx.x - g0 * ev) * 4;
  const int GV = 1024 / ev;                             // groups of the 1024-thread per-step launch (E = 512: 8) = 2 GR
  float4 hm[2][8];                                      // RES: memory rows of this thread's two groups, frames g, g + GV, ..
  if (RES) {
    for (int i = threadIdx.x * 4; i < S * A; i += PD_THREADS * 4)
      *reinterpret_cast<float4*>(Pl + i) = *reinterpret_cast<const float4*>(Pg + i);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int s0 = g0 + k2 * GR + k * GV;
        hm[k2][k] = (g0 < GR && s0 < S) ? *reinterpret_cast<const float4*>(Hn + (long)s0 * E + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    __syncthreads();
  }
  const float* P = RES ? Pl : Pg;
  // a lane's share of A: elements lane * 4 + 256 i (A <= 2048); v once, the step's query once per step (ONE round trip)
  float4 vv[8], qv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int a = lane * 4 + 256 * i;
    vv[i] = a < A ? *reinterpret_cast<const float4*>(v + a) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int t = 0; t < p.Tc; ++t) {
    if (!pd_wait(p.cnt + PD_C_D1Q * p.Tc + t, (unsigned)(A / 32), p.abort_word, p.spin_limit, s_flag)) return;
    const long qi = (long)n * p.Tc * A + (long)t * A;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int a = lane * 4 + 256 * i;
      qv[i] = a < A ? ld_sc1_4(p.qd, qi + a) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- scores: wave per frame, lanes over A
    for (int s = wave; s < S; s += SK_WAVES) {
      const float* pr = P + (long)s * A;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int a = lane * 4 + 256 * i;
        if (a < A) {
          const float4 pv = *reinterpret_cast<const float4*>(pr + a);
          acc += vv[i].x * tanh_att(qv[i].x + pv.x) + vv[i].y * tanh_att(qv[i].y + pv.y) + vv[i].z * tanh_att(qv[i].z + pv.z) +
                 vv[i].w * tanh_att(qv[i].w + pv.w);
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) sc[s] = (s < len) ? acc : -1e10f;
    }
    __syncthreads();
    // ---- softmax over S (S <= PD_THREADS: one element per thread, as with the 1024-thread launch of the per-step path)
    float m = -INFINITY;
    for (int s = threadIdx.x; s < S; s += PD_THREADS) m = fmaxf(m, sc[s]);
    m = block_max(m, red);
    float sum = 0.f;
    for (int s = threadIdx.x; s < S; s += PD_THREADS) {
      const float e = expf(sc[s] - m);
      sc[s] = e;
      sum += e;
    }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    float* wout = p.attn_w + (long)n * p.Tc * S + (long)t * S;
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += PD_THREADS) {
      const float w = sc[s] * inv;
      sc[s] = w;
      wout[s] = w;
    }
    __syncthreads();
    // ---- context: group g takes frames g, g + GV, ..; partial sums meet in LDS in group order
    if (g0 < GR) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int g = g0 + k2 * GR;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RES) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int s0 = g + k * GV;
            if (s0 < S) {
              const float4 h = hm[k2][k];
              const float w = sc[s0];
              acc.x += w * h.x; acc.y += w * h.y; acc.z += w * h.z; acc.w += w * h.w;
            }
          }
        } else {
          for (int s0 = g; s0 < S; s0 += GV) {
            const float4 h = *reinterpret_cast<const float4*>(Hn + (long)s0 * E + e4);
            const float w = sc[s0];
            acc.x += w * h.x; acc.y += w * h.y; acc.z += w * h.z; acc.w += w * h.w;
          }
        }
        *reinterpret_cast<float4*>(part + (long)g * E + e4) = acc;
      }
    }
    __syncthreads();
    float* c = p.rnn_d + (long)n * p.Tc * 3 * E + (long)t * 3 * E + E;
    for (int e = threadIdx.x; e < E; e += PD_THREADS) {
      float acc = 0.f;
      for (int k = 0; k < GV; ++k) acc += part[(long)k * E + e];
      st_sc1(c + e, acc);
    }
    pd_arrive(p.cnt + PD_C_D2 * p.Tc + t);
  }
}

// ---------------------------------------------------------------- D3: gi += ctx . Wih[:, E:2E]^T, GRU cell (rnn.hip gru_fwd_kernel)
__device__ void role_d3(const PdParams& p, int slice, PdSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int H = p.H, E = p.E, u0 = slice * 16;
  // tile 1 columns: r of units u0..u0+15 | z of the same units;  tile 2: n of the units | 16 unused columns (row 0)
  const int row1 = li < 16 ? u0 + li : H + u0 + (li - 16);
  const int row2 = li < 16 ? 2 * H + u0 + li : 0;
  const float* bp1 = p.w_ih + (long)row1 * 3 * E + E + 4 * lh;
  const float* bp2 = p.w_ih + (long)row2 * 3 * E + E + 4 * lh;
  const int arow = li < p.N ? li : 0;
  PdFrag fb1, fb2;
  pd_fetch_b(fb1, bp1, E, wave);
  pd_fetch_b(fb2, bp2, E, wave);
  const int mm = threadIdx.x >> 4, j = threadIdx.x & 15, u = u0 + j;      // this thread's cell element (row, unit)
  const bool mine = mm < p.N;
  float h = 0.f;                                                          // h_{t-1}[mm][u]
  for (int t = 0; t < p.Tc; ++t) {
    // off the critical path: the hoisted projection of this step, and gh (D1 finished it before the attention started)
    float g_r = 0.f, g_z = 0.f, g_n = 0.f, b_r = 0.f, b_z = 0.f, ghn = 0.f;
    if (mine) {
      const float* gi = p.gi + (long)mm * p.Tc * 3 * H + (long)t * 3 * H;
      g_r = gi[u]; g_z = gi[H + u]; g_n = gi[2 * H + u];
    }
    if (!pd_wait(p.cnt + PD_C_D1H * p.Tc + t, (unsigned)(3 * H / 32), p.abort_word, p.spin_limit, &sm.flag)) return;
    if (mine) {
      const float* gh = p.gh + (long)mm * 3 * H;
      b_r = ld_sc1(gh + u); b_z = ld_sc1(gh + H + u); ghn = ld_sc1(gh + 2 * H + u);
    }
    if (!pd_wait(p.cnt + PD_C_D2 * p.Tc + t, (unsigned)p.N, p.abort_word, p.spin_limit, &sm.flag)) return;
    const long ai = (long)arow * p.Tc * 3 * E + (long)t * 3 * E + E + 4 * lh;
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    if (E <= 64 * PD_U) {                       // the context rows cross the chip ONCE and feed both tiles
      float4 a[PD_U];
      pd_load_a<true>(a, p.rnn_d, ai, E, wave);
      __builtin_amdgcn_sched_barrier(0);
      pd_mfma(acc, a, fb1, E, wave);
      pd_mfma(acc2, a, fb2, E, wave);
    } else {
      pd_accumulate<true>(acc, p.rnn_d, ai, bp1, E, wave);
      pd_accumulate<true>(acc2, p.rnn_d, ai, bp2, E, wave);
    }
    pd_stash(sm.red, acc, wave, li, lh);
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += PD_THREADS) sm.keep[e >> 5][e & 31] = pd_sum(sm.red, e >> 5, e & 31);
    __syncthreads();
    pd_stash(sm.red, acc2, wave, li, lh);
    __syncthreads();
    if (mine) {
      const float a_r = sm.keep[mm][j] + g_r;               // the per-step path: v = sum over waves; v += *p (accumulate)
      const float a_z = sm.keep[mm][16 + j] + g_z;
      const float a_n = pd_sum(sm.red, mm, j) + g_n;
      const float r = sigmoidf_(a_r + b_r);
      const float z = sigmoidf_(a_z + b_z);
      const float nn = tanhf(a_n + r * ghn);
      const float hn = (1.f - z) * nn + z * h;
      p.hprev_d[(long)mm * p.Tc * H + (long)t * H + u] = h;
      float* sv = p.gru_save + (long)mm * p.Tc * 4 * H + (long)t * 4 * H;
      sv[u] = r; sv[H + u] = z; sv[2 * H + u] = nn; sv[3 * H + u] = ghn;
      st_sc1(p.outputs + (long)mm * p.Tc * H + (long)t * H + u, hn);
      h = hn;
    }
    pd_arrive(p.cnt + PD_C_D3 * p.Tc + t);
  }
}

// ---------------------------------------------------------------- P1: gates += z . Wih[:, 2E:3E]^T + hp . Whh^T + bhh, LSTM cell
__device__ void role_p1(const PdParams& p, int slice, PdSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int Hp = p.E, E = p.E, u0 = slice * 8;
  const int row = (li >> 3) * Hp + u0 + (li & 7);            // column li = (gate li / 8, unit li % 8)
  const float* bz = p.pw_ih + (long)row * 3 * E + 2 * E + 4 * lh;
  const float* bh = p.pw_hh + (long)row * Hp + 4 * lh;
  const int arow = li < p.N ? li : 0;
  PdFrag fbz, fbh;
  pd_fetch_b(fbz, bz, E, wave);
  pd_fetch_b(fbh, bh, Hp, wave);
  const int mm = threadIdx.x >> 3, j = threadIdx.x & 7, u = u0 + j;      // cell element of threads 0..255
  const bool mine = threadIdx.x < 256 && mm < p.N;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (mine)
#pragma unroll
    for (int k = 0; k < 4; ++k) bias[k] = p.pb_hh[k * Hp + u];
  float c = 0.f;                                                          // c_{t-1}[mm][u]
  for (int t = 0; t < p.Tc; ++t) {
    float gpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (mine) {
      const float* g = p.gates_p + (long)mm * p.Tc * 4 * Hp + (long)t * 4 * Hp;
#pragma unroll
      for (int k = 0; k < 4; ++k) gpre[k] = g[k * Hp + u];
    }
    const long zi = (long)arow * p.Tc * 3 * E + (long)t * 3 * E + 2 * E + 4 * lh;
    const float* hprev = t ? p.hp_all + (long)(t - 1) * Hp : p.zeros;
    const long ldh = t ? (long)p.Tc * Hp : (long)Hp;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (E <= 64 * PD_U) {
      // hp_{t-1} is complete long before z_{t-1} (P2 works from it): fetch it behind this role's own counter, then wait for z
      float4 ah[PD_U], azv[PD_U];
      if (t > 0 && !pd_wait(p.cnt + PD_C_P1 * p.Tc + (t - 1), (unsigned)p.n_p1, p.abort_word, p.spin_limit, &sm.flag)) return;
      if (t) pd_load_a<true>(ah, hprev, arow * ldh + 4 * lh, Hp, wave);
      else pd_load_a<false>(ah, hprev, arow * ldh + 4 * lh, Hp, wave);
      if (t > 0 && !pd_wait(p.cnt + PD_C_P2 * p.Tc + (t - 1), (unsigned)p.n_p2, p.abort_word, p.spin_limit, &sm.flag)) return;
      if (t) pd_load_a<true>(azv, p.rnn_p, zi, E, wave);
      else pd_load_a<false>(azv, p.rnn_p, zi, E, wave);
      __builtin_amdgcn_sched_barrier(0);
      pd_mfma(acc, azv, fbz, E, wave);             // the per-step path's order: the z product, then the hp product
      pd_mfma(acc, ah, fbh, Hp, wave);
    } else {
      if (t > 0 && !pd_wait(p.cnt + PD_C_P2 * p.Tc + (t - 1), (unsigned)p.n_p2, p.abort_word, p.spin_limit, &sm.flag)) return;
      if (t) {
        pd_product<true>(acc, p.rnn_p, zi, bz, E, wave, fbz);
        pd_product<true>(acc, hprev, arow * ldh + 4 * lh, bh, Hp, wave, fbh);
      } else {
        pd_product<false>(acc, p.rnn_p, zi, bz, E, wave, fbz);
        pd_product<false>(acc, hprev, arow * ldh + 4 * lh, bh, Hp, wave, fbh);
      }
    }
    pd_stash(sm.red, acc, wave, li, lh);
    __syncthreads();
    if (mine) {
      float gv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = pd_sum(sm.red, mm, k * 8 + j);
        v += bias[k];
        v += gpre[k];
        gv[k] = v;
      }
      const float ig = sigmoidf_(gv[0]), fg = sigmoidf_(gv[1]), gg = tanhf(gv[2]), og = sigmoidf_(gv[3]);
      const float c2 = fg * c + ig * gg;
      const float tc = tanhf(c2);
      st_sc1(p.hp_all + (long)mm * p.Tc * Hp + (long)t * Hp + u, og * tc);
      p.c_all[(long)mm * p.Tc * Hp + (long)t * Hp + u] = c2;
      float* s = p.lstm_save + (long)mm * p.Tc * 5 * Hp + (long)t * 5 * Hp;
      s[u] = ig; s[Hp + u] = fg; s[2 * Hp + u] = gg; s[3 * Hp + u] = og; s[4 * Hp + u] = tc;
      c = c2;
    }
    pd_arrive(p.cnt + PD_C_P1 * p.Tc + t);
  }
}

// ---------------------------------------------------------------- P2: [mu | logvar] = hp . Wml^T + b, re-parameterisation
__device__ void role_p2(const PdParams& p, int slice, PdSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int Hp = p.E, E = p.E, e0 = slice * 16;
  const int row = li < 16 ? e0 + li : E + e0 + (li - 16);
  const float* bp = p.w_ml + (long)row * Hp + 4 * lh;
  const int arow = li < p.N ? li : 0;
  PdFrag fb;
  pd_fetch_b(fb, bp, Hp, wave);
  const int mm = threadIdx.x >> 4, j = threadIdx.x & 15, e = e0 + j;
  const bool mine = mm < p.N;
  const float b_mu = p.b_ml[e], b_lv = p.b_ml[E + e];
  for (int t = 0; t < p.Tc; ++t) {
    const float eps = mine ? p.eps_p[(long)t * p.N * E + (long)mm * E + e] : 0.f;
    if (!pd_wait(p.cnt + PD_C_P1 * p.Tc + t, (unsigned)p.n_p1, p.abort_word, p.spin_limit, &sm.flag)) return;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    pd_product<true>(acc, p.hp_all, (long)arow * p.Tc * Hp + (long)t * Hp + 4 * lh, bp, Hp, wave, fb);
    pd_stash(sm.red, acc, wave, li, lh);
    __syncthreads();
    if (mine) {
      const float mu = pd_sum(sm.red, mm, j) + b_mu;
      const float lv = pd_sum(sm.red, mm, 16 + j) + b_lv;
      const float zz = eps * expf(.5f * lv) + mu;
      const long o = (long)mm * p.Tc * E + (long)t * E + e;
      p.p_means[o] = mu; p.p_logs[o] = lv; p.p_z[o] = zz;
      if (t + 1 < p.Tc) st_sc1(p.rnn_p + (long)mm * p.Tc * 3 * E + (long)(t + 1) * 3 * E + 2 * E + e, zz);
    }
    pd_arrive(p.cnt + PD_C_P2 * p.Tc + t);
  }
}

__global__ __launch_bounds__(PD_THREADS) void decode_persist_kernel(PdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pd_smem_raw[];
  PdSmem& sm = *reinterpret_cast<PdSmem*>(pd_smem_raw);
  int b = blockIdx.x;
  if (b < p.n_d1) { role_d1(p, b, sm); return; }
  b -= p.n_d1;
  if (b < p.N) {
    float* sm2 = reinterpret_cast<float*>(pd_smem_raw) + 4;
    int* fl = reinterpret_cast<int*>(pd_smem_raw);
    if (p.att_resident) role_d2<true>(p, b, sm2, fl); else role_d2<false>(p, b, sm2, fl);
    return;
  }
  b -= p.N;
  if (b < p.n_d3) { role_d3(p, b, sm); return; }
  b -= p.n_d3;
  if (b < p.n_p1) { role_p1(p, b, sm); return; }
  b -= p.n_p1;
  role_p2(p, b, sm);
}


// =====================================================================================================================
// Backward through time of the same loop (acvae_decode_bwd's dec_bptt / prior_bptt, decoder.hip) as ONE persistent launch.
// 512 threads per workgroup (the K range of a product is dealt over 8 wavefronts; 1024 threads leave 128 registers per
// thread and the attention role spills); roles, t = Tc-1 .. 0:
//
//   decoder chain                                                     prior chain
//   RA [H/32]  dh = dh.z + dgh[t+1].Whh + dqd[t+1].Watt_q;            PA [E/16]  [dhp | dlz] = dgates[t+1].[Whh | Wih_z];
//              GRU cell backward -> dgi[t], dgh[t]                               re-parameterisation backward -> dml[t]
//   RB [E/32]  dctx = dgi[t].Wih_ctx                                  PB [Hp/32] dhp += dml[t].Wml; LSTM cell backward -> dgates[t]
//   RC [N]     attention backward of clip n: dqd[t], running sums
//              of d encproj / d memory / dv in registers
//
// The dgh product of RA (three quarters of its K) does not depend on the attention: it runs while RB and RC work on the
// step.  Same hand-off protocol as the forward launch.  Products are summed over 8 wave shares (one pass) instead of the per-step
// path's split-K slabs and the attention sums its frames in one sweep, so results agree with that path to rounding (not
// bit for bit); the parity test bounds the difference.
constexpr int PB_THREADS = 512, PB_WAVES = 8;

struct PbSmem {
  float red[PB_WAVES][32][33];
  int flag;
};
__device__ __forceinline__ void pb_stash(float (*red)[32][33], const f32x16& acc, int wave, int li, int lh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
}
__device__ __forceinline__ float pb_sum(float (*red)[32][33], int mm, int nn) {
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < PB_WAVES; ++w) v += red[w][mm][nn];
  return v;
}
// acc += A[32 rows][K] . B[32 rows][K]^T, this wave's K-groups (wave, wave + 8, ..) in batches of eight, software-pipelined by
// one batch: the loads of batch i + 1 are in flight while the MFMAs of batch i run (a batch's round trip to handed-off data is
// 3-4 us; taken one after the other - load, wait, multiply, load - a K = 1536 product spent 15 us on it).  A handed off
// inside the launch (coherent 16-byte loads), B = weights (plain loads).
struct PbBatch { float4 a[8], b[8]; };
// K a multiple of 512 (every width of the reference configuration): every K-group of a batch exists, so the eight fragment
// addresses of a batch are ONE lane base plus compile-time steps (an immediate of the load) and nothing is masked.  Round 3's
// general form computed a clamped address per fragment; hipcc hoisted those sixteen address registers out of the step loop and
// spilled them (28-36 B per lane of scratch, reloaded in front of every batch on the critical path of a step).  Other widths
// (small models) take pb_gemm_tail: one K-group at a time, same order of additions, no registers to hoist.
__device__ __forceinline__ void pb_load(PbBatch& f, const float* abase, long aidx, const float* bp, int g) {
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, 0x7fffffff, 0x00020000);
  const int vo = (int)((aidx + (long)g * 8) * 4);
  const float* bq = bp + (long)g * 8;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    f.b[u] = *reinterpret_cast<const float4*>(bq + u * PB_WAVES * 8);
    const pd_v4u v = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, u * PB_WAVES * 32, 16 /* sc1 */);
    f.a[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
}
__device__ __forceinline__ void pb_mfma(f32x16& acc, const PbBatch& f) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[u].x, f.b[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[u].y, f.b[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[u].z, f.b[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[u].w, f.b[u].w, acc, 0, 0, 0);
  }
}
// any K (a multiple of 8): this wave's groups wave, wave + 8, .. one at a time
__device__ __forceinline__ void pb_gemm_tail(f32x16& acc, const float* abase, long aidx, const float* bp, int G, int wave) {
#pragma clang loop unroll(disable)
  for (int g = wave; g < G; g += PB_WAVES) {
    const float4 b = *reinterpret_cast<const float4*>(bp + (long)g * 8);
    const float4 a = ld_sc1_4(abase, aidx + (long)g * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}
// K <= 512: the wave's whole share is one batch
__device__ __forceinline__ void pb_gemm1(f32x16& acc, const float* abase, long aidx, const float* bp, int K, int wave) {
  const int G = K / 8;
  if (G == 8 * PB_WAVES) {
    PbBatch f;
    pb_load(f, abase, aidx, bp, wave);
    __builtin_amdgcn_sched_barrier(0);
    pb_mfma(acc, f);
  } else {
    pb_gemm_tail(acc, abase, aidx, bp, G, wave);
  }
}
// K <= 512 with the handed-off operand arriving as `ns` (2 or 3) shares `sstride` floats apart (the attention role split
// over the frames of a long clip): every share of a fragment is fetched in the same round trip, the shares are added in share
// order; `keep` (lane's row of the summed operand, or null) receives the sum for the products behind the launch.
__device__ __forceinline__ void pb_gemm1_shares(f32x16& acc, const float* abase, long aidx, long sstride, int ns, const float* bp,
                                                int K, int wave, float* keep) {
  const int G = K / 8;
  if (G == 8 * PB_WAVES) {
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, 0x7fffffff, 0x00020000);
    const int vo = (int)((aidx + (long)wave * 8) * 4);
    const float* bq = bp + (long)wave * 8;
    pd_v4u r[3][8];
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      b[u] = *reinterpret_cast<const float4*>(bq + u * PB_WAVES * 8);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        r[q][u] = __builtin_amdgcn_raw_buffer_load_b128(ars, vo + (int)((q < ns ? q : 0) * sstride * 4), u * PB_WAVES * 32, 16 /* sc1 */);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float4 v = make_float4(__uint_as_float(r[0][u].x), __uint_as_float(r[0][u].y), __uint_as_float(r[0][u].z), __uint_as_float(r[0][u].w));
#pragma unroll
      for (int q = 1; q < 3; ++q)
        if (q < ns) {
          v.x += __uint_as_float(r[q][u].x); v.y += __uint_as_float(r[q][u].y);
          v.z += __uint_as_float(r[q][u].z); v.w += __uint_as_float(r[q][u].w);
        }
      if (keep) *reinterpret_cast<float4*>(keep + (wave + u * PB_WAVES) * 8) = v;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, b[u].w, acc, 0, 0, 0);
    }
  } else {
#pragma clang loop unroll(disable)
    for (int g = wave; g < G; g += PB_WAVES) {
      const float4 b = *reinterpret_cast<const float4*>(bp + (long)g * 8);
      float4 v = ld_sc1_4(abase, aidx + (long)g * 8);
      for (int q = 1; q < ns; ++q) {
        const float4 w = ld_sc1_4(abase, aidx + q * sstride + (long)g * 8);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      if (keep) *reinterpret_cast<float4*>(keep + g * 8) = v;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, b.w, acc, 0, 0, 0);
    }
  }
}
__device__ __forceinline__ void pb_gemm(f32x16& acc, const float* abase, long aidx, const float* bp, int K, int wave) {
  const int G = K / 8;
  constexpr int STEP = 8 * PB_WAVES;
  if (G % STEP != 0) { pb_gemm_tail(acc, abase, aidx, bp, G, wave); return; }
  PbBatch f0, f1;
  pb_load(f0, abase, aidx, bp, wave);
  for (int g = wave; g < G; g += 2 * STEP) {
    if (g + STEP < G) pb_load(f1, abase, aidx, bp, g + STEP);
    __builtin_amdgcn_sched_barrier(0);
    pb_mfma(acc, f0);
    if (g + STEP < G) {
      if (g + 2 * STEP < G) pb_load(f0, abase, aidx, bp, g + 2 * STEP);
      __builtin_amdgcn_sched_barrier(0);
      pb_mfma(acc, f1);
    }
  }
}

// The same product with ALL of the wave's handed-off operand fragments fetched in one round trip (NB batches of eight K-groups:
// 32 registers each) and the weight fragments streamed batch by batch from L2 behind them, one batch ahead: a role whose K
// spans several batches pays the 3-4 us to handed-off data once instead of once per batch (K = 2048: 21 -> 11 us).
// NP > 1: the operand arrives as NP shares `pstride` floats apart (a K-split producer); they are summed in share order.
// `keep` (optional): the lane's row of the summed operand is stored there for the fragments f (0 .. 8 NB - 1) with
// f % keep_mod == keep_rem - the readers of one operand share the work of keeping it.
template <int NB, int NP = 1>
__device__ __forceinline__ void pb_gemm_apre(f32x16& acc, const float* abase, long aidx, const float* bp, int wave, long pstride = 0,
                                             float* keep = nullptr, int keep_mod = 1, int keep_rem = 0) {
  // K = 512 NB exactly: every K-group exists, all offsets are a per-wave base plus constants
  constexpr int STEP = 8 * PB_WAVES;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, 0x7fffffff, 0x00020000);
  const int vbase = (int)((aidx + (long)wave * 8) * 4);
  const float* bw = bp + wave * 8;
  float4 a[NB][8];
  {
    pd_v4u r[NP][NB][8];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          r[q][nb][u] = __builtin_amdgcn_raw_buffer_load_b128(ars, vbase + (int)(q * pstride * 4), (nb * STEP + u * PB_WAVES) * 32, 16 /* sc1 */);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float4 v = make_float4(__uint_as_float(r[0][nb][u].x), __uint_as_float(r[0][nb][u].y), __uint_as_float(r[0][nb][u].z),
                               __uint_as_float(r[0][nb][u].w));
#pragma unroll
        for (int q = 1; q < NP; ++q) {
          v.x += __uint_as_float(r[q][nb][u].x); v.y += __uint_as_float(r[q][nb][u].y);
          v.z += __uint_as_float(r[q][nb][u].z); v.w += __uint_as_float(r[q][nb][u].w);
        }
        a[nb][u] = v;
        if (keep && (nb * 8 + u) % keep_mod == keep_rem) *reinterpret_cast<float4*>(keep + (wave + nb * STEP + u * PB_WAVES) * 8) = v;
      }
  }
  float4 b0[8], b1[8];
  auto load_b = [&](float4 (&b)[8], int nb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(bw + (nb * STEP + u * PB_WAVES) * 8);
  };
  auto mfma = [&](const float4 (&av)[8], const float4 (&b)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, b[u].w, acc, 0, 0, 0);
    }
  };
  if (NB <= 2) {                 // two weight sets: the next batch's fragments fly during this batch's MFMAs
    load_b(b0, 0);
#pragma unroll
    for (int nb = 0; nb < NB; nb += 2) {
      if (nb + 1 < NB) load_b(b1, nb + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(a[nb], b0);
      if (nb + 1 < NB) {
        if (nb + 2 < NB) load_b(b0, nb + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma(a[nb + 1], b1);
      }
    }
  } else {                       // 4 x 32 operand registers leave room for one weight set
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      load_b(b0, nb);
      __builtin_amdgcn_sched_barrier(0);
      mfma(a[nb], b0);
    }
  }
}

// ---------------------------------------------------------------- RA: dh_{t} for 32 units, GRU cell backward (rnn.hip gru_bwd_kernel)
// (32-wide slices: every reader of a handed-off tensor pulls all of it across the chip - 196 KB of dgh per workgroup and
// step - so fewer, full-width tiles halve that traffic at the same MFMA time per workgroup)
__device__ void role_ra(const PbParams& p, int slice, PbSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int H = p.H, A = p.A, u0 = slice * 32;
  const float* b_hh = p.wt_dhh + (long)(u0 + li) * 3 * H + 4 * lh;   // [H][3H]
  const float* b_att = p.wt_datt + (long)(u0 + li) * A + 4 * lh;     // [H + E][A], rows 0..H = query half
  const int arow = li < p.N ? li : 0;
  // cell elements of this thread: rows mm0 and mm0 + 16, unit u
  const int mm0 = threadIdx.x >> 5, j = threadIdx.x & 31, u = u0 + j;
  float carry[2] = {0.f, 0.f};                                  // dh.z of the step before
  for (int t = p.Tc - 1; t >= 0; --t) {
    float v[2] = {0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if (t < p.Tc - 1) {
      if (!pd_wait(p.cnt + PB_C_RA * p.Tc + (t + 1), (unsigned)p.n_ra, p.abort_word, p.spin_limit, &sm.flag)) return;
      pb_gemm(acc, p.dgh, (long)arow * p.Tc * 3 * H + (long)(t + 1) * 3 * H + 4 * lh, b_hh, 3 * H, wave);
    }
    // this step's cell inputs do not depend on the launch (fetched here, behind the long product: its two batches of
    // fragments and these would not fit the register file together)
    float dout[2], r[2], z[2], nn[2], ghn[2], hp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = mm0 + 16 * i;
      dout[i] = r[i] = z[i] = nn[i] = ghn[i] = hp[i] = 0.f;
      if (mm < p.N) {
        dout[i] = p.d_out[(long)mm * p.Tc * H + (long)t * H + u];
        const float* sv = p.gru_save + (long)mm * p.Tc * 4 * H + (long)t * 4 * H;
        r[i] = sv[u]; z[i] = sv[H + u]; nn[i] = sv[2 * H + u]; ghn[i] = sv[3 * H + u];
        hp[i] = p.hprev_d[(long)mm * p.Tc * H + (long)t * H + u];
      }
    }
    if (t < p.Tc - 1) {
      if (!pd_wait(p.cnt + PB_C_RC * p.Tc + (t + 1), (unsigned)(p.N * p.rc_splits), p.abort_word, p.spin_limit, &sm.flag)) return;
      const long qidx = (long)arow * p.Tc * A + (long)(t + 1) * A + 4 * lh;
      if (p.rc_splits == 1) pb_gemm1(acc, p.dqd, qidx, b_att, A, wave);   // A <= 512 (decode_persist_bwd_ok)
      else pb_gemm1_shares(acc, p.dqd_part, qidx, (long)p.N * p.Tc * A, p.rc_splits, b_att, A, wave,
                           (slice == 0 && li < p.N) ? p.dqd + qidx : nullptr);
      pb_stash(sm.red, acc, wave, li, lh);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = pb_sum(sm.red, mm0 + 16 * i, j);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = mm0 + 16 * i;
      if (mm < p.N) {
        float dh = v[i] + carry[i];
        dh += dout[i];
        const float dn = dh * (1.f - z[i]);
        const float dz = dh * (hp[i] - nn[i]);
        const float dnp = dn * (1.f - nn[i] * nn[i]);
        const float drp = dnp * ghn[i] * r[i] * (1.f - r[i]);
        const float dzp = dz * z[i] * (1.f - z[i]);
        float* a = p.dgi + (long)mm * p.Tc * 3 * H + (long)t * 3 * H;
        float* b = p.dgh + (long)mm * p.Tc * 3 * H + (long)t * 3 * H;
        st_sc1(a + u, drp); st_sc1(a + H + u, dzp); st_sc1(a + 2 * H + u, dnp);
        st_sc1(b + u, drp); st_sc1(b + H + u, dzp); st_sc1(b + 2 * H + u, dnp * r[i]);
        carry[i] = dh * z[i];
      }
    }
    pd_arrive(p.cnt + PB_C_RA * p.Tc + t);
  }
}

// ---------------------------------------------------------------- RB: dctx = dgi[t] . Wih[:, E:2E] for 32 context columns
__device__ void role_rb(const PbParams& p, int slice, PbSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  // K = 3H split over ks_rb workgroups per tile: a round trip to handed-off data costs 3-4 us per batch of eight K-groups per
  // wavefront whatever its size, so three workgroups with one batch each (one round trip + 64 MFMAs) replace one workgroup with
  // three (15 -> 6 us per step); the partial tiles are summed in split order by their reader (RC).
  const int H = p.H, E = p.E, tile = slice / p.ks_rb, ks = slice - tile * p.ks_rb, e0 = tile * 32;
  const int Kc = 3 * H / p.ks_rb, k0 = ks * Kc;
  const float* bp = p.wt_dih + (long)(E + e0 + li) * 3 * H + k0 + 4 * lh;       // [3E][3H]
  const int arow = li < p.N ? li : 0;
  const int mm0 = threadIdx.x >> 5, j = threadIdx.x & 31;
  float* part = p.dctx_part + (long)ks * p.N * E;
  for (int t = p.Tc - 1; t >= 0; --t) {
    if (!pd_wait(p.cnt + PB_C_RA * p.Tc + t, (unsigned)p.n_ra, p.abort_word, p.spin_limit, &sm.flag)) return;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const long aidx = (long)arow * p.Tc * 3 * H + (long)t * 3 * H + k0 + 4 * lh;
    if (Kc <= 512) pb_gemm1(acc, p.dgi, aidx, bp, Kc, wave);
    else pb_gemm(acc, p.dgi, aidx, bp, Kc, wave);
    pb_stash(sm.red, acc, wave, li, lh);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = mm0 + 16 * i;
      if (mm < p.N) st_sc1(part + (long)mm * E + e0 + j, pb_sum(sm.red, mm, j));
    }
    pd_arrive(p.cnt + PB_C_RB * p.Tc + t);
  }
}

// ---------------------------------------------------------------- RC: attention backward of clip n (attention.hip: score / accum / reduce)
// thread = channel a (A, E <= 512): the projected memory of the workgroup's frames in LDS, their memory rows and the running sums
// of d encproj in registers - 64 frames per workgroup.  The memory's own gradient, sum_t w[t][s] dctx[t][e], needs no
// recurrence: dctx is kept per step and attn_dmem_kernel forms it behind the launch.
// Round 4: clips of more than 64 frames (BASELINE configs[3]: S = 187) are split over p.rc_splits = ceil(S / 64) workgroups,
// share j owning frames 64 j .. 64 j + 63.  The softmax backward's only sum over ALL frames, sum_s w[s] dw[s], equals
// <dctx, ctx> (dw[s] = dctx . mem[s], ctx = sum_s w[s] mem[s]: the forward's context, saved in rnn_d), so the shares never
// talk to each other: each hands over its part of dq[t] and the reader (RA) adds the shares in share order.
__device__ void role_rc(const PbParams& p, int n, int share, float* smem, int* s_flag) {
  const int A = p.A, E = p.E;
  const int s0 = share * 64, S = min(64, p.S - s0);          // this workgroup's frames: s0 .. s0 + S - 1
  const bool split = p.rc_splits > 1;
  float* w_s = smem;                  // [64]
  float* ds_s = smem + 64;            // [64]
  float* dwred = smem + 128;          // [8][64]
  float* dc_s = smem + 128 + 512;     // [512]
  float* Pl = smem + 128 + 512 + 512; // [S][A]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = threadIdx.x;
  const int len = (int)p.mem_lens[n];
  for (int i = threadIdx.x * 4; i < S * A; i += PB_THREADS * 4)
    *reinterpret_cast<float4*>(Pl + i) = *reinterpret_cast<const float4*>(p.encproj + ((long)n * p.S + s0) * A + i);
  // dw[s] = dctx . mem[s] with lanes over the frames: thread (wave w, lane s) holds mem[s][64 w .. 64 w + 63] and sums its 64
  // products in registers - no cross-lane reduction (62 wave-wide shuffle reductions per step cost 8 us here); the eight
  // wave shares meet in LDS.  The running sums of d encproj stay per channel: thread a, all frames.
  float m[64], dPa[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const int e = wave * 64 + k;
    m[k] = (lane < S && e < E) ? p.mem[((long)n * p.S + s0 + lane) * E + e] : 0.f;
    dPa[k] = 0.f;
  }
  const float va = a < A ? p.att_v[a] : 0.f;
  float dva = 0.f;
  float* dq_out = split ? p.dqd_part + (long)share * p.N * p.Tc * A : p.dqd;
  __syncthreads();
  for (int t = p.Tc - 1; t >= 0; --t) {
    const float qa = a < A ? p.qd[(long)n * p.Tc * A + (long)t * A + a] : 0.f;
    if (threadIdx.x < S) w_s[threadIdx.x] = p.attn_w[(long)n * p.Tc * p.S + (long)t * p.S + s0 + threadIdx.x];
    // split: the forward context of the step (lane l of wavefront 0: elements l, l + 64, ..), fetched ahead of the wait
    float cx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (split && wave == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (lane + 64 * k < E) cx[k] = p.ctx[(long)n * p.Tc * 3 * E + (long)t * 3 * E + lane + 64 * k];
    }
    if (!pd_wait(p.cnt + PB_C_RB * p.Tc + t, (unsigned)p.n_rb, p.abort_word, p.spin_limit, s_flag)) return;
    {   // the step's context gradient: the K-split partials of RB in split order; kept per step for attn_dmem_kernel
      float dc = 0.f;
      if (a < E) {
        float pv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < p.ks_rb) pv[q] = ld_sc1(p.dctx_part + ((long)q * p.N + n) * E + a);
        dc = ((pv[0] + pv[1]) + pv[2]) + pv[3];
        if (share == 0) p.dctx[(long)n * p.Tc * E + (long)t * E + a] = dc;
      }
      dc_s[a] = dc;
    }
    __syncthreads();
    {
      float part = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const float4 d4 = *reinterpret_cast<const float4*>(dc_s + wave * 64 + k4 * 4);      // the same address in every lane: broadcast
        part += d4.x * m[k4 * 4] + d4.y * m[k4 * 4 + 1] + d4.z * m[k4 * 4 + 2] + d4.w * m[k4 * 4 + 3];
      }
      dwred[wave * 64 + lane] = part;
    }
    __syncthreads();
    // dscore[s] = w[s] (dw[s] - sum_j w[j] dw[j]), 0 behind the clip's last frame: one wave
    if (wave == 0) {
      const int s = lane;
      float dw = 0.f, w = 0.f;
      if (s < S) {
#pragma unroll
        for (int q = 0; q < 8; ++q) dw += dwred[q * 64 + s];
        w = w_s[s];
      }
      float dot;
      if (split) {                      // sum over ALL frames of the clip = <dctx, ctx>
        float pd = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) pd += (lane + 64 * k < E) ? dc_s[lane + 64 * k] * cx[k] : 0.f;
        dot = wave_sum(pd);
      } else {
        dot = wave_sum(w * dw);
      }
      if (s < S) ds_s[s] = (s0 + s < len) ? w * (dw - dot) : 0.f;
    }
    __syncthreads();
    float dq = 0.f;
    if (a < A) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        if (k < S) {
          const float g = ds_s[k];
          const float th = tanh_att(qa + Pl[(long)k * A + a]);
          const float du = g * va * (1.f - th * th);
          dPa[k] += du;
          dq += du;
          dva += g * th;
        }
      }
      st_sc1(dq_out + (long)n * p.Tc * A + (long)t * A + a, dq);
    }
    pd_arrive(p.cnt + PB_C_RC * p.Tc + t);          // (its barrier also protects w_s / ds_s / dwred against the next step's writes)
  }
  if (a < A) {
#pragma unroll
    for (int k = 0; k < 64; ++k)
      if (k < S) p.dencproj[((long)n * p.S + s0 + k) * A + a] = dPa[k];
    p.dvpart[((long)share * p.N + n) * A + a] = dva;         // [rc_splits][N][A]: summed with the rows by the caller's column sum
  }
}

// ---------------------------------------------------------------- PA: [dhp | dlz] for 16 units, re-parameterisation backward
__device__ void role_pa(const PbParams& p, int slice, PbSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  // K = 4Hp split over ks_pa workgroups per tile: the product is bound by ONE CU's matrix pipe (32 x 32 x 2048: 7 us), so
  // two CUs take half each.  Everything downstream of the product is linear in it: each split hands over ITS share of dhp and
  // of the two halves of dml (split 0 adds the terms that do not come from the product), the reader (PB) sums the shares.
  const int E = p.E, Hp = p.E, tile = slice / p.ks_pa, ks = slice - tile * p.ks_pa, u0 = tile * 16;
  const int Kc = 4 * Hp / p.ks_pa, k0 = ks * Kc;
  // tile columns 0..15: dhp of units u0.. (rows of Whh^T), 16..31: d last_z of the same indices (rows 2E.. of Wih^T)
  const float* bp = (li < 16 ? p.wt_phh + (long)(u0 + li) * 4 * Hp : p.wt_pih + (long)(2 * E + u0 + li - 16) * 4 * Hp) + k0 + 4 * lh;
  const int arow = li < p.N ? li : 0;
  const int mm = threadIdx.x >> 4, j = threadIdx.x & 15, e = u0 + j;
  const bool mine = mm < p.N;
  float* dhp_o = p.dhp_part + ((long)ks * p.N + mm) * Hp + e;
  float* dml_o = p.dml_part + ((long)ks * p.N + mm) * 2 * E;
  for (int t = p.Tc - 1; t >= 0; --t) {
    const long o = (long)mm * p.Tc * E + (long)t * E + e;
    float dpz = 0.f, dme = 0.f, dle = 0.f, lv = 0.f, eps = 0.f;
    if (mine) {
      if (ks == 0) {
        if (p.d_p_z) dpz = p.d_p_z[o];
        if (p.d_p_means) dme = p.d_p_means[o];
        if (p.d_p_logs) dle = p.d_p_logs[o];
      }
      lv = p.p_logs[o];
      eps = p.eps_p[(long)t * p.N * E + (long)mm * E + e];
    }
    float vhp = 0.f, vlz = 0.f;
    if (t < p.Tc - 1) {
      if (!pd_wait(p.cnt + PB_C_PB * p.Tc + (t + 1), (unsigned)p.n_pb, p.abort_word, p.spin_limit, &sm.flag)) return;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const long aidx = (long)arow * p.Tc * 4 * Hp + (long)(t + 1) * 4 * Hp + k0 + 4 * lh;
      if (Kc == 1024) pb_gemm_apre<2>(acc, p.dgates, aidx, bp, wave);
      else if (Kc <= 512) pb_gemm1(acc, p.dgates, aidx, bp, Kc, wave);
      else pb_gemm(acc, p.dgates, aidx, bp, Kc, wave);
      pb_stash(sm.red, acc, wave, li, lh);
      __syncthreads();
      if (mine) { vhp = pb_sum(sm.red, mm, j); vlz = pb_sum(sm.red, mm, 16 + j); }
    }
    if (mine) {
      const float g = vlz + dpz;                                        // this split's share of d z_t
      st_sc1(dml_o + e, g + dme);
      st_sc1(dml_o + E + e, g * eps * .5f * expf(.5f * lv) + dle);
      st_sc1(dhp_o, vhp);
    }
    pd_arrive(p.cnt + PB_C_PA * p.Tc + t);
  }
}

// ---------------------------------------------------------------- PB: dhp += dml[t] . Wml, LSTM cell backward (rnn.hip lstm_bwd_kernel), 32 units
__device__ void role_pb(const PbParams& p, int slice, PbSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int E = p.E, Hp = p.E, u0 = slice * 32;
  const float* bp = p.wt_pml + (long)(u0 + li) * 2 * E + 4 * lh;          // [Hp][2E]
  const int arow = li < p.N ? li : 0;
  const int mm0 = threadIdx.x >> 5, j = threadIdx.x & 31, u = u0 + j;
  float dc_next[2] = {0.f, 0.f};
  for (int t = p.Tc - 1; t >= 0; --t) {
    float ig[2], fg[2], gg[2], og[2], tc[2], c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = mm0 + 16 * i;
      ig[i] = fg[i] = gg[i] = og[i] = tc[i] = c[i] = 0.f;
      if (mm < p.N) {
        const float* s = p.lstm_save + (long)mm * p.Tc * 5 * Hp + (long)t * 5 * Hp;
        ig[i] = s[u]; fg[i] = s[Hp + u]; gg[i] = s[2 * Hp + u]; og[i] = s[3 * Hp + u]; tc[i] = s[4 * Hp + u];
        c[i] = t ? p.c_all[(long)mm * p.Tc * Hp + (long)(t - 1) * Hp + u] : 0.f;
      }
    }
    if (!pd_wait(p.cnt + PB_C_PA * p.Tc + t, (unsigned)p.n_pa, p.abort_word, p.spin_limit, &sm.flag)) return;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // dhp and dml arrive as the ks_pa shares of PA's K-split: summed here in share order; dhp is fetched with the product's
    // operands, not behind them (one round trip less per step).  The summed d [mean | logvar] of the step is kept for the
    // parameter products behind the launch (every workgroup stores its share of the fragments).
    float dhp_in[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (mm0 + 16 * i < p.N) {
        float v0 = ld_sc1(p.dhp_part + (long)(mm0 + 16 * i) * Hp + u), v1 = 0.f;
        if (p.ks_pa == 2) v1 = ld_sc1(p.dhp_part + ((long)p.N + mm0 + 16 * i) * Hp + u);
        dhp_in[i] = v0 + v1;
      }
    {
      const bool keeper = 16 % p.n_pb == 0 ? slice < 16 : slice == 0;          // every workgroup keeps 16 / n_pb of the 16 fragment columns
      float* keep = (keeper && li < p.N) ? p.dml_all + (long)li * p.Tc * 2 * E + (long)t * 2 * E + 4 * lh : nullptr;
      const long aidx = (long)arow * 2 * E + 4 * lh;
      if (p.ks_pa == 2 && 2 * E == 1024)
        pb_gemm_apre<2, 2>(acc, p.dml_part, aidx, bp, wave, (long)p.N * 2 * E, keep, 16 % p.n_pb == 0 ? p.n_pb : 1, 16 % p.n_pb == 0 ? slice : 0);
      else {                     // other widths: one share (ks_pa = 1), the streaming product, the step's row copied by slice 0
        pb_gemm(acc, p.dml_part, aidx, bp, 2 * E, wave);
        if (slice == 0)
          for (int i = threadIdx.x; i < p.N * 2 * E; i += PB_THREADS) {
            const int n_ = i / (2 * E), c_ = i - n_ * 2 * E;
            p.dml_all[(long)n_ * p.Tc * 2 * E + (long)t * 2 * E + c_] = ld_sc1(p.dml_part + i);
          }
      }
    }
    pb_stash(sm.red, acc, wave, li, lh);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = mm0 + 16 * i;
      if (mm < p.N) {
        float d = pb_sum(sm.red, mm, j);
        d += dhp_in[i];
        float dc = d * og[i] * (1.f - tc[i] * tc[i]);
        dc += dc_next[i];
        float* g = p.dgates + (long)mm * p.Tc * 4 * Hp + (long)t * 4 * Hp;
        st_sc1(g + u, dc * gg[i] * ig[i] * (1.f - ig[i]));
        st_sc1(g + Hp + u, dc * c[i] * fg[i] * (1.f - fg[i]));
        st_sc1(g + 2 * Hp + u, dc * ig[i] * (1.f - gg[i] * gg[i]));
        st_sc1(g + 3 * Hp + u, d * tc[i] * og[i] * (1.f - og[i]));
        dc_next[i] = dc * fg[i];
      }
    }
    pd_arrive(p.cnt + PB_C_PB * p.Tc + t);
  }
}

// dmem[n][s][e] = sum_t w[n][t][s] dctx[n][t][e]   (the decoder attention's direct share of the memory gradient)
__global__ void attn_dmem_kernel(const float* __restrict__ w, const float* __restrict__ dctx, float* __restrict__ dmem, int Tc,
                                 int S, int E) {
  const int n = blockIdx.x / S, s = blockIdx.x % S;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float acc = 0.f;
    for (int t = Tc - 1; t >= 0; --t) acc += w[((long)n * Tc + t) * S + s] * dctx[((long)n * Tc + t) * E + e];
    dmem[((long)n * S + s) * E + e] = acc;
  }
}

__global__ __launch_bounds__(PB_THREADS) void decode_persist_bwd_kernel(PbParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem_raw[];
  PbSmem& sm = *reinterpret_cast<PbSmem*>(pb_smem_raw);
  int b = blockIdx.x;
  if (b < p.n_ra) { role_ra(p, b, sm); return; }
  b -= p.n_ra;
  if (b < p.n_rb) { role_rb(p, b, sm); return; }
  b -= p.n_rb;
  if (b < p.N * p.rc_splits) {
    role_rc(p, b / p.rc_splits, b % p.rc_splits, reinterpret_cast<float*>(pb_smem_raw) + 4, reinterpret_cast<int*>(pb_smem_raw));
    return;
  }
  b -= p.N * p.rc_splits;
  if (b < p.n_pa) { role_pa(p, b, sm); return; }
  b -= p.n_pa;
  role_pb(p, b, sm);
}

}  // namespace

namespace {
// =====================================================================================================================
// Posterior BiGRU, persistent (PqParams / PqbParams in decode_persist.h).  Arithmetic of the per-step path (acvae_posterior_fwd
// in decoder.hip): gemm_skinny's K order for every 32 x 32 tile, gru_fwd_kernel's / gru_bwd_kernel's cell formulas.
// =====================================================================================================================
struct PqSmem {
  float red[SK_WAVES][32][33];
  int flag;
};

__global__ __launch_bounds__(PD_THREADS) void posterior_persist_fwd_kernel(PqParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pq_smem_raw[];
  PqSmem& sm = *reinterpret_cast<PqSmem*>(pq_smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int Hq = p.Hq, nwg = Hq / 32;
  const int dir = blockIdx.x / nwg, u0 = (blockIdx.x - dir * nwg) * 32;
  const float* W = p.w_hh[dir];
  const int arow = li < p.N ? li : 0;
  // weight fragments of the three gate tiles: resident for all steps (K = Hq <= 512: one batch per wavefront)
  PdFrag fb[3];
  const float* bp[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    bp[g] = W + (long)(g * Hq + u0 + li) * Hq + 4 * lh;
    pd_fetch_b(fb[g], bp[g], Hq, wave);
  }
  // the two elements (row mm, unit u0 + nn) this thread owns in the cell
  const int nn = threadIdx.x & 31, m0 = threadIdx.x >> 5;          // rows m0, m0 + 16
  const float br = p.b_hh[dir][u0 + nn], bz = p.b_hh[dir][Hq + u0 + nn], bn_ = p.b_hh[dir][2 * Hq + u0 + nn];
  float hreg[2] = {0.f, 0.f};
  int len[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) len[i] = (m0 + 16 * i) < p.N ? (int)p.lens1[m0 + 16 * i] : 0;
  unsigned* cnt = p.cnt + (long)dir * p.Tc;
  float* hb = p.hbuf + (long)dir * 2 * p.N * Hq;
  for (int k = 0; k < p.Tc; ++k) {
    const int t = dir ? p.Tc - 1 - k : k;
    if (k > 0 && !pd_wait(cnt + (k - 1), (unsigned)nwg, p.abort_word, p.spin_limit, &sm.flag)) return;
    const float* hin = hb + (long)((k + 1) & 1) * p.N * Hq;      // step 0: parity 1 = the zeros the caller put there
    float4 a[PD_U];
    pd_load_a<true>(a, hin, (long)arow * Hq + 4 * lh, Hq, wave);
    __builtin_amdgcn_sched_barrier(0);
    float ghv[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      pd_mfma(acc, a, fb[g], Hq, wave);
      if (g) __syncthreads();                      // the previous tile's sums have been taken
      pd_stash(sm.red, acc, wave, li, lh);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) ghv[g][i] = pd_sum(sm.red, m0 + 16 * i, nn);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = m0 + 16 * i;
      if (mm < p.N) {
        const long row = (long)mm * p.Tc + t;
        const float h = hreg[i];
        p.hprev[dir][row * Hq + u0 + nn] = h;
        float hn = h, os = 0.f;
        if (t < len[i]) {
          const float* gi = p.gi[dir] + row * 3 * Hq + u0 + nn;
          const float r = sigmoidf_(gi[0] + (ghv[0][i] + br));
          const float z = sigmoidf_(gi[Hq] + (ghv[1][i] + bz));
          const float ghn = ghv[2][i] + bn_;
          const float nv = tanhf(gi[2 * Hq] + r * ghn);
          hn = (1.f - z) * nv + z * h;
          os = hn;
          float* sv = p.save[dir] + row * 4 * Hq + u0 + nn;
          sv[0] = r; sv[Hq] = z; sv[2 * Hq] = nv; sv[3 * Hq] = ghn;
        }
        hreg[i] = hn;
        st_sc1(hb + (long)(k & 1) * p.N * Hq + (long)mm * Hq + u0 + nn, hn);
        p.hid[row * 2 * Hq + dir * Hq + u0 + nn] = os;
      }
    }
    pd_arrive(cnt + k);
  }
}

// backward: t runs against the forward's order.  dh(k) = dh(k-1) . z(k-1)  [own elements, registers]  +  dgh(k-1) . Whh  [all of the
// direction's dgh of the previous step: the hand-off];  then gru_bwd_kernel's formulas.
__global__ __launch_bounds__(PD_THREADS) void posterior_persist_bwd_kernel(PqbParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pq_smem_raw[];
  PqSmem& sm = *reinterpret_cast<PqSmem*>(pq_smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int Hq = p.Hq, nwg = Hq / 32;
  const int dir = blockIdx.x / nwg, u0 = (blockIdx.x - dir * nwg) * 32;
  const int arow = li < p.N ? li : 0;
  const float* bp = p.wt[dir] + (long)(u0 + li) * 3 * Hq + 4 * lh;         // row = output unit, K = 3Hq contiguous
  const int nn = threadIdx.x & 31, m0 = threadIdx.x >> 5;
  float dhd[2] = {0.f, 0.f};               // dh . z of the previous step (own elements)
  int len[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) len[i] = (m0 + 16 * i) < p.N ? (int)p.lens1[m0 + 16 * i] : 0;
  unsigned* cnt = p.cnt + (long)dir * p.Tc;
  int tprev = 0;
  for (int k = 0; k < p.Tc; ++k) {
    const int t = dir ? k : p.Tc - 1 - k;
    float prod[2] = {0.f, 0.f};
    if (k > 0) {
      if (!pd_wait(cnt + (k - 1), (unsigned)nwg, p.abort_word, p.spin_limit, &sm.flag)) return;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // A = dgh[:, tprev, :] of this direction ([N][Tc][3Hq]: row stride Tc * 3Hq), handed over
      pd_accumulate<true>(acc, p.dgh[dir], ((long)arow * p.Tc + tprev) * 3 * Hq + 4 * lh, bp, 3 * Hq, wave);
      pd_stash(sm.red, acc, wave, li, lh);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) prod[i] = pd_sum(sm.red, m0 + 16 * i, nn);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = m0 + 16 * i;
      if (mm < p.N) {
        const long row = (long)mm * p.Tc + t;
        float dh = dhd[i] + prod[i];
        float* a = p.dgi[dir] + row * 3 * Hq + u0 + nn;
        float* b = p.dgh[dir] + row * 3 * Hq + u0 + nn;
        if (!(t < len[i])) {
          a[0] = a[Hq] = a[2 * Hq] = 0.f;
          st_sc1(b, 0.f); st_sc1(b + Hq, 0.f); st_sc1(b + 2 * Hq, 0.f);
          dhd[i] = dh;
        } else {
          dh += p.dhid[row * 2 * Hq + dir * Hq + u0 + nn];
          const float* sv = p.save[dir] + row * 4 * Hq + u0 + nn;
          const float r = sv[0], z = sv[Hq], nv = sv[2 * Hq], ghn = sv[3 * Hq];
          const float h = p.hprev[dir][row * Hq + u0 + nn];
          const float dn = dh * (1.f - z);
          const float dz = dh * (h - nv);
          const float dnp = dn * (1.f - nv * nv);
          const float drp = dnp * ghn * r * (1.f - r);
          const float dzp = dz * z * (1.f - z);
          a[0] = drp; a[Hq] = dzp; a[2 * Hq] = dnp;
          st_sc1(b, drp); st_sc1(b + Hq, dzp); st_sc1(b + 2 * Hq, dnp * r);
          dhd[i] = dh * z;
        }
      }
    }
    tprev = t;
    pd_arrive(cnt + k);
  }
}
}  // namespace

// =====================================================================================================================
// Launchers.  A persistent launch only makes progress while ALL of its workgroups are resident (every role spins on what
// another role produces), so a launcher
//   (1) asks the occupancy calculator whether the whole grid fits the device with the launch's LDS and registers, and
//       reports "does not fit" to its caller BEFORE anything is queued (the per-step path of decoder.hip runs instead);
//   (2) chains the device's persistent launches behind each other (one event per device, recorded after every such launch
//       and waited for by the next one, whatever stream or host thread it comes from): two spin-wait grids of one process
//       never share the chip;
//   (3) queues a one-workgroup tail kernel behind the launch that reads the launch's abort word - set when a bounded wait ran
//       out all the same (another PROCESS holds the CUs, a CU mask the occupancy query does not see) - and, if it is set,
//       overwrites the launch's outputs with NaN and raises the device's status word (acvae_persist_status_register):
//       an aborted launch can neither be trained on nor go unnoticed.
// The slot below (event, status pointer, CU count) is the library's per-device mutable state, listed in acvae_hip.h.
// =====================================================================================================================
namespace {
constexpr int PERSIST_MAX_DEV = 64;
enum { PK_DECODE_FWD, PK_DECODE_BWD, PK_POST_FWD, PK_POST_BWD, PK_COUNT };
struct PersistSlot {
  std::mutex mu;
  hipEvent_t done = nullptr;                      // completion of the device's latest persistent launch
  std::atomic<unsigned*> status{nullptr};         // registered status words (device-visible), or null
  std::atomic<int> cus{0};
  std::atomic<bool> raised[PK_COUNT];             // dynamic-LDS attribute of kernel k raised on this device
};
PersistSlot g_slot[PERSIST_MAX_DEV];

struct PoisonList { float* p[8]; long n[8]; int count; };
inline void poison_add(PoisonList& l, float* p, long n) {
  if (p && n > 0 && l.count < 8) { l.p[l.count] = p; l.n[l.count] = n; ++l.count; }
}
__global__ void persist_tail_kernel(const unsigned* abort_word, unsigned* status, int kind, PoisonList list) {
  if (__hip_atomic_load(abort_word, PD_RLX_AGENT) == 0u) return;
  const float qnan = __uint_as_float(0x7fc00000u);
  for (int k = 0; k < list.count; ++k)
    for (long i = threadIdx.x; i < list.n[k]; i += blockDim.x) list.p[k][i] = qnan;
  if (status && threadIdx.x == 0) {
    __hip_atomic_store(status + kind, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(status + PK_COUNT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int persist_device(int& dev) {
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  return (dev < 0 || dev >= PERSIST_MAX_DEV) ? ACVAE_EUNSUPPORTED : ACVAE_OK;
}
// does a grid of `grid` workgroups of 512 threads with `shm` bytes of dynamic LDS fit the current device all at once?
template <class K>
bool persist_fits(K kernel, int kid, int grid, size_t shm) {
  int dev = 0;
  if (persist_device(dev) != ACVAE_OK) return false;
  PersistSlot& sl = g_slot[dev];
  if (shm > 64 * 1024 && !sl.raised[kid].load()) {       // more than 64 KB of dynamic LDS needs the attribute, once per device
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    sl.raised[kid].store(true);
  }
  int cus = sl.cus.load();
  if (cus == 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      (void)hipGetLastError();
      return false;
    }
    sl.cus.store(cus);
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PD_THREADS, shm) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return (long)per_cu * cus >= grid;
}
// memset of the counters, the launch and its tail, chained behind the device's previous persistent launch
template <class K, class P>
int persist_launch(K kernel, int kid, const P& p, int grid, size_t shm, long counter_words, const PoisonList& poison, hipStream_t st,
                   bool zeroed = false) {
  int dev = 0;
  ACVAE_TRY(persist_device(dev));
  PersistSlot& sl = g_slot[dev];
  std::lock_guard<std::mutex> lock(sl.mu);
  if (!sl.done) {
    if (hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
  } else if (hipStreamWaitEvent(st, sl.done, 0) != hipSuccess) {
    return (int)hipGetLastError();
  }
  if (!zeroed && hipMemsetAsync(p.cnt, 0, (size_t)counter_words * sizeof(unsigned), st) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(PD_THREADS), shm, st, p);
  hipLaunchKernelGGL(persist_tail_kernel, dim3(1), dim3(1024), 0, st, p.abort_word, sl.status.load(), kid, poison);
  ACVAE_LAUNCH_CHECK();
  if (hipEventRecord(sl.done, st) != hipSuccess) return (int)hipGetLastError();
  return ACVAE_OK;
}
// ACVAE_FLAG_TEST_STALL: one workgroup short and a short spin limit - the roles that wait for the missing workgroup run into
// their bound, exactly as when part of a grid is not resident (tests/test_decode_persist_gpu.py)
inline void persist_test_stall(int flags, int& grid, unsigned& spin_limit) {
  spin_limit = PD_SPIN_LIMIT;
  if (flags & ACVAE_FLAG_TEST_STALL) { grid -= 1; spin_limit = 1u << 12; }
}
}  // namespace

extern "C" int acvae_persist_status_register(int device, void* status_words_host) {
  if (device < 0 || device >= PERSIST_MAX_DEV) return ACVAE_EINVAL;
  unsigned* dptr = nullptr;
  if (status_words_host) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, status_words_host, 0) != hipSuccess) { (void)hipGetLastError(); return ACVAE_EINVAL; }
    dptr = static_cast<unsigned*>(d);
  }
  g_slot[device].status.store(dptr);
  return ACVAE_OK;
}

namespace acvae {

// ---- decode forward
static void decode_fwd_geometry(int N, int S, int E, int H, int A, bool resident, int& grid, size_t& shm, int& att_resident) {
  grid = (A / 32 + 3 * H / 32) + N + H / 16 + E / 8 + E / 16;
  shm = sizeof(PdSmem);
  size_t att = (size_t)(4 + ((S + 16 + 3) & ~3) + 4096) * sizeof(float);      // context partials: (1024 / (E/4)) x E
  // the clip's projected memory in LDS and its memory rows in registers (8 frames for each of a thread's 2 context groups)
  const int GV = 1024 / (E / 4);
  att_resident = (resident && S <= 8 * GV && att + (size_t)S * A * sizeof(float) <= 150 * 1024) ? 1 : 0;
  if (att_resident) att += (size_t)S * A * sizeof(float);
  if (att > shm) shm = att;
}
static bool decode_fwd_shape_ok(int N, int Tc, int S, int E, int H, int A) {
  // one 32-row tile of clips; whole 32 / 16 / 8-wide slices and K-groups of 8; one score per thread in the softmax; E a
  // power of two so that the context groups of the per-step attention kernel (1024 / (E / 4)) can be replayed exactly
  return N >= 1 && N <= 32 && Tc >= 1 && S >= 1 && S <= PD_THREADS && E >= 32 && E <= 2048 && (E & (E - 1)) == 0 &&
         H % 32 == 0 && A % 32 == 0;
}
// picks the attention form (memory resident on the CU or streamed) whose grid fits the device; false: neither does
static bool decode_fwd_plan(int N, int Tc, int S, int E, int H, int A, int& grid, size_t& shm, int& att_resident) {
  if (!decode_fwd_shape_ok(N, Tc, S, E, H, A)) return false;
  for (int resident = 1; resident >= 0; --resident) {
    decode_fwd_geometry(N, S, E, H, A, resident != 0, grid, shm, att_resident);
    if (resident && !att_resident) continue;
    if (persist_fits(decode_persist_kernel, PK_DECODE_FWD, grid, shm)) return true;
  }
  return false;
}
bool decode_persist_ok(int N, int Tc, int S, int E, int H, int A) {
  int grid, res; size_t shm;
  return decode_fwd_plan(N, Tc, S, E, H, A, grid, shm, res);
}
long decode_persist_counter_words(int Tc) { return ((long)PD_C_COUNT * Tc + 1 + 3) & ~3L; }

int decode_persist_fwd(PdParams p, hipStream_t st, int flags) {
  int grid; size_t shm;
  if (!decode_fwd_plan(p.N, p.Tc, p.S, p.E, p.H, p.A, grid, shm, p.att_resident)) return ACVAE_EUNSUPPORTED;
  p.n_d1 = p.A / 32 + 3 * p.H / 32;
  p.n_d3 = p.H / 16;
  p.n_p1 = p.E / 8;
  p.n_p2 = p.E / 16;
  const long words = decode_persist_counter_words(p.Tc);
  p.abort_word = p.cnt + (long)PD_C_COUNT * p.Tc;
  persist_test_stall(flags, grid, p.spin_limit);
  const long R = (long)p.N * p.Tc;
  PoisonList poison{};
  poison_add(poison, p.outputs, R * p.H); poison_add(poison, p.p_means, R * p.E); poison_add(poison, p.p_logs, R * p.E);
  poison_add(poison, p.p_z, R * p.E); poison_add(poison, p.attn_w, R * p.S);
  return persist_launch(decode_persist_kernel, PK_DECODE_FWD, p, grid, shm, words, poison, st, (flags & ACVAE_FLAG_INT_CNT_ZEROED) != 0);
}

// ---- decode backward
static size_t decode_bwd_shm(int S, int A) {
  size_t shm = sizeof(PbSmem);
  const size_t att = (size_t)(4 + 128 + 512 + 512 + (long)(S < 64 ? S : 64) * A) * sizeof(float);   // 64 frames per attention workgroup
  return att > shm ? att : shm;
}
int decode_persist_bwd_rc_splits(int S) { return (S + 63) / 64; }
static void decode_bwd_splits(int E, int H, int& ks_rb, int& ks_pa) {
  // K-splits: one resident batch (K <= 512) per workgroup where the K of the product divides that way, at most 4
  ks_rb = (3 * H) % 512 == 0 && 3 * H / 512 <= 4 ? 3 * H / 512 : 1;
  ks_pa = E == 512 ? 2 : 1;               // K = 4Hp = 2048 = 2 x 1024 and 2E = 1024: the shapes the split products are written for
}
bool decode_persist_bwd_ok(int N, int Tc, int S, int E, int H, int A) {
  // an attention workgroup keeps 64 frames in 2 x 32 register slots and its channels in 512 thread columns; a clip takes up to
  // three of them (S <= 192: BASELINE configs[3] has 187)
  if (!(decode_fwd_shape_ok(N, Tc, S, E, H, A) && S <= 192 && E <= 512 && A <= 512 && H % 32 == 0 && H == E &&
        decode_bwd_shm(S, A) <= 150 * 1024))
    return false;
  int ks_rb, ks_pa;
  decode_bwd_splits(E, H, ks_rb, ks_pa);
  const int grid = H / 32 + (E / 32) * ks_rb + N * decode_persist_bwd_rc_splits(S) + (E / 16) * ks_pa + E / 32;
  return persist_fits(decode_persist_bwd_kernel, PK_DECODE_BWD, grid, decode_bwd_shm(S, A));
}
long decode_persist_bwd_counter_words(int Tc) { return ((long)PB_C_COUNT * Tc + 1 + 3) & ~3L; }
long decode_persist_bwd_part_floats(int N, int E, int H) { return 4L * N * E + 4L * N * E + 4L * N * 2 * E; }

int decode_persist_bwd(PbParams p, hipStream_t st, int flags) {
  if (!decode_persist_bwd_ok(p.N, p.Tc, p.S, p.E, p.H, p.A)) return ACVAE_EUNSUPPORTED;
  decode_bwd_splits(p.E, p.H, p.ks_rb, p.ks_pa);
  p.n_ra = p.H / 32; p.n_rb = (p.E / 32) * p.ks_rb; p.n_pa = (p.E / 16) * p.ks_pa; p.n_pb = p.E / 32;
  p.rc_splits = decode_persist_bwd_rc_splits(p.S);
  if (p.rc_splits > 1 && (!p.dqd_part || !p.ctx)) return ACVAE_EINVAL;
  const long words = decode_persist_bwd_counter_words(p.Tc);
  p.abort_word = p.cnt + (long)PB_C_COUNT * p.Tc;
  int grid = p.n_ra + p.n_rb + p.N * p.rc_splits + p.n_pa + p.n_pb;
  persist_test_stall(flags, grid, p.spin_limit);
  const long R = (long)p.N * p.Tc;
  PoisonList poison{};
  poison_add(poison, p.dgi, R * 3 * p.H); poison_add(poison, p.dgh, R * 3 * p.H); poison_add(poison, p.dgates, R * 4 * p.E);
  poison_add(poison, p.dml_all, R * 2 * p.E); poison_add(poison, p.dctx, R * p.E); poison_add(poison, p.dqd, R * p.A);
  poison_add(poison, p.dencproj, (long)p.N * p.S * p.A);
  ACVAE_TRY(persist_launch(decode_persist_bwd_kernel, PK_DECODE_BWD, p, grid, decode_bwd_shm(p.S, p.A), words, poison, st,
                           (flags & ACVAE_FLAG_INT_CNT_ZEROED) != 0));
  // (dctx poisoned = NaN in dmem: attn_dmem_kernel forms it from dctx behind the tail)
  hipLaunchKernelGGL(attn_dmem_kernel, dim3(p.N * p.S), dim3(256), 0, st, p.attn_w, p.dctx, p.dmem, p.Tc, p.S, p.E);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// ---- posterior
bool posterior_persist_ok(int N, int Tc, int Hq) {
  // one 32-row tile of clips; 32 hidden units per workgroup; a wavefront's share of K = Hq is one resident batch
  if (!(N >= 1 && N <= 32 && Tc >= 1 && Hq >= 32 && Hq <= 64 * PD_U && Hq % 32 == 0)) return false;
  return persist_fits(posterior_persist_fwd_kernel, PK_POST_FWD, 2 * (Hq / 32), sizeof(PqSmem)) &&
         persist_fits(posterior_persist_bwd_kernel, PK_POST_BWD, 2 * (Hq / 32), sizeof(PqSmem));
}
long posterior_persist_counter_words(int Tc) { return (2L * Tc + 1 + 3) & ~3L; }

int posterior_persist_fwd(PqParams p, hipStream_t st, int flags) {
  if (!posterior_persist_ok(p.N, p.Tc, p.Hq)) return ACVAE_EUNSUPPORTED;
  p.abort_word = p.cnt + 2L * p.Tc;
  int grid = 2 * (p.Hq / 32);
  persist_test_stall(flags, grid, p.spin_limit);
  PoisonList poison{};
  poison_add(poison, p.hid, (long)p.N * p.Tc * 2 * p.Hq);
  return persist_launch(posterior_persist_fwd_kernel, PK_POST_FWD, p, grid, sizeof(PqSmem), posterior_persist_counter_words(p.Tc),
                        poison, st, (flags & ACVAE_FLAG_INT_CNT_ZEROED) != 0);
}
int posterior_persist_bwd(PqbParams p, hipStream_t st, int flags) {
  if (!posterior_persist_ok(p.N, p.Tc, p.Hq)) return ACVAE_EUNSUPPORTED;
  p.abort_word = p.cnt + 2L * p.Tc;
  int grid = 2 * (p.Hq / 32);
  persist_test_stall(flags, grid, p.spin_limit);
  PoisonList poison{};
  for (int dir = 0; dir < 2; ++dir) {
    poison_add(poison, p.dgi[dir], (long)p.N * p.Tc * 3 * p.Hq);
    poison_add(poison, p.dgh[dir], (long)p.N * p.Tc * 3 * p.Hq);
  }
  return persist_launch(posterior_persist_bwd_kernel, PK_POST_BWD, p, grid, sizeof(PqSmem), posterior_persist_counter_words(p.Tc),
                        poison, st, (flags & ACVAE_FLAG_INT_CNT_ZEROED) != 0);
}
}  // namespace acvae
