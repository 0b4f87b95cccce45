// Dense fp32 products on MFMA for the Linear layers of the path (attention projections, recurrent
// input/hidden projections, mean/log-variance heads, vocabulary classifier) and their gradients.
// Replaces torch.nn.Linear / F.linear at models/attn_model.py:32, models/decoder.py:198,
// models/text_encoder.py:192,255, models/vae_model.py:726 and the GEMMs inside nn.GRU / nn.LSTM.
#include <cstdlib>
#include "mfma_tile.h"
#include "transpose_batch.h"
#include "../../include/acvae_hip.h"
#include "conv.h"

namespace {
using namespace mfma;

template <int BN, bool VEC4>
__global__ __launch_bounds__(nt_threads<128>(), 2) void gemm_nt_kernel(const float* __restrict__ A, long lda,
                                                                     const float* __restrict__ B, long ldb,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ C, long ldc, int M, int N,
                                                                     int K, int accumulate) {
  __shared__ NtSmem<128, BN> sm;
  PlainLoader<VEC4, 4> al{A, lda, M, K};
  PlainLoader<VEC4, BN / 32> bl{B, ldb, N, K};
  PlainEpilogue ep{C, ldc, bias, accumulate};
  int bm, bn;
  xcd_tile(gridDim.x, gridDim.y, bm, bn);
  nt_block<128, BN>(al, bl, M, N, K, bm, bn, ep, sm);
}

template <int WM, int WN, bool VEC4>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const float* __restrict__ A, long lda,
                                                         const float* __restrict__ B, long ldb, float* __restrict__ C,
                                                         long ldc, int M, int N, int K, int k_per, int accumulate,
                                                         long slab_stride) {
  __shared__ TnSmem<WM, WN> sm;
  PlainKMajorLoader<VEC4> al{A, lda, M, K}, bl{B, ldb, N, K};
  const int kb = blockIdx.z * k_per;
  const int ke = min(K, kb + k_per);
  float* out = C + (long)blockIdx.z * slab_stride;
  tn_block<WM, WN>(al, bl, M, N, kb, ke, blockIdx.x, blockIdx.y, out, ldc, accumulate, sm);
}

// C[m][n] (+)= sum_z slab[z][m][n]   (fixed order: deterministic)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, long slab_stride, int nsplit, float* __restrict__ C,
                                   long ldc, int M, int N, int accumulate) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    float acc = 0.f;
    for (int z = 0; z < nsplit; ++z) acc += slab[z * slab_stride + i];
    float* p = C + (long)m * ldc + n;
    *p = accumulate ? *p + acc : acc;
  }
}

// gemm_tn with the slab sum inside (the composite calls' form: 12 launches of slab_reduce_kernel per step): every K-slice
// writes its partial tile to its slab and takes a ticket of the TILE; the block that draws the last one sums the tile's slices
// IN SLICE ORDER - the arithmetic of slab_reduce_kernel, bit for bit - into C.  Hand-off without fences: the partial tile is
// stored write-through (sc1), drained (vmcnt(0)) in front of the barrier and the ticket, and read back with sc1 loads.  (With
// __threadfence() around the ticket - a write-back and an invalidate of the XCD's whole L2 per workgroup - the Winograd weight
// gradient running beside these products on the first stream lost 18 %: 0.55 -> 0.65 ms per launch, +0.8 ms per step.)
// Tickets: one word per output tile, zeroed once per composite call, reset by the reducer.
template <int WM, int WN, bool VEC4>
__global__ __launch_bounds__(256, 2) void gemm_tn_fused_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                               long ldb, float* __restrict__ slab, float* __restrict__ C, long ldc,
                                                               int M, int N, int K, int k_per, int accumulate, long slab_stride,
                                                               unsigned* __restrict__ tickets) {
  __shared__ TnSmem<WM, WN> sm;
  __shared__ int s_last;
  PlainKMajorLoader<VEC4> al{A, lda, M, K}, bl{B, ldb, N, K};
  const int kb = blockIdx.z * k_per;
  const int ke = min(K, kb + k_per);
  tn_block<WM, WN, true>(al, bl, M, N, kb, ke, blockIdx.x, blockIdx.y, slab + (long)blockIdx.z * slab_stride, (long)N, 0, sm);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* tk = tickets + blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.z - 1 ? 1 : 0;
    if (s_last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  constexpr int TM = WM * 64, TN = WN * 64;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const int nsplit = gridDim.z;
  for (int i = threadIdx.x; i < TM * TN; i += 256) {
    const int m = m0 + i / TN, n = n0 + i % TN;
    if (m >= M || n >= N) continue;
    const long e = (long)m * N + n;
    float acc = 0.f;
    for (int z0 = 0; z0 < nsplit; z0 += 8) {          // eight loads in flight; the additions stay in slice order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = z0 + u < nsplit ? __hip_atomic_load(slab + (z0 + u) * slab_stride + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u < nsplit) acc += v[u];
    }
    float* p = C + (long)m * ldc + n;
    *p = accumulate ? *p + acc : acc;
  }
}

// gridDim.z = S K-slices per output tile.  S == 1: the block owns the tile.  S > 1: every slice writes its partial
// tile to a slab and takes a ticket; the block that draws the last ticket sums the S slabs IN SLICE ORDER (so the
// result does not depend on which block was last: deterministic) and runs the epilogue.  Hand-off = the agent-scope
// release/acquire counter recipe of cdna_hip_programming.md §6 Guideline 16; the counter is reset by the reducer
// (and zeroed once per composite call by the host: acvae_skinny_ws_reset).
constexpr int SK_MAX_TILES = 1024;
template <bool VEC4>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_kernel(const float* __restrict__ A1, long lda1,
                                                                 const float* __restrict__ B1, long ldb1, int K1,
                                                                 const float* __restrict__ A2, long lda2,
                                                                 const float* __restrict__ B2, long ldb2, int K2,
                                                                 const float* __restrict__ bias, float* __restrict__ C,
                                                                 long ldc, int M, int N, int accumulate,
                                                                 float* __restrict__ slabs, unsigned* __restrict__ cnt) {
  __shared__ float red[SK_WAVES][32][33];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int S = gridDim.z, z = blockIdx.z;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // K-groups (of 8) are dealt round-robin to the S*8 (slice, wave) pairs
  sk_accumulate<VEC4>(acc, A1, lda1, B1, ldb1, M, N, K1, m0, n0, z * SK_WAVES + wave, S * SK_WAVES, li, lh);
  if (A2) sk_accumulate<VEC4>(acc, A2, lda2, B2, ldb2, M, N, K2, m0, n0, z * SK_WAVES + wave, S * SK_WAVES, li, lh);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  __syncthreads();
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  float* my = slabs + ((long)tile * S + z) * 1024;
  // Split-K hand-off without fences (round 4; an agent release fence writes back the XCD's whole L2 and the acquire
  // invalidates the CU's L1 - measured on the column-sum kernel: 6-10 us per launch): the partial tile is stored write-through
  // (8-byte agent-scope atomic stores, `sc1`), every wave drains its stores, ONE lane takes the ticket behind the barrier, and
  // the block whose ticket came last reads all S partial tiles with agent-scope loads (`sc1`: never from its L1) - the first
  // row of MI355X_MICROARCH.md's table of hand-offs measured with `sc1` loads in place of the acquire.
  if (S > 1) {
    {
      const int e = threadIdx.x * 2;                 // SK_THREADS = 512: one pair of neighbours per thread
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) { v0 += red[w][e >> 5][e & 31]; v1 += red[w][e >> 5][(e & 31) + 1]; }
      const unsigned long long bits = (unsigned long long)__float_as_uint(v0) | ((unsigned long long)__float_as_uint(v1) << 32);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(my + e), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned t = __hip_atomic_fetch_add(&cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == (unsigned)(S - 1));
      if (last) __hip_atomic_store(&cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // the S partial tiles in slice order, eight loads in flight
    const float* base = slabs + (long)tile * S * 1024;
    const int e = threadIdx.x * 2;
    float v0 = 0.f, v1 = 0.f;
    for (int q0 = 0; q0 < S; q0 += 8) {
      unsigned long long b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        b[u] = q0 + u < S ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(base + (long)(q0 + u) * 1024 + e),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q0 + u < S) { v0 += __uint_as_float((unsigned)b[u]); v1 += __uint_as_float((unsigned)(b[u] >> 32)); }
    }
    const int mm = e >> 5, m = m0 + mm;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int n = n0 + (e & 31) + k;
      if (m < M && n < N) {
        float v = k ? v1 : v0;
        if (bias) v += bias[n];
        float* p = C + (long)m * ldc + n;
        if (accumulate) v += *p;
        *p = v;
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < 1024; e += SK_THREADS) {
    const int mm = e >> 5, nn = e & 31;
    const int m = m0 + mm, n = n0 + nn;
    if (m < M && n < N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += red[w][mm][nn];
      if (bias) v += bias[n];
      float* p = C + (long)m * ldc + n;
      if (accumulate) v += *p;
      *p = v;
    }
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, long ld_in, float* __restrict__ out, long ld_out,
                                 int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(long)c * ld_out + r] = tile[threadIdx.x][j];
  }
}

__global__ void transpose_batch_kernel(TransposeBatch b) {
  __shared__ float tile[32][33];
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.tile0[m + 1]) ++m;
  const int t = blockIdx.x - b.tile0[m];
  const int tcols = (b.cols[m] + 31) / 32;
  const int c0 = (t % tcols) * 32, r0 = (t / tcols) * 32;
  const float* in = b.in[m];
  float* out = b.out[m];
  const int rows = b.rows[m], cols = b.cols[m];
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * b.ld_in[m] + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(long)c * b.ld_out[m] + r] = tile[threadIdx.x][j];
  }
}

inline bool vec_ok(const float* p, int64_t ld, int k) { return aligned16(p) && (ld & 3) == 0 && (k & 3) == 0; }

}  // namespace

int acvae_transpose_batch(TransposeBatch& b, hipStream_t st) {
  if (b.n <= 0) return ACVAE_OK;
  b.tile0[0] = 0;
  for (int m = 0; m < b.n; ++m) {
    if (!b.in[m] || !b.out[m] || b.rows[m] <= 0 || b.cols[m] <= 0) return ACVAE_EINVAL;
    b.tile0[m + 1] = b.tile0[m] + cdiv(b.cols[m], 32) * cdiv(b.rows[m], 32);
  }
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(b.tile0[b.n]), dim3(32, 8), 0, st, b);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// Internal C++ entry (also used by the composite encoder/decoder drivers).
long acvae_skinny_ws_floats() { return SK_MAX_TILES + (long)SK_MAX_TILES * 1024; }
long acvae_skinny_ticket_words() { return SK_MAX_TILES; }
int acvae_skinny_ws_reset(float* ws, hipStream_t st) {   // zero the ticket counters (first SK_MAX_TILES words)
  return hipMemsetAsync(ws, 0, SK_MAX_TILES * sizeof(unsigned), st) == hipSuccess ? ACVAE_OK : (int)hipGetLastError();
}

namespace {
// Two independent skinny products in ONE launch (no split-K): the x-tiles of the first, then those of the second.  For
// the serial decode steps, where two short-K products hang off the same input (h . W_att^T and h . W_hh^T in the decoder
// step): each kernel on such a chain lasts 7-20 us - its own load latency, not its work - so running the two side by
// side takes one kernel's time instead of two.  Each tile runs exactly the arithmetic of gemm_skinny_kernel with
// gridDim.z = 1 (long-K products gain more from the split-K path and stay separate).
struct SkProblem {
  const float* A; long lda;
  const float* B; long ldb;
  int K;
  const float* bias;
  float* C; long ldc;
  int N, accumulate;
};
template <bool VEC4>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_pair_kernel(SkProblem p0, SkProblem p1, int nx0, int M) {
  __shared__ float red[SK_WAVES][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const bool second = (int)blockIdx.x >= nx0;
  const SkProblem& p = second ? p1 : p0;
  const int n0 = ((int)blockIdx.x - (second ? nx0 : 0)) * 32, m0 = blockIdx.y * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  sk_accumulate<VEC4>(acc, p.A, p.lda, p.B, p.ldb, M, p.N, p.K, m0, n0, wave, SK_WAVES, li, lh);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += SK_THREADS) {
    const int mm = e >> 5, nn = e & 31;
    const int m = m0 + mm, n = n0 + nn;
    if (m < M && n < p.N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += red[w][mm][nn];
      if (p.bias) v += p.bias[n];
      float* o = p.C + (long)m * p.ldc + n;
      if (p.accumulate) v += *o;
      *o = v;
    }
  }
}
}  // namespace

int acvae_gemm_nt_pair(const float* A0, int64_t lda0, const float* B0, int64_t ldb0, int K0, const float* bias0, float* C0,
                       int64_t ldc0, int N0, int acc0, const float* A1, int64_t lda1, const float* B1, int64_t ldb1, int K1,
                       const float* bias1, float* C1, int64_t ldc1, int N1, int acc1, int M, hipStream_t st) {
  if (!A0 || !B0 || !C0 || !A1 || !B1 || !C1 || M <= 0 || M > 64 || N0 <= 0 || N1 <= 0 || K0 <= 0 || K1 <= 0)
    return ACVAE_EINVAL;
  const bool vec = vec_ok(A0, lda0, K0) && vec_ok(B0, ldb0, K0) && vec_ok(A1, lda1, K1) && vec_ok(B1, ldb1, K1);
  const SkProblem p0{A0, lda0, B0, ldb0, K0, bias0, C0, ldc0, N0, acc0}, p1{A1, lda1, B1, ldb1, K1, bias1, C1, ldc1, N1, acc1};
  const int nx0 = cdiv(N0, 32);
  dim3 grid(nx0 + cdiv(N1, 32), cdiv(M, 32));
  if (vec) hipLaunchKernelGGL(gemm_skinny_pair_kernel<true>, grid, dim3(SK_THREADS), 0, st, p0, p1, nx0, M);
  else hipLaunchKernelGGL(gemm_skinny_pair_kernel<false>, grid, dim3(SK_THREADS), 0, st, p0, p1, nx0, M);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

int acvae_gemm_nt_dual(const float* A1, int64_t lda1, const float* B1, int64_t ldb1, int K1, const float* A2,
                       int64_t lda2, const float* B2, int64_t ldb2, int K2, const float* bias, float* C, int64_t ldc,
                       int M, int N, int accumulate, hipStream_t st, float* skws) {
  if (!A1 || !B1 || !C || M <= 0 || N <= 0 || K1 <= 0) return ACVAE_EINVAL;
  if (A2 && (!B2 || K2 <= 0)) return ACVAE_EINVAL;
  bool vec = vec_ok(A1, lda1, K1) && vec_ok(B1, ldb1, K1);
  if (A2) vec = vec && vec_ok(A2, lda2, K2) && vec_ok(B2, ldb2, K2);
  // The 128-row tile kernel needs >= ~100 workgroups to fill the chip; mid-sized products (M = N*Tc = 672 rows
  // against 512..2048 columns) would launch 24-96 of them and run 100-140 us, so they take the 32x32-tile kernel too.
  const long big_blocks = (long)cdiv(M, 128) * cdiv(N, N <= 64 ? 64 : 128);
  if (M <= 64 || A2 || (big_blocks < 100 && M <= 4096)) {
    const int tiles = cdiv(N, 32) * cdiv(M, 32);
    // split K over more workgroups when there are few tiles and K is long (the serial decode/BPTT steps)
    int S = 1;
    const int Ktot = K1 + (A2 ? K2 : 0);
    static const bool splitk_on = !(getenv("ACVAE_SKINNY_SPLITK") && getenv("ACVAE_SKINNY_SPLITK")[0] == '0');  // tuning switch
    // Only the long-K, few-tile products of the BPTT steps gain: the release/acquire hand-off costs a few us
    // (measured: 48 tiles x K=512 went 6.8 -> 9.8 us with S=2, 16 tiles x K=2048 went 25 -> 13.7 us with S=8).
    if (skws && splitk_on && tiles <= 16 && Ktot >= 1024) {
      S = 128 / tiles;
      const int maxs = Ktot / 256;       // keep >= 256 k per slice
      if (S > maxs) S = maxs;
      if (S > 8) S = 8;
      if (S < 1) S = 1;
      if ((long)tiles * S > SK_MAX_TILES) S = 1;
    }
    dim3 grid(cdiv(N, 32), cdiv(M, 32), S);
    unsigned* cnt = (unsigned*)skws;
    float* slabs = skws ? skws + SK_MAX_TILES : nullptr;
    if (vec)
      hipLaunchKernelGGL(gemm_skinny_kernel<true>, grid, dim3(SK_THREADS), 0, st, A1, lda1, B1, ldb1, K1, A2, lda2, B2,
                         ldb2, K2, bias, C, ldc, M, N, accumulate, slabs, cnt);
    else
      hipLaunchKernelGGL(gemm_skinny_kernel<false>, grid, dim3(SK_THREADS), 0, st, A1, lda1, B1, ldb1, K1, A2, lda2,
                         B2, ldb2, K2, bias, C, ldc, M, N, accumulate, slabs, cnt);
    ACVAE_LAUNCH_CHECK();
    return ACVAE_OK;
  }
  const bool bn64 = (N <= 64);
  dim3 grid(cdiv(M, 128), cdiv(N, bn64 ? 64 : 128));
#define LAUNCH_NT(BN_, V_)                                                                                     \
  hipLaunchKernelGGL((gemm_nt_kernel<BN_, V_>), grid, dim3(nt_threads<128>()), 0, st, A1, lda1, B1, ldb1, bias, C, ldc, M, \
                     N, K1, accumulate)
  if (bn64) { if (vec) LAUNCH_NT(64, true); else LAUNCH_NT(64, false); }
  else      { if (vec) LAUNCH_NT(128, true); else LAUNCH_NT(128, false); }
#undef LAUNCH_NT
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                             int64_t ldc, int M, int N, int K, int accumulate, void* stream) {
  return acvae_gemm_nt_dual(A, lda, B, ldb, K, nullptr, 0, nullptr, 0, 0, bias, C, ldc, M, N, accumulate,
                            (hipStream_t)stream, nullptr);
}

static int tn_splits(int M, int N, int K) {
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
  // one round of workgroups (256 CUs, two blocks per CU would fit): these products run on the second stream beside the
  // MFMA-bound encoder backward, where more slices only mean more slab traffic (347 MB per step with the earlier target of
  // 1024 workgroups, 139 MB now) and a reduce launch for the three largest products that no longer need one
  int s = (int)(256 / (tiles > 0 ? tiles : 1));
  const int maxs = cdiv(K, 4 * BKT);  // at least 64 k per slice
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return s;
}

extern "C" int64_t acvae_gemm_tn_workspace_bytes(int M, int N, int K) {
  const int s = tn_splits(M, N, K);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int acvae_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                             int N, int K, int accumulate, float* slab_ws, int64_t slab_ws_bytes, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ACVAE_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int s = tn_splits(M, N, K);
  if (s > 1 && (!slab_ws || slab_ws_bytes < (int64_t)s * M * N * (int64_t)sizeof(float))) s = 1;
  int k_per = cdiv(cdiv(K, s), BKT) * BKT;
  s = cdiv(K, k_per);
  const bool vec = aligned16(A) && aligned16(B) && (lda & 3) == 0 && (ldb & 3) == 0 && (M & 3) == 0 && (N & 3) == 0;
  const bool narrow = (M <= 64);
  dim3 grid(cdiv(M, narrow ? 64 : 128), cdiv(N, narrow ? 256 : 128), s);
  float* out = s > 1 ? slab_ws : C;
  const long out_ld = s > 1 ? N : ldc;
  const long slab_stride = s > 1 ? (long)M * N : 0;
  const int acc_k = s > 1 ? 0 : accumulate;
#define LAUNCH_TN(WM_, WN_, V_)                                                                              \
  hipLaunchKernelGGL((gemm_tn_kernel<WM_, WN_, V_>), grid, dim3(256), 0, st, A, lda, B, ldb, out, out_ld, M, N, K, \
                     k_per, acc_k, slab_stride)
  if (narrow) { if (vec) LAUNCH_TN(1, 4, true); else LAUNCH_TN(1, 4, false); }
  else        { if (vec) LAUNCH_TN(2, 2, true); else LAUNCH_TN(2, 2, false); }
#undef LAUNCH_TN
  if (s > 1) {
    const long total = (long)M * N;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)),
                       dim3(256), 0, st, slab_ws, slab_stride, s, C, ldc, M, N, accumulate);
  }
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// ws = [TN_TICKETS words of tickets (zeroed once per composite call) | slabs]
int acvae_gemm_tn_fused(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                        int accumulate, float* ws, int64_t ws_bytes, hipStream_t st) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ACVAE_EINVAL;
  int s = tn_splits(M, N, K);
  const bool narrow = (M <= 64);
  dim3 grid(cdiv(M, narrow ? 64 : 128), cdiv(N, narrow ? 256 : 128), 1);
  if (s > 1 && (!ws || ws_bytes < (int64_t)TN_TICKETS * 4 + (int64_t)s * M * N * (int64_t)sizeof(float) ||
                (long)grid.x * grid.y > TN_TICKETS))
    return acvae_gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, accumulate, nullptr, 0, st);     // one slice per tile, no workspace
  int k_per = cdiv(cdiv(K, s), BKT) * BKT;
  s = cdiv(K, k_per);
  if (s <= 1) return acvae_gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, accumulate, nullptr, 0, st);
  grid.z = s;
  const bool vec = aligned16(A) && aligned16(B) && (lda & 3) == 0 && (ldb & 3) == 0 && (M & 3) == 0 && (N & 3) == 0;
  unsigned* tickets = reinterpret_cast<unsigned*>(ws);
  float* slab = ws + TN_TICKETS;
#define LAUNCH_TNF(WM_, WN_, V_)                                                                                        \
  hipLaunchKernelGGL((gemm_tn_fused_kernel<WM_, WN_, V_>), grid, dim3(256), 0, st, A, lda, B, ldb, slab, C, ldc, M, N, K, k_per, \
                     accumulate, (long)M * N, tickets)
  if (narrow) { if (vec) LAUNCH_TNF(1, 4, true); else LAUNCH_TNF(1, 4, false); }
  else        { if (vec) LAUNCH_TNF(2, 2, true); else LAUNCH_TNF(2, 2, false); }
#undef LAUNCH_TNF
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_transpose(const float* in, int64_t ld_in, float* out, int64_t ld_out, int rows, int cols,
                               void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0) return ACVAE_EINVAL;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, (hipStream_t)stream, in,
                     ld_in, out, ld_out, rows, cols);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
