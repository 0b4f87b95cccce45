// fp32 MFMA tile machinery shared by the dense-product kernels (gemm.hip) and the implicit-GEMM
// convolutions (conv.hip).  gfx950 only.
//
// Instruction: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD, 4096 FLOP) — the MI355X fp32 matrix
// peak of 157 TFLOP/s (MI355X_MICROARCH.md §Matrix cores).  Operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&31][k = l>>5],  B: lane l holds B[k = l>>5][j = l&31],
//   D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// NT block kernel (C[M,N] = A[M,K] . B[N,K]^T), 256 threads = 4 wavefronts as 2(M) x 2(N):
//   block tile 128 x BN (BN = 128 or 64), BK = 32, wave tile 64 x BN/2 = 2 x (BN/64) MFMA tiles.
//   Both operands are staged global -> registers -> LDS ([row][BK+4] floats: the +4 pad makes every
//   ds_read_b128 of 16 consecutive rows hit 64 distinct banks) in a two-stage ring, one barrier per
//   K-step; a lane reads 4 consecutive k with one ds_read_b128 and feeds 4 MFMAs with it: in MFMA e
//   of k-group g, lanes 0-31 supply k = 8g+e and lanes 32-63 supply k = 8g+4+e for BOTH operands,
//   which is a legal permutation of the summation index.
#pragma once
#include <type_traits>
#include "common.h"

namespace mfma {

constexpr int BK = 32;            // K-step
constexpr int LDS_LD = BK + 4;    // floats per LDS row (+4 pad: every ds_read_b128 lane group covers 64 distinct banks)
constexpr int KT = BK / 4;        // float4 columns per tile row = loader threads per row
constexpr int NT_LOADERS = 256;   // loader threads per workgroup (4 wavefronts)
constexpr int RPP = NT_LOADERS / KT;  // tile rows fetched per pass of the loader threads (32)

// NT block kernel (C[M,N] = A[M,K] . B[N,K]^T), warp-specialised:
//   BMT x BN block tile (BMT = 128 or 256, BN = 128 or 64), K-step 32;
//   wavefronts 0 .. BMT/32-1 are MATRIX waves in a (BMT/64) x 2 grid, each owning a 64 x BN/2 tile
//   (2 x BN/64 MFMA tiles): they only read LDS and issue MFMAs;
//   the last 4 wavefronts are LOADER waves: they fetch the next tile global -> registers, apply the operand's
//   activation, and store it into the other half of a two-deep LDS ring.  One barrier per K-step.
// (Two identical waves sharing a SIMD finish their MFMA phase together and then both leave the matrix pipe idle
//  while they stage: 124 vs 148 TFLOP/s measured without staging; with split roles the pipe always has a wave ready.)
// A lane reads 4 consecutive k with one ds_read_b128 and feeds 4 MFMAs with it: in MFMA e of k-group g, lanes 0-31
// supply k = 8g+e and lanes 32-63 supply k = 8g+4+e for BOTH operands — a legal permutation of the summation index.
template <int BMT, int BN>
struct alignas(16) NtSmem {
  alignas(16) float a[2][BMT * LDS_LD];
  alignas(16) float b[2][BN * LDS_LD];
};
template <int BMT>
constexpr int nt_threads() { return (BMT / 32 + 4) * 64; }

// XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin to the 8 XCDs (observed,
// not contractual: only speed depends on it), so give each XCD a CONTIGUOUS run of the tile sequence; neighbouring
// M-tiles share im2col halo rows / operand panels and then meet in the same 4 MiB L2.  Bijective for any count.
__device__ __forceinline__ void xcd_tile(int nm, int nn, int& bm, int& bn) {
#ifdef ACVAE_NO_XCD_REMAP
  bm = blockIdx.x; bn = blockIdx.y; return;
#endif
  const int total = nm * nn;
  const int b = blockIdx.x + blockIdx.y * nm;
  const int q = total >> 3, r = total & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int t = xcd * q + (xcd < r ? xcd : r) + idx;   // XCD x owns q (+1 if x < r) consecutive tiles
  // N fastest inside the run: the N-tiles of one M-tile (same A rows, different weight panel) execute back to back
  // on one XCD, so the A operand is fetched from HBM once, not once per N-tile; consecutive M-tiles share halo rows
  bn = t % nn;
  bm = t / nn;
}

// Operand loaders work in two phases so that the global loads of K-step k+1 are in flight while the matrix waves
// run K-step k: issue() only computes (clamped, always-legal) addresses and starts the loads; finish() applies
// masks / activations just before the LDS stores.  Nothing in issue() may consume a loaded value.
// A loader thread lt (0..255) owns tile rows (lt / 8) + 32*j and the float4 column lt % 8.
template <int ROWS>
struct Pending {
  float4 v[ROWS];
  float4 sc, sh;      // optional per-channel affine (conv loaders)
  unsigned mask;      // VEC4: bit j = row j valid; scalar: bits 4j..4j+3 = elements of row j valid
};

// plain row-major operand: NR = rows per loader thread (tile rows / 32)
template <bool VEC4, int NR>
struct PlainLoader {
  static_assert(VEC4 || NR <= 8, "scalar mask holds 8 rows");
  const float* base;
  long ld;
  int rows, K;
  __device__ __forceinline__ void init(int, int) {}
  __device__ __forceinline__ void issue(int row0, int kstep, int lt, Pending<NR>& p) const {
    const int c = (lt % KT) * 4 + kstep * BK;
    p.mask = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = row0 + (lt / KT) + RPP * j;
      if (VEC4) {
        const bool ok = (r < rows) && (c + 4 <= K);
        p.v[j] = *reinterpret_cast<const float4*>(ok ? base + (long)r * ld + c : base);
        p.mask |= (ok ? 1u : 0u) << j;
      } else {
        const bool rok = r < rows;
        const float* q = base + (rok ? (long)r * ld : 0);
        const bool o0 = rok && c + 0 < K, o1 = rok && c + 1 < K, o2 = rok && c + 2 < K, o3 = rok && c + 3 < K;
        p.v[j].x = q[o0 ? c + 0 : 0]; p.v[j].y = q[o1 ? c + 1 : 0];
        p.v[j].z = q[o2 ? c + 2 : 0]; p.v[j].w = q[o3 ? c + 3 : 0];
        p.mask |= ((o0 ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u)) << (4 * j);
      }
    }
  }
  __device__ __forceinline__ void finish(Pending<NR>& p) const {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (VEC4) {
        if (!((p.mask >> j) & 1u)) p.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const unsigned m = p.mask >> (4 * j);
        if (!(m & 1u)) p.v[j].x = 0.f;
        if (!(m & 2u)) p.v[j].y = 0.f;
        if (!(m & 4u)) p.v[j].z = 0.f;
        if (!(m & 8u)) p.v[j].w = 0.f;
      }
    }
  }
};

// ---- epilogues -------------------------------------------------------------------------------
// An epilogue's run() receives the matrix wave's accumulators; element (i, j, r) of the wave at (wm, wn) is
//   m = row0 + wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*lh,   n = col0 + wn*(BN/2) + j*32 + li.
// It is called by EVERY wavefront (loader waves pass matrix_wave = false) so that it may use barriers.
struct PlainEpilogue {
  float* C;
  long ldc;
  const float* bias;
  int accumulate;
  template <int BMT, int BN, int NTN>
  __device__ __forceinline__ void run(const f32x16 (&acc)[2][NTN], int row0, int col0, int wm, int wn, int li, int lh,
                                      int M, int N, float*, bool matrix_wave) const {
    if (!matrix_wave) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTN; ++j) {
        const int n = col0 + wn * (BN / 2) + j * 32 + li;
        const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M && n < N) {
            float* p = C + (long)m * ldc + n;
            float v = acc[i][j][r] + bv;
            if (accumulate) v += *p;
            *p = v;
          }
        }
      }
  }
};

template <int BMT, int BN, class ALoader, class BLoader, class Epilogue>
__device__ __forceinline__ void nt_block(ALoader al, BLoader bl, int M, int N, int K, int block_m, int block_n,
                                         const Epilogue& ep, NtSmem<BMT, BN>& sm) {
  constexpr int NTN = BN / 64;     // MFMA tiles per matrix wave along N
  constexpr int NMW = BMT / 32;    // matrix wavefronts (BMT/64 x 2)
  constexpr int AR = BMT / RPP;    // A rows per loader thread
  constexpr int BR = BN / RPP;     // B rows per loader thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool matrix_wave = wave < NMW;
  const int wm = (wave >> 1) % (BMT / 64), wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int row0 = block_m * BMT, col0 = block_n * BN;
  const int nk = (K + BK - 1) / BK;

  f32x16 acc[2][NTN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (!matrix_wave) {
    // ------------------------------------------------------------------ loader wavefronts
    const int lt = tid - NMW * 64;
    const int srow = lt / KT, scol = (lt % KT) * 4;
    Pending<AR> pa;
    Pending<BR> pb;
    al.init(row0, lt);
    bl.init(col0, lt);
    auto stash = [&](int buf) {
      al.finish(pa);
      bl.finish(pb);
#pragma unroll
      for (int j = 0; j < AR; ++j) *reinterpret_cast<float4*>(&sm.a[buf][(srow + RPP * j) * LDS_LD + scol]) = pa.v[j];
#pragma unroll
      for (int j = 0; j < BR; ++j) *reinterpret_cast<float4*>(&sm.b[buf][(srow + RPP * j) * LDS_LD + scol]) = pb.v[j];
    };
    al.issue(row0, 0, lt, pa);
    bl.issue(col0, 0, lt, pb);
    stash(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      if (ks + 1 < nk) {
        al.issue(row0, ks + 1, lt, pa);
        bl.issue(col0, ks + 1, lt, pb);
        stash((ks + 1) & 1);
      }
      __syncthreads();
    }
  } else {
    // ------------------------------------------------------------------ matrix wavefronts
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const int cur = ks & 1;
      const float* As = sm.a[cur] + (wm * 64 + li) * LDS_LD + 4 * lh;
      const float* Bs = sm.b[cur] + (wn * (BN / 2) + li) * LDS_LD + 4 * lh;
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        float4 af[2], bf[NTN];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(As + i * 32 * LDS_LD + g * 8);
#pragma unroll
        for (int j = 0; j < NTN; ++j) bf[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDS_LD + g * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NTN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
    }
  }
  // every wave is past the last LDS access (barrier above): LDS is free for the epilogue
  ep.template run<BMT, BN, NTN>(acc, row0, col0, wm, wn, li, lh, M, N, &sm.a[0][0], matrix_wave);
}

// =================================================================================================
// TN block kernel:  C[M,N] = sum_k A[k][m] * B[k][n]   (both operands k-major: weight gradients,
// K = rows / pixels).  256 threads = 4 wavefronts as WM x WN (2x2 -> 128x128 tile, 1x4 -> 64x256),
// wave tile 64 x 64, BKT = 16.  LDS holds [k][m] / [k][n] rows; a lane reads TWO adjacent m (n) of
// one k with a ds_read_b64 and uses them for two MFMA tiles, so MFMA tile e of a wave covers the rows
// m = 2*rho + e (rho = MFMA row index): the permutation is undone in the epilogue.
// Split-K over gridDim.z: slice z covers k in [z*k_per, (z+1)*k_per) and writes its own slab.
// =================================================================================================
constexpr int BKT = 16;

template <int WM, int WN>
struct alignas(16) TnSmem {
  alignas(16) float a[2][BKT * 64 * WM];
  alignas(16) float b[2][BKT * 64 * WN];
};

// float4 slots a thread holds for a [BKT][W] tile (W/4 threads per k-row, 256 threads)
template <int W>
constexpr int tn_its() { return (BKT + 256 / (W / 4) - 1) / (256 / (W / 4)); }

// plain k-major loader: tile [BKT][W] floats, W = 64*WX; thread loads float4 at k = (tid / (W/4)) + it*(256/(W/4)), col = (tid % (W/4))*4
template <bool VEC4>
struct PlainKMajorLoader {
  const float* base;
  long ld;
  int cols, K;  // cols = M or N extent
  template <int W>
  __device__ __forceinline__ void init(int) {}
  template <int W>
  __device__ __forceinline__ void issue(int col0, int k0, Pending<4>& p) const {
    constexpr int TPR = W / 4;           // threads per k-row
    constexpr int ROWS_PER_IT = 256 / TPR;
    constexpr int ITS = tn_its<W>();
    constexpr bool EXACT = (ROWS_PER_IT * TPR == 256) && (BKT % ROWS_PER_IT == 0);   // W = 192: 240 threads, 4th pass partial
    const int c = col0 + (threadIdx.x % TPR) * 4;
    p.mask = 0;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int kr = threadIdx.x / TPR + it * ROWS_PER_IT;
      const int k = k0 + kr;
      const bool slot = EXACT || (threadIdx.x < ROWS_PER_IT * TPR && kr < BKT);
      if (VEC4) {
        const bool ok = slot && (k < K) && (c + 4 <= cols);
        const float* q = ok ? base + (long)k * ld + c : base;
        p.v[it] = *reinterpret_cast<const float4*>(q);
        p.mask |= (ok ? 1u : 0u) << it;
      } else {
        const bool kok = slot && k < K;
        const float* q = base + (kok ? (long)k * ld : 0);
        const bool o0 = kok && c + 0 < cols, o1 = kok && c + 1 < cols, o2 = kok && c + 2 < cols, o3 = kok && c + 3 < cols;
        p.v[it].x = q[o0 ? c + 0 : 0]; p.v[it].y = q[o1 ? c + 1 : 0];
        p.v[it].z = q[o2 ? c + 2 : 0]; p.v[it].w = q[o3 ? c + 3 : 0];
        p.mask |= ((o0 ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u)) << (4 * it);
      }
    }
  }
  template <int W>
  __device__ __forceinline__ void finish(Pending<4>& p) const {
    constexpr int ITS = tn_its<W>();
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      if (VEC4) {
        if (!((p.mask >> it) & 1u)) p.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const unsigned m = p.mask >> (4 * it);
        if (!(m & 1u)) p.v[it].x = 0.f;
        if (!(m & 2u)) p.v[it].y = 0.f;
        if (!(m & 4u)) p.v[it].z = 0.f;
        if (!(m & 8u)) p.v[it].w = 0.f;
      }
    }
  }
};

// WT: the tile is stored write-through (relaxed agent-scope atomic stores: `sc1`) - the form for a partial tile that another
// workgroup will read in the same launch (gemm.hip: gemm_tn_fused_kernel)
template <int WM, int WN, bool WT = false, class ALoader, class BLoader>
__device__ __forceinline__ void tn_block(ALoader al, BLoader bl, int M, int N, int k_begin, int k_end,
                                         int block_m, int block_n, float* C, long ldc, int accumulate,
                                         TnSmem<WM, WN>& sm) {
  constexpr int TM = 64 * WM, TN_ = 64 * WN;
  constexpr int AITS = BKT / (256 / (TM / 4)), BITS = BKT / (256 / (TN_ / 4));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int row0 = block_m * TM, col0 = block_n * TN_;
  const int nk = (k_end - k_begin + BKT - 1) / BKT;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  static_assert(AITS <= 4 && BITS <= 4, "Pending holds 4 float4");
  Pending<4> pa, pb;
  al.template init<TM>(row0);
  bl.template init<TN_>(col0);
  auto stash = [&](int buf) {
    al.template finish<TM>(pa);
    bl.template finish<TN_>(pb);
#pragma unroll
    for (int it = 0; it < AITS; ++it) {
      const int k = tid / (TM / 4) + it * (256 / (TM / 4));
      *reinterpret_cast<float4*>(&sm.a[buf][k * TM + (tid % (TM / 4)) * 4]) = pa.v[it];
    }
#pragma unroll
    for (int it = 0; it < BITS; ++it) {
      const int k = tid / (TN_ / 4) + it * (256 / (TN_ / 4));
      *reinterpret_cast<float4*>(&sm.b[buf][k * TN_ + (tid % (TN_ / 4)) * 4]) = pb.v[it];
    }
  };
#define TN_ISSUE_A(k0_) al.template issue<TM>(row0, k0_, pa)
#define TN_ISSUE_B(k0_) bl.template issue<TN_>(col0, k0_, pb)
  if (nk > 0) {
    TN_ISSUE_A(k_begin);
    TN_ISSUE_B(k_begin);
    stash(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      TN_ISSUE_A(k_begin + (ks + 1) * BKT);
      TN_ISSUE_B(k_begin + (ks + 1) * BKT);
    }
    const float* As = sm.a[cur] + lh * TM + wm * 64 + 2 * li;
    const float* Bs = sm.b[cur] + lh * TN_ + wn * 64 + 2 * li;
#pragma unroll
    for (int kk = 0; kk < BKT / 2; ++kk) {
      const float2 af = *reinterpret_cast<const float2*>(As + kk * 2 * TM);
      const float2 bf = *reinterpret_cast<const float2*>(Bs + kk * 2 * TN_);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[1][1], 0, 0, 0);
    }
    if (ks + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }
  // epilogue: acc[em][en][r] -> C[row0 + wm*64 + 2*rho + em][col0 + wn*64 + 2*li + en]
#pragma unroll
  for (int em = 0; em < 2; ++em)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = row0 + wm * 64 + 2 * rho + em;
      if (m >= M) continue;
#pragma unroll
      for (int en = 0; en < 2; ++en) {
        const int n = col0 + wn * 64 + 2 * li + en;
        if (n < N) {
          float* p = C + (long)m * ldc + n;
          float v = acc[em][en][r];
          if (accumulate) v += *p;
          if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *p = v;
        }
      }
    }
}

#undef TN_ISSUE_A
#undef TN_ISSUE_B

// =================================================================================================
// Skinny NT kernel for the serial decode steps: M <= 32 per block row (batch), one 32x32 output tile
// per 512-thread workgroup, the K range interleaved over its 8 wavefronts in groups of 8 k, operand
// fragments loaded straight from global/L2 as float4 (no LDS round trip: each weight row is used by
// exactly one workgroup), partial tiles reduced through LDS in fixed order.
// Computes C = A1.B1^T (+ A2.B2^T) (+ bias) (+ C).
// =================================================================================================
constexpr int SK_THREADS = 512;
constexpr int SK_WAVES = 8;

template <bool VEC4>
__device__ __forceinline__ float4 sk_load(const float* p, int k, int K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (VEC4) {
    if (k + 4 <= K) v = *reinterpret_cast<const float4*>(p + k);
  } else {
    if (k + 0 < K) v.x = p[k];
    if (k + 1 < K) v.y = p[k + 1];
    if (k + 2 < K) v.z = p[k + 2];
    if (k + 3 < K) v.w = p[k + 3];
  }
  return v;
}

template <bool VEC4>
__device__ __forceinline__ void sk_accumulate(f32x16& acc, const float* A, long lda, const float* B, long ldb, int M,
                                              int N, int K, int m0, int n0, int wave, int nwaves, int li, int lh) {
  // rows beyond M / N are never stored, so their lanes may read row 0 instead of being zeroed (no branches)
  const float* ap = A + (long)((m0 + li) < M ? m0 + li : 0) * lda + 4 * lh;
  const float* bp = B + (long)((n0 + li) < N ? n0 + li : 0) * ldb + 4 * lh;
  const int G = (K + 7) / 8;
  const int Gfull = VEC4 ? K / 8 : 0;   // k-groups that need no bounds check
  // The loop is latency-bound (weights arrive from the Infinity Cache / HBM: ~1-2 us per dependent round), so keep
  // SK_U k-groups = 2*SK_U float4 loads in flight per lane and predicate instead of branching: a slot past the end
  // re-reads group 0 and is zeroed by a select.
  constexpr int SK_U = 8;
  int g = wave;
  for (; g < Gfull; g += SK_U * nwaves) {
    float4 a[SK_U], b[SK_U];
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
      const int gu = g + u * nwaves;
      const long off = (gu < Gfull) ? (long)gu * 8 : 0;
      a[u] = *reinterpret_cast<const float4*>(ap + off);
      b[u] = *reinterpret_cast<const float4*>(bp + off);
    }
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
      const bool ok = (g + u * nwaves) < Gfull;
      const float4 av = ok ? a[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[u].w, acc, 0, 0, 0);
    }
  }
  // tail groups (K % 8 != 0) or the scalar path: bounds-checked loads
  g = Gfull + ((wave - Gfull % nwaves) % nwaves + nwaves) % nwaves;
  for (; g < G; g += nwaves) {
    const int k = g * 8 + 4 * lh;
    const float4 a = sk_load<VEC4>(ap - 4 * lh, k, K);   // k >= K reads give zeros
    const float4 b = sk_load<VEC4>(bp - 4 * lh, k, K);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}

}  // namespace mfma
