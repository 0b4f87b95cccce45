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
#include "common.h"

namespace mfma {

#ifndef ACVAE_BK
#define ACVAE_BK 32
#endif
#ifndef ACVAE_NT_OCC
#define ACVAE_NT_OCC 2          // workgroups per CU the NT kernels are register-budgeted for
#endif
constexpr int BM = 128;
constexpr int BK = ACVAE_BK;
constexpr int LDS_LD = BK + 4;  // floats per LDS row
constexpr int NT_THREADS = 256;   // loader (and compute) threads per workgroup
constexpr int KT = BK / 4;      // float4 columns per tile row (threads per row)
constexpr int RPP = NT_THREADS / KT;  // rows loaded per pass

template <int BN>
struct alignas(16) NtSmem {
  alignas(16) float a[2][BM * LDS_LD];
  alignas(16) float b[2][BN * LDS_LD];
};

// Operand loaders are split in two phases so that the global loads of K-step k+1 are all in flight while the
// MFMAs of K-step k run: issue() only computes (clamped, always-legal) addresses and starts the loads;
// finish() — called after the MFMAs, just before the LDS stores — applies masks / activations.  Nothing in
// issue() may consume a loaded value (that would force an s_waitcnt vmcnt(0) in front of the MFMAs).
// Loader threads: in the warp-specialised kernel only wavefronts 4..7 load; their lane id within the loader
// group is threadIdx.x - NT_LOADER_BASE.
#ifndef ACVAE_NT_WS
#define ACVAE_NT_WS 1
#endif
#if ACVAE_NT_WS
#define NT_LOADER_BASE 256
#define NT_BLOCK_THREADS 512
#else
#define NT_LOADER_BASE 0
#define NT_BLOCK_THREADS 256
#endif
__device__ __forceinline__ int ltid() { return (int)threadIdx.x - NT_LOADER_BASE; }

// XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin to the 8 XCDs (observed,
// not contractual: only speed depends on it), so give each XCD a CONTIGUOUS run of the tile sequence; neighbouring
// M-tiles share im2col halo rows / operand panels and then meet in the same 4 MiB L2.  Bijective for any count.
__device__ __forceinline__ void xcd_tile(int nm, int nn, int& bm, int& bn) {
  const int total = nm * nn;
  const int b = blockIdx.x + blockIdx.y * nm;
  const int q = total >> 3, r = total & 7;
  const int xcd = b & 7, idx = b >> 3;
  // XCD x owns q (+1 if x < r) consecutive tiles starting at x*q + min(x, r)
  const int t = xcd * q + (xcd < r ? xcd : r) + idx;
  bm = t % nm;
  bn = t / nm;
}

struct Pending {
  float4 v[4];
  float4 sc, sh;      // optional per-channel affine (conv loaders)
  unsigned mask;      // VEC4: bit j = row j valid; scalar: bits 4j..4j+3 = elements of row j valid
};

// ---- plain row-major operand loader: 4 rows per thread (r = row0 + (tid>>3) + 32*j), float4 at k = kstep*32 + (tid&7)*4
template <bool VEC4>
struct PlainLoader {
  const float* base;
  long ld;
  int rows, K;
  __device__ __forceinline__ void init(int) {}
  __device__ __forceinline__ void issue(int row0, int kstep, int nrow_iters, Pending& p) const {
    const int c = (ltid() % KT) * 4 + kstep * BK;
    p.mask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nrow_iters) break;
      const int r = row0 + (ltid() / KT) + RPP * j;
      if (VEC4) {
        const bool ok = (r < rows) && (c + 4 <= K);
        const float* q = ok ? base + (long)r * ld + c : base;
        p.v[j] = *reinterpret_cast<const float4*>(q);
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
  __device__ __forceinline__ void finish(int nrow_iters, Pending& p) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nrow_iters) break;
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
// An epilogue's run() receives the wave's accumulators; element (i, j, r) of a wave at (wm, wn) is
//   m = row0 + wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*lh,   n = col0 + wn*(BN/2) + j*32 + li.
struct PlainEpilogue {
  float* C;
  long ldc;
  const float* bias;
  int accumulate;
  template <int BN, int NTN>
  __device__ __forceinline__ void run(const f32x16 (&acc)[2][NTN], int row0, int col0, int wm, int wn, int li, int lh,
                                      int M, int N, float*, bool compute_wave) const {
    if (!compute_wave) return;
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

// The block-level mainloop.  ALoader must provide load(row0, kstep, 4, regs).
template <int BN, class ALoader, class BLoader, class Epilogue>
__device__ __forceinline__ void nt_block(ALoader al, BLoader bl, int M, int N, int K, int block_m, int block_n,
                                         const Epilogue& ep, NtSmem<BN>& sm) {
  constexpr int NTN = BN / 64;  // MFMA tiles per wave along N
  constexpr int AROWS = BM / RPP;  // A row iterations per thread
  constexpr int BROWS = BN / RPP;  // B row iterations per thread
  static_assert(AROWS <= 4 && BROWS <= 4 && BROWS >= 1, "Pending holds 4 float4");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int row0 = block_m * BM, col0 = block_n * BN;
  const int nk = (K + BK - 1) / BK;

  f32x16 acc[2][NTN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#if !ACVAE_NT_WS
  al.init(row0);
  bl.init(col0);
#endif
  const int srow = tid / KT, scol = (tid % KT) * 4;
  [[maybe_unused]] auto stash = [&](int buf, Pending& pa, Pending& pb) {
    al.finish(AROWS, pa);
    bl.finish(BROWS, pb);
#pragma unroll
    for (int j = 0; j < AROWS; ++j) *reinterpret_cast<float4*>(&sm.a[buf][(srow + RPP * j) * LDS_LD + scol]) = pa.v[j];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) *reinterpret_cast<float4*>(&sm.b[buf][(srow + RPP * j) * LDS_LD + scol]) = pb.v[j];
  };
  auto compute = [&](int cur) {
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
  };

#if ACVAE_NT_WS
  // Warp-specialised schedule: wavefronts 0-3 (one per SIMD) only read LDS and issue MFMAs; wavefronts 4-7 (their
  // SIMD partners) only load, activate and store the next tile.  Two identical waves sharing a SIMD finish their MFMA
  // phase together and then both stall the matrix pipe while they stage (measured: 124 vs 148 TFLOP/s without
  // staging); with split roles the matrix pipe always has a wave ready and the staging VALU/VMEM work co-issues.
  if (wave >= 4) {
    // loader wavefronts: two register sets, tiles requested TWO K-steps before they are written to the 2-deep LDS ring
    Pending pa0, pb0, pa1, pb1;
    al.init(row0);
    bl.init(col0);
    const int lt = ltid();
    const int srow_l = lt / KT, scol_l = (lt % KT) * 4;
    auto stash_l = [&](int buf, Pending& pa, Pending& pb) {
      al.finish(AROWS, pa);
      bl.finish(BROWS, pb);
#pragma unroll
      for (int j = 0; j < AROWS; ++j)
        *reinterpret_cast<float4*>(&sm.a[buf][(srow_l + RPP * j) * LDS_LD + scol_l]) = pa.v[j];
#pragma unroll
      for (int j = 0; j < BROWS; ++j)
        *reinterpret_cast<float4*>(&sm.b[buf][(srow_l + RPP * j) * LDS_LD + scol_l]) = pb.v[j];
    };
    al.issue(row0, 0, AROWS, pa0);
    bl.issue(col0, 0, BROWS, pb0);
    if (nk > 1) {
      al.issue(row0, 1, AROWS, pa1);
      bl.issue(col0, 1, BROWS, pb1);
    }
    stash_l(0, pa0, pb0);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
      if (ks + 2 < nk) {
        al.issue(row0, ks + 2, AROWS, pa0);
        bl.issue(col0, ks + 2, BROWS, pb0);
      }
      if (ks + 1 < nk) stash_l(1, pa1, pb1);
      __syncthreads();
      if (ks + 1 >= nk) break;
      if (ks + 3 < nk) {
        al.issue(row0, ks + 3, AROWS, pa1);
        bl.issue(col0, ks + 3, BROWS, pb1);
      }
      if (ks + 2 < nk) stash_l(0, pa0, pb0);
      __syncthreads();
    }
  } else {
#ifdef ACVAE_WS_PRIO
    __builtin_amdgcn_s_setprio(ACVAE_WS_PRIO);   // matrix waves win issue arbitration against their loader partners
#endif
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      compute(ks & 1);
      __syncthreads();
    }
#ifdef ACVAE_WS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  }
#else
  // Global loads run TWO K-steps ahead of the MFMAs (two register sets, statically named so nothing is
  // dynamically indexed), the LDS ring is two deep: tile ks+2 is requested before the MFMAs of tile ks, tile ks+1
  // (requested one iteration earlier) is activated + written to LDS after them.  One barrier per K-step.
  Pending pa0, pb0, pa1, pb1;
  al.issue(row0, 0, AROWS, pa0);
  bl.issue(col0, 0, BROWS, pb0);
  if (nk > 1) {
    al.issue(row0, 1, AROWS, pa1);
    bl.issue(col0, 1, BROWS, pb1);
  }
  stash(0, pa0, pb0);
  __syncthreads();
  for (int ks = 0; ks < nk; ks += 2) {
    // even step: compute tile ks from LDS[0]; slot 0 registers are free -> request tile ks+2 into them
    if (ks + 2 < nk) {
      al.issue(row0, ks + 2, AROWS, pa0);
      bl.issue(col0, ks + 2, BROWS, pb0);
    }
    compute(0);
    if (ks + 1 < nk) stash(1, pa1, pb1);
    __syncthreads();
    if (ks + 1 >= nk) break;
    // odd step: compute tile ks+1 from LDS[1]; request tile ks+3 into slot 1
    if (ks + 3 < nk) {
      al.issue(row0, ks + 3, AROWS, pa1);
      bl.issue(col0, ks + 3, BROWS, pb1);
    }
    compute(1);
    if (ks + 2 < nk) stash(0, pa0, pb0);
    __syncthreads();
  }

#endif

  // all waves are past the last LDS read (barrier above): LDS is free for the epilogue
#if ACVAE_NT_WS
  ep.template run<BN, NTN>(acc, row0, col0, wm, wn, li, lh, M, N, &sm.a[0][0], wave < 4);
#else
  ep.template run<BN, NTN>(acc, row0, col0, wm, wn, li, lh, M, N, &sm.a[0][0], true);
#endif
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

// plain k-major loader: tile [BKT][W] floats, W = 64*WX; thread loads float4 at k = (tid / (W/4)) + it*(256/(W/4)), col = (tid % (W/4))*4
template <bool VEC4>
struct PlainKMajorLoader {
  const float* base;
  long ld;
  int cols, K;  // cols = M or N extent
  template <int W>
  __device__ __forceinline__ void init(int) {}
  template <int W>
  __device__ __forceinline__ void issue(int col0, int k0, Pending& p) const {
    constexpr int TPR = W / 4;           // threads per k-row
    constexpr int ROWS_PER_IT = 256 / TPR;
    constexpr int ITS = BKT / ROWS_PER_IT;
    const int c = col0 + (threadIdx.x % TPR) * 4;
    p.mask = 0;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int k = k0 + threadIdx.x / TPR + it * ROWS_PER_IT;
      if (VEC4) {
        const bool ok = (k < K) && (c + 4 <= cols);
        const float* q = ok ? base + (long)k * ld + c : base;
        p.v[it] = *reinterpret_cast<const float4*>(q);
        p.mask |= (ok ? 1u : 0u) << it;
      } else {
        const bool kok = k < K;
        const float* q = base + (kok ? (long)k * ld : 0);
        const bool o0 = kok && c + 0 < cols, o1 = kok && c + 1 < cols, o2 = kok && c + 2 < cols, o3 = kok && c + 3 < cols;
        p.v[it].x = q[o0 ? c + 0 : 0]; p.v[it].y = q[o1 ? c + 1 : 0];
        p.v[it].z = q[o2 ? c + 2 : 0]; p.v[it].w = q[o3 ? c + 3 : 0];
        p.mask |= ((o0 ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u)) << (4 * it);
      }
    }
  }
  template <int W>
  __device__ __forceinline__ void finish(Pending& p) const {
    constexpr int ITS = BKT / (256 / (W / 4));
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

template <int WM, int WN, class ALoader, class BLoader>
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
  Pending pa, pb;
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
  if (nk > 0) {
    al.template issue<TM>(row0, k_begin, pa);
    bl.template issue<TN_>(col0, k_begin, pb);
    stash(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      al.template issue<TM>(row0, k_begin + (ks + 1) * BKT, pa);
      bl.template issue<TN_>(col0, k_begin + (ks + 1) * BKT, pb);
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
          *p = v;
        }
      }
    }
}

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
