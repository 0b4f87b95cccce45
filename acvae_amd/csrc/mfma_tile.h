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

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;  // floats per LDS row
constexpr int NT_THREADS = 256;

template <int BN>
struct NtSmem {
  float a[2][BM * LDS_LD];
  float b[2][BN * LDS_LD];
};

// ---- plain row-major operand loader: rows r0.., 4 rows per thread (r = (tid>>3) + 32*j), float4 at k = kstep*32 + (tid&7)*4
template <bool VEC4>
struct PlainLoader {
  const float* base;
  long ld;
  int rows, K;
  __device__ __forceinline__ void init(int) {}
  __device__ __forceinline__ void load(int row0, int kstep, int nrow_iters, float4* regs) const {
    const int c = (threadIdx.x & 7) * 4 + kstep * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nrow_iters) break;
      const int r = row0 + (threadIdx.x >> 3) + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows) {
        const float* p = base + (long)r * ld + c;
        if (VEC4) {
          if (c + 4 <= K) v = *reinterpret_cast<const float4*>(p);
        } else {
          if (c + 0 < K) v.x = p[0];
          if (c + 1 < K) v.y = p[1];
          if (c + 2 < K) v.z = p[2];
          if (c + 3 < K) v.w = p[3];
        }
      }
      regs[j] = v;
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
                                      int M, int N, float*) const {
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
  constexpr int BROWS = BN / 32;  // B row iterations per thread
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

  float4 ra[4], rb[4];
  al.init(row0);
  bl.init(col0);
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&sm.a[buf][(srow + 32 * j) * LDS_LD + scol]) = ra[j];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) *reinterpret_cast<float4*>(&sm.b[buf][(srow + 32 * j) * LDS_LD + scol]) = rb[j];
  };

  al.load(row0, 0, 4, ra);
  bl.load(col0, 0, BROWS, rb);
  stash(0);
  __syncthreads();

  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      al.load(row0, ks + 1, 4, ra);
      bl.load(col0, ks + 1, BROWS, rb);
    }
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
    if (ks + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }

  // all waves are past the last LDS read (barrier above): LDS is free for the epilogue
  ep.template run<BN, NTN>(acc, row0, col0, wm, wn, li, lh, M, N, &sm.a[0][0]);
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
struct TnSmem {
  float a[2][BKT * 64 * WM];
  float b[2][BKT * 64 * WN];
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
  __device__ __forceinline__ void load(int col0, int k0, float4* regs) const {
    constexpr int TPR = W / 4;           // threads per k-row
    constexpr int ROWS_PER_IT = 256 / TPR;
    constexpr int ITS = BKT / ROWS_PER_IT;
    const int c = col0 + (threadIdx.x % TPR) * 4;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int k = k0 + threadIdx.x / TPR + it * ROWS_PER_IT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K) {
        const float* p = base + (long)k * ld + c;
        if (VEC4) {
          if (c + 4 <= cols) v = *reinterpret_cast<const float4*>(p);
        } else {
          if (c + 0 < cols) v.x = p[0];
          if (c + 1 < cols) v.y = p[1];
          if (c + 2 < cols) v.z = p[2];
          if (c + 3 < cols) v.w = p[3];
        }
      }
      regs[it] = v;
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

  float4 ra[AITS], rb[BITS];
  al.template init<TM>(row0);
  bl.template init<TN_>(col0);
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < AITS; ++it) {
      const int k = tid / (TM / 4) + it * (256 / (TM / 4));
      *reinterpret_cast<float4*>(&sm.a[buf][k * TM + (tid % (TM / 4)) * 4]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < BITS; ++it) {
      const int k = tid / (TN_ / 4) + it * (256 / (TN_ / 4));
      *reinterpret_cast<float4*>(&sm.b[buf][k * TN_ + (tid % (TN_ / 4)) * 4]) = rb[it];
    }
  };
  if (nk > 0) {
    al.template load<TM>(row0, k_begin, ra);
    bl.template load<TN_>(col0, k_begin, rb);
    stash(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      al.template load<TM>(row0, k_begin + (ks + 1) * BKT, ra);
      bl.template load<TN_>(col0, k_begin + (ks + 1) * BKT, rb);
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
                                              int N, int K, int m0, int n0, int wave, int li, int lh) {
  const bool aval = (m0 + li) < M, bval = (n0 + li) < N;
  const float* ap = A + (long)(aval ? m0 + li : 0) * lda;
  const float* bp = B + (long)(bval ? n0 + li : 0) * ldb;
  const int G = (K + 7) / 8;
  for (int g = wave; g < G; g += 4 * SK_WAVES) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (g + u * SK_WAVES) * 8 + 4 * lh;
      a[u] = aval ? sk_load<VEC4>(ap, k, K) : make_float4(0.f, 0.f, 0.f, 0.f);  // k >= K reads give zeros
      b[u] = bval ? sk_load<VEC4>(bp, k, K) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
}

}  // namespace mfma
