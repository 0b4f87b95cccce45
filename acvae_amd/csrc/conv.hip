// A1: the Cnn10 audio encoder's convolution stack (models/encoder.py:606-707) on gfx950.
//
// Data layout in HBM: activations are NHWC = [clip][time][mel][channel] fp32 (the reference is NCHW);
// with channels innermost the K index of the implicit GEMM (tap, ci) is contiguous per tap, so every
// im2col row segment is a coalesced 128-B read, and the GEMM's N index (cout) is contiguous in the
// output.  Weights are repacked once per step from the state-dict's OIHW to [cout][tap][cin].
//
// conv3x3 as implicit GEMM on fp32 MFMA:  Y[p, co] = sum_{tap,ci} act(X[p+tap, ci]) * W[co, tap, ci]
//   M = N*H*W pixels, N = Cout, K = 9*Cin.  `act` is the PREVIOUS layer's BatchNorm+ReLU applied while
//   staging the operand (relu(x*scale[ci]+shift[ci]), zero outside the image), so the normalised
//   tensor is never written to HBM; the epilogue writes the raw conv output and this layer's
//   per-channel sum / sum-of-squares partials for the batch statistics.
// The same kernel computes the data gradient (X = dY, W = flipped/transposed weights, no act, no stats).
// The weight gradient is the k-major (TN) MFMA kernel with the shifted/activated operand loader,
// split over pixels (split-K) into slabs that are reduced in fixed order.
#include "mfma_tile.h"
#include "../../include/acvae_hip.h"
#include "conv.h"
#include "prof.h"

namespace {
using namespace mfma;

// ------------------------------------------------------------------ im2col loaders
// x / d for 0 <= x < 2^40 / d by one 64-bit multiply (mul = ceil(2^40 / d)): the pixel -> (row, column) split of the
// weight-gradient loader runs every K-step, where two integer divisions cost more issue slots than the loads they feed
struct FastDiv {
  unsigned long long mul;
  static FastDiv make(int d) { return FastDiv{((1ULL << 40) + (unsigned long long)d - 1) / (unsigned long long)d}; }
  __device__ __forceinline__ int div(int x) const { return (int)(((unsigned long long)(unsigned)x * mul) >> 40); }
};

struct ConvKMajorLoader {  // TN B-operand: k = pixel, n = (tap, ci)
  const float* X;
  const float* scale;
  const float* shift;
  int H, W, C, M, NC;  // NC = 9*C
  FastDiv dW, dH;
  int dy, dx, ci;
  bool colok;
  float4 sc, sh;
  template <int WT>
  __device__ __forceinline__ void init(int col0) {
    const int col = col0 + (threadIdx.x % (WT / 4)) * 4;
    colok = col < NC;
    const int tap = colok ? col / C : 0;
    ci = colok ? col - tap * C : 0;
    dy = tap / 3 - 1; dx = tap % 3 - 1;
    sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale && colok) {
      sc = *reinterpret_cast<const float4*>(scale + ci);
      sh = *reinterpret_cast<const float4*>(shift + ci);
    }
  }
  template <int WT>
  __device__ __forceinline__ void issue(int, int k0, Pending<4>& p) const {
    constexpr int TPR = WT / 4, RPI = 256 / TPR, ITS = tn_its<WT>();
    constexpr bool EXACT = (RPI * TPR == 256) && (BKT % RPI == 0);
    p.mask = 0;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int kr = threadIdx.x / TPR + it * RPI;
      const int px = k0 + kr;
      bool ok = colok && px < M && (EXACT || (threadIdx.x < RPI * TPR && kr < BKT));
      const int row = dW.div(px);
      const int w = px - row * W, h = row - dH.div(row) * H;
      const int hh = h + dy, ww = w + dx;
      ok = ok && hh >= 0 && hh < H && ww >= 0 && ww < W;
      p.v[it] = *reinterpret_cast<const float4*>(ok ? X + ((long)px + dy * W + dx) * C + ci : X);
      p.mask |= (ok ? 1u : 0u) << it;
    }
  }
  template <int WT>
  __device__ __forceinline__ void finish(Pending<4>& p) const {
    constexpr int ITS = tn_its<WT>();
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      float4 v = p.v[it];
      if (scale) {
        v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
        v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
      }
      p.v[it] = ((p.mask >> it) & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// ------------------------------------------------------------------ conv epilogue: raw store + BN partials
struct ConvStatsEpilogue {
  float* Y;          // [M][Cout]
  float* partials;   // [gridM][2][Cout] or nullptr
  int Cout;
  template <int BMT, int BN, int NTN>
  __device__ __forceinline__ void run(const f32x16 (&acc)[2][NTN], int row0, int col0, int wm, int wn, int li, int lh,
                                      int M, int N, float* lds, bool matrix_wave) const {
    constexpr int WMS = BMT / 64;   // matrix waves along M
    if (matrix_wave) {
#pragma unroll
      for (int j = 0; j < NTN; ++j) {
        const int nl = wn * (BN / 2) + j * 32 + li;
        const int n = col0 + nl;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = acc[i][j][r];
            if (m < M && n < N) Y[(long)m * Cout + n] = v;
            if (m < M) { s += v; q += v * v; }   // tile rows past the last pixel may hold a neighbour's shifted data
          }
        if (partials) {
          s += __shfl_xor(s, 32, 64);
          q += __shfl_xor(q, 32, 64);
          if (lh == 0) {
            lds[(wm * 2 + 0) * BN + nl] = s;
            lds[(wm * 2 + 1) * BN + nl] = q;
          }
        }
      }
    }
    if (partials) {
      __syncthreads();
      const int t = threadIdx.x;
      if (t < BN && col0 + t < N) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WMS; ++w) { s += lds[(w * 2 + 0) * BN + t]; q += lds[(w * 2 + 1) * BN + t]; }
        float* out = partials + (long)(row0 / BMT) * 2 * Cout + col0 + t;
        out[0] = s;
        out[Cout] = q;
      }
    }
  }
};

// ------------------------------------------------------------------ conv3x3 NT block with horizontal-tap reuse
// The A tile of tap (dy, dx) is the A tile of tap (dy, 0) shifted by dx pixels, so one staged strip of
// CV_ROWS = 128 + 2 pixel rows (pixels row0-1 .. row0+128 of the flattened image, at vertical offset dy) serves the
// three horizontal taps: the matrix waves read it at row offsets 0 / 1 / 2 and zero the lanes whose pixel would step
// over the left / right image border.  A-operand global loads drop to a third (they cost 14 % of the kernel's time
// when measured alone, tools/ablate.sh); the weight panel of each tap still streams through its own two-deep ring.
// K order: dy (3) x 32-channel chunk (C/32) x dx (3); weights stay in the (tap, ci)-major layout of repack_fwd.
constexpr int CV_BMT = 128;
constexpr int CV_ROWS = CV_BMT + 2;
constexpr int CV_AR = 5;            // float4 slots per loader thread for the strip: 4 full passes + rows 128, 129
// 53.8 KB for BN = 64 with LDS-DMA panels (three workgroups per CU), 74.9 KB for BN = 128 (two)
template <int BN, bool BDMA>
struct alignas(16) ConvSmem {
  alignas(16) float a[2][CV_ROWS * LDS_LD];
  alignas(16) float b[2][BN * (BDMA ? BK : LDS_LD)];
};

struct ConvStripLoader {   // strip row j <-> pixel q = row0 - 1 + j, read at vertical tap offset dy
  const float* X;
  const float* scale;  // nullable: act = identity
  const float* shift;
  int H, W, C, M;
  int qh[CV_AR];
  long qbase[CV_AR];
  int cq;
  __device__ __forceinline__ void init(int row0, int lt) {
    cq = (lt % KT) * 4;
#pragma unroll
    for (int j = 0; j < CV_AR; ++j) {
      const int r = (lt / KT) + RPP * j;                 // strip row; slot 4 exists only for rows 128, 129
      const long q = (long)row0 - 1 + r;
      const bool live = r < CV_ROWS && q >= 0 && q < M;
      qh[j] = live ? (int)((q / W) % H) : -100000;
      qbase[j] = live ? q * C : 0;
    }
  }
  // group = (dy index, 32-channel chunk)
  __device__ __forceinline__ void issue(int dyi, int chunk, Pending<CV_AR>& p) const {
    const int ci = chunk * BK + cq;
    const int dy = dyi - 1;
    const long off = (long)dy * W * C + ci;
    if (scale) {
      p.sc = *reinterpret_cast<const float4*>(scale + ci);
      p.sh = *reinterpret_cast<const float4*>(shift + ci);
    }
    p.mask = 0;
#pragma unroll
    for (int j = 0; j < CV_AR; ++j) {
      const int hh = qh[j] + dy;
      const bool ok = hh >= 0 && hh < H;
      p.v[j] = *reinterpret_cast<const float4*>(ok ? X + qbase[j] + off : X);   // always a legal address
      p.mask |= (ok ? 1u : 0u) << j;
    }
  }
  __device__ __forceinline__ void finish(Pending<CV_AR>& p) const {
#pragma unroll
    for (int j = 0; j < CV_AR; ++j) {
      float4 v = p.v[j];
      if (scale) {
        v.x = fmaxf(v.x * p.sc.x + p.sh.x, 0.f); v.y = fmaxf(v.y * p.sc.y + p.sh.y, 0.f);
        v.z = fmaxf(v.z * p.sc.z + p.sh.z, 0.f); v.w = fmaxf(v.w * p.sc.w + p.sh.w, 0.f);
      }
      p.v[j] = ((p.mask >> j) & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// BDMA: the weight panel goes global -> LDS by LDS-DMA (no register round trip): the LDS image of a panel is then
// lane-linear, [BN rows][32 floats] unpadded, and the bank spread comes from an XOR swizzle of the 16-byte chunk index
// with the row (chunk c of row r sits at position c ^ (r & 7)), applied on the SOURCE address by the loader and on the
// read address by the matrix waves.  Needs N % BN == 0 (an LDS-DMA cannot zero-fill).
template <int BN, bool BDMA, class Epilogue>
__device__ __forceinline__ void conv_nt_block(ConvStripLoader al, PlainLoader<true, BN / 32> bl, int M, int N, int C,
                                              int W, int block_m, int block_n, const Epilogue& ep,
                                              ConvSmem<BN, BDMA>& sm) {
  // Matrix wavefronts 2 x 2: (64-row half, BN/2-column half).  For BN = 64 a variant whose second index split the
  // 32-channel K-step instead (every wave a 64 x 64 tile, fewer LDS fragment reads per MFMA, halves added through LDS
  // before the epilogue) measured 1353 -> 1390 us: that kernel is not bound by LDS reads (nor by occupancy: three
  // workgroups per CU instead of two changed nothing either) but by its strip fetches - K = 9 * 64 is only 18 stages
  // per tile and every tile reads its image rows three times.
  constexpr int NTN = BN / 64;     // MFMA tiles per matrix wave along N
  constexpr int NG = BK / 8;       // 8-channel groups per K-step
  constexpr int NMW = CV_BMT / 32;
  constexpr int BR = BN / RPP;     // B rows per loader thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool matrix_wave = wave < NMW;
  const int wm = (wave >> 1) % (CV_BMT / 64), wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int row0 = block_m * CV_BMT, col0 = block_n * BN;
  const int nchunk = C / BK;
  const int ngrp = 3 * nchunk;      // (dy, chunk) groups, three horizontal taps each

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
    Pending<CV_AR> pa;
    Pending<BR> pb;
    al.init(row0, lt);
    bl.init(col0, lt);
    auto put_a = [&](int buf) {
      al.finish(pa);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&sm.a[buf][(srow + RPP * j) * LDS_LD + scol]) = pa.v[j];
      if (srow < CV_ROWS - 4 * RPP) *reinterpret_cast<float4*>(&sm.a[buf][(srow + RPP * 4) * LDS_LD + scol]) = pa.v[4];
    };
    auto put_b = [&](int buf) {
      if (BDMA) return;
      bl.finish(pb);
#pragma unroll
      for (int j = 0; j < BR; ++j) *reinterpret_cast<float4*>(&sm.b[buf][(srow + RPP * j) * LDS_LD + scol]) = pb.v[j];
    };
    // one LDS-DMA instruction = 8 rows x 128 B: wave lw takes the row groups lw, lw+4, ...
    auto fetch_b = [&](int buf, int kstep) {
      if (!BDMA) { bl.issue(col0, kstep, lt, pb); return; }
      const int lw = lt >> 6, l = lt & 63;
      const int chunk = (l & 7) ^ (l >> 3);
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        const int rg = lw + 4 * j;
        const float* src = bl.base + (long)(col0 + rg * 8 + (l >> 3)) * bl.ld + kstep * BK + chunk * 4;
        __builtin_amdgcn_global_load_lds(src, &sm.b[buf][rg * 8 * BK], 16, 0, 0);
      }
    };
    // weight K-step of sub-stage (grp, dxi): tap = dyi*3 + dxi, channels chunk*32..
    auto kstep_of = [&](int grp, int dxi) { const int dyi = grp / nchunk; return (dyi * 3 + dxi) * nchunk + (grp - dyi * nchunk); };
    // An LDS-DMA is ordered for the readers only by the issuing wave's vmcnt wait followed by a barrier; hipcc places
    // that wait in front of __syncthreads() by itself, the explicit one keeps the kernel independent of that.
    auto dma_wait = [&]() { if (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    al.issue(0, 0, pa);
    fetch_b(0, kstep_of(0, 0));
    put_a(0);
    put_b(0);
    dma_wait();
    __syncthreads();
    for (int grp = 0; grp < ngrp; ++grp) {
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        const int s = grp * 3 + dxi;
        const int g1 = grp + 1;
        if (dxi == 0 && g1 < ngrp) {             // the next strip's loads fly for three sub-stages
          const int dyi = g1 / nchunk;
          al.issue(dyi, g1 - dyi * nchunk, pa);
        }
        if (dxi < 2) {
          fetch_b((s + 1) & 1, kstep_of(grp, dxi + 1));
          put_b((s + 1) & 1);
        } else if (g1 < ngrp) {
          fetch_b((s + 1) & 1, kstep_of(g1, 0));
          put_a(g1 & 1);
          put_b((s + 1) & 1);
        }
        dma_wait();
        __syncthreads();
      }
    }
  } else {
    // ------------------------------------------------------------------ matrix wavefronts
    // border masks of this lane's two tile rows: pixel p = row0 + wm*64 + i*32 + li
    bool okl[2], okr[2];
    // 16-byte chunk of this lane's first weight fragment in the swizzled panel row: k pair ^ (row & 7); the group index
    // g * 2 occupies other bits than lh, so it can be xor-ed in
    const int bsw = lh ^ (li & 7);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = row0 + wm * 64 + i * 32 + li;
      const int w = px % W;
      okl[i] = w > 0; okr[i] = w < W - 1;
    }
    __syncthreads();
    for (int grp = 0; grp < ngrp; ++grp) {
      const float* Ag = sm.a[grp & 1] + (wm * 64 + li) * LDS_LD + 4 * lh;
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        const float* As = Ag + dxi * LDS_LD;                 // strip row of pixel p + (dxi - 1) is (p - row0) + dxi
        const float* Bs = sm.b[(grp * 3 + dxi) & 1] + (wn * (BN / 2) + li) * (BDMA ? BK : LDS_LD) + (BDMA ? 0 : 4 * lh);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          float4 af[2], bf[NTN];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            af[i] = *reinterpret_cast<const float4*>(As + i * 32 * LDS_LD + g * 8);
            if (dxi == 0 && !okl[i]) af[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dxi == 2 && !okr[i]) af[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < NTN; ++j)
            bf[j] = BDMA ? *reinterpret_cast<const float4*>(Bs + j * 32 * BK + (((g * 2) ^ bsw) << 2))
                         : *reinterpret_cast<const float4*>(Bs + j * 32 * LDS_LD + g * 8);
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
  }
  // every wave is past the last LDS access (barrier above): LDS is free for the epilogue
  ep.template run<CV_BMT, BN, NTN>(acc, row0, col0, wm, wn, li, lh, M, N, &sm.a[0][0], matrix_wave);
}

template <int BN, bool BDMA>
__global__ __launch_bounds__(nt_threads<CV_BMT>(), 2) void conv_igemm3_kernel(ConvStripLoader al,
                                                                              const float* __restrict__ Wp,
                                                                              ConvStatsEpilogue ep, int M, int Cout,
                                                                              int K) {
  __shared__ ConvSmem<BN, BDMA> sm;
  PlainLoader<true, BN / 32> bl{Wp, K, Cout, K};
  int bm, bn;
  xcd_tile(gridDim.x, gridDim.y, bm, bn);
  conv_nt_block<BN, BDMA>(al, bl, M, Cout, al.C, al.W, bm, bn, ep, sm);
}

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ dY, ConvKMajorLoader bl,
                                                            float* __restrict__ slab, int M, int Cout, int NC,
                                                            int k_per) {
  __shared__ TnSmem<WM, WN> sm;
  PlainKMajorLoader<true> al{dY, Cout, Cout, M};
  // XCD-aware order: all (m, n) tiles of one pixel slice z read the same dY rows and the same (shifted) activation
  // rows, so run them back to back on ONE XCD (workgroups are dealt round-robin: linear id % 8): XCD c takes the
  // slices z = c, c+8, ...  (gridDim.z is a multiple of 8; only speed depends on the placement)
  const int tiles = gridDim.x * gridDim.y;
  const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xcd = b & 7, idx = b >> 3;
  const int z = (idx / tiles) * 8 + xcd, tile = idx % tiles;
  const int bx = tile % gridDim.x, by = tile / gridDim.x;
  const int kb = z * k_per;
  const int ke = min(M, kb + k_per);
  tn_block<WM, WN>(al, bl, Cout, NC, kb, ke, bx, by, slab + (long)z * Cout * NC, NC, 0, sm);
}

// ------------------------------------------------------------------ weight gradient with horizontal-tap reuse
// One workgroup owns a 128 (or 64) x 192 tile of dW whose columns are the THREE horizontal taps of one vertical tap dy
// for 64 input channels.  Per K-step of 16 pixels it stages ONE strip of 18 pixels x 64 channels (pixels k0-1 .. k0+16 at
// vertical offset dy, BN+ReLU applied once) and the three taps read it at row offsets 0 / 1 / 2; pixels whose horizontal
// neighbour would cross the image border are switched off by per-pixel flags staged beside the strip (the border
// depends on the reduction index here, not on the lane).  Activation loads per K-step: 18 x 64 instead of 16 x 192.
template <int EM>
struct alignas(16) WgStripSmem {
  alignas(16) float a[2][BKT * EM * 64];
  alignas(16) float s[2][(BKT + 2) * 64];
  alignas(16) float mk[2][2 * BKT];       // [0..BKT) left-neighbour ok, [BKT..2BKT) right-neighbour ok (1.0 / 0.0)
};

template <int EM>
__global__ __launch_bounds__(256, 3) void conv_wgrad3_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ slab,
                                                             int M, int Cout, int H, int W, int C, FastDiv dW_,
                                                             FastDiv dH_, int k_per) {
  constexpr int TM = EM * 64;
  constexpr int AITS = tn_its<TM>();
  __shared__ WgStripSmem<EM> sm;
  const int NC = 9 * C;
  // XCD-aware order as in conv_wgrad_kernel: the tiles of one pixel slice run back to back on one XCD
  const int tiles = gridDim.x * gridDim.y;
  const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xcd = b & 7, idx = b >> 3;
  const int z = (idx / tiles) * 8 + xcd, tile = idx % tiles;
  const int bx = tile % gridDim.x, by = tile / gridDim.x;
  const int ncg = C / 64;
  const int dyi = by / ncg, c0 = (by - dyi * ncg) * 64, dy = dyi - 1;
  const int k_begin = z * k_per, k_end = min(M, k_begin + k_per);
  float* Cs = slab + (long)z * Cout * NC;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int row0 = bx * TM;
  const int nk = (k_end - k_begin + BKT - 1) / BKT;

  // Every wave holds 64 rows x 96 columns (paired LDS reads need two row tiles and two channel tiles per wave).  EM = 2:
  // the waves split the 128 rows.  EM = 1 (64 output channels): there are no second 64 rows to give to wm, and one row
  // tile per wave costs 6 LDS reads per 3 MFMAs, which saturates the LDS (measured 101 TFLOP/s); instead wm splits the
  // K-step - wave (wm, wn) takes the pixel pairs kk = wm, wm + 2, ... - and the two halves go to two slabs of their own
  // (the slabs of this layer are 147 KB each), summed by wgrad_reduce_kernel like any other pair of slices.
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  PlainKMajorLoader<true> al{dY, Cout, Cout, M};
  Pending<4> pa;
  // strip slots of this thread: slot 0 = row tid/16 (0..15), slot 1 = rows 16, 17 for tid < 32; 4 channels at (tid%16)*4
  const int sc4 = (tid & 15) * 4;
  float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale) {
    bsc = *reinterpret_cast<const float4*>(scale + c0 + sc4);
    bsh = *reinterpret_cast<const float4*>(shift + c0 + sc4);
  }
  float4 sv[2];
  bool sok[2];
  float mkv = 0.f;
  auto issue = [&](int k0) {
    al.template issue<TM>(row0, k0, pa);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int r = (tid >> 4) + 16 * sl;
      const long q = (long)k0 - 1 + r;
      bool ok = r < BKT + 2 && q >= 0 && q < M;
      if (ok) {
        const int rowq = dW_.div((int)q);
        const int h = rowq - dH_.div(rowq) * H + dy;
        ok = h >= 0 && h < H;
      }
      sok[sl] = ok;
      sv[sl] = *reinterpret_cast<const float4*>(ok ? X + (q + (long)dy * W) * C + c0 + sc4 : X);
    }
    if (tid < 2 * BKT) {            // border flags of the BKT pixels of this K-step
      const int k = tid < BKT ? tid : tid - BKT;
      const int p = k0 + k;
      const int w = p - dW_.div(p) * W;
      mkv = (p < M && (tid < BKT ? w > 0 : w < W - 1)) ? 1.f : 0.f;
    }
  };
  auto stash = [&](int buf) {
    al.template finish<TM>(pa);
#pragma unroll
    for (int it = 0; it < AITS; ++it) {
      const int k = tid / (TM / 4) + it * (256 / (TM / 4));
      *reinterpret_cast<float4*>(&sm.a[buf][k * TM + (tid % (TM / 4)) * 4]) = pa.v[it];
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int r = (tid >> 4) + 16 * sl;
      if (r < BKT + 2) {
        float4 v = sv[sl];
        if (scale) {
          v.x = fmaxf(v.x * bsc.x + bsh.x, 0.f); v.y = fmaxf(v.y * bsc.y + bsh.y, 0.f);
          v.z = fmaxf(v.z * bsc.z + bsh.z, 0.f); v.w = fmaxf(v.w * bsc.w + bsh.w, 0.f);
        }
        if (!sok[sl]) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&sm.s[buf][r * 64 + sc4]) = v;
      }
    }
    if (tid < 2 * BKT) sm.mk[buf][tid] = mkv;
  };
  if (nk > 0) { issue(k_begin); stash(0); }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) issue(k_begin + (ks + 1) * BKT);
    {
      // paired reads (as in tn_block): one ds_read_b64 gives two adjacent output rows / two adjacent channels, so MFMA tile
      // e of the A side covers rows 2*rho + e and the two tiles of the wave's full tap (dx = 0 for wn = 0, dx = 2 for
      // wn = 1) cover channels 2*li + e; the third tile is the wave's half of the centre tap (channels wn*32 + li)
      const float* As = sm.a[cur] + lh * TM + (EM == 2 ? wm * 64 : 0) + 2 * li;
      const float* Sp = sm.s[cur] + (lh + 2 * wn) * 64 + 2 * li;      // strip row k + dx, dx = 2*wn
      const float* Sc = sm.s[cur] + (lh + 1) * 64 + wn * 32 + li;     // centre tap
      const float* Mk = sm.mk[cur] + wn * BKT + lh;
#pragma unroll
      for (int kq = 0; kq < BKT / 2 / (EM == 2 ? 1 : 2); ++kq) {
        const int kk = EM == 2 ? kq : 2 * kq + wm;
        const float2 af = *reinterpret_cast<const float2*>(As + kk * 2 * TM);
        float2 bp = *reinterpret_cast<const float2*>(Sp + kk * 2 * 64);
        const float bc = Sc[kk * 2 * 64];
        const float m = Mk[kk * 2];
        bp.x *= m; bp.y *= m;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bp.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bp.y, acc[0][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bc, acc[0][2], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bp.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bp.y, acc[1][1], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bc, acc[1][2], 0, 0, 0);
      }
    }
    if (ks + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }
  // slab[co][(dy*3 + dx)*C + c]; EM = 1: the K-halves of the two wave rows go to slabs 2z and 2z + 1
  float* Cw = EM == 2 ? Cs : slab + (long)(2 * z + wm) * Cout * NC;
#pragma unroll
  for (int em = 0; em < 2; ++em)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = row0 + (EM == 2 ? wm * 64 : 0) + 2 * rho + em;
      if (m >= Cout) continue;
      float* out = Cw + (long)m * NC + dyi * 3 * C + c0;
      out[2 * wn * C + 2 * li + 0] = acc[em][0][r];
      out[2 * wn * C + 2 * li + 1] = acc[em][1][r];
      out[C + wn * 32 + li] = acc[em][2][r];
    }
}

// dW_oihw[co][ci][tap] = sum_z slab[z][co][tap*Cin + ci]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, float* __restrict__ dW, int Cout,
                                    int Cin) {
  const long total = (long)Cout * 9 * Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // four independent partial sums keep 4 loads in flight; the combination order is fixed -> deterministic
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int z = 0;
    for (; z + 3 < nsplit; z += 4) {
      a0 += slab[(long)z * total + i]; a1 += slab[(long)(z + 1) * total + i];
      a2 += slab[(long)(z + 2) * total + i]; a3 += slab[(long)(z + 3) * total + i];
    }
    for (; z < nsplit; ++z) a0 += slab[(long)z * total + i];
    const float acc = (a0 + a1) + (a2 + a3);
    const int co = (int)(i / (9 * Cin)), rem = (int)(i % (9 * Cin));
    const int tap = rem / Cin, ci = rem % Cin;
    dW[((long)co * Cin + ci) * 9 + tap] = acc;
  }
}

// ------------------------------------------------------------------ weight repacks
// fwd:  Wf[co][tap*Cin + ci] = W[co][ci][tap];   dgrad: Wd[ci][tap'*Cout + co] = W[co][ci][8 - tap']
template <class T>
__global__ void repack_fwd_kernel(const float* __restrict__ W, T* __restrict__ Wf, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i / (9 * Cin)), rem = (int)(i % (9 * Cin));
    const int tap = rem / Cin, ci = rem % Cin;
    Wf[i] = (T)W[((long)co * Cin + ci) * 9 + tap];
  }
}
template <class T>
__global__ void repack_dgrad_kernel(const float* __restrict__ W, T* __restrict__ Wd, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i / (9 * Cout)), rem = (int)(i % (9 * Cout));
    const int tp = rem / Cout, co = rem % Cout;
    Wd[i] = (T)W[((long)co * Cin + ci) * 9 + (8 - tp)];
  }
}

// all layers' repacks in one launch (the bf16 encoder queued fourteen 7-us launches per step in front of its convolutions)
template <class T>
__global__ void repack_batch_kernel(acvae::RepackBatch b) {
  const long total = b.start[b.n];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < b.n && i >= b.start[l + 1]) ++l;
    const long k = i - b.start[l];
    const int Cout = b.Cout[l], Cin = b.Cin[l];
    const float* W = b.W[l];
    T* out = reinterpret_cast<T*>(b.dst[l]);
    if (!b.dgrad[l]) {
      const int co = (int)(k / (9 * Cin)), rem = (int)(k % (9 * Cin));
      const int tap = rem / Cin, ci = rem % Cin;
      out[k] = (T)W[((long)co * Cin + ci) * 9 + tap];
    } else {
      const int ci = (int)(k / (9 * Cout)), rem = (int)(k % (9 * Cout));
      const int tp = rem / Cout, co = rem % Cout;
      out[k] = (T)W[((long)co * Cin + ci) * 9 + (8 - tp)];
    }
  }
}

// ------------------------------------------------------------------ BatchNorm statistics
// bn0 (models/encoder.py:679-681: BatchNorm2d(64) over the mel axis): per-mel sums over all (clip, frame).
__global__ __launch_bounds__(256) void bn0_stats_kernel(const float* __restrict__ x, float* __restrict__ partials,
                                                        long rows, int F, int rows_per_block) {
  __shared__ float red[2][4][64];
  const int w = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  float s = 0.f, q = 0.f;
  if (w < F)
    for (long r = r0 + g; r < r0 + rows_per_block && r < rows; r += 4) {
      const float v = x[r * F + w];
      s += v; q += v * v;
    }
  red[0][g][w] = s; red[1][g][w] = q;
  __syncthreads();
  if (threadIdx.x < 64 && w < F) {
    float* out = partials + (long)blockIdx.x * 2 * F;
    out[w] = red[0][0][w] + red[0][1][w] + red[0][2][w] + red[0][3][w];
    out[F + w] = red[1][0][w] + red[1][1][w] + red[1][2][w] + red[1][3][w];
  }
}

// Column sums of a [P][width] fp32 partial matrix in two fixed-order stages (deterministic) inside ONE launch:
// stage 1: grid (width/64, R) blocks, block r sums rows p = r, r+R, ... in fp64 -> dpart[r][width];
// stage 2: the block that arrives LAST at its column block's ticket sums the R group sums in group order and finishes
// (plain sums, or the BatchNorm statistics -> scale / shift / running buffers).  Round 4: the second stage used to be a
// launch of its own - 26 colsum_stage2 + 9 bn_finalize launches per training step, each a kernel boundary and 6-10 us of
// one-wave latency on the critical path between two convolutions.
// Tickets: CS_TICKETS words at the head of the caller's `dpart` scratch, zero between launches (the last arriver resets its
// word); a composite driver zeroes them ONCE per call (colsum_tickets_reset) - calls that may run side by side (two streams)
// have a dpart each.
constexpr int CS_R = 64;          // most row groups of stage 1 (scratch: CS_R x width doubles)
constexpr int CS_TICKETS = 128;   // ticket words (64 doubles) in front of the group sums: column blocks of widths up to 8192
// Row groups for P partial rows: 16 up to 1024 rows, 32 up to 4096, 64 beyond - with 16 groups the 8000-row partials of the
// 64-channel layers were read by 32 workgroups (42 us for 4 MB); the second stage reads the groups' sums.
inline int cs_groups(int P) { return P < 1 ? 1 : P < 16 ? P : P <= 1024 ? 16 : P <= 4096 ? 32 : CS_R; }

struct BnFinalizeArgs {
  double count;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  int64_t* nbt;
  float *scale, *shift, *mean_out, *invstd_out;
};

// Group sums are handed over WITHOUT fences (a release fence writes back the XCD's whole L2 and an acquire invalidates the
// CU's L1: the fenced form of this kernel took 20 us where the two launches it replaced took 14): every group sum is stored
// write-through (agent-scope atomic store, `sc1`) and read back by the last arriver with agent-scope loads (`sc1`: served by
// L2 / memory, never by the reader's L1); the storing wave drains its stores (s_waitcnt vmcnt(0)) in front of the barrier
// behind which ONE lane adds to the ticket (MI355X_MICROARCH.md, hand-offs measured with `sc1` loads in place of the acquire,
// first row: one lane signals for the workgroup, the workgroup whose add returned last reads).
__device__ __forceinline__ void cs_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double cs_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
// true in every thread of the block that arrived last at `ticket` (which expects `expect` arrivals); the ticket is zero again
__device__ __forceinline__ bool cs_last_arriver(unsigned* ticket, unsigned expect, int* s_last) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == expect - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = last;
  }
  __syncthreads();
  return *s_last != 0;
}

// FINALIZE = false: out[i] = sum_r dsum[r][i]; entries >= split go to out2 (bn: dbeta | dgamma).
// FINALIZE = true (width = 2C = sums | sums of squares): batch mean / biased var, scale / shift for the fused activation,
// running stats (momentum 0.1, unbiased var: torch BatchNorm2d); one ticket per 64 channels, 2R arrivals.
template <bool FINALIZE>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ partials, int P, int width, double* __restrict__ dpart,
                                                     float* __restrict__ out, float* __restrict__ out2, int split,
                                                     BnFinalizeArgs bn) {
  __shared__ double red[4][64];
  __shared__ int s_last;
  unsigned* tickets = reinterpret_cast<unsigned*>(dpart);
  double* dsum = dpart + CS_TICKETS / 2;
  const int il = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  const int R = gridDim.y;
  double a = 0.0;
  if (i < width)
    for (int p = blockIdx.y + g * R; p < P; p += 4 * R) a += (double)partials[(long)p * width + i];
  red[g][il] = a;
  __syncthreads();
  if (g == 0 && i < width) cs_store(dsum + (long)blockIdx.y * width + i, red[0][il] + red[1][il] + red[2][il] + red[3][il]);
  if (!FINALIZE) {
    if (!cs_last_arriver(tickets + blockIdx.x, (unsigned)R, &s_last)) return;
    if (g != 0 || i >= width) return;
    double t = 0.0;
    for (int r0 = 0; r0 < R; r0 += 8) {          // eight loads in flight; the additions stay in group order
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r0 + u < R ? cs_load(dsum + (long)(r0 + u) * width + i) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < R) t += v[u];
    }
    if (split > 0 && i >= split) out2[i - split] = (float)t;
    else out[i] = (float)t;
  } else {
    const int C = width >> 1, cb = C >> 6;                 // C is a multiple of 64 (checked by the launcher)
    const int xc = blockIdx.x % cb;
    if (!cs_last_arriver(tickets + xc, (unsigned)(2 * R), &s_last)) return;
    if (g != 0) return;
    const int c = xc * 64 + il;
    double sm = 0.0, q = 0.0;
    for (int r0 = 0; r0 < R; r0 += 8) {          // sixteen loads in flight; the additions stay in group order
      double v[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = r0 + u < R ? cs_load(dsum + (long)(r0 + u) * width + c) : 0.0;
        w[u] = r0 + u < R ? cs_load(dsum + (long)(r0 + u) * width + C + c) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < R) { sm += v[u]; q += w[u]; }
    }
    const double mean = sm / bn.count;
    double var = q / bn.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float sc = bn.gamma[c] * invstd;
    bn.scale[c] = sc; bn.shift[c] = bn.beta[c] - (float)mean * sc;
    bn.mean_out[c] = (float)mean; bn.invstd_out[c] = invstd;
    const double unbiased = bn.count > 1.0 ? var * bn.count / (bn.count - 1.0) : var;
    bn.running_mean[c] = 0.9f * bn.running_mean[c] + 0.1f * (float)mean;
    bn.running_var[c] = 0.9f * bn.running_var[c] + 0.1f * (float)unbiased;
    if (c == 0 && bn.nbt) bn.nbt[0] += 1;
  }
}

// colsum_kernel<false> for several matrices at once (conv.h: ColsumBatch): block x -> (job, column block), block y -> row group
// of the job (blocks past the job's groups leave at once); tickets are indexed by the launch's block x, group sums lie one job
// behind the other.
__global__ __launch_bounds__(256) void colsum_batch_kernel(acvae::ColsumBatch b, double* __restrict__ dpart) {
  __shared__ double red[4][64];
  __shared__ int s_last;
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;
  const int R = b.R[j];
  if ((int)blockIdx.y >= R) return;
  const float* __restrict__ partials = b.x[j];
  const int P = b.P[j], width = b.width[j];
  unsigned* tickets = reinterpret_cast<unsigned*>(dpart);
  double* dsum = dpart + CS_TICKETS / 2 + b.d0[j];
  const int il = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = ((int)blockIdx.x - b.blk0[j]) * 64 + il;
  double a = 0.0;
  if (i < width)
    for (int p = blockIdx.y + g * R; p < P; p += 4 * R) a += (double)partials[(long)p * width + i];
  red[g][il] = a;
  __syncthreads();
  if (g == 0 && i < width) cs_store(dsum + (long)blockIdx.y * width + i, red[0][il] + red[1][il] + red[2][il] + red[3][il]);
  if (!cs_last_arriver(tickets + blockIdx.x, (unsigned)R, &s_last)) return;
  if (g != 0 || i >= width) return;
  double t = 0.0;
  for (int r0 = 0; r0 < R; r0 += 8) {          // eight loads in flight; the additions stay in group order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r0 + u < R ? cs_load(dsum + (long)(r0 + u) * width + i) : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r0 + u < R) t += v[u];
  }
  b.out[j][i] = (float)t;
  if (b.out_b[j]) b.out_b[j][i] = (float)t;
}

// evaluation mode: scale / shift from the running statistics (no batch statistics, nothing to reduce)
__global__ __launch_bounds__(64) void bn_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                     float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + 1e-5f);
  const float sc = gamma[c] * invstd;
  scale[c] = sc; shift[c] = beta[c] - running_mean[c] * sc;
  mean_out[c] = running_mean[c]; invstd_out[c] = invstd;
}

// ------------------------------------------------------------------ first conv (Cin = 1), direct
// Y[n,t,w,co] = sum_tap W1[co][tap] * xin(t+dy, w+dx),  xin = x*scale0[w] + shift0[w] inside the image.
// Block = RB time rows of one clip; 256 threads = 16 pixels x 16 cout-quads per pass.
constexpr int C1_RB = 8;
template <class TY>
__global__ __launch_bounds__(256) void conv1_first_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ scale0,
                                                              const float* __restrict__ shift0,
                                                              const float* __restrict__ W1, TY* __restrict__ Y,
                                                              float* __restrict__ partials, int T, int F) {
  __shared__ float wl[64 * 9];
  __shared__ float patch[(C1_RB + 2) * 66];
  __shared__ float red[2][16][64];
  const int nblk_t = (T + C1_RB - 1) / C1_RB;
  const int n = blockIdx.x / nblk_t, t0 = (blockIdx.x % nblk_t) * C1_RB;
  for (int i = threadIdx.x; i < 64 * 9; i += 256) wl[i] = W1[i];
  for (int i = threadIdx.x; i < (C1_RB + 2) * 66; i += 256) {
    const int r = i / 66, cidx = i % 66;
    const int t = t0 + r - 1, w = cidx - 1;
    float v = 0.f;
    if (t >= 0 && t < T && w >= 0 && w < F) v = x[((long)n * T + t) * F + w] * scale0[w] + shift0[w];
    patch[i] = v;
  }
  __syncthreads();
  const int cq = threadIdx.x & 15, pp = threadIdx.x >> 4;
  float wreg[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wreg[c][k] = wl[(cq * 4 + c) * 9 + k];
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (int pix = pp; pix < C1_RB * 64; pix += 16) {
    const int r = pix >> 6, w = pix & 63;
    const int t = t0 + r;
    if (t >= T || w >= F) continue;
    float in[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) in[k] = patch[(r + k / 3) * 66 + w + k % 3];
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) a += wreg[c][k] * in[k];
      // bf16 storage: the statistics describe the tensor as stored (rounded), so that BatchNorm is exact on it
      if (!std::is_same<TY, float>::value) a = round_bf16(a);
      o[c] = a; s[c] += a; q[c] += a * a;
    }
    store4(Y + (((long)n * T + t) * F + w) * 64 + cq * 4, make_float4(o[0], o[1], o[2], o[3]));
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) { red[0][pp][cq * 4 + c] = s[c]; red[1][pp][cq * 4 + c] = q[c]; }
  __syncthreads();
  if (partials && threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, co = threadIdx.x & 63;
    float a = 0.f;
    for (int i = 0; i < 16; ++i) a += red[which][i][co];
    partials[(long)blockIdx.x * 128 + which * 64 + co] = a;
  }
}

// Backward of the first conv: dW1 partials [blocks][64*9] and bn0 grad partials [blocks][2][64]
// (sum over pixels of dxin and dxin*xhat per mel bin), dxin[q] = sum_tap D[q - tap][tap],
// D[p][tap] = sum_co dY[p][co] * W1[co][tap].
template <class TY>
__global__ __launch_bounds__(256) void conv1_first_bwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ scale0,
                                                              const float* __restrict__ shift0,
                                                              const float* __restrict__ mean0,
                                                              const float* __restrict__ invstd0,
                                                              const float* __restrict__ W1,
                                                              const TY* __restrict__ dY,
                                                              float* __restrict__ dw_part, float* __restrict__ bn_part,
                                                              int T, int F) {
  __shared__ float wl[64 * 9];
  __shared__ float patch[(C1_RB + 2) * 66];
  // D (13.5 KB) is dead once the input gradient has been gathered from it; the weight-gradient partials then reuse the
  // space in two halves of 8 pixel groups (18 KB instead of 14 + 37 KB: 6 instead of 2 workgroups per CU)
  constexpr int D_FLOATS = (C1_RB + 2) * 64 * 9, WACC_FLOATS = 8 * 576;
  __shared__ float scratch[D_FLOATS > WACC_FLOATS ? D_FLOATS : WACC_FLOATS];
  float* D = scratch;
  float (*wacc)[576] = reinterpret_cast<float (*)[576]>(scratch);
  const int nblk_t = (T + C1_RB - 1) / C1_RB;
  const int n = blockIdx.x / nblk_t, t0 = (blockIdx.x % nblk_t) * C1_RB;
  for (int i = threadIdx.x; i < 64 * 9; i += 256) wl[i] = W1[i];
  for (int i = threadIdx.x; i < (C1_RB + 2) * 66; i += 256) {
    const int r = i / 66, cidx = i % 66;
    const int t = t0 + r - 1, w = cidx - 1;
    float v = 0.f;
    if (t >= 0 && t < T && w >= 0 && w < F) v = x[((long)n * T + t) * F + w] * scale0[w] + shift0[w];
    patch[i] = v;
  }
  __syncthreads();
  const int cq = threadIdx.x & 15, pp = threadIdx.x >> 4;
  float wreg[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wreg[c][k] = wl[(cq * 4 + c) * 9 + k];
  float dwacc[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[c][k] = 0.f;
  // rows t0-1 .. t0+RB (halo rows contribute to D only).  The loop is latency-bound on the dY stream (one 16-B load per
  // lane and pixel): the loads of the next four pixels are in flight while these four are reduced.
  constexpr int C1_IT = (C1_RB + 2) * 64 / 16, C1_UN = 4;
  static_assert(C1_IT % C1_UN == 0, "pixel loop is unrolled by four");
  auto fetch = [&](int it, float4& g) {
    const int pix = pp + it * 16;
    const int r = pix >> 6, w = pix & 63;
    const int t = t0 + r - 1;
    g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < T && w < F) g = load4(dY + (((long)n * T + t) * F + w) * 64 + cq * 4);
  };
  float4 gq[C1_UN], gn[C1_UN];
#pragma unroll
  for (int u = 0; u < C1_UN; ++u) fetch(u, gq[u]);
  for (int it0 = 0; it0 < C1_IT; it0 += C1_UN) {
    if (it0 + C1_UN < C1_IT) {
#pragma unroll
      for (int u = 0; u < C1_UN; ++u) fetch(it0 + C1_UN + u, gn[u]);
    }
#pragma unroll
    for (int u = 0; u < C1_UN; ++u) {
      const int pix = pp + (it0 + u) * 16;
      const int r = pix >> 6, w = pix & 63;
      const int t = t0 + r - 1;
      const bool inimg = (t >= 0 && t < T && w < F);
      const float gv[4] = {gq[u].x, gq[u].y, gq[u].z, gq[u].w};
      float d[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) d[k] = gv[0] * wreg[0][k] + gv[1] * wreg[1][k] + gv[2] * wreg[2][k] + gv[3] * wreg[3][k];
      // reduce over the 16 cout-quads (16 consecutive lanes)
#pragma unroll
      for (int k = 0; k < 9; ++k) d[k] = row16_sum(d[k]);
      if (cq == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) D[(r * 64 + w) * 9 + k] = d[k];
      }
      if (inimg && r >= 1 && r <= C1_RB) {  // own rows: weight gradient
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float in = patch[(r - 1 + k / 3) * 66 + w + k % 3];
#pragma unroll
          for (int c = 0; c < 4; ++c) dwacc[c][k] += gv[c] * in;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < C1_UN; ++u) gq[u] = gn[u];
  }
  __syncthreads();   // D complete
  // dxin for own pixels and the per-mel reductions
  const int w = threadIdx.x & 63, rr = threadIdx.x >> 6;
  float sg = 0.f, sgx = 0.f;
  if (w < F)
    for (int r = rr; r < C1_RB; r += 4) {
      const int t = t0 + r;
      if (t >= T) break;
      float dx = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        // Y[p] uses xin[p + (dy,dx)], so dxin[q] += D[q - (dy,dx)][tap]
        const int dy = k / 3 - 1, dxx = k % 3 - 1;
        const int pr = r + 1 - dy, pw2 = w - dxx;
        if (pw2 >= 0 && pw2 < 64) dx += D[(pr * 64 + pw2) * 9 + k];
      }
      const float xhat = (x[((long)n * T + t) * F + w] - mean0[w]) * invstd0[w];
      sg += dx; sgx += dx * xhat;
    }
  // weight-gradient partials of the 16 pixel groups, summed in group order (two halves through the shared scratch)
  float asum[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();   // D (first pass) / the previous half's partials are no longer read
    if ((pp >> 3) == half) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) wacc[pp & 7][cq * 36 + c * 9 + k] = dwacc[c][k];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = threadIdx.x + q * 256;
      if (i < 576)
        for (int j = 0; j < 8; ++j) asum[q] += wacc[j][i];
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int i = threadIdx.x + q * 256;
    if (i < 576) dw_part[(long)blockIdx.x * 576 + i] = asum[q];   // index = co*9 + tap (cq*36 + c*9 + k)
  }
  __syncthreads();
  float* rb = scratch;
  rb[rr * 128 + w] = sg; rb[rr * 128 + 64 + w] = sgx;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, ww = threadIdx.x & 63;
    bn_part[(long)blockIdx.x * 128 + which * 64 + ww] =
        rb[0 * 128 + which * 64 + ww] + rb[1 * 128 + which * 64 + ww] + rb[2 * 128 + which * 64 + ww] +
        rb[3 * 128 + which * 64 + ww];
  }
}

// ------------------------------------------------------------------ BN + ReLU + 2x2 avg-pool + dropout
__device__ __forceinline__ float4 bnrelu4(float4 v, float4 sc, float4 sh) {
  return make_float4(fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f), fmaxf(v.z * sc.z + sh.z, 0.f),
                     fmaxf(v.w * sc.w + sh.w, 0.f));
}
// keep-multiplier (0 or 1/(1-p)) for the 4 channels of pooled element (n,ho,wo,c4*4..+3)
__device__ __forceinline__ float4 drop4(const DropoutSpec& d, long idx4, int n, int ho, int wo, int c, int Ho, int Wo,
                                        int C) {
  if (d.p <= 0.f) return make_float4(1.f, 1.f, 1.f, 1.f);
  const float k = 1.0f / (1.0f - d.p);
  if (d.mask) {  // explicit mask in the reference's NCHW order (parity tests)
    const long b = (((long)n * C + c) * Ho + ho) * Wo + wo;
    const long cs = (long)Ho * Wo;
    return make_float4(d.mask[b] ? k : 0.f, d.mask[b + cs] ? k : 0.f, d.mask[b + 2 * cs] ? k : 0.f,
                       d.mask[b + 3 * cs] ? k : 0.f);
  }
  const uint4 r = philox4x32(d.seed, (uint64_t)idx4, d.site);
  const float u = 1.0f / 16777216.0f;
  return make_float4((float)(r.x >> 8) * u >= d.p ? k : 0.f, (float)(r.y >> 8) * u >= d.p ? k : 0.f,
                     (float)(r.z >> 8) * u >= d.p ? k : 0.f, (float)(r.w >> 8) * u >= d.p ? k : 0.f);
}

template <class T>
__global__ void bn_relu_pool_kernel(const T* __restrict__ Y, const float* __restrict__ scale,
                                    const float* __restrict__ shift, T* __restrict__ P, int N, int H, int W, int C,
                                    DropoutSpec drop) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
    const T* y = Y + ((((long)n * H + 2 * ho) * W + 2 * wo) * C + c4 * 4);
    const float4 a = bnrelu4(load4(y), sc, sh);
    const float4 b = bnrelu4(load4(y + C), sc, sh);
    const float4 c = bnrelu4(load4(y + (long)W * C), sc, sh);
    const float4 d = bnrelu4(load4(y + (long)W * C + C), sc, sh);
    const float4 m = drop4(drop, i, n, ho, wo, c4 * 4, Ho, Wo, C);
    // F.avg_pool2d then F.dropout: (sum * 0.25) * mask * 1/(1-p)
    float4 o;
    o.x = ((a.x + b.x + c.x + d.x) * 0.25f) * m.x; o.y = ((a.y + b.y + c.y + d.y) * 0.25f) * m.y;
    o.z = ((a.z + b.z + c.z + d.z) * 0.25f) * m.z; o.w = ((a.w + b.w + c.w + d.w) * 0.25f) * m.w;
    store4(P + 4 * i, o);
  }
}

// pool_size (1,1): BN + ReLU + dropout at full resolution
template <class T>
__global__ void bn_relu_drop_kernel(const T* __restrict__ Y, const float* __restrict__ scale,
                                    const float* __restrict__ shift, T* __restrict__ P, int N, int H, int W, int C,
                                    DropoutSpec drop) {
  const int C4 = C / 4;
  const long total = (long)N * H * W * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
    const float4 a = bnrelu4(load4(Y + 4 * i), sc, sh);
    const float4 m = drop4(drop, i, n, h, w, c4 * 4, H, W, C);
    store4(P + 4 * i, make_float4(a.x * m.x, a.y * m.y, a.z * m.z, a.w * m.w));
  }
}

// ReLU decisions of a BN+ReLU site exactly as the backward kernels take them (y * scale + shift > 0), written in the
// reference's NCHW order: test / debugging aid (acvae_encoder_relu_mask)
template <class T>
__global__ void relu_mask_kernel(const T* __restrict__ Y, const float* __restrict__ scale,
                                 const float* __restrict__ shift, uint8_t* __restrict__ out, int N, int H, int W, int C) {
  const long total = (long)N * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    const float y = load1(Y + i);
    out[(((long)n * C + c) * H + h) * W + w] = (y * scale[c] + shift[c] <= 0.f) ? 0 : 1;
  }
}

// ------------------------------------------------------------------ BN backward
// g = relu'(bn(Y)) * upstream;  UP_POOL: upstream = dP[n,h/2,w/2,c] * dropmask * 0.25 (0 for a trailing odd row/col)
template <int UP, class T>
__device__ __forceinline__ float4 upstream4(const T* __restrict__ dO, const DropoutSpec& drop, int n, int h, int w,
                                            int c, int H, int W, int C) {
  if (UP == UP_PLAIN) return load4(dO + ((((long)n * H + h) * W + w) * C + c));
  if (UP == UP_DROP) {
    const long i4 = ((((long)n * H + h) * W + w) * C + c) >> 2;
    float4 v = load4(dO + 4 * i4);
    const float4 m = drop4(drop, i4, n, h, w, c, H, W, C);
    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    return v;
  }
  const int Ho = H / 2, Wo = W / 2;
  const int ho = h >> 1, wo = w >> 1;
  if (ho >= Ho || wo >= Wo) return make_float4(0.f, 0.f, 0.f, 0.f);
  const long i4 = ((((long)n * Ho + ho) * Wo + wo) * C + c) >> 2;
  float4 v = load4(dO + 4 * i4);
  const float4 m = drop4(drop, i4, n, ho, wo, c, Ho, Wo, C);
  v.x *= m.x * 0.25f; v.y *= m.y * 0.25f; v.z *= m.z * 0.25f; v.w *= m.w * 0.25f;
  return v;
}

// partials [blocks][2][C]: sum g | sum g*yhat.  Block = 256 threads = (Cc/4) channel-quads x (1024/Cc) pixels of the
// channel chunk blockIdx.y (Cc = min(C, 1024) channels per chunk).
template <int UP, class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ Y, const T* __restrict__ dO,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            float* __restrict__ partials, int N, int H, int W, int C,
                                                            int pix_per_block, DropoutSpec drop) {
  extern __shared__ float red[];  // [256][8]
  const int Cc = C < 1024 ? C : 1024;
  const int C4 = Cc / 4;
  const int cq = threadIdx.x % C4, pl = threadIdx.x / C4, npl = 256 / C4;
  const long M = (long)N * H * W;
  const long p0 = (long)blockIdx.x * pix_per_block;
  const int c = blockIdx.y * Cc + cq * 4;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (pl < npl) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    for (long p = p0 + pl; p < p0 + pix_per_block && p < M; p += npl) {
      int w = 0, h = 0, n = 0;
      if (UP != UP_PLAIN) {                       // (the plain upstream is the same element of dO: no coordinates)
        w = (int)(p % W);
        const long t = p / W;
        h = (int)(t % H); n = (int)(t / H);
      }
      const float4 y = load4(Y + p * C + c);
      float4 g = UP == UP_PLAIN ? load4(dO + p * C + c) : upstream4<UP>(dO, drop, n, h, w, c, H, W, C);
      if (y.x * sc.x + sh.x <= 0.f) g.x = 0.f;
      if (y.y * sc.y + sh.y <= 0.f) g.y = 0.f;
      if (y.z * sc.z + sh.z <= 0.f) g.z = 0.f;
      if (y.w * sc.w + sh.w <= 0.f) g.w = 0.f;
      s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
      q[0] += g.x * ((y.x - mu.x) * is.x); q[1] += g.y * ((y.y - mu.y) * is.y);
      q[2] += g.z * ((y.z - mu.z) * is.z); q[3] += g.w * ((y.w - mu.w) * is.w);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[k * 256 + threadIdx.x] = s[k]; red[(4 + k) * 256 + threadIdx.x] = q[k]; }
  __syncthreads();
  if (threadIdx.x < C4) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < npl; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += red[k * 256 + j * C4 + threadIdx.x];
    float* out = partials + (long)blockIdx.x * 2 * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) { out[c + k] = a[k]; out[C + c + k] = a[4 + k]; }
  }
}

// dY = scale * (g - sum_g/n - yhat * sum_gy/n)      (scale = gamma * invstd)
template <int UP, class T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ Y, const T* __restrict__ dO,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ sum_g, const float* __restrict__ sum_gy,
                                    T* __restrict__ dYout, int N, int H, int W, int C, DropoutSpec drop,
                                    int batch_stats) {
  const int C4 = C / 4;
  const long total = (long)N * H * W * C4;
  // evaluation-mode BatchNorm normalises with the running statistics, which do not depend on the batch: dY = scale * g
  const float invn = batch_stats ? 1.0f / (float)((long)N * H * W) : 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long p = i / C4;
    const int w = (int)(p % W);
    const long t = p / W;
    const int h = (int)(t % H), n = (int)(t / H);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 sg = *reinterpret_cast<const float4*>(sum_g + c), sgy = *reinterpret_cast<const float4*>(sum_gy + c);
    const float4 y = load4(Y + 4 * i);
    float4 g = upstream4<UP>(dO, drop, n, h, w, c, H, W, C);
    if (y.x * sc.x + sh.x <= 0.f) g.x = 0.f;
    if (y.y * sc.y + sh.y <= 0.f) g.y = 0.f;
    if (y.z * sc.z + sh.z <= 0.f) g.z = 0.f;
    if (y.w * sc.w + sh.w <= 0.f) g.w = 0.f;
    float4 o;
    o.x = sc.x * (g.x - sg.x * invn - ((y.x - mu.x) * is.x) * (sgy.x * invn));
    o.y = sc.y * (g.y - sg.y * invn - ((y.y - mu.y) * is.y) * (sgy.y * invn));
    o.z = sc.z * (g.z - sg.z * invn - ((y.z - mu.z) * is.z) * (sgy.z * invn));
    o.w = sc.w * (g.w - sg.w * invn - ((y.w - mu.w) * is.w) * (sgy.w * invn));
    store4(dYout + 4 * i, o);
  }
}

// ---- round 3: the same two passes with the index arithmetic and the dropout hash off the per-element path.
// The generic kernels above spend ~150 VALU instructions per float4 on 64-bit divisions / remainders (and, through a pool, a
// Philox hash per PIXEL: four per pooled element, in both passes); they ran at 4.5-5.2 TB/s where bn_relu_pool_kernel, which
// hashes once per pooled element, streams at 7.4.
//   UP_PLAIN: a thread keeps ONE channel quad (the grid stride is a multiple of C/4), its six per-channel constants live in
//             registers, the element index is the only loop variable; two elements in flight.
//   UP_POOL : a thread owns one POOLED element (n, ho, wo, c4): one hash, one upstream load, four pixels of Y.  Rows / columns
//             behind the last full 2x2 window (odd H or W) have no upstream gradient but still get their dY.
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_plain_kernel(const T* __restrict__ Y, const T* __restrict__ dO,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ sum_g, const float* __restrict__ sum_gy,
                                                                 T* __restrict__ dYout, long total, int C4, float invn) {
  const long tid = blockIdx.x * 256L + threadIdx.x, stride = gridDim.x * 256L;       // stride % C4 == 0 (launcher)
  const int c = (int)(tid % C4) * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
  const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
  float4 a = *reinterpret_cast<const float4*>(sum_g + c), b = *reinterpret_cast<const float4*>(sum_gy + c);
  a.x *= invn; a.y *= invn; a.z *= invn; a.w *= invn;
  b.x *= invn; b.y *= invn; b.z *= invn; b.w *= invn;
  auto one = [&](float4 y, float4 g) {
    if (y.x * sc.x + sh.x <= 0.f) g.x = 0.f;
    if (y.y * sc.y + sh.y <= 0.f) g.y = 0.f;
    if (y.z * sc.z + sh.z <= 0.f) g.z = 0.f;
    if (y.w * sc.w + sh.w <= 0.f) g.w = 0.f;
    float4 o;
    o.x = sc.x * (g.x - a.x - ((y.x - mu.x) * is.x) * b.x);
    o.y = sc.y * (g.y - a.y - ((y.y - mu.y) * is.y) * b.y);
    o.z = sc.z * (g.z - a.z - ((y.z - mu.z) * is.z) * b.z);
    o.w = sc.w * (g.w - a.w - ((y.w - mu.w) * is.w) * b.w);
    return o;
  };
  long i = tid;
  for (; i + stride < total; i += 2 * stride) {
    const float4 y0 = load4(Y + 4 * i), g0 = load4(dO + 4 * i);
    const float4 y1 = load4(Y + 4 * (i + stride)), g1 = load4(dO + 4 * (i + stride));
    store4(dYout + 4 * i, one(y0, g0));
    store4(dYout + 4 * (i + stride), one(y1, g1));
  }
  if (i < total) store4(dYout + 4 * i, one(load4(Y + 4 * i), load4(dO + 4 * i)));
}

// pooled element i -> (n, ho, wo, c4) over the grid of 2x2 windows INCLUDING the partial ones (Hc = ceil(H/2), Wc = ceil(W/2))
struct PoolIdx { int n, ho, wo, c4; };
__device__ __forceinline__ PoolIdx pool_idx(unsigned i, unsigned C4, unsigned Wc, unsigned Hc) {
  PoolIdx r;
  r.c4 = (int)(i % C4);
  unsigned t = i / C4;
  r.wo = (int)(t % Wc); t /= Wc;
  r.ho = (int)(t % Hc);
  r.n = (int)(t / Hc);
  return r;
}
// upstream gradient of the four pixels of window (n, ho, wo): dP * dropout mask * 1/4 (0 for a partial window)
template <class T>
__device__ __forceinline__ float4 pool_upstream(const T* __restrict__ dO, const DropoutSpec& drop, const PoolIdx& q, int Ho, int Wo,
                                                int C) {
  if (q.ho >= Ho || q.wo >= Wo) return make_float4(0.f, 0.f, 0.f, 0.f);
  const long i4 = ((((long)q.n * Ho + q.ho) * Wo + q.wo) * C + q.c4 * 4) >> 2;
  float4 v = load4(dO + 4 * i4);
  const float4 m = drop4(drop, i4, q.n, q.ho, q.wo, q.c4 * 4, Ho, Wo, C);
  v.x *= m.x * 0.25f; v.y *= m.y * 0.25f; v.z *= m.z * 0.25f; v.w *= m.w * 0.25f;
  return v;
}
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const T* __restrict__ Y, const T* __restrict__ dO,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ sum_g, const float* __restrict__ sum_gy,
                                                                T* __restrict__ dYout, int N, int H, int W, int C, DropoutSpec drop,
                                                                float invn) {
  const int C4 = C / 4, Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const unsigned total = (unsigned)N * Hc * Wc * C4;          // < 2^31 (launcher)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const PoolIdx q = pool_idx(i, C4, Wc, Hc);
    const int c = q.c4 * 4;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    float4 a = *reinterpret_cast<const float4*>(sum_g + c), b = *reinterpret_cast<const float4*>(sum_gy + c);
    a.x *= invn; a.y *= invn; a.z *= invn; a.w *= invn;
    b.x *= invn; b.y *= invn; b.z *= invn; b.w *= invn;
    const long base = (((long)q.n * H + 2 * q.ho) * W + 2 * q.wo) * C + c;
    const bool hx = 2 * q.ho + 1 < H, wx = 2 * q.wo + 1 < W;
    const long off[4] = {0, (long)C, (long)W * C, (long)W * C + C};
    const bool ok[4] = {true, wx, hx, hx && wx};
    float4 y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = ok[k] ? load4(Y + base + off[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 gp = pool_upstream(dO, drop, q, Ho, Wo, C);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!ok[k]) continue;
      float4 g = gp;
      if (y[k].x * sc.x + sh.x <= 0.f) g.x = 0.f;
      if (y[k].y * sc.y + sh.y <= 0.f) g.y = 0.f;
      if (y[k].z * sc.z + sh.z <= 0.f) g.z = 0.f;
      if (y[k].w * sc.w + sh.w <= 0.f) g.w = 0.f;
      float4 o;
      o.x = sc.x * (g.x - a.x - ((y[k].x - mu.x) * is.x) * b.x);
      o.y = sc.y * (g.y - a.y - ((y[k].y - mu.y) * is.y) * b.y);
      o.z = sc.z * (g.z - a.z - ((y[k].z - mu.z) * is.z) * b.z);
      o.w = sc.w * (g.w - a.w - ((y[k].w - mu.w) * is.w) * b.w);
      store4(dYout + base + off[k], o);
    }
  }
}
// partials [blocks][2][C] as bn_bwd_reduce_kernel; a block owns `win_per_block` FULL windows (partial ones carry no gradient)
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const T* __restrict__ Y, const T* __restrict__ dO,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ partials, int N, int H, int W, int C,
                                                                 int win_per_block, DropoutSpec drop) {
  extern __shared__ float red[];  // [256][8]
  const int Cc = C < 1024 ? C : 1024;
  const int C4 = Cc / 4, Ho = H / 2, Wo = W / 2;
  const int cq = threadIdx.x % C4, pl = threadIdx.x / C4, npl = 256 / C4;
  const unsigned M = (unsigned)N * Ho * Wo;
  const unsigned p0 = blockIdx.x * (unsigned)win_per_block;
  const int c = blockIdx.y * Cc + cq * 4;
  float s[4] = {0, 0, 0, 0}, qv[4] = {0, 0, 0, 0};
  if (pl < npl) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    for (unsigned p = p0 + pl; p < p0 + win_per_block && p < M; p += npl) {
      PoolIdx q;
      q.c4 = c >> 2;
      q.wo = (int)(p % (unsigned)Wo);
      const unsigned t = p / (unsigned)Wo;
      q.ho = (int)(t % (unsigned)Ho); q.n = (int)(t / (unsigned)Ho);
      const long base = (((long)q.n * H + 2 * q.ho) * W + 2 * q.wo) * C + c;
      const float4 y0 = load4(Y + base), y1 = load4(Y + base + C), y2 = load4(Y + base + (long)W * C),
                   y3 = load4(Y + base + (long)W * C + C);
      const float4 gp = pool_upstream(dO, drop, q, Ho, Wo, C);
      const float4 ys[4] = {y0, y1, y2, y3};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 g = gp;
        const float4 y = ys[k];
        if (y.x * sc.x + sh.x <= 0.f) g.x = 0.f;
        if (y.y * sc.y + sh.y <= 0.f) g.y = 0.f;
        if (y.z * sc.z + sh.z <= 0.f) g.z = 0.f;
        if (y.w * sc.w + sh.w <= 0.f) g.w = 0.f;
        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
        qv[0] += g.x * ((y.x - mu.x) * is.x); qv[1] += g.y * ((y.y - mu.y) * is.y);
        qv[2] += g.z * ((y.z - mu.z) * is.z); qv[3] += g.w * ((y.w - mu.w) * is.w);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[k * 256 + threadIdx.x] = s[k]; red[(4 + k) * 256 + threadIdx.x] = qv[k]; }
  __syncthreads();
  if (threadIdx.x < C4) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < npl; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += red[k * 256 + j * C4 + threadIdx.x];
    float* out = partials + (long)blockIdx.x * 2 * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) { out[c + k] = a[k]; out[C + c + k] = a[4 + k]; }
  }
}

// ------------------------------------------------------------------ encoder tail
// audio_embeds[n,s,c] = mean_f P4[n,s,f,c]  (torch.mean(x, dim=3), encoder.py:691)
template <class T>
__global__ void freq_mean_kernel(const T* __restrict__ P, float* __restrict__ out, long rows, int Fp, int C) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i % C);
    float a = 0.f;
    for (int f = 0; f < Fp; ++f) a += load1(P + (r * Fp + f) * C + c);
    out[i] = a / (float)Fp;
  }
}
// dP4[n,s,f,c] = d_ae[n,s,c] / Fp
template <class T>
__global__ void freq_mean_bwd_kernel(const float* __restrict__ dae, T* __restrict__ dP, long rows, int Fp, int C) {
  const long total = rows * Fp * C;
  const float k = 1.0f / (float)Fp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / ((long)Fp * C);
    dP[i] = (T)(dae[r * C + c] * k);
  }
}
// pooled_in[n,c] = dropout(max_s ae + mean_s ae)   (encoder.py:693-696, unmasked over s)
__global__ void time_pool_kernel(const float* __restrict__ ae, float* __restrict__ out, int N, int S, int C,
                                 DropoutSpec drop) {
  const int total = N * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / C, c = i % C;
    float mx = -INFINITY, sm = 0.f;
    for (int s = 0; s < S; ++s) {
      const float v = ae[((long)n * S + s) * C + c];
      mx = fmaxf(mx, v); sm += v;
    }
    float o = mx + sm / (float)S;
    if (drop.p > 0.f) {
      const float k = 1.0f / (1.0f - drop.p);
      const bool keep = drop.mask ? drop.mask[i] != 0 : dropout_keep(drop.seed, drop.site, (uint64_t)i, drop.p);
      o = keep ? o * k : 0.f;
    }
    out[i] = o;
  }
}
__global__ void relu_dropout_kernel(float* __restrict__ x, int total, DropoutSpec drop) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float o = fmaxf(x[i], 0.f);
    if (drop.p > 0.f) {
      const float k = 1.0f / (1.0f - drop.p);
      const bool keep = drop.mask ? drop.mask[i] != 0 : dropout_keep(drop.seed, drop.site, (uint64_t)i, drop.p);
      o = keep ? o * k : 0.f;
    }
    x[i] = o;
  }
}

inline int ew_grid(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

// =============================================================================================
// host-side launchers (C++ internal API, declared in conv.h)
// =============================================================================================
namespace acvae {

// M-tile height of the conv kernels.  256-row tiles (8 matrix waves, 1 workgroup per CU, -25 % operand traffic per
// flop) were tried and measured SLOWER (encoder fwd 9.2 vs 8.2 ms): a single workgroup per CU has nobody to cover its
// barrier bubbles.  128 rows x 2 workgroups per CU it is.
constexpr int CONV_BMT = 128;

int conv3x3_igemm(const float* X, const float* scale, const float* shift, const float* Wp, float* Y, float* partials,
                  int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  if (!X || !Wp || !Y) return ACVAE_EINVAL;
  if (Cin % 32 != 0 || Cout % 4 != 0) return ACVAE_EUNSUPPORTED;
  if (!aligned16(X) || !aligned16(Wp) || !aligned16(Y)) return ACVAE_EALIGN;
  const int M = N * H * W, K = 9 * Cin;
  ConvStatsEpilogue ep{Y, partials, Cout};
  // weight panels by LDS-DMA whenever the output channels fill whole tiles (an LDS-DMA cannot zero-fill)
  prof_begin(ACVAE_PROF_CONV_IGEMM, st);
  ConvStripLoader al{X, scale, shift, H, W, Cin, M};
  const int bn = Cout <= 64 ? 64 : 128;
  const dim3 grid(cdiv(M, CV_BMT), cdiv(Cout, bn)), block(nt_threads<CV_BMT>());
  const bool dma = Cout % bn == 0;
  if (bn == 64) {
    if (dma) hipLaunchKernelGGL((conv_igemm3_kernel<64, true>), grid, block, 0, st, al, Wp, ep, M, Cout, K);
    else hipLaunchKernelGGL((conv_igemm3_kernel<64, false>), grid, block, 0, st, al, Wp, ep, M, Cout, K);
  } else {
    if (dma) hipLaunchKernelGGL((conv_igemm3_kernel<128, true>), grid, block, 0, st, al, Wp, ep, M, Cout, K);
    else hipLaunchKernelGGL((conv_igemm3_kernel<128, false>), grid, block, 0, st, al, Wp, ep, M, Cout, K);
  }
  prof_end(ACVAE_PROF_CONV_IGEMM, st);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
int conv_partials_rows(int N, int H, int W) { return cdiv((long)N * H * W, CONV_BMT); }

// Strip kernel (conv_wgrad3_kernel) whenever the input channels come in chunks of 64 (every layer but conv1): against the
// one-tap-per-tile kernel it measured 1.62 -> 1.48 ms on the 64-output-channel layer and, with the paired LDS reads of
// its 128-row variant, 1.335 -> 1.285 ms / 0.69 -> 0.655 ms on the wider ones.
static bool wgrad_strip(int NC, int Cout) { (void)Cout; return NC % 576 == 0; }
// slabs per pixel slice: the 64-row strip kernel writes the two K-halves of its wave rows separately
static int wgrad_slabs_per_slice(int NC, int Cout) { return wgrad_strip(NC, Cout) && Cout <= 64 ? 2 : 1; }
static int wgrad_splits(int M, int Cout, int NC) {
  const bool narrow = Cout <= 64, w192 = wgrad_strip(NC, Cout);
  const long tiles = w192 ? (long)cdiv(Cout, narrow ? 64 : 128) * (NC / 192)
                          : (long)cdiv(Cout, narrow ? 64 : 128) * cdiv(NC, narrow ? 256 : 128);
  // Resident workgroups per CU of the kernel variant (256 threads; from the compiler's resource report: 120 / 111 VGPRs
  // and 32 KB LDS -> 4, 165 / 134 VGPRs -> 3): the grid should fill whole rounds of 256 x that many slots, because a few
  // blocks over a round boundary cost a whole extra round (1040 blocks ran 18 % slower than 2030).  The slice count is a
  // multiple of 8 (one group of pixel slices per XCD, conv_wgrad_kernel); every slice adds one slab of Cout x NC floats
  // that is written and then read by the reduce, so: minimise (MFMA time / fill of the last round) + slab traffic.
  const long slots = 256L * (wgrad_strip(NC, Cout) ? 3 : ((w192 ? !narrow : narrow) ? 3 : 4));   // strip kernels: 161 / 168 VGPRs
  const int maxs = cdiv(M, 16 * BKT) & ~7;
  const double t_mfma = 2.0 * (double)M * Cout * NC / 1.1e14;
  // measured: a slab costs ~6x its bytes / HBM rate
  const double t_slab = 2.0 * (double)Cout * NC * 4.0 / 0.67e12 * wgrad_slabs_per_slice(NC, Cout);
  int best = 8;
  double best_t = 1e30;
  for (int k = 8; k <= (maxs < 8 ? 8 : maxs) && tiles * k <= 5 * slots; k += 8) {
    const long b = tiles * k;
    const double fill = (double)b / (double)(slots * ((b + slots - 1) / slots));
    // under ~2 rounds there is nothing to balance slow blocks against: price a single round like a 3/4-full one
    const double t = t_mfma / (2 * b < 3 * slots ? fill * 0.75 : fill) + k * t_slab;
    if (t < best_t) { best_t = t; best = k; }
  }
  return best;
}
long conv3x3_wgrad_slab_floats(int N, int H, int W, int Cin, int Cout) {
  return (long)wgrad_splits(N * H * W, Cout, 9 * Cin) * wgrad_slabs_per_slice(9 * Cin, Cout) * Cout * 9 * Cin;
}
int conv3x3_wgrad(const float* dY, const float* X, const float* scale, const float* shift, float* dW_oihw, float* slab,
                  int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  if (!dY || !X || !dW_oihw || !slab) return ACVAE_EINVAL;
  if (Cin % 4 != 0 || Cout % 4 != 0) return ACVAE_EUNSUPPORTED;
  const int M = N * H * W, NC = 9 * Cin;
  const int s = wgrad_splits(M, Cout, NC);            // slices past the last pixel just write zero slabs
  const int k_per = cdiv(cdiv(M, s), BKT) * BKT;
  if ((long)(M + 4096) * (W > H ? W : H) >= (1L << 40)) return ACVAE_EUNSUPPORTED;   // FastDiv range
  ConvKMajorLoader bl{X, scale, shift, H, W, Cin, M, NC, FastDiv::make(W), FastDiv::make(H)};
  prof_begin(ACVAE_PROF_CONV_WGRAD, st);
  if (wgrad_strip(NC, Cout)) {
    const FastDiv fw = FastDiv::make(W), fh = FastDiv::make(H);
    if (Cout <= 64) {
      dim3 grid(cdiv(Cout, 64), NC / 192, s);
      hipLaunchKernelGGL((conv_wgrad3_kernel<1>), grid, dim3(256), 0, st, dY, X, scale, shift, slab, M, Cout, H, W, Cin,
                         fw, fh, k_per);
    } else {
      dim3 grid(cdiv(Cout, 128), NC / 192, s);
      hipLaunchKernelGGL((conv_wgrad3_kernel<2>), grid, dim3(256), 0, st, dY, X, scale, shift, slab, M, Cout, H, W, Cin,
                         fw, fh, k_per);
    }
  } else if (Cout <= 64) {
    dim3 grid(cdiv(Cout, 64), cdiv(NC, 256), s);
    hipLaunchKernelGGL((conv_wgrad_kernel<1, 4>), grid, dim3(256), 0, st, dY, bl, slab, M, Cout, NC, k_per);
  } else {
    dim3 grid(cdiv(Cout, 128), cdiv(NC, 128), s);
    hipLaunchKernelGGL((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, st, dY, bl, slab, M, Cout, NC, k_per);
  }
  prof_end(ACVAE_PROF_CONV_WGRAD, st);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid((long)Cout * NC)), dim3(256), 0, st, slab,
                     s * wgrad_slabs_per_slice(NC, Cout), dW_oihw, Cout, Cin);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

int wgrad_reduce(const float* slab, int nsplit, float* dW_oihw, int Cout, int Cin, hipStream_t st) {
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid((long)Cout * 9 * Cin)), dim3(256), 0, st, slab, nsplit, dW_oihw, Cout,
                     Cin);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

template <class T>
int repack_weights(const float* W_oihw, T* Wf, T* Wd, int Cout, int Cin, hipStream_t st) {
  const long total = (long)Cout * Cin * 9;
  if (Wf) hipLaunchKernelGGL(repack_fwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st, W_oihw, Wf, Cout, Cin);
  if (Wd) hipLaunchKernelGGL(repack_dgrad_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st, W_oihw, Wd, Cout, Cin);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int repack_weights<float>(const float*, float*, float*, int, int, hipStream_t);
template int repack_weights<bf16_t>(const float*, bf16_t*, bf16_t*, int, int, hipStream_t);
template <class T>
int repack_weights_batch(RepackBatch& b, hipStream_t st) {
  if (b.n <= 0) return ACVAE_OK;
  b.start[0] = 0;
  for (int l = 0; l < b.n; ++l) {
    if (!b.W[l] || !b.dst[l]) return ACVAE_EINVAL;
    b.start[l + 1] = b.start[l] + (long)b.Cout[l] * b.Cin[l] * 9;
  }
  const long total = b.start[b.n];
  hipLaunchKernelGGL(repack_batch_kernel<T>, dim3(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256)), dim3(256), 0, st, b);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int repack_weights_batch<float>(RepackBatch&, hipStream_t);
template int repack_weights_batch<bf16_t>(RepackBatch&, hipStream_t);

int bn0_stats(const float* x, float* partials, long rows, int F, int* nparts, hipStream_t st) {
  if (F > 64) return ACVAE_EUNSUPPORTED;
  const int rpb = 256;
  const int nb = cdiv(rows, rpb);
  hipLaunchKernelGGL(bn0_stats_kernel, dim3(nb), dim3(256), 0, st, x, partials, rows, F, rpb);
  *nparts = nb;
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
int bn0_partials_rows(long rows) { return cdiv(rows, 256); }

int colsum2(const float* partials, int P, int width, double* dpart, float* out, float* out2, int split,
            hipStream_t st) {
  if (cdiv(width, 64) > CS_TICKETS) return ACVAE_EUNSUPPORTED;
  const int R = cs_groups(P);
  hipLaunchKernelGGL(colsum_kernel<false>, dim3(cdiv(width, 64), R), dim3(256), 0, st, partials, P, width, dpart, out, out2, split,
                     BnFinalizeArgs{});
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
int colsum_batch(ColsumBatch& b, double* dpart, long dpart_doubles, hipStream_t st) {
  if (b.n <= 0) return ACVAE_OK;
  if (b.n > ColsumBatch::MAXJ) return ACVAE_EINVAL;
  long d = 0;
  int blocks = 0, maxr = 1;
  for (int j = 0; j < b.n; ++j) {
    b.R[j] = cs_groups(b.P[j]);
    b.blk0[j] = blocks; b.d0[j] = d;
    blocks += cdiv(b.width[j], 64);
    d += (long)b.R[j] * b.width[j];
    if (b.R[j] > maxr) maxr = b.R[j];
  }
  b.blk0[b.n] = blocks;
  if (b.n == 1 || blocks > CS_TICKETS || CS_TICKETS / 2 + d > dpart_doubles) {         // one launch each, as before
    for (int j = 0; j < b.n; ++j) {
      ACVAE_TRY(colsum2(b.x[j], b.P[j], b.width[j], dpart, b.out[j], nullptr, 0, st));
      if (b.out_b[j]) ACVAE_TRY(colsum2(b.x[j], b.P[j], b.width[j], dpart, b.out_b[j], nullptr, 0, st));
    }
    return ACVAE_OK;
  }
  hipLaunchKernelGGL(colsum_batch_kernel, dim3(blocks, maxr), dim3(256), 0, st, b, dpart);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
long colsum_scratch_doubles(int width) { return CS_TICKETS / 2 + (long)CS_R * width; }
// the tickets at the head of a dpart scratch start at zero: once per composite call, in front of its first column sum
long colsum_ticket_words() { return CS_TICKETS; }
int colsum_tickets_reset(double* dpart, hipStream_t st) {
  return hipMemsetAsync(dpart, 0, CS_TICKETS * sizeof(unsigned), st) == hipSuccess ? ACVAE_OK : (int)hipGetLastError();
}

int bn_finalize(const float* partials, int P, int C, double count, const float* gamma, const float* beta,
                float* running_mean, float* running_var, int64_t* nbt, int training, float* scale, float* shift,
                float* mean, float* invstd, double* dpart, hipStream_t st) {
  if (training) {
    if (C % 64 != 0 || cdiv(2 * C, 64) > CS_TICKETS) return ACVAE_EUNSUPPORTED;
    const int R = cs_groups(P);
    hipLaunchKernelGGL(colsum_kernel<true>, dim3(2 * C / 64, R), dim3(256), 0, st, partials, P, 2 * C, dpart, nullptr, nullptr, 0,
                       BnFinalizeArgs{count, gamma, beta, running_mean, running_var, nbt, scale, shift, mean, invstd});
  } else {
    hipLaunchKernelGGL(bn_eval_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, C, gamma, beta, running_mean, running_var, scale,
                       shift, mean, invstd);
  }
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

int conv1_first_blocks(int N, int T) { return N * cdiv(T, C1_RB); }
template <class TY>
int conv1_first_fwd(const float* x, const float* scale0, const float* shift0, const float* W1, TY* Y,
                    float* partials, int N, int T, int F, hipStream_t st) {
  if (F != 64) return ACVAE_EUNSUPPORTED;
  hipLaunchKernelGGL(conv1_first_fwd_kernel<TY>, dim3(conv1_first_blocks(N, T)), dim3(256), 0, st, x, scale0, shift0, W1,
                     Y, partials, T, F);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int conv1_first_fwd<float>(const float*, const float*, const float*, const float*, float*, float*, int, int, int, hipStream_t);
template int conv1_first_fwd<bf16_t>(const float*, const float*, const float*, const float*, bf16_t*, float*, int, int, int, hipStream_t);
template <class TY>
int conv1_first_bwd(const float* x, const float* scale0, const float* shift0, const float* mean0, const float* invstd0,
                    const float* W1, const TY* dY, float* dw_part, float* bn_part, float* dW1, float* dgamma0,
                    float* dbeta0, double* dpart, int N, int T, int F, hipStream_t st) {
  if (F != 64) return ACVAE_EUNSUPPORTED;
  const int nb = conv1_first_blocks(N, T);
  hipLaunchKernelGGL(conv1_first_bwd_kernel<TY>, dim3(nb), dim3(256), 0, st, x, scale0, shift0, mean0, invstd0, W1, dY,
                     dw_part, bn_part, T, F);
  ACVAE_TRY(colsum2(dw_part, nb, 576, dpart, dW1, nullptr, 0, st));
  // bn0: y = xhat*gamma + beta with xin = scale0*x + shift0  ->  dbeta = sum dxin, dgamma = sum dxin*xhat
  ACVAE_TRY(colsum2(bn_part, nb, 128, dpart, dbeta0, dgamma0, 64, st));
  return ACVAE_OK;
}
template int conv1_first_bwd<float>(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*, double*, int, int, int, hipStream_t);
template int conv1_first_bwd<bf16_t>(const float*, const float*, const float*, const float*, const float*, const float*, const bf16_t*, float*, float*, float*, float*, float*, double*, int, int, int, hipStream_t);

template <class T>
int bn_relu_pool(const T* Y, const float* scale, const float* shift, T* P, int N, int H, int W, int C,
                 DropoutSpec drop, hipStream_t st, bool pool) {
  const long total = pool ? (long)N * (H / 2) * (W / 2) * (C / 4) : (long)N * H * W * (C / 4);
  if (total <= 0 || C % 4 != 0) return ACVAE_EINVAL;
  if (pool)
    hipLaunchKernelGGL(bn_relu_pool_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st, Y, scale, shift, P, N, H, W, C, drop);
  else
    hipLaunchKernelGGL(bn_relu_drop_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, st, Y, scale, shift, P, N, H, W, C, drop);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int bn_relu_pool<float>(const float*, const float*, const float*, float*, int, int, int, int, DropoutSpec, hipStream_t, bool);
template int bn_relu_pool<bf16_t>(const bf16_t*, const float*, const float*, bf16_t*, int, int, int, int, DropoutSpec, hipStream_t, bool);

// Pixels per reduce block: a block's 256 threads are (C/4 channel quads) x (256 / (C/4) pixel lanes) and every thread should
// walk ~32 pixels - with a fixed 512 pixels per block the 512-channel layers ran 63 blocks of 256-iteration threads
// (207 us for a 65 MB tensor; 4x that many blocks of short loops: HBM-bound like the wide layers).
static int bnb_pix(int C) {
  const int Cc = C < 1024 ? C : 1024;
  const int npl = 256 / (Cc / 4) > 0 ? 256 / (Cc / 4) : 1;
  return npl * 32 < 32 ? 32 : npl * 32;
}
int bn_bwd_blocks(int N, int H, int W, int C) { return cdiv((long)N * H * W, bnb_pix(C)); }
template <class T>
int bn_bwd(const T* Y, const T* dO, int upstream, const float* scale, const float* shift, const float* mean,
           const float* invstd, float* partials, float* sum_g, float* sum_gy, T* dY, double* dpart, int N, int H,
           int W, int C, DropoutSpec drop, hipStream_t st, bool batch_stats, int ready_rows) {
  const int Cc = C < 1024 ? C : 1024;
  if (C % 4 != 0 || 1024 % Cc != 0 || C % Cc != 0) return ACVAE_EUNSUPPORTED;
  if (ready_rows > 0 && upstream != UP_PLAIN) return ACVAE_EINVAL;
  const dim3 rgrid(bn_bwd_blocks(N, H, W, C), C / Cc);
  const int nb = rgrid.x;
  const size_t shm = 256 * 8 * sizeof(float);
  const long total = (long)N * H * W * (C / 4);
#define BN_BWD_LAUNCH(UP_)                                                                                             \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<UP_, T>), rgrid, dim3(256), shm, st, Y, dO, scale, shift, mean, invstd, partials, \
                     N, H, W, C, bnb_pix(C), drop);                                                                       \
  ACVAE_TRY(colsum2(partials, nb, 2 * C, dpart, sum_g, sum_gy, C, st)); /* sum_g (= dbeta) | sum_gy (= dgamma) */        \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<UP_, T>), dim3(ew_grid(total)), dim3(256), 0, st, Y, dO, scale, shift, mean,  \
                     invstd, sum_g, sum_gy, dY, N, H, W, C, drop, batch_stats ? 1 : 0)
  const float invn = batch_stats ? 1.0f / (float)((long)N * H * W) : 0.f;
  const bool small = (long)N * H * W * C < (1L << 31);
  if (upstream == UP_POOL && small && H >= 2 && W >= 2) {
    // a thread per pooled element: one dropout hash and one upstream load for four pixels
    const long wins = (long)N * (H / 2) * (W / 2);
    const int wpb = bnb_pix(C) / 4 > 0 ? bnb_pix(C) / 4 : 1;
    const dim3 pgrid((unsigned)cdiv(wins, wpb), C / Cc);            // <= nb of the layout: a window is four pixels
    hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T>), pgrid, dim3(256), shm, st, Y, dO, scale, shift, mean, invstd, partials, N, H, W,
                       C, wpb, drop);
    ACVAE_TRY(colsum2(partials, (int)pgrid.x, 2 * C, dpart, sum_g, sum_gy, C, st));
    const long totalp = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T>), dim3(ew_grid(totalp)), dim3(256), 0, st, Y, dO, scale, shift, mean, invstd,
                       sum_g, sum_gy, dY, N, H, W, C, drop, invn);
  } else if (upstream == UP_POOL) { BN_BWD_LAUNCH(UP_POOL); }
  else if (upstream == UP_DROP) { BN_BWD_LAUNCH(UP_DROP); }
  else {
    if (ready_rows <= 0)
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<UP_PLAIN, T>), rgrid, dim3(256), shm, st, Y, dO, scale, shift, mean, invstd, partials,
                         N, H, W, C, bnb_pix(C), drop);
    ACVAE_TRY(colsum2(partials, ready_rows > 0 ? ready_rows : nb, 2 * C, dpart, sum_g, sum_gy, C, st));
    // grid stride a multiple of C/4 (256 is one for C <= 1024; an even grid makes it one for C = 2048)
    int g = ew_grid((total + 1) / 2);
    if ((C / 4) > 256 && (g & 1)) ++g;
    if ((256L * g) % (C / 4) == 0)
      hipLaunchKernelGGL((bn_bwd_apply_plain_kernel<T>), dim3(g), dim3(256), 0, st, Y, dO, scale, shift, mean, invstd, sum_g, sum_gy,
                         dY, total, C / 4, invn);
    else
      hipLaunchKernelGGL((bn_bwd_apply_kernel<UP_PLAIN, T>), dim3(ew_grid(total)), dim3(256), 0, st, Y, dO, scale, shift, mean,
                         invstd, sum_g, sum_gy, dY, N, H, W, C, drop, batch_stats ? 1 : 0);
  }
#undef BN_BWD_LAUNCH
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int bn_bwd<float>(const float*, const float*, int, const float*, const float*, const float*, const float*, float*, float*, float*, float*, double*, int, int, int, int, DropoutSpec, hipStream_t, bool, int);
template int bn_bwd<bf16_t>(const bf16_t*, const bf16_t*, int, const float*, const float*, const float*, const float*, float*, float*, float*, bf16_t*, double*, int, int, int, int, DropoutSpec, hipStream_t, bool, int);

template <class T>
int relu_mask(const T* Y, const float* scale, const float* shift, uint8_t* out, int N, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(relu_mask_kernel<T>, dim3(ew_grid((long)N * H * W * C)), dim3(256), 0, st, Y, scale, shift, out, N, H, W, C);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int relu_mask<float>(const float*, const float*, const float*, uint8_t*, int, int, int, int, hipStream_t);
template int relu_mask<bf16_t>(const bf16_t*, const float*, const float*, uint8_t*, int, int, int, int, hipStream_t);

template <class T>
int freq_mean(const T* P, float* out, long rows, int Fp, int C, hipStream_t st) {
  hipLaunchKernelGGL(freq_mean_kernel<T>, dim3(ew_grid(rows * C)), dim3(256), 0, st, P, out, rows, Fp, C);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template <class T>
int freq_mean_bwd(const float* dae, T* dP, long rows, int Fp, int C, hipStream_t st) {
  hipLaunchKernelGGL(freq_mean_bwd_kernel<T>, dim3(ew_grid(rows * Fp * C)), dim3(256), 0, st, dae, dP, rows, Fp, C);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
template int freq_mean<float>(const float*, float*, long, int, int, hipStream_t);
template int freq_mean<bf16_t>(const bf16_t*, float*, long, int, int, hipStream_t);
template int freq_mean_bwd<float>(const float*, float*, long, int, int, hipStream_t);
template int freq_mean_bwd<bf16_t>(const float*, bf16_t*, long, int, int, hipStream_t);
int time_pool(const float* ae, float* out, int N, int S, int C, DropoutSpec drop, hipStream_t st) {
  hipLaunchKernelGGL(time_pool_kernel, dim3(ew_grid((long)N * C)), dim3(256), 0, st, ae, out, N, S, C, drop);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
int relu_dropout(float* x, int total, DropoutSpec drop, hipStream_t st) {
  hipLaunchKernelGGL(relu_dropout_kernel, dim3(ew_grid(total)), dim3(256), 0, st, x, total, drop);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

}  // namespace acvae
