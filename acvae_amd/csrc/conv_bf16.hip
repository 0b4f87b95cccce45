// bf16 convolution kernels of the Cnn10 / Cnn14 stack for BASELINE configs[2] ("bf16 forward / fp32 loss"): activations
// and repacked weights are STORED in bf16 (NHWC, as in conv.hip), products run on v_mfma_f32_32x32x16_bf16 (16x the fp32
// MFMA rate), accumulation, BatchNorm statistics, weight gradients, the text side and the loss stay fp32.
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md §3): lane l, r = l & 31, h = l >> 5:
//   A: A[row r][k = 8h + j], B: B[k = 8h + j][col r], j = 0..7 (8 bf16 = one 16-byte fragment per lane),
//   D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * h                       (as the fp32 forms).
//
// conv_igemm_bf16_kernel - conv3x3 forward AND data gradient (implicit GEMM, NT):  Y[p][co] = sum_{tap,ci} act(X)[p+tap][ci] W[co][tap][ci]
//   128 pixels x BN output channels per workgroup, 4 wavefronts as 2 x 2 (64 x BN/2 each), K order dy x 64-channel chunk x dx.
//   As in the fp32 kernel ONE activation strip of 130 pixels per (dy, chunk) serves the three horizontal taps (read at row
//   offsets 0 / 1 / 2, border lanes zeroed); the previous layer's BatchNorm+ReLU is applied while staging the strip
//   (bf16 -> fp32 affine, relu -> bf16).  LDS rows are 64 channels = 128 B padded to 144 B: every ds_read_b128 lane group
//   then covers 16 distinct 16-byte slots (conflict-free).  Epilogue: the fp32 accumulators are rounded to bf16, the
//   per-channel sum / sum-of-squares partials are taken from the ROUNDED values (the statistics describe the tensor that
//   is stored), two neighbouring channels are packed per lane and written as dwords.
//
// conv_wgrad_bf16_kernel - weight gradient (TN):  dW[co][tap][ci] = sum_p dY[p][co] act(X)[p+tap][ci]
//   Both operands are pixel-major in memory while the MFMA wants the reduction index (pixels) innermost per lane, so the
//   LDS tiles stay [pixel][channel] as loaded and the fragments are read with ds_read_b64_tr_b16 (the hardware transposing
//   read: 4 pixels x 16 channels per 16-lane group).  Tile 128 output channels x 192 columns (3 horizontal taps x 64 input
//   channels of one vertical tap) from ONE 34-pixel strip per 32-pixel K-step; pixels whose horizontal neighbour leaves
//   the image row are switched off by AND-masks staged beside the strip.  Split over pixel slices into fp32 slabs that
//   wgrad_reduce_kernel (conv.hip) sums in fixed order.
#include <cstdlib>
#include "mfma_tile.h"
#include "../../include/acvae_hip.h"
#include "conv.h"
#include "prof.h"

namespace {

constexpr int KC = 64;              // channels per K chunk (128 B of bf16)
constexpr int LDR = KC + 8;         // LDS row length in bf16 elements: 144 B
constexpr int BMT = 128;            // pixels per tile
constexpr int SROWS = BMT + 2;      // strip rows

struct FastDivB {
  unsigned long long mul;
  static FastDivB make(int d) { return FastDivB{((1ULL << 40) + (unsigned long long)d - 1) / (unsigned long long)d}; }
  __device__ __forceinline__ int div(int x) const { return (int)(((unsigned long long)(unsigned)x * mul) >> 40); }
};

// relu(x * sc + sh) on 8 packed bf16 (one 16-byte chunk); sc / sh: 8 floats as two float4
__device__ __forceinline__ uint4 bnrelu8(uint4 v, float4 s0, float4 s1, float4 h0, float4 h1) {
  uint4 o;
  o.x = pack_bf16(fmaxf(bf16_lo(v.x) * s0.x + h0.x, 0.f), fmaxf(bf16_hi(v.x) * s0.y + h0.y, 0.f));
  o.y = pack_bf16(fmaxf(bf16_lo(v.y) * s0.z + h0.z, 0.f), fmaxf(bf16_hi(v.y) * s0.w + h0.w, 0.f));
  o.z = pack_bf16(fmaxf(bf16_lo(v.z) * s1.x + h1.x, 0.f), fmaxf(bf16_hi(v.z) * s1.y + h1.y, 0.f));
  o.w = pack_bf16(fmaxf(bf16_lo(v.w) * s1.z + h1.z, 0.f), fmaxf(bf16_hi(v.w) * s1.w + h1.w, 0.f));
  return o;
}

// BDMA: the weight panel goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write - the
// VGPR -> LDS store path, ~79 B/clk/CU, is what the register-staged panel spends most of its LDS time on).  A DMA piece is
// lane-linear (8 rows x 128 B per wave instruction), so the panel image has unpadded 128-byte rows and the bank spread
// comes from an XOR swizzle of the 16-byte chunk index with (row >> 1) & 7, applied to the SOURCE address by the loader and
// to the read address by the MFMA fragment reads: every ds_read_b128 lane group then covers 16 distinct 16-byte slots.
template <int BN, bool BDMA>
struct alignas(16) IgSmem {
  alignas(16) bf16_t a[2][SROWS * LDR];
  alignas(16) bf16_t b[2][BN * (BDMA ? KC : LDR)];
};

// grid: x = pixel tiles, y = cout tiles (XCD-aware order as in the fp32 kernel).  256 threads, every wave loads and multiplies;
// two workgroups per CU cover each other's staging.
template <int BN, bool BDMA>
__device__ __forceinline__ void conv_igemm_bf16_body(const bf16_t* __restrict__ X, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const bf16_t* __restrict__ Wp,
                                                     bf16_t* __restrict__ Y, float* __restrict__ partials, int M, int H,
                                                     int W, int C, int Cout, IgSmem<BN, BDMA>& sm) {
  constexpr int NTN = BN / 64;       // 32-column MFMA tiles per wave along N
  constexpr int BR = BN / 32;        // weight-panel 16-byte slots per thread
  int bm, bn;
  __builtin_amdgcn_s_setprio(3);      // prologue / epilogue beside the other workgroup's main loop (conv_wino.hip, round 4): -0.3 % of the bf16 step
  mfma::xcd_tile(gridDim.x, gridDim.y, bm, bn);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int row0 = bm * BMT, col0 = bn * BN;
  const int K = 9 * C;
  const int nchunk = C / KC;
  const int NS = 9 * nchunk;          // sub-stages: (dy, chunk) groups x 3 horizontal taps

  // ---- loader state: 16-byte slot (row, c8) of the strip / panel
  const int c8 = (tid & 7) * 8, lrow = tid >> 3;
  int qh[5];
  long qbase[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int r = lrow + 32 * j;                      // strip row; slot 4 exists for rows 128, 129 (tid < 16)
    const long q = (long)row0 - 1 + r;
    const bool live = r < SROWS && q >= 0 && q < M;
    qh[j] = live ? (int)((q / W) % H) : -100000;
    qbase[j] = live ? q * C : 0;
  }
  uint4 pa[5], pb[BR];
  unsigned amask = 0;
  float4 s0, s1, h0, h1;
  auto issue_a = [&](int grp) {
    const int dyi = grp / nchunk, chunk = grp - dyi * nchunk, dy = dyi - 1;
    const int ci = chunk * KC + c8;
    const long off = (long)dy * W * C + ci;
    if (scale) {
      s0 = *reinterpret_cast<const float4*>(scale + ci); s1 = *reinterpret_cast<const float4*>(scale + ci + 4);
      h0 = *reinterpret_cast<const float4*>(shift + ci); h1 = *reinterpret_cast<const float4*>(shift + ci + 4);
    }
    amask = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int hh = qh[j] + dy;
      const bool ok = hh >= 0 && hh < H;
      pa[j] = *reinterpret_cast<const uint4*>(ok ? X + qbase[j] + off : X);      // always a legal address
      amask |= (ok ? 1u : 0u) << j;
    }
  };
  auto put_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      uint4 v = pa[j];
      if (scale) v = bnrelu8(v, s0, s1, h0, h1);
      if (!((amask >> j) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
      const int r = lrow + 32 * j;
      if (j < 4 || r < SROWS) *reinterpret_cast<uint4*>(&sm.a[buf][r * LDR + c8]) = v;
    }
  };
  // weight panel of sub-stage s: tap = dyi*3 + dxi, channels chunk*64..: Wp[co][tap*C + ci]
  auto issue_b = [&](int s) {
    const int grp = s / 3, dxi = s - grp * 3;
    const int dyi = grp / nchunk, chunk = grp - dyi * nchunk;
    if (BDMA) {
      // wave w moves the 8-row pieces w, w + 4, ...: lane l -> row piece*8 + (l >> 3), LDS position l & 7, source chunk
      // (l & 7) ^ ((row >> 1) & 7); rows past Cout re-read the last row (their output columns are never stored)
#pragma unroll
      for (int j = 0; j < BN / 32; ++j) {
        const int row = (wave + 4 * j) * 8 + (lane >> 3);
        const int chunkc = (lane & 7) ^ ((row >> 1) & 7);
        const int co = col0 + row < Cout ? col0 + row : Cout - 1;
        const bf16_t* src = Wp + (long)co * K + (long)(dyi * 3 + dxi) * C + chunk * KC + chunkc * 8;
        __builtin_amdgcn_global_load_lds(src, &sm.b[s & 1][(wave + 4 * j) * 8 * KC], 16, 0, 0);
      }
      return;
    }
    const long koff = (long)(dyi * 3 + dxi) * C + chunk * KC + c8;
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int co = col0 + lrow + 32 * j;
      pb[j] = *reinterpret_cast<const uint4*>(Wp + (long)(co < Cout ? co : 0) * K + koff);
    }
  };
  auto put_b = [&](int buf) {
    if (BDMA) return;
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int r = lrow + 32 * j;
      *reinterpret_cast<uint4*>(&sm.b[buf][r * LDR + c8]) = (col0 + r < Cout) ? pb[j] : make_uint4(0u, 0u, 0u, 0u);
    }
  };

  // ---- matrix state
  f32x16 acc[2][NTN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bool okl[2], okr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int px = row0 + wm * 64 + i * 32 + li;
    const int w = px % W;
    okl[i] = w > 0; okr[i] = w < W - 1;
  }

  // An LDS-DMA is ordered for other waves' reads only by the issuing wave's vmcnt wait followed by a barrier
  auto dma_wait = [&]() { if (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  issue_a(0);
  issue_b(0);
  put_a(0);
  put_b(0);
  dma_wait();
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  for (int s = 0; s < NS; ++s) {
    const int grp = s / 3, dxi = s - grp * 3;
    const bool more = s + 1 < NS;
    const bool next_grp = more && dxi == 2;
    if (more) issue_b(s + 1);
    if (next_grp) issue_a(grp + 1);
    {
      const bf16_t* As = sm.a[grp & 1] + (wm * 64 + li + dxi) * LDR + lh * 8;   // strip row of pixel p + (dxi - 1) = (p - row0) + dxi
      const int brow = wn * (BN / 2) + li;                 // swizzle term (row >> 1) & 7 is the same for brow and brow + 32
      const bf16_t* Bs = BDMA ? sm.b[s & 1] + brow * KC : sm.b[s & 1] + brow * LDR + lh * 8;
      const int bsw = (brow >> 1) & 7;
      const bool zl = dxi == 0, zr = dxi == 2;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        bf16x8 af[2], bfr[NTN];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          uint4 v = *reinterpret_cast<const uint4*>(As + i * 32 * LDR + ks * 16);
          if ((zl && !okl[i]) || (zr && !okr[i])) v = make_uint4(0u, 0u, 0u, 0u);
          af[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < NTN; ++j)
          bfr[j] = BDMA ? *reinterpret_cast<const bf16x8*>(Bs + j * 32 * KC + (((ks * 2 + lh) ^ bsw) << 3))
                        : *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDR + ks * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NTN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) put_b((s + 1) & 1);
    if (next_grp) put_a((grp + 1) & 1);
    dma_wait();
    __syncthreads();
  }

  // ---- epilogue: round to bf16, statistics of the rounded values, two channels per dword
  __builtin_amdgcn_s_setprio(3);
  float* red = reinterpret_cast<float*>(&sm.a[0][0]);     // LDS is free (barrier above)
#pragma unroll
  for (int j = 0; j < NTN; ++j) {
    const int nl = wn * (BN / 2) + j * 32 + li;
    const int n = col0 + nl;
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        // lane pair (li even, li + 1): the even lane stores the pair of register r, the odd lane that of register r + 1
        const float v0 = round_bf16(acc[i][j][r]), v1 = round_bf16(acc[i][j][r + 1]);
        const int m0 = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m0 < M) { sum += v0; sq += v0 * v0; }
        if (m0 + 1 < M) { sum += v1; sq += v1 * v1; }
        const float o0 = __shfl_xor(v0, 1, 64), o1 = __shfl_xor(v1, 1, 64);
        const bool odd = li & 1;
        const int m = odd ? m0 + 1 : m0;
        const unsigned packed = odd ? pack_bf16(o1, v1) : pack_bf16(v0, o0);
        const int nn = n & ~1;
        if (m < M && nn + 1 < Cout) *reinterpret_cast<unsigned*>(Y + (long)m * Cout + nn) = packed;
        else if (m < M && nn < Cout) Y[(long)m * Cout + nn] = (bf16_t)(odd ? o1 : v0);
      }
    if (partials) {
      sum += __shfl_xor(sum, 32, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (lh == 0) {
        red[(wm * 2 + 0) * BN + nl] = sum;
        red[(wm * 2 + 1) * BN + nl] = sq;
      }
    }
  }
  if (partials) {
    __syncthreads();
    if (tid < BN && col0 + tid < Cout) {
      float* out = partials + (long)bm * 2 * Cout + col0 + tid;
      out[0] = red[0 * BN + tid] + red[2 * BN + tid];
      out[Cout] = red[1 * BN + tid] + red[3 * BN + tid];
    }
  }
}

// (The body lives in a __device__ function on purpose: with the inline asm / LDS-DMA builtin directly inside the templated
// __global__ function hipcc 7.2 emitted no host stub for it - an undefined __device_stub__ symbol at load time, no diagnostic.)
template <int BN, bool BDMA>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16_kernel(const bf16_t* __restrict__ X,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const bf16_t* __restrict__ Wp, bf16_t* __restrict__ Y,
                                                                 float* __restrict__ partials, int M, int H, int W, int C,
                                                                 int Cout) {
  __shared__ IgSmem<BN, BDMA> sm;
  conv_igemm_bf16_body<BN, BDMA>(X, scale, shift, Wp, Y, partials, M, H, W, C, Cout, sm);
}

// ------------------------------------------------------------------------------------------ weight gradient
// EM = 2: 128 output channels per tile, 32 pixels per K-step, the wave rows split the channels.
// EM = 1: 64 output channels (the 64-channel layers: a 128-row tile would multiply zeros half the time), 64 pixels per
//         K-step, the wave rows split the K-step (k-steps wm, wm + 2) and write slabs of their own (2z + wm).
constexpr int S_LD = KC + 32;           // strip row: 128 B + 64 B pad -> the 4 pixel rows of a transposed read hit 4 bank quarters
template <int EM>
struct alignas(16) WgSmem {
  static constexpr int TM = 64 * EM, KPT = EM == 1 ? 64 : 32, A_LD = TM + 32;   // dY tile row + 64 B pad
  alignas(16) bf16_t a[2][KPT * A_LD];
  alignas(16) bf16_t s[2][(KPT + 2) * S_LD];
  alignas(16) unsigned short mk[2][2][KPT];    // [side 0 = left neighbour ok, 1 = right][pixel]: 0xffff / 0
};

__device__ __forceinline__ bf16x4 tr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int EM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const bf16_t* __restrict__ dY,
                                                                 const bf16_t* __restrict__ X,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 float* __restrict__ slab, int M, int Cout, int H, int W,
                                                                 int C, FastDivB dW_, FastDivB dH_, int k_per) {
  constexpr int TM = WgSmem<EM>::TM, KP = WgSmem<EM>::KPT, A_LD = WgSmem<EM>::A_LD;
  constexpr int CPR = TM / 8, RPP = 256 / CPR;      // dY tile: 16-byte chunks per pixel row, pixel rows per loader pass (2 passes)
  constexpr int SJ = (KP + 2 + 31) / 32;            // strip slots per thread
  __shared__ WgSmem<EM> sm;
  const int NC = 9 * C;
  const int tiles = gridDim.x * gridDim.y;
  const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xcd = b & 7, idx = b >> 3;                       // slices of one XCD share the operand rows in its L2
  const int z = (idx / tiles) * 8 + xcd, tile = idx % tiles;
  const int bx = tile % gridDim.x, by = tile / gridDim.x;
  const int ncg = C / KC;
  const int dyi = by / ncg, c0 = (by - dyi * ncg) * KC, dy = dyi - 1;
  const int k_begin = z * k_per, k_end = min(M, k_begin + k_per);
  const int nk = k_end > k_begin ? (k_end - k_begin + KP - 1) / KP : 0;
  const int row0 = bx * TM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- loader: dY tile KP pixels x TM channels = 512 slots of 16 B (2 per thread); strip (KP + 2) x 64 channels
  const int ac8 = (tid % CPR) * 8, arow = tid / CPR;        // dY: rows arow, arow + RPP
  const int sc8 = (tid & 7) * 8, srow = tid >> 3;           // strip: rows srow + 32 j
  float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0;
  if (scale) {
    s0 = *reinterpret_cast<const float4*>(scale + c0 + sc8); s1 = *reinterpret_cast<const float4*>(scale + c0 + sc8 + 4);
    h0 = *reinterpret_cast<const float4*>(shift + c0 + sc8); h1 = *reinterpret_cast<const float4*>(shift + c0 + sc8 + 4);
  }
  uint4 pa[2], ps[SJ];
  bool aok[2], sok[SJ];
  unsigned short mkv = 0;
  auto issue = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = k0 + arow + RPP * j;
      const int co = row0 + ac8;
      aok[j] = p < k_end && co < Cout;
      pa[j] = *reinterpret_cast<const uint4*>(aok[j] ? dY + (long)p * Cout + co : dY);
    }
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const int r = srow + 32 * j;
      const long q = (long)k0 - 1 + r;
      // a strip row is needed only by pixels of this slice: rows beyond the slice end may stay zero (their dY rows are zero)
      bool ok = r < KP + 2 && q >= 0 && q < M;
      if (ok) {
        const int rowq = dW_.div((int)q);
        const int h = rowq - dH_.div(rowq) * H + dy;
        ok = h >= 0 && h < H;
      }
      sok[j] = ok;
      ps[j] = *reinterpret_cast<const uint4*>(ok ? X + (q + (long)dy * W) * C + c0 + sc8 : X);
    }
    if (tid < 2 * KP) {
      const int k = tid < KP ? tid : tid - KP;
      const int p = k0 + k;
      const int w = p - dW_.div(p) * W;
      mkv = (p < M && (tid < KP ? w > 0 : w < W - 1)) ? 0xffffu : 0u;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<uint4*>(&sm.a[buf][(arow + RPP * j) * A_LD + ac8]) = aok[j] ? pa[j] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const int r = srow + 32 * j;
      if (r < KP + 2) {
        uint4 v = ps[j];
        if (scale) v = bnrelu8(v, s0, s1, h0, h1);
        if (!sok[j]) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(&sm.s[buf][r * S_LD + sc8]) = v;
      }
    }
    if (tid < 2 * KP) sm.mk[buf][tid < KP ? 0 : 1][tid < KP ? tid : tid - KP] = mkv;
  };

  // transposed-read lane addressing: 16-lane group g = lane >> 4 covers columns 16 * (g & 1) .. + 15 and the pixel rows
  // 8 * (g >> 1) + {0..3} (second read: + 4); lane 4q + p of the group supplies row q, columns 4p .. 4p + 3
  const int grp16 = lane >> 4, l16 = lane & 15;
  const int trow = 8 * (grp16 >> 1) + (l16 >> 2), tcol = 16 * (grp16 & 1) + 4 * (l16 & 3);

  if (nk > 0) { issue(k_begin); stash(0); }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) issue(k_begin + (ks + 1) * KP);
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      const int kk = EM == 2 ? kq : 2 * kq + wm;     // EM = 1: the wave rows take alternate k-steps of the 64-pixel K-step
      bf16x8 af[2], bfr[3];
      const bf16_t* Ap = sm.a[cur] + (kk * 16 + trow) * A_LD + (EM == 2 ? wm * 64 : 0) + tcol;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = cat8(tr_read(Ap + i * 32), tr_read(Ap + 4 * A_LD + i * 32));
      // this lane's 8 pixels of the K-step: k = kk*16 + 8*lh + j -> border masks, 8 x 16 bit
      const uint4 ml = *reinterpret_cast<const uint4*>(&sm.mk[cur][0][kk * 16 + 8 * lh]);
      const uint4 mr = *reinterpret_cast<const uint4*>(&sm.mk[cur][1][kk * 16 + 8 * lh]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int t = wn * 3 + j;                    // column tile 0..5: horizontal tap t >> 1, channel half t & 1
        const int dxi = t >> 1;
        const bf16_t* Sp = sm.s[cur] + (kk * 16 + trow + dxi) * S_LD + 32 * (t & 1) + tcol;
        uint4 v = __builtin_bit_cast(uint4, cat8(tr_read(Sp), tr_read(Sp + 4 * S_LD)));
        if (dxi == 0) { v.x &= ml.x; v.y &= ml.y; v.z &= ml.z; v.w &= ml.w; }
        if (dxi == 2) { v.x &= mr.x; v.y &= mr.y; v.z &= mr.z; v.w &= mr.w; }
        bfr[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }
  // slab[z][co][(dy*3 + dx)*C + c]; EM = 1: the K-halves of the two wave rows go to slabs 2z and 2z + 1
  float* Cs = slab + (long)(EM == 2 ? z : 2 * z + wm) * Cout * NC;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = row0 + (EM == 2 ? wm * 64 : 0) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= Cout) continue;
      float* out = Cs + (long)m * NC + dyi * 3 * C + c0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int t = wn * 3 + j;
        out[(t >> 1) * C + 32 * (t & 1) + li] = acc[i][j][r];
      }
    }
}

}  // namespace

namespace acvae {

int conv3x3_igemm_bf16(const bf16_t* X, const float* scale, const float* shift, const bf16_t* Wp, bf16_t* Y,
                       float* partials, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  if (!X || !Wp || !Y) return ACVAE_EINVAL;
  if (Cin % KC != 0 || Cout % 2 != 0) return ACVAE_EUNSUPPORTED;
  if (!aligned16(X) || !aligned16(Wp) || (reinterpret_cast<uintptr_t>(Y) & 3u)) return ACVAE_EALIGN;
  const int M = N * H * W;
  prof_begin(ACVAE_PROF_CONV_IGEMM, st);
  // A/B switch: ACVAE_BF16_BDMA=0 keeps the register-staged weight panel, 2 uses the DMA for the 64-column tiles too.
  // Measured in one job (B=32, T=1000): 128-column launches 3-4 % faster with the DMA panel (fwd 237 -> 228, 216 -> 207 us),
  // the 64-column ones 1-3 % slower (339 -> 349 us): default 1 = DMA for the 128-column tiles only.
  static const int bdma_mode = getenv("ACVAE_BF16_BDMA") ? atoi(getenv("ACVAE_BF16_BDMA")) : 1;
  const bool bdma = bdma_mode == 2 || (bdma_mode == 1 && Cout > 64);
#define IG_LAUNCH(BN_, D_)                                                                                               \
  hipLaunchKernelGGL((conv_igemm_bf16_kernel<BN_, D_>), dim3(cdiv(M, BMT), cdiv(Cout, BN_)), dim3(256), 0, st, X, scale,  \
                     shift, Wp, Y, partials, M, H, W, Cin, Cout)
  if (Cout <= 64) { if (bdma) IG_LAUNCH(64, true); else IG_LAUNCH(64, false); }
  else            { if (bdma) IG_LAUNCH(128, true); else IG_LAUNCH(128, false); }
#undef IG_LAUNCH
  prof_end(ACVAE_PROF_CONV_IGEMM, st);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// pixel slices: whole rounds of the 2 x 256 resident workgroups, at least 8 K-steps per slice, a multiple of 8 (one group of
// slices per XCD); every slice costs one fp32 slab of Cout x 9 Cin that is written and read once more
static int wgrad_bf16_splits(int M, int Cout, int NC) {
  const int TM = Cout <= 64 ? 64 : 128, KP = Cout <= 64 ? 64 : 32;
  const long tiles = (long)cdiv(Cout, TM) * (NC / 192);
  const int maxs = (cdiv(M, 8 * KP) / 8) * 8;
  int best = 8;
  double best_t = 1e30;
  for (int k = 8; k <= (maxs < 8 ? 8 : maxs) && k <= 512; k += 8) {
    const long nb = tiles * k;
    const long rounds = (nb + 511) / 512;
    static const double slab_cost = getenv("ACVAE_WGB_SLABCOST") ? atof(getenv("ACVAE_WGB_SLABCOST")) : 0.5;   // measured: 6 -> 0.5 took the seven launches from 2.26 to 1.56 ms
    const double t = (double)rounds * ((double)M / k) + slab_cost * k * 32.0;    // K-steps per round x rounds + slab cost (in pixel units)
    if (t < best_t) { best_t = t; best = k; }
  }
  return best;
}
// slabs the reduce has to sum: one per pixel slice, two for the 64-channel tile (its wave rows split the K-step)
int conv3x3_wgrad_bf16_splits(int N, int H, int W, int Cin, int Cout) {
  return wgrad_bf16_splits(N * H * W, Cout, 9 * Cin) * (Cout <= 64 ? 2 : 1);
}
long conv3x3_wgrad_bf16_slab_floats(int N, int H, int W, int Cin, int Cout) {
  return (long)conv3x3_wgrad_bf16_splits(N, H, W, Cin, Cout) * Cout * 9 * Cin;
}

int conv3x3_wgrad_bf16_launch(const bf16_t* dY, const bf16_t* X, const float* scale, const float* shift, float* slab, int N,
                              int H, int W, int Cin, int Cout, hipStream_t st) {
  if (!dY || !X || !slab) return ACVAE_EINVAL;
  if (Cin % KC != 0 || Cout % 8 != 0) return ACVAE_EUNSUPPORTED;
  if (!aligned16(dY) || !aligned16(X)) return ACVAE_EALIGN;
  const int M = N * H * W, NC = 9 * Cin;
  if ((long)(M + 4096) * (W > H ? W : H) >= (1L << 40)) return ACVAE_EUNSUPPORTED;   // FastDiv range
  const int s = wgrad_bf16_splits(M, Cout, NC);
  const int KP = Cout <= 64 ? 64 : 32;
  const int k_per = cdiv(cdiv(M, s), KP) * KP;
  prof_begin(ACVAE_PROF_CONV_WGRAD, st);
  if (Cout <= 64) {
    const dim3 grid(cdiv(Cout, 64), NC / 192, s);
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel<1>, grid, dim3(256), 0, st, dY, X, scale, shift, slab, M, Cout, H, W, Cin,
                       FastDivB::make(W), FastDivB::make(H), k_per);
  } else {
    const dim3 grid(cdiv(Cout, 128), NC / 192, s);
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel<2>, grid, dim3(256), 0, st, dY, X, scale, shift, slab, M, Cout, H, W, Cin,
                       FastDivB::make(W), FastDivB::make(H), k_per);
  }
  prof_end(ACVAE_PROF_CONV_WGRAD, st);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

}  // namespace acvae
