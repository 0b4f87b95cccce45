// A1 (models/encoder.py:606-649 ConvBlock): the 3x3 convolutions of the encoder as Winograd F(2x2, 3x3) on the fp32
// matrix pipe - 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e. 2.25x fewer MFMA flops than the
// implicit GEMM of conv.hip for the same result (fp32 throughout; the transforms only add and halve, their rounding is
// of the order of the accumulation's own - bounds in tests/test_kernels_gpu.py).
//
//   Y = At [ sum_ci (G g G^T) .* (Bt d B) ] A      d: 4x4 input window, g: 3x3 filter, Y: 2x2 outputs
//
// One workgroup = 32 output tiles (R tile rows x TW tile columns of ONE clip, TW = W/2, R = 32/TW) x 64 output channels
// x all 16 transform positions: 16 GEMMs [32 tiles x Cin] x [Cin x 64] on FOUR wavefronts of 128 accumulators (218-230
// registers each), so that TWO workgroups share a CU (one wavefront of each per SIMD).  Until round 4 one workgroup of eight
// wavefronts (64 tiles) had the CU to itself, and its prologue (first window from HBM: 1.4-2.6 us), its epilogue (2.0-2.4 us)
// and the dispatch gap behind it (0.5 us) ran with the matrix pipes idle - 19 % of a 64-channel workgroup, 3 % of a
// 512-channel one (in-kernel timeline, profiles/r04_wino_lab.txt).  With two independent workgroups per CU the one's
// main loop runs under the other's prologue / epilogue.  Nothing transformed ever touches HBM or LDS:
//   * the activation window (2R+2 pixel rows x W+2 columns, 16 channels per stage) is staged ONCE, with the previous
//     layer's BatchNorm+ReLU and the zero padding applied on the way, into four (row parity, column parity) planes, so
//     that the 16 tiles a 16-lane LDS read group serves are 16 consecutive 16-byte slots;
//   * every wavefront owns ONE vertical frequency of all 32 tiles and both 32-column halves: it builds the Bt d B fragments of
//     its 4 positions in registers from 8 ds_read_b128 (two window rows) and 16 v_pk_add_f32 per 32 MFMAs - a fragment feeds two
//     MFMA groups (until round 4: two frequencies, one column half: 12 reads and 32 packed adds per 32 MFMAs);
//   * G g G^T is precomputed per step (wino_weights_kernel, all layers in one launch) into a 32 KB-per-8-channel-chunk image
//     whose fragments go from L2 STRAIGHT into the MFMA operand registers (raw_buffer_load_b128 into a ring of four
//     float4 per lane, reloaded four position groups ahead): they never pass through LDS (round 3; the LDS-DMA path
//     cost ~470 cycles per chunk);
//   * the epilogue applies At . A in registers, the four frequencies of a tile meet through LDS once per tile, and the
//     BatchNorm batch statistics of the raw output are reduced exactly like conv.hip's epilogue does.
// The same kernel is the data gradient (X = dY, filter flipped and transposed by wino_weights_kernel, no activation); the data
// gradient of a block's second convolution also reduces the BatchNorm + ReLU backward that consumes it (conv_wino_bnred_kernel).
#include "mfma_tile.h"
#include "../../include/acvae_hip.h"
#include "conv.h"
#include "prof.h"
#include <type_traits>

namespace {
using namespace mfma;

constexpr int WN_THREADS = 256;
constexpr int WN_TILES = 32;      // output tiles (2x2 pixels each) per workgroup
constexpr int WN_TN = 64;         // output channels per workgroup
// activation window in LDS, float4 slots: [quad WN_SQ][row parity WN_SR][column parity WN_SC][row (R+1)][col RW]; a plane
// holds (R+1)*RW <= 66 slots.  The strides are padded so that the 16 lanes of a staging write (4 pixels x 4 quads) land in
// 16 different 16-byte bank groups: WN_SC = 2 (mod 4) spreads the 4 pixels, WN_SQ = 4 (mod 16) the quads.
constexpr int WN_SC = 66, WN_SR = 2 * WN_SC, WN_SQ = 276;
constexpr int WN_BCHUNK = 16 * 2 * 64;   // float4 per weight chunk image: [position][k half][column]
constexpr int WN_MAXC = 2048;            // input channels whose BatchNorm scale / shift fit the LDS copy of the ACT kernel
#ifndef WN_PRIO_EDGE
#define WN_PRIO_EDGE 3                   // wavefront priority in prologue and epilogue (main loop: 0)
#endif
#ifndef WN_PF_DIST
#define WN_PF_DIST 64                    // prefetch distance in the XCD's run of tiles (2 workgroups on each of its 32 CUs)
#endif

// LDS: one array - the two window buffers (2 x 17664 B), then (ACT build) the scale / shift copy; the epilogue's exchange
// (6 sets x 8 KB) and the BatchNorm sums (1 KB) overlay it behind the main loop.  50-52 KB per workgroup, two per CU.
constexpr int WN_RAWBUF = 4 * WN_SQ;        // float4 per stage buffer
constexpr int WN_EX_F4 = 6 * 512;           // float4 of the exchange area: [set 6][row 16][lane 64] pairs
constexpr int WN_LDS_F4 = WN_EX_F4 + 64;    // float4 of a workgroup's array
static_assert(2 * WN_RAWBUF <= WN_LDS_F4 && 2 * WN_RAWBUF + 2 * WN_MAXC / 4 >= WN_LDS_F4, "exchange + sums overlay the window buffers (+ scale / shift)");

// Tile (row of the block, column) of MFMA row `row` (0..31, = lane & 31 of the A operand).  A ds_read_b128
// is served in four groups of 16 lanes - NOT consecutive ones: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS table) - and the 16 tiles of a group must sit in 16 different 16-byte bank groups.  So the
// lane is first mapped to (group, rank in the group); then TW >= 16: a group reads 16 neighbours of one tile row; TW < 16:
// group g reads the 16/TW tile rows g, g + 2, g + 4 ..., and the row pitch RW is chosen so that their slots tile the 256-byte
// bank span (TW = 8: RW = 12, rows 2 apart -> 384 = 128 B mod 256; TW = 4: RW = 6 -> 192 B; TW = 2: RW = 3 -> 96 B steps).
__device__ __forceinline__ void wino_tile(int row, int tw_shift, int& tyl, int& tx) {
  // 4-lane blocks 0..7 -> group * 16 + first rank: 0, 16, 20, 4, 24, 8, 12, 28
  const int cell = (int)((0x1c0c081804141000ULL >> (8 * (row >> 2))) & 0xff) + (row & 3);
  if (tw_shift >= 4) {
    tyl = cell >> tw_shift;
    tx = cell & ((1 << tw_shift) - 1);
  } else {
    const int g = cell >> 4, l16 = cell & 15;
    tx = l16 & ((1 << tw_shift) - 1);
    tyl = g + 2 * (l16 >> tw_shift);
  }
}
__host__ __device__ __forceinline__ int wino_row_pitch(int tw_shift) {
  return tw_shift == 3 ? 12 : tw_shift == 2 ? 6 : (1 << tw_shift) + 1;
}

struct WinoParams {
  const float* X;       // [N][H][W][C]
  const float* scale;   // nullable: no activation on the operand
  const float* shift;
  const float* U;       // wino_weights_kernel image
  float* Y;             // [N][H][W][Cout]
  float* partials;      // [blocks][2][Cout] or nullptr
  // data gradient feeding a BatchNorm + ReLU backward (WinoBnReduce): the tensor that BatchNorm normalised (same shape as Y),
  // its scale / shift / mean / invstd
  const float* rY;
  const float* rscale;
  const float* rshift;
  const float* rmean;
  const float* rinvstd;
  int N, H, W, C, Cout;
  int tw_shift;         // TW = W/2 = 1 << tw_shift
  int R;                // tile rows per workgroup = 32 / TW
  int bpc;              // workgroups (row blocks) per clip = ceil(ceil(H/2) / R)
  // divisions of the prologue as multiply-shifts (the launcher computes the constants; every wavefront of every workgroup
  // ran ~7 integer divisions by run-time values, ~30 instructions each, before its first load - round 4 timeline: 0.8 us of a
  // 22 us workgroup on the 64-channel layers)
  int w_shift;          // W = 1 << w_shift
  unsigned bpc_magic;   // x / bpc = umulhi(x, bpc_magic) for x < 2^20, bpc >= 2
  int nn_shift;         // log2(Cout / 64) when that is a power of two, else -1 (generic division)
  int bn_group;         // order of an XCD's run of tiles: 0 = column block fastest; G = 1, 2, 4: groups of G column blocks, inside a
                        // group the row block runs through all its values with the G column blocks fastest (see wino_tile_of)
};
// (row block, column block) of position t in the launch's tile order.  bn_outer = 0: the column block runs fastest - the
// workgroups an XCD runs side by side share their input window (one read from HBM, nn - 1 from its L2) and use ALL nn weight
// images at once: right while those fit the 4 MB L2 (nn x 4096 x C bytes).  Where they do not (256 -> 256 channels and wider:
// 4 - 17 MB) every workgroup fetched its 1 - 2 MB image from the Infinity Cache again - 2048 workgroups x 2 MB per launch on
// the 512-channel layers, FETCH_SIZE 3.0x the algorithmic reads (profiles/r04_e_traffic_conv_igemm.json).  bn_group = G: an XCD
// works through ALL row blocks with only G column blocks (as many images as fit 2 MB of its L2), then the next G: the input
// is read nn / G times instead - 771 -> 495 MB per launch on the 512-channel layers (G = 1).
__device__ __forceinline__ void wino_tile_of(const WinoParams& p, int t, int nm, int nn, int& bm, int& bn) {
  if (p.bn_group) {
    const int G = p.bn_group, per = nm * G;              // positions of a group of G column blocks
    int grp = (int)__fdividef((float)t, (float)per);     // t < 2^23: exact up to +-1, corrected below
    int rem = t - grp * per;
    if (rem < 0) { --grp; rem += per; } else if (rem >= per) { ++grp; rem -= per; }
    bm = rem >> (G >> 1);                                // G is 1, 2 or 4: log2 G = G >> 1
    bn = grp * G + (rem & (G - 1));
  } else if (p.nn_shift >= 0) { bn = t & (nn - 1); bm = t >> p.nn_shift; }
  else { bn = t % nn; bm = t / nn; }
}
// position of this workgroup: mfma_tile.h xcd_tile with the launcher's constants instead of divisions
__device__ __forceinline__ int wino_xcd_tile(const WinoParams& p, int nm, int nn, int& bm, int& bn) {
  const int total = nm * nn;
  const int b = blockIdx.x + blockIdx.y * nm;
  const int q = total >> 3, r = total & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int t = xcd * q + (xcd < r ? xcd : r) + idx;   // XCD x owns q (+1 if x < r) consecutive positions
  wino_tile_of(p, t, nm, nn, bm, bn);
  return t;
}

// The transform adds as v_pk_add_f32: two values per VALU issue slot.  In-kernel counters and the A/B below say the SIMD's issue
// slots are what a chunk runs out of (every instruction a wavefront issues, MFMA or not, costs its cycles; halving the 64 adds
// of a chunk: -4 %).  Written as inline asm: from <2 x float> IR the backend scalarises them again because every element ends
// up as a separate MFMA operand.  IEEE add / subtract, the same rounding as the scalar instructions.  The compiler's hazard
// recogniser does not look inside inline asm, so the block whose results feed MFMAs ends in the wait it would have inserted
// between a VALU write and an MFMA reading the register as SrcA (without it the MFMA reads stale values - seen).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define WN_PK_SUB(d, a, b) "v_pk_add_f32 " d ", " a ", " b " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define WN_PK_ADD(d, a, b) "v_pk_add_f32 " d ", " a ", " b "\n\t"
// the vertical half: its results only feed the block below (VALU to VALU: no wait needed)
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) {
  f32x2 lo, hi;
  asm(WN_PK_SUB("%0", "%2", "%4") WN_PK_SUB("%1", "%3", "%5")
      : "=&v"(lo), "=&v"(hi) : "v"(f32x2{a.x, a.y}), "v"(f32x2{a.z, a.w}), "v"(f32x2{b.x, b.y}), "v"(f32x2{b.z, b.w}));
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  f32x2 lo, hi;
  asm(WN_PK_ADD("%0", "%2", "%4") WN_PK_ADD("%1", "%3", "%5")
      : "=&v"(lo), "=&v"(hi) : "v"(f32x2{a.x, a.y}), "v"(f32x2{a.z, a.w}), "v"(f32x2{b.x, b.y}), "v"(f32x2{b.z, b.w}));
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
// the horizontal half of Bt d B for one vertical frequency: v = [t0 - t2, t1 + t2, t2 - t1, t1 - t3], MFMA operands
__device__ __forceinline__ void wino_htrans(const float4 (&t)[4], float4 (&v)[4]) {
  f32x2 o[8];
  asm(WN_PK_SUB("%0", "%8", "%12") WN_PK_SUB("%1", "%9", "%13")
      WN_PK_ADD("%2", "%10", "%12") WN_PK_ADD("%3", "%11", "%13")
      WN_PK_SUB("%4", "%12", "%10") WN_PK_SUB("%5", "%13", "%11")
      WN_PK_SUB("%6", "%10", "%14") WN_PK_SUB("%7", "%11", "%15")
      "s_nop 1"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(f32x2{t[0].x, t[0].y}), "v"(f32x2{t[0].z, t[0].w}), "v"(f32x2{t[1].x, t[1].y}), "v"(f32x2{t[1].z, t[1].w}),
        "v"(f32x2{t[2].x, t[2].y}), "v"(f32x2{t[2].z, t[2].w}), "v"(f32x2{t[3].x, t[3].y}), "v"(f32x2{t[3].z, t[3].w}));
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = make_float4(o[2 * k][0], o[2 * k][1], o[2 * k + 1][0], o[2 * k + 1][1]);
}
// in-place forms ("+v": the result replaces an operand, no third register set): a -= b / a += b
__device__ __forceinline__ void pk_sub_ip(float4& a, const float4& b) {
  f32x2 lo{a.x, a.y}, hi{a.z, a.w};
  asm(WN_PK_SUB("%0", "%0", "%2") WN_PK_SUB("%1", "%1", "%3") : "+v"(lo), "+v"(hi) : "v"(f32x2{b.x, b.y}), "v"(f32x2{b.z, b.w}));
  a = make_float4(lo[0], lo[1], hi[0], hi[1]);
}
__device__ __forceinline__ void pk_add_ip(float4& a, const float4& b) {
  f32x2 lo{a.x, a.y}, hi{a.z, a.w};
  asm(WN_PK_ADD("%0", "%0", "%2") WN_PK_ADD("%1", "%1", "%3") : "+v"(lo), "+v"(hi) : "v"(f32x2{b.x, b.y}), "v"(f32x2{b.z, b.w}));
  a = make_float4(lo[0], lo[1], hi[0], hi[1]);
}
// horizontal half, three of the four results in place of their operands: v = [t0 - t2, t1 + t2, t2 - t1, t1 - t3] with
// v0 in t0's registers, v2 in t2's, v3 in t3's and v1 in a new pair of pairs; t is dead afterwards
__device__ __forceinline__ void wino_htrans_ip(float4 (&t)[4], float4 (&v)[4]) {
  f32x2 a0{t[0].x, t[0].y}, a1{t[0].z, t[0].w}, c0{t[2].x, t[2].y}, c1{t[2].z, t[2].w}, e0{t[3].x, t[3].y}, e1{t[3].z, t[3].w};
  f32x2 n0, n1;
  asm(WN_PK_SUB("%0", "%0", "%2") WN_PK_SUB("%1", "%1", "%3")          // t0 - t2
      WN_PK_ADD("%6", "%8", "%2") WN_PK_ADD("%7", "%9", "%3")          // t1 + t2
      WN_PK_SUB("%2", "%2", "%8") WN_PK_SUB("%3", "%3", "%9")          // t2 - t1
      WN_PK_SUB("%4", "%8", "%4") WN_PK_SUB("%5", "%9", "%5")          // t1 - t3
      "s_nop 1"
      : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1), "+v"(e0), "+v"(e1), "=&v"(n0), "=&v"(n1)
      : "v"(f32x2{t[1].x, t[1].y}), "v"(f32x2{t[1].z, t[1].w}));
  v[0] = make_float4(a0[0], a0[1], a1[0], a1[1]);
  v[1] = make_float4(n0[0], n0[1], n1[0], n1[1]);
  v[2] = make_float4(c0[0], c0[1], c1[0], c1[1]);
  v[3] = make_float4(e0[0], e0[1], e1[0], e1[1]);
}
// first use of an accumulator: C = 0 as the instruction's inline constant instead of 16 v_mov per accumulator in the prologue
__device__ __forceinline__ void mfma4_first(f32x16& c, float4 a, float4 b) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, z, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma4(f32x16& c, float4 a, float4 b) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
}

// Epilogue of the three kernels.  A wavefront holds ONE vertical frequency XI of its 32 tiles x 64 output channels: per
// column half nh and accumulator row r it applies the horizontal half of At . A (4 positions -> the tile's 2 output columns:
// F = [m0 + m1 + m2, m1 - m2 - m3]) and the four wavefronts meet ONCE through LDS for the vertical half:
//   output row 0 = (F0 + F1) + F2,  output row 1 = F1 - (F3 + F2)        (F_x: the pair of wavefront x)
// Wavefront XI produces output row XI >> 1 of column half XI & 1, keeps its own pairs of that half in registers and reads the
// two others; six of the eight (frequency, half) sets are ever read by another wavefront - 6 x 8 KB of LDS over the window
// buffers (and, in the BatchNorm + ReLU build, its scale / shift copy), 16-32 writes and 32 reads of 8 bytes per lane, one
// barrier.  (First version: two rounds, one per column half, inside the 35 KB of the window buffers: three more barriers and
// 4.3-6.9 us per workgroup against 2.6-3.9; in-kernel timeline.)
// Addressing (round 4): the 16 tiles of a lane are four groups (k = r >> 2) of four tiles whose coordinates differ by a
// WAVE-UNIFORM step (wino_tile: cell = T[2k + h] + (r & 3), T a multiple of 4): so the lane computes one byte offset into its
// clip per group, the per-tile steps live in SGPRs, and the stores are buffer stores (32-bit lane offset + scalar offset)
// against a descriptor of the CLIP whose size makes the hardware drop rows past the clip's end - no address arithmetic, no
// branch per tile.
// MODE 1 (forward convolutions): the launch also returns the BatchNorm partial sums of its raw output (sum, sum of squares).
// MODE 2 (the data gradient of a block's second convolution): its output dA is the upstream gradient of the first
// convolution's BatchNorm + ReLU, whose backward starts with the sums  sum g, sum g * yhat  (g = dA where the activation
// was positive, yhat = (y - mean) * invstd) over the whole batch - bn_bwd_reduce_kernel, a pass over two tensors (0.40 ms per
// step in four launches).  The epilogue has dA in registers: it reads y at the pixels it stores (32 loads per lane, issued in
// front of the exchange) and returns the partial sums in the layout of bn_bwd_reduce_kernel ([blocks][2][Cout]).
template <int XI, int MODE>
__device__ __forceinline__ void wino_epilogue(const WinoParams& p, const f32x16 (&acc)[8], float2* ex, float* red, int li, int h,
                                              int lane, int n, int ty0, int bm, int bn) {
  constexpr int ROW = XI >> 1, NH = XI & 1;             // output row of the tile / column half this wavefront stores
  // exchange sets: 0: F0[1], 1: F1[0], 2: F1[1], 3: F2[0], 4: F2[1], 5: F3[0]
  constexpr int SET_OTHER = XI == 0 ? 0 : XI == 1 ? 1 : XI == 2 ? 4 : 5;      // this wavefront's pairs of the half it does not store
  constexpr int SET_OWN = XI == 1 ? 2 : XI == 2 ? 3 : -1;                      // ... of its own half (frequencies 1, 2: needed by the other row)
  constexpr int SET_A = XI == 0 ? 1 : XI == 1 ? 0 : XI == 2 ? 1 : 2, SET_B = XI == 0 ? 3 : XI == 1 ? 4 : XI == 2 ? 5 : 4;
  constexpr bool STATS = MODE != 0;
  const int tid = XI * 64 + lane;                       // = threadIdx.x, from the values taken afresh behind the main loop
  const int H = p.H, W = p.W, Cout = p.Cout;
  const int cout = bn * WN_TN + NH * 32 + li;
  // per group k: pixel (y, x) of the tile with r & 3 == 0 (y already the output row of the tile), byte offset into the clip
  int yk[4], voff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int tyl, tx;
    wino_tile(8 * k + 4 * h, p.tw_shift, tyl, tx);
    yk[k] = 2 * (ty0 + tyl) + ROW;
    voff[k] = ((yk[k] * W + 2 * tx) * Cout + cout) * 4;
  }
  // wave-uniform step of tile j = r & 3 inside a group: TW >= 4: two pixels to the right per tile; TW = 2 (wino_tile: the
  // tile column is bit 0 of the rank, bit 1 selects the tile row two further down): x + 2 (j & 1), y + 4 (j >> 1)
  int dyj[4], soff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dyj[j] = p.tw_shift == 1 ? 4 * (j >> 1) : 0;
    const int dx = p.tw_shift == 1 ? 2 * (j & 1) : 2 * j;
    soff[j] = (dyj[j] * W + dx) * Cout * 4;
  }
  const int pix = Cout * 4;                                  // bytes from a pixel to its right neighbour
  float2 yv[MODE == 2 ? 16 : 1];
  float rsc = 0.f, rsh = 0.f, rmu = 0.f, ris = 0.f;
  if (MODE == 2) {
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rY) + (long)n * H * W * Cout, 0, H * W * Cout * 4, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      yv[r].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, voff[r >> 2], soff[r & 3], 0));
      yv[r].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, voff[r >> 2], soff[r & 3] + pix, 0));
    }
    rsc = p.rscale[cout]; rsh = p.rshift[cout]; rmu = p.rmean[cout]; ris = p.rinvstd[cout];
  }
  float2 keep[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const float m0 = acc[0 + nh][r], m1 = acc[2 + nh][r], m2 = acc[4 + nh][r], m3 = acc[6 + nh][r];
      const float2 f = make_float2(m0 + m1 + m2, m1 - m2 - m3);
      if (nh == NH) {
        keep[r] = f;
        if (SET_OWN >= 0) ex[(SET_OWN * 16 + r) * 64 + lane] = f;
      } else {
        ex[(SET_OTHER * 16 + r) * 64 + lane] = f;
      }
    }
  }
  __syncthreads();
  // descriptor of clip n: stores whose offset lies past H * W * Cout * 4 bytes (rows >= H of a partial block) are dropped
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.Y + (long)n * H * W * Cout, 0, H * W * Cout * 4, 0x00020000);
  float s = 0.f, qq = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = r >> 2, j = r & 3;
    const float2 a = ex[(SET_A * 16 + r) * 64 + lane], b = ex[(SET_B * 16 + r) * 64 + lane];
    // the order of round 3's two-wavefront exchange (bit-identical results): (F0 + F1) + F2 and F1 - (F3 + F2)
    const float o0 = ROW ? a.x - (b.x + keep[r].x) : (keep[r].x + a.x) + b.x, o1 = ROW ? a.y - (b.y + keep[r].y) : (keep[r].y + a.y) + b.y;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), yrs, voff[k], soff[j], 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1), yrs, voff[k], soff[j] + pix, 0);
    if (MODE == 1) {
      const bool ok = yk[k] + dyj[j] < H;
      const float a0 = ok ? o0 : 0.f, a1 = ok ? o1 : 0.f;
      s += a0 + a1;
      qq += a0 * a0 + a1 * a1;
    }
    if (MODE == 2) {
      // bn_bwd_reduce_kernel's arithmetic on the two pixels (rows past the clip: y = 0 from the bounds-checked load, masked anyway)
      const bool ok = yk[k] + dyj[j] < H;
      const float g0 = (ok && yv[r].x * rsc + rsh > 0.f) ? o0 : 0.f, g1 = (ok && yv[r].y * rsc + rsh > 0.f) ? o1 : 0.f;
      s += g0; s += g1;
      qq += g0 * ((yv[r].x - rmu) * ris); qq += g1 * ((yv[r].y - rmu) * ris);
    }
  }
  if (STATS && p.partials) {
    s += __shfl_xor(s, 32, 64);
    qq += __shfl_xor(qq, 32, 64);
    if (h == 0) {
      red[(ROW * 2 + 0) * 64 + NH * 32 + li] = s;
      red[(ROW * 2 + 1) * 64 + NH * 32 + li] = qq;
    }
    __syncthreads();
    if (tid < WN_TN) {
      const float ts = red[(0 * 2 + 0) * 64 + tid] + red[(1 * 2 + 0) * 64 + tid], tq = red[(0 * 2 + 1) * 64 + tid] + red[(1 * 2 + 1) * 64 + tid];
      float* out = p.partials + (long)bm * 2 * p.Cout + bn * WN_TN + tid;
      out[0] = ts;
      out[p.Cout] = tq;
    }
  }
}

// XI: vertical frequency of this wavefront.  Window rows (of the tile's four) and sign: 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r1 - r3
template <int XI, bool ACT, int STATS>
__device__ __forceinline__ void conv_wino_body(const WinoParams& p, float4* raw0, float4* raw1, float4* scsh) {
  // Two workgroups share a CU: this wavefront's prologue / epilogue run beside a wavefront of the OTHER workgroup that is
  // inside its main loop, and the SIMD's round-robin issue gives each of their ~600 vector instructions one slot per 64-cycle
  // MFMA of the neighbour: 4-7 us of latency for ~1.5 us of work (in-kernel timeline).  WN_PRIO_EDGE raises the priority
  // outside the main loop so that these short phases run through and the wavefront joins the MFMA stream again.
  __builtin_amdgcn_s_setprio(WN_PRIO_EDGE);
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  int bm, bn;
  const int tpos = wino_xcd_tile(p, gridDim.x, gridDim.y, bm, bn);
  const int RW = wino_row_pitch(p.tw_shift), R = p.R;
  const int n = p.bpc == 1 ? bm : (int)__umulhi((unsigned)bm, p.bpc_magic), ty0 = (bm - n * p.bpc) * R;   // (2^32 / 1 does not fit the constant)
  const int H = p.H, W = p.W, C = p.C;

  // ---------------------------------------------------------------- staging items of this thread (fixed over stages)
  // item e = tid + 256 j -> pixel e / 4 of the window's W real columns, channel quad e % 4 (= tid % 4): at most 4 rows x 64
  // columns x 4 quads = 4 items per thread.  The two padding columns of the window never change: zeroed once, below.
  const int nitems = ((2 * R + 2) << p.w_shift) * 4;
  const int q = tid & 3;
  for (int i = tid; i < (2 * R + 2) * 8; i += WN_THREADS) {
    const int ry = i >> 3, side = (i >> 2) & 1, qi = i & 3;
    const int rx = side ? W + 1 : 0;
    const int slot = qi * WN_SQ + (ry & 1) * WN_SR + (rx & 1) * WN_SC + (ry >> 1) * RW + (rx >> 1);
    raw0[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    raw1[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned goff[4];          // element offsets into X (host side: the tensor has < 2^32 elements)
  int loff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + WN_THREADS * j;
    const bool lv = e < nitems;
    const int px = e >> 2;
    const int ry = px >> p.w_shift, x = px & (W - 1), rx = x + 1;
    const int y = 2 * ty0 - 1 + ry;
    const bool ok = lv && y >= 0 && y < H;
    goff[j] = ok ? (unsigned)((((long)(n * H + y) * W + x) * C) + q * 4) : 0u;     // always a legal address
    // Which items are zero padding (rows above / below the clip) does not change over the stages of a
    // workgroup: their slots are zeroed ONCE in both buffers here, and afterwards these items - like the ones past the
    // window - write a slot nobody reads.  The staging code then has neither a branch nor a mask.
    const int slot = q * WN_SQ + (ry & 1) * WN_SR + (rx & 1) * WN_SC + (ry >> 1) * RW + (rx >> 1);
    if (lv && !ok) {
      raw0[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
      raw1[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    loff[j] = ok ? slot : WN_RAWBUF - 1 - (tid & 7);
  }
  float4 pv[4];
  auto issue_raw = [&](int st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pv[j] = *reinterpret_cast<const float4*>(p.X + goff[j] + st * 16);
  };
  // ACT: scale / shift of all C input channels sit in LDS (scsh: [C/4] scale quads, then [C/4] shift quads) - a
  // global load inside a filler slot would wait out its whole latency there.  Staged in the prologue BEHIND the first
  // window loads and the first weight loads, so that the three latencies run side by side.
  auto stage_scsh = [&]() {
    if (ACT) {
      for (int i = tid; i < (C >> 2); i += WN_THREADS) {
        scsh[i] = *reinterpret_cast<const float4*>(p.scale + 4 * i);
        scsh[(C >> 2) + i] = *reinterpret_cast<const float4*>(p.shift + 4 * i);
      }
      __syncthreads();
    }
  };
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  auto read_scsh = [&](int st) {
    if (ACT) {
      sc = scsh[st * 4 + q];
      sh = scsh[(C >> 2) + st * 4 + q];
    }
  };
  auto put_raw_part = [&](float4* raw, int j0) {         // items j0, j0 + 1
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j) {
      float4 v = pv[j];
      if (ACT) {
        v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
        v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
      }
      raw[loff[j]] = v;
    }
  };
  auto put_raw = [&](float4* raw) { put_raw_part(raw, 0); put_raw_part(raw, 2); };
  // weight chunk c: 32 KB image [position][k half][column] of float4 (4 k-steps).  The fragments go from L2 / L1 STRAIGHT into
  // the MFMA operand registers: lane (h, li) needs one float4 per position, column half and chunk, at
  // (pos * 128 + h * 64 + nh * 32 + li) - two 512-byte runs per wave instruction.
  const int nchunk = C >> 3;
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.U + ((long)bn * nchunk) * (WN_BCHUNK * 4)), 0, 0x7fffffff, 0x00020000);
  // ---------------------------------------------------------------- this lane's tile and fragment addresses
  int tyl_a, tx_a;
  wino_tile(li, p.tw_shift, tyl_a, tx_a);
  const int abase = tyl_a * RW + tx_a;
  constexpr int RA = XI == 0 ? 0 : XI == 2 ? 2 : 1, RB = XI == 0 ? 2 : XI == 1 ? 2 : XI == 2 ? 1 : 3;    // t = row RA -/+ row RB
  const int rowa = (RA & 1) * WN_SR + (RA >> 1) * RW + abase, rowb = (RB & 1) * WN_SR + (RB >> 1) * RW + abase;
  const int bvoff = (h * 64 + li) * 16;
  // group g (0..7) of chunk c: horizontal position g >> 1 (image position XI * 4 + (g >> 1)), column half g & 1
  auto load_b = [&](int c, int g) {
    typedef unsigned wn_v4u __attribute__((ext_vector_type(4)));
    const wn_v4u v = __builtin_amdgcn_raw_buffer_load_b128(urs, bvoff, c * (WN_BCHUNK * 16) + (XI * 4 + (g >> 1)) * 2048 + (g & 1) * 512, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };

  f32x16 acc[8];             // [horizontal position nu][column half nh] = acc[2 nu + nh]; first written by chunk 0 (C = 0)

  // ---------------------------------------------------------------- main loop
  // Per 8-channel chunk a wavefront issues 8 groups of 4 MFMAs (one accumulator each; a transformed fragment serves the two
  // column halves) and, pinned BETWEEN the groups by sched_barriers, the work that does not need the matrix pipe.  A chunk
  // starts with its operands in registers: the transformed window of its frequency (va / vb by chunk parity; the window reads
  // and the transform of the NEXT chunk - 8 ds_read_b128, 16 v_pk_add_f32 - are software-pipelined into this one) and the
  // weight fragments in a ring of four float4 (slot g & 3 for group g): the slot a group has just used is reloaded with the
  // fragment four groups ahead (>= 1500 cycles of flight for an L2 hit of 300-500).  Round 4: on this part an fp32 MFMA and
  // any other vector instruction of the SIMD add up (a chunk measured 4096 + 2 x 256 cycles = the MFMAs plus the 32 packed adds
  // of each of the SIMD's two wavefronts, whatever else was in the program), so the transform is the cost to cut: one vertical
  // frequency per wavefront and both column halves halves the adds per MFMA (until round 4: two frequencies, one column half).
  // ONE barrier per 16-channel stage (for the window buffers):
  //   raw window of stage s+1: loaded (global -> registers) in chunk (s-1, 0), written to LDS in chunk (s, 0), first read
  //   (prefetch for chunk (s+1, 0)) in chunk (s, 1); its buffer's last readers were the prefetch reads in chunk (s-1, 0);
  //   the barrier at the end of chunk (s, 0) separates both pairs.
  const int nstage = C >> 4;
  const float4* const aq0 = raw0 + h * WN_SQ;        // + 2 * sub * WN_SQ + row + (j & 1) * WN_SC + (j >> 1)
  const float4* const aq1 = raw1 + h * WN_SQ;
  float4 va[4], vb[4], B[4];
  auto read_row = [&](const float4* rq, int row, float4 (&d)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = rq[row + (j & 1) * WN_SC + (j >> 1)];
  };
  auto vertical = [&](float4 (&da)[4], float4 (&db)[4]) {     // into da
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (XI == 1) pk_add_ip(da[j], db[j]); else pk_sub_ip(da[j], db[j]);
    }
  };

#pragma unroll
  for (int g = 0; g < 4; ++g) B[g] = load_b(0, g);
  issue_raw(0);
  stage_scsh();
  read_scsh(0);
  put_raw(raw0);
  issue_raw(nstage > 1 ? 1 : 0);
  __syncthreads();
  {
    float4 da[4], db[4];
    read_row(aq0, rowa, da);
    read_row(aq0, rowb, db);
    vertical(da, db);
    wino_htrans_ip(da, va);
  }
#define WN_SB() __builtin_amdgcn_sched_barrier(0)
  // No run-time condition anywhere in a chunk: past the last stage the staging repeats the last stage (loads that hit L2,
  // stores into a buffer nobody reads any more), past the last chunk the weight ring reloads the last chunk.
  auto step = [&](int c, auto kk, auto first) {
    constexpr int K = decltype(kk)::value;            // c % 4
    constexpr bool FIRST = decltype(first)::value;    // chunk 0: the accumulators start here
    auto mm = [&](f32x16& a_, float4 x_, float4 y_) { if (FIRST) mfma4_first(a_, x_, y_); else mfma4(a_, x_, y_); };
    constexpr int sub = K & 1, sp = K >> 1;
    float4 (&vc)[4] = sub ? vb : va;                   // this chunk's fragments / the next chunk's
    float4 (&vn)[4] = sub ? va : vb;
    const int st = c >> 1;
    float4* const rnxt = sp ? raw0 : raw1;
    // the next chunk's window: the other half of this stage's buffer, or the first half of the next stage's
    const float4* const nq = sub ? (sp ? aq0 : aq1) : (sp ? aq1 : aq0) + 2 * WN_SQ;
    const int cn = c + 1 < nchunk ? c + 1 : c;          // last chunk: a harmless repeat instead of a branch
    float4 da[4], db[4];
    const int st1 = st + 1 < nstage ? st + 1 : nstage - 1, st2 = st + 2 < nstage ? st + 2 : nstage - 1;
    if (sub == 0) read_scsh(st1);
    WN_SB();
    mm(acc[0], vc[0], B[0]);
    WN_SB();
    B[0] = load_b(c, 4);
    // first chunk of a stage: the staging of stage st + 1 (loaded during the previous stage), then the loads of stage st + 2
    if (sub == 0) put_raw_part(rnxt, 0);
    WN_SB();
    mm(acc[1], vc[0], B[1]);
    WN_SB();
    B[1] = load_b(c, 5);
    if (sub == 0) put_raw_part(rnxt, 2);
    WN_SB();
    mm(acc[2], vc[1], B[2]);
    WN_SB();
    B[2] = load_b(c, 6);
    if (sub == 0) issue_raw(st2);                     // stored in chunk (st + 1, 0)
    WN_SB();
    mm(acc[3], vc[1], B[3]);
    WN_SB();
    B[3] = load_b(c, 7);
    read_row(nq, rowa, da);
    WN_SB();
    mm(acc[4], vc[2], B[0]);
    WN_SB();
    B[0] = load_b(cn, 0);
    read_row(nq, rowb, db);
    WN_SB();
    mm(acc[5], vc[2], B[1]);
    WN_SB();
    B[1] = load_b(cn, 1);
    WN_SB();
    mm(acc[6], vc[3], B[2]);
    WN_SB();
    B[2] = load_b(cn, 2);
    vertical(da, db);
    WN_SB();
    mm(acc[7], vc[3], B[3]);
    WN_SB();
    B[3] = load_b(cn, 3);
    wino_htrans_ip(da, vn);
    WN_SB();
    if (sub == 0) __syncthreads();
  };
#undef WN_SB
  using std::integral_constant;
  using std::false_type;
  __builtin_amdgcn_s_setprio(0);
  // the first group of chunks apart (chunk 0's MFMAs take C = 0), then the loop; the number of chunks is even
  step(0, integral_constant<int, 0>(), std::true_type());
  step(1, integral_constant<int, 1>(), false_type());
  if (2 < nchunk) {
    step(2, integral_constant<int, 2>(), false_type());
    step(3, integral_constant<int, 3>(), false_type());
  }
  for (int c = 4; c < nchunk; c += 4) {
    step(c, integral_constant<int, 0>(), false_type());
    step(c + 1, integral_constant<int, 1>(), false_type());
    if (c + 2 < nchunk) {
      step(c + 2, integral_constant<int, 2>(), false_type());
      step(c + 3, integral_constant<int, 3>(), false_type());
    }
  }
  __builtin_amdgcn_s_setprio(WN_PRIO_EDGE);
  __syncthreads();          // the epilogue reuses the window buffers
  // Lane coordinates are taken afresh here (lane id from mbcnt): carried over from the top of the kernel they are live across
  // the main loop, which has no register to spare.
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int li_e = lane_e & 31, h_e = lane_e >> 5;
  const int tid_e = XI * 64 + lane_e;

  // ---------------------------------------------------------------- prefetch for the workgroup that follows on this XCD
  // A workgroup's prologue waits out one HBM round trip for the first stage of its window (2-3 us under load: the tensor was
  // written by the previous kernel and is far larger than the caches).
  // The XCD works through its run of tiles in dispatch order, 32 CUs x 2 workgroups at a time: the workgroup that starts when
  // this one ends is most often tile t + WN_PF_DIST of the run (in-kernel timeline, tools/lab_wino.py full) - and whichever CU of
  // the XCD gets that tile, it reads through the same L2.  So, with its own loads done, every workgroup touches the first
  // 128-byte line of each window pixel of that tile (channels 0-31: stages 0 and 1; one dword load per pixel, the value is
  // discarded at the end of the epilogue): the line is on its way into the XCD's L2 while this epilogue runs.
  // Only where the operand carries no activation: the BatchNorm + ReLU build lost 2-6 % in its main loop with the prefetch in
  // the program - more than the shorter prologue gave back.  With two workgroups per CU (the one's prologue under the other's
  // main loop) the gain is within the noise (profiles/r04_wino_lab.txt); kept, it costs nothing.
  float pfv = 0.f;
  if (!ACT) {
    const int nn_ = (int)gridDim.y;
    const int t2 = tpos + WN_PF_DIST * ((int)gridDim.x * nn_ >= 8 * WN_PF_DIST ? 1 : 0);
    int bm2 = bm, bn2 = bn;
    if (t2 < (int)gridDim.x * nn_) wino_tile_of(p, t2, (int)gridDim.x, nn_, bm2, bn2);
    const int n2 = p.bpc == 1 ? bm2 : (int)__umulhi((unsigned)bm2, p.bpc_magic), ty2 = (bm2 - n2 * p.bpc) * R;
    const int ry = tid_e >> p.w_shift, x = tid_e & (W - 1);
    const int y = 2 * ty2 - 1 + ry;
    if (bm2 < (int)gridDim.x && bm2 != bm && ry < 2 * R + 2 && y >= 0 && y < H)
      pfv = p.X[((long)(n2 * H + y) * W + x) * C];        // default cache policy: the line is to stay in L2
  }

  // ---------------------------------------------------------------- epilogue
  float2* exb = reinterpret_cast<float2*>(raw0);        // [set][16][64 lanes] pairs over the workgroup's whole array, then the sums
  wino_epilogue<XI, STATS>(p, acc, exb, reinterpret_cast<float*>(raw0 + WN_EX_F4),
                           li_e, h_e, lane_e, n, ty0, bm, bn);
  if (!ACT) asm volatile("" :: "v"(pfv));      // the prefetched value is dropped here: the load stays in the program, its register reserved
}

// ACT: the operand carries the previous layer's BatchNorm + ReLU (p.scale / p.shift); STATS: BatchNorm partial sums of the output.
// Three builds: the data gradient (neither), a block's first convolution (statistics), its second one (both).
template <bool ACT, int STATS>
__device__ __forceinline__ void conv_wino_entry(const WinoParams& p, float4* raw, float4* scsh) {
  switch (threadIdx.x >> 6) {
    case 0: conv_wino_body<0, ACT, STATS>(p, raw, raw + WN_RAWBUF, scsh); break;
    case 1: conv_wino_body<1, ACT, STATS>(p, raw, raw + WN_RAWBUF, scsh); break;
    case 2: conv_wino_body<2, ACT, STATS>(p, raw, raw + WN_RAWBUF, scsh); break;
    default: conv_wino_body<3, ACT, STATS>(p, raw, raw + WN_RAWBUF, scsh); break;
  }
}
// (256 threads, 2 wavefronts per SIMD): at most 256 registers per wavefront, so that two workgroups share a CU
__global__ __launch_bounds__(WN_THREADS, 2) void conv_wino_kernel(WinoParams p) {
  __shared__ float4 raw[WN_LDS_F4];
  conv_wino_entry<false, 0>(p, raw, nullptr);
}
__global__ __launch_bounds__(WN_THREADS, 2) void conv_wino_bnred_kernel(WinoParams p) {
  __shared__ float4 raw[WN_LDS_F4];
  conv_wino_entry<false, 2>(p, raw, nullptr);
}
__global__ __launch_bounds__(WN_THREADS, 2) void conv_wino_stats_kernel(WinoParams p) {
  __shared__ float4 raw[WN_LDS_F4];
  conv_wino_entry<false, 1>(p, raw, nullptr);
}
__global__ __launch_bounds__(WN_THREADS, 2) void conv_wino_act_kernel(WinoParams p) {
  __shared__ float4 raw[2 * WN_RAWBUF + 2 * WN_MAXC / 4];
  conv_wino_entry<true, 1>(p, raw, raw + 2 * WN_RAWBUF);
}

// U = G g G^T for every (input channel, output channel) pair, written as the LDS image of the weight chunks:
//   [output block of 64][8-channel chunk][position 16][k half 2][column 64][4 channels]
// dgrad: the data gradient's filter is the forward one flipped and transposed (its input channels are the layer's outputs)
// one (output channel, input-channel quad) item of a layer's image
__device__ __forceinline__ void wino_weights_item(const float* __restrict__ Wt, float4* __restrict__ U, int Cout, int Cin, int dgrad,
                                                  long idx) {
  const int K = dgrad ? Cout : Cin, NO = dgrad ? Cin : Cout;
  {
    const int out = (int)(idx % NO), kq = (int)(idx / NO);
    float u[16][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int in = kq * 4 + jj;
      double g[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
          g[a][b] = dgrad ? (double)Wt[((long)in * Cin + out) * 9 + (2 - a) * 3 + (2 - b)]
                          : (double)Wt[((long)out * Cin + in) * 9 + a * 3 + b];
      double t[4][3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
      }
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        u[xi * 4 + 0][jj] = (float)t[xi][0];
        u[xi * 4 + 1][jj] = (float)(0.5 * (t[xi][0] + t[xi][1] + t[xi][2]));
        u[xi * 4 + 2][jj] = (float)(0.5 * (t[xi][0] - t[xi][1] + t[xi][2]));
        u[xi * 4 + 3][jj] = (float)t[xi][2];
      }
    }
    const int cb = out >> 6, col = out & 63, chunk = kq >> 1, hh = kq & 1;
    float4* dst = U + (((long)cb * (K >> 3) + chunk) * 16 * 2 + hh) * 64 + col;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) dst[(long)pos * 128] = make_float4(u[pos][0], u[pos][1], u[pos][2], u[pos][3]);
  }
}
__global__ void wino_weights_kernel(const float* __restrict__ Wt, float4* __restrict__ U, int Cout, int Cin, int dgrad) {
  const int K = dgrad ? Cout : Cin, NO = dgrad ? Cin : Cout;
  const long total = (long)NO * (K >> 2);
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
    wino_weights_item(Wt, U, Cout, Cin, dgrad, idx);
}
// The images of several layers (forward and / or data gradient) in ONE launch: the encoder builds all of a step's images in
// front of its first convolution instead of one 8-12 us launch in front of each of the 14 convolutions.
__global__ void wino_weights_batch_kernel(acvae::WinoWeightsBatch b) {
  const long total = b.start[b.n];
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < b.n && idx >= b.start[l + 1]) ++l;
    wino_weights_item(b.W[l], reinterpret_cast<float4*>(b.U[l]), b.Cout[l], b.Cin[l], b.dgrad[l], idx - b.start[l]);
  }
}

inline int tile_rows_per_block(int W) { return WN_TILES / (W / 2); }
}  // namespace

namespace acvae {

bool conv3x3_wino_ok(int H, int W, int Cin, int Cout) {
  if (H < 1 || W < 4 || W > 64 || (W & (W - 1)) != 0) return false;      // TW = W/2 in {2,..,32}, a power of two
  return Cin % 16 == 0 && Cin <= WN_MAXC && Cout % WN_TN == 0;
}
long conv3x3_wino_weight_floats(int Cin, int Cout) { return 16L * Cin * Cout; }
int conv_wino_partials_rows(int N, int H, int W) { return N * cdiv(cdiv(H, 2), tile_rows_per_block(W)); }

int conv3x3_wino_weights(const float* W_oihw, float* U, int Cout, int Cin, bool dgrad, hipStream_t st) {
  if (!W_oihw || !U) return ACVAE_EINVAL;
  const int K = dgrad ? Cout : Cin, NO = dgrad ? Cin : Cout;
  if (K % 16 != 0 || NO % WN_TN != 0) return ACVAE_EUNSUPPORTED;
  const long total = (long)NO * (K / 4);
  hipLaunchKernelGGL(wino_weights_kernel, dim3(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256)), dim3(256), 0, st, W_oihw,
                     reinterpret_cast<float4*>(U), Cout, Cin, dgrad ? 1 : 0);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

int conv3x3_wino_weights_batch(WinoWeightsBatch& b, hipStream_t st) {
  if (b.n <= 0) return ACVAE_OK;
  b.start[0] = 0;
  for (int l = 0; l < b.n; ++l) {
    if (!b.W[l] || !b.U[l]) return ACVAE_EINVAL;
    const int K = b.dgrad[l] ? b.Cout[l] : b.Cin[l], NO = b.dgrad[l] ? b.Cin[l] : b.Cout[l];
    if (K % 16 != 0 || NO % WN_TN != 0) return ACVAE_EUNSUPPORTED;
    b.start[l + 1] = b.start[l] + (long)NO * (K / 4);
  }
  const long total = b.start[b.n];
  hipLaunchKernelGGL(wino_weights_batch_kernel, dim3(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256)), dim3(256), 0, st, b);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

// Y[N][H][W][Cout] = conv3x3(act(X[N][H][W][Cin])), pad 1; act = relu(x * scale + shift) when scale != nullptr.
// U: conv3x3_wino_weights image for (Cin -> Cout).  partials (nullable): [conv_wino_partials_rows][2][Cout].
int conv3x3_wino(const float* X, const float* scale, const float* shift, const float* U, float* Y, float* partials, int N,
                 int H, int W, int Cin, int Cout, hipStream_t st, const WinoBnReduce* red) {
  if (!X || !U || !Y) return ACVAE_EINVAL;
  if (red && (scale || !partials || !red->Y || !red->scale || !red->shift || !red->mean || !red->invstd)) return ACVAE_EINVAL;
  if (!conv3x3_wino_ok(H, W, Cin, Cout)) return ACVAE_EUNSUPPORTED;
  if ((long)N * H * W * Cin >= (1L << 32)) return ACVAE_EUNSUPPORTED;      // 32-bit element offsets into X
  if (!aligned16(X) || !aligned16(U) || !aligned16(Y) || (scale && (!aligned16(scale) || !aligned16(shift)))) return ACVAE_EALIGN;
  WinoParams p;
  p.X = X; p.scale = scale; p.shift = shift; p.U = U; p.Y = Y; p.partials = partials;
  p.rY = red ? red->Y : nullptr; p.rscale = red ? red->scale : nullptr; p.rshift = red ? red->shift : nullptr;
  p.rmean = red ? red->mean : nullptr; p.rinvstd = red ? red->invstd : nullptr;
  p.N = N; p.H = H; p.W = W; p.C = Cin; p.Cout = Cout;
  const int TW = W / 2;
  p.tw_shift = __builtin_ctz(TW);
  p.R = WN_TILES / TW;
  p.bpc = cdiv(cdiv(H, 2), p.R);
  p.w_shift = p.tw_shift + 1;
  p.bpc_magic = (unsigned)((0x100000000ULL + (unsigned)p.bpc - 1) / (unsigned)p.bpc);   // ceil(2^32 / bpc): exact for x * bpc < 2^32
  const int nn = Cout / WN_TN;
  p.nn_shift = (nn & (nn - 1)) == 0 ? __builtin_ctz(nn) : -1;
  {   // column blocks whose weight images (4096 x Cin bytes each) share 2 MB of an XCD's L2; all of them: the plain order
    int G = 1;
    while (G < 4 && (long)(2 * G) * 4096 * Cin <= (2L << 20)) G *= 2;
    p.bn_group = (G < nn && nn % G == 0 && (long)nn * 4096 * Cin > (2L << 20) && (long)N * p.bpc * nn < (1L << 23)) ? G : 0;
  }
  if ((long)N * p.bpc >= (1L << 20)) return ACVAE_EUNSUPPORTED;
  const dim3 grid(N * p.bpc, Cout / WN_TN);
  prof_begin(ACVAE_PROF_CONV_IGEMM, st);
  if (red) hipLaunchKernelGGL(conv_wino_bnred_kernel, grid, dim3(WN_THREADS), 0, st, p);
  else if (scale) hipLaunchKernelGGL(conv_wino_act_kernel, grid, dim3(WN_THREADS), 0, st, p);
  else if (partials) hipLaunchKernelGGL(conv_wino_stats_kernel, grid, dim3(WN_THREADS), 0, st, p);
  else hipLaunchKernelGGL(conv_wino_kernel, grid, dim3(WN_THREADS), 0, st, p);
  prof_end(ACVAE_PROF_CONV_IGEMM, st);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

}  // namespace acvae

namespace {
using namespace mfma;
// =====================================================================================================================
// Weight gradient as Winograd F(2x2, 3x3):   dW = G^T [ sum_tiles (Bt d B) .* (A dY_tile A^T) ] G
//   d: 4x4 window of act(X) around the tile, dY_tile: the 2x2 output gradients of the tile, A = At^T.
// 16 GEMMs [Cin x tiles] x [tiles x Cout] (K = tiles = pixels / 4), again 16/36 of the implicit GEMM's multiplies.  One
// workgroup owns a 64 x 64 (ci, co) block of all 16 positions (the same 65536 accumulators as the forward kernel) over a
// contiguous range of 16-tile stages; the K split over workgroups goes through fp32 slabs [z][16][Cin][Cout] that
// wino_wgrad_reduce_kernel sums in fixed order (in double) and folds through G^T . G into dW[co][ci][3][3].
//   * both operands are transforms of raw data and are built in registers: per pair of tiles (one MFMA k-step: lanes
//     0-31 take the even tile, 32-63 the odd one) a lane reads its input channel at the 8 window pixels of ITS vertical
//     frequency and the 2 x 4 gradient pixels of both output-channel halves (ds_read_b32, channel-contiguous over lanes) and
//     spends 12-16 adds (6-8 v_pk_add_f32) for 8 MFMAs; all signs of A and Bt are folded into the order of subtractions;
//   * LDS holds the raw window [pixel][64 ci] and the gradients [tile][2x2][64 co]; the channel index is XOR-ed with 32
//     for odd tile columns / odd tiles so that the two lane halves of a read never meet in a bank.
constexpr int WG_XW_PIX = 136;                 // window pixels per stage: 4 x 34 (W >= 32) .. 10 x 10 (W = 8)
constexpr int WG_XW_WORDS = WG_XW_PIX * 64;
constexpr int WG_DY_WORDS = 16 * 4 * 64;

struct WinoWgradParams {
  const float* dY;      // [N][H][W][Cout]
  const float* X;       // [N][H][W][Cin]
  const float* scale;   // nullable: X is used as is
  const float* shift;
  float* slab;          // [Z][16][Cin][Cout]
  int N, H, W, Cin, Cout;
  int st_shift;         // tiles per stage row = 1 << st_shift = min(W/2, 16)
  int RS;               // tile rows per stage = 16 >> st_shift
  int segs;             // stages per tile row (W = 64: 2) else 1
  int spc;              // stages per clip
  int total;            // N * spc
  int per;              // stages per workgroup
};

// SS = log2(tiles per stage row) as a template parameter: every window offset of the eight tile pairs of a stage is then an
// immediate of its ds_read (the issue slots are what the kernel runs out of; the per-pair address arithmetic was a quarter
// of its non-MFMA instructions).
template <int XI, int SS>
__device__ __forceinline__ void wino_wgrad_body(const WinoWgradParams& p, float* xw0, float* xw1, float* dy0, float* dy1) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // the same value as a scalar (for the code behind the main loop)
  const int ah = wave >> 2;                              // ci half of this wavefront; XI = wave & 3: its vertical frequency
  const int z = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;
  const int H = p.H, W = p.W;
  constexpr int STW = 1 << SS, WC = 2 * STW + 2;
  const int WR = 2 * p.RS + 2;
  const int g_begin = z * p.per, g_end = min(p.total, g_begin + p.per);

  // ---------------------------------------------------------------- staging items (window coordinates are per thread)
  const int quad = tid & 15;
  int xyx[5], xlo[5];        // window (row << 8 | column) and LDS word of the thread's items
  unsigned xlive = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int e = tid + 512 * j, wp = e >> 4;
    const bool lv = wp < WR * WC;
    xlive |= (lv ? 1u : 0u) << j;
    const int wy = wp / WC, wx = wp - wy * WC;
    xyx[j] = (wy << 8) | wx;
    xlo[j] = wp * 64 + ((quad * 4) ^ (32 * ((wx >> 1) & 1)));
  }
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool act = p.scale != nullptr;
  if (act) {
    sc = *reinterpret_cast<const float4*>(p.scale + cib * 64 + quad * 4);
    sh = *reinterpret_cast<const float4*>(p.shift + cib * 64 + quad * 4);
  }
  // The gradient tiles go global -> LDS by LDS-DMA (no transform to apply, no registers, no ds_write): a DMA instruction of a
  // wavefront moves one tile = 4 pixels x 64 channels, lane l = (pixel l / 16, 16-byte slot l % 16), and the XOR swizzle of the
  // channel with the tile parity sits on the SOURCE address.  A DMA cannot zero-fill: lanes whose pixel lies below the image
  // (odd H / the clip's last stage) are masked out of the DMA and store zeros themselves - nobody reads that buffer before
  // the next barrier.
  const int WCin = W * p.Cin;
  float4 px[5];
  unsigned xok = 0;
  // stage g = (clip n, stage s of the clip): advanced incrementally (wave-uniform scalars; the two integer divisions per stage
  // were a visible part of the stage's issue time)
  const int seg_shift = p.segs == 2 ? 1 : 0;
  auto issue_dy = [&](int n, int s, float* dys) {
    const int trow = s >> seg_shift, seg = s & (p.segs - 1);
    const int ty0 = trow * p.RS, tx0 = seg * 16;          // first tile of the stage
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = wave * 2 + i, pp = lane >> 4, pq = lane & 15;
      const int y = 2 * (ty0 + (t >> p.st_shift)) + (pp >> 1), x = 2 * (tx0 + (t & (STW - 1))) + (pp & 1);
      float* dst = dys + t * 256;
      if (y < H) {
        const float* src = p.dY + (unsigned)(((n * H + y) * W + x) * p.Cout + cob * 64 + ((pq ^ (8 * (t & 1))) << 2));
        __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
      } else {
        *reinterpret_cast<float4*>(dst + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto issue_x = [&](int n, int s) {
    const int trow = s >> seg_shift, seg = s & (p.segs - 1);
    const int ty0 = trow * p.RS, tx0 = seg * 16;
    // 32-bit element offsets (the launcher refuses tensors of 2^31 elements or more): a wave-uniform base per stage plus a
    // per-item part - no 64-bit multiplies per item
    const int xbase = ((n * H + 2 * ty0 - 1) * W + (2 * tx0 - 1)) * p.Cin + cib * 64 + quad * 4;
    xok = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int wy = xyx[j] >> 8, wx = xyx[j] & 255;
      const int y = 2 * ty0 - 1 + wy, x = 2 * tx0 - 1 + wx;
      const bool ok = ((xlive >> j) & 1u) && y >= 0 && y < H && x >= 0 && x < W;
      xok |= (ok ? 1u : 0u) << j;
      const unsigned off = ok ? (unsigned)(xbase + wy * WCin + wx * p.Cin) : 0u;     // always a legal address
      px[j] = *reinterpret_cast<const float4*>(p.X + off);
    }
  };
  auto put_part = [&](float* xw, int j0, int j1) {
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      float4 v = px[j];
      if (act) {
        v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
        v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
      }
      if (!((xok >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((xlive >> j) & 1u) *reinterpret_cast<float4*>(xw + xlo[j]) = v;
    }
  };

  // ---------------------------------------------------------------- fragment addresses of this lane
  // tile of pair j for this lane half: t = 2j + kh -> tile row (2j) >> SS, tile column ((2j) & (STW-1)) + kh: kh only shifts the
  // window by two pixel columns and flips the channel swizzle (window columns 0,1 carry the tile's own column parity = kh,
  // columns 2,3 the other one)
  // Round 4: a wavefront owns ONE vertical frequency of its 32 input channels and BOTH halves of the 64 output channels (until
  // then: two frequencies, one half).  Its window transform then needs two rows instead of three and is no longer repeated by
  // the wavefront of the other half: 8 packed adds per 8 MFMAs instead of 11 (the SIMD's issue time is what the kernel runs out of).
  const int ci = ah * 32 + li;
  const int xo01 = kh * 128 + (ci ^ (32 * kh)), xo23 = kh * 128 + (ci ^ (32 * (1 - kh)));
  constexpr int rowstride = WC * 64;
  const int bbase0 = kh * 256 + (li ^ (32 * kh)), bbase1 = kh * 256 + ((32 + li) ^ (32 * kh));
  // window rows of the tile's four that frequency XI combines (signs folded as before): 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r3 - r1
  constexpr int RA = XI == 0 ? 0 : 1, RB = XI == 0 ? 2 : XI == 3 ? 3 : 2;

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // Two register sets: the 16 LDS reads of tile pair j + 1 are issued before the 8 MFMAs of pair j (sched_barriers keep
  // them there; left alone the compiler reads each value just in front of the MFMA that takes it).
  struct Frag { float d[2][4]; float y[2][4]; };
  auto load = [&](auto jj, const float* xw, const float* dys, Frag& f) {
    constexpr int j = decltype(jj)::value;
    constexpr int OFF = ((2 * ((2 * j) >> SS)) * WC + 2 * ((2 * j) & (STW - 1))) * 64;
    const float* x01 = xw + xo01 + OFF;
    const float* x23 = xw + xo23 + OFF;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      constexpr int ROW[2] = {RA, RB};
      f.d[a][0] = x01[ROW[a] * rowstride + 0 * 64];
      f.d[a][1] = x01[ROW[a] * rowstride + 1 * 64];
      f.d[a][2] = x23[ROW[a] * rowstride + 2 * 64];
      f.d[a][3] = x23[ROW[a] * rowstride + 3 * 64];
    }
    const float* ya = dys + bbase0 + j * 512;
    const float* yb = dys + bbase1 + j * 512;
    f.y[0][0] = ya[0]; f.y[0][1] = ya[64]; f.y[0][2] = ya[128]; f.y[0][3] = ya[192];
    f.y[1][0] = yb[0]; f.y[1][1] = yb[64]; f.y[1][2] = yb[128]; f.y[1][3] = yb[192];
  };
  // Both operand transforms of a tile pair as ONE block of v_pk_add_f32 (8 for the frequencies 1 and 2, 6 for 0 and 3; the scalar
  // form needs twice as many adds; issue slots are what the kernel runs out of).  Register pairs: TL / TH = the vertical
  // combination of the window, columns 0,1 / 2,3 (overwrites row RA's registers); per output-channel half h: Y0 / Y1 = the
  // gradient rows, R = their vertical combination.  op_sel / op_sel_hi pick the dword of a pair per lane, neg_lo / neg_hi its sign:
  //   A operands (a0,a1) = (t[0] - t[2], t[1] + t[2]),  (a2,a3) = (t[2] - t[1], t[3] - t[1])
  //   B operands (b1,b2) = (r[0] + r[1], r[0] - r[1]),  b0 = r[0], b3 = r[1]
  // The block ends in the wait between a VALU write and an MFMA read of the register that the compiler's hazard recogniser
  // would insert if it could see inside.
#define WG_A01(D, TL, TH) "v_pk_add_f32 " D ", " TL ", " TH " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
#define WG_A23(D, TL, TH) "v_pk_add_f32 " D ", " TH ", " TL " op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define WG_B12(D, R) "v_pk_add_f32 " D ", " R ", " R " op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]\n\t"
#define WG_SUB(D, A, B) "v_pk_add_f32 " D ", " A ", " B " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define WG_ADD(D, A, B) "v_pk_add_f32 " D ", " A ", " B "\n\t"
  auto mm = [&](const Frag& f) {
    f32x2 a01, a23, b12a, b12b, ra, rb;
    f32x2 tl = {f.d[0][0], f.d[0][1]}, th = {f.d[0][2], f.d[0][3]};           // row RA; overwritten by the vertical combination
    const f32x2 ul = {f.d[1][0], f.d[1][1]}, uh = {f.d[1][2], f.d[1][3]};     // row RB
    const f32x2 y0a = {f.y[0][0], f.y[0][1]}, y1a = {f.y[0][2], f.y[0][3]}, y0b = {f.y[1][0], f.y[1][1]}, y1b = {f.y[1][2], f.y[1][3]};
    if constexpr (XI == 0) {                 // t = r0 - r2, r = Y0
      asm(WG_SUB("%4", "%4", "%6") WG_SUB("%5", "%5", "%7")
          WG_A01("%0", "%4", "%5") WG_A23("%1", "%4", "%5") WG_B12("%2", "%8") WG_B12("%3", "%9") "s_nop 1"
          : "=&v"(a01), "=&v"(a23), "=&v"(b12a), "=&v"(b12b), "+v"(tl), "+v"(th) : "v"(ul), "v"(uh), "v"(y0a), "v"(y0b));
      ra = y0a; rb = y0b;
    } else if constexpr (XI == 3) {          // t = r3 - r1, r = Y1
      asm(WG_SUB("%4", "%6", "%4") WG_SUB("%5", "%7", "%5")
          WG_A01("%0", "%4", "%5") WG_A23("%1", "%4", "%5") WG_B12("%2", "%8") WG_B12("%3", "%9") "s_nop 1"
          : "=&v"(a01), "=&v"(a23), "=&v"(b12a), "=&v"(b12b), "+v"(tl), "+v"(th) : "v"(ul), "v"(uh), "v"(y1a), "v"(y1b));
      ra = y1a; rb = y1b;
    } else if constexpr (XI == 1) {          // t = r1 + r2, r = Y0 + Y1
      asm(WG_ADD("%6", "%6", "%8") WG_ADD("%7", "%7", "%9") WG_ADD("%4", "%10", "%11") WG_ADD("%5", "%12", "%13")
          WG_A01("%0", "%6", "%7") WG_A23("%1", "%6", "%7") WG_B12("%2", "%4") WG_B12("%3", "%5") "s_nop 1"
          : "=&v"(a01), "=&v"(a23), "=&v"(b12a), "=&v"(b12b), "=&v"(ra), "=&v"(rb), "+v"(tl), "+v"(th)
          : "v"(ul), "v"(uh), "v"(y0a), "v"(y1a), "v"(y0b), "v"(y1b));
    } else {                                 // t = r2 - r1, r = Y0 - Y1
      asm(WG_SUB("%6", "%8", "%6") WG_SUB("%7", "%9", "%7") WG_SUB("%4", "%10", "%11") WG_SUB("%5", "%12", "%13")
          WG_A01("%0", "%6", "%7") WG_A23("%1", "%6", "%7") WG_B12("%2", "%4") WG_B12("%3", "%5") "s_nop 1"
          : "=&v"(a01), "=&v"(a23), "=&v"(b12a), "=&v"(b12b), "=&v"(ra), "=&v"(rb), "+v"(tl), "+v"(th)
          : "v"(ul), "v"(uh), "v"(y0a), "v"(y1a), "v"(y0b), "v"(y1b));
    }
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[0], ra[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[0], rb[0], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[1], b12a[0], acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[1], b12b[0], acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23[0], b12a[1], acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23[0], b12b[1], acc[5], 0, 0, 0);
    acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23[1], ra[1], acc[6], 0, 0, 0);
    acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23[1], rb[1], acc[7], 0, 0, 0);
  };
#undef WG_A01
#undef WG_A23
#undef WG_B12
#undef WG_SUB
#undef WG_ADD
  // One stage = 8 tile pairs = 8 blocks of 8 MFMAs per wavefront.  Everything that does not need the matrix pipe sits
  // BETWEEN the blocks (sched_barriers pin it there), where the other wavefront of the SIMD covers it with its own MFMAs:
  // the next stage's gradient DMA behind block 0, its window loads behind block 1, the staging of that window (BatchNorm +
  // ReLU + ds_write) behind blocks 4-6, after the loads have had three blocks to land.  (Before, all eight wavefronts issued
  // their loads together in front of the stage's first MFMA and staged together behind its last: 2560 + 1420 cycles of a
  // 15 900-cycle stage with the pipe idle - in-kernel counters of round 2.)
#define WG_SB() __builtin_amdgcn_sched_barrier(0)
  auto compute = [&](const float* xw, const float* dys, bool more, int nn, int ns, float* xw_n, float* dy_n) {
    Frag fa, fb;
    load(std::integral_constant<int, 0>(), xw, dys, fa);
    load(std::integral_constant<int, 1>(), xw, dys, fb); WG_SB(); mm(fa); WG_SB();
    if (more) issue_dy(nn, ns, dy_n);
    WG_SB();
    load(std::integral_constant<int, 2>(), xw, dys, fa); WG_SB(); mm(fb); WG_SB();
    if (more) issue_x(nn, ns);
    WG_SB();
    load(std::integral_constant<int, 3>(), xw, dys, fb); WG_SB(); mm(fa); WG_SB();
    load(std::integral_constant<int, 4>(), xw, dys, fa); WG_SB(); mm(fb); WG_SB();
    load(std::integral_constant<int, 5>(), xw, dys, fb); WG_SB(); mm(fa); WG_SB();
    if (more) put_part(xw_n, 0, 2);
    WG_SB();
    load(std::integral_constant<int, 6>(), xw, dys, fa); WG_SB(); mm(fb); WG_SB();
    if (more) put_part(xw_n, 2, 4);
    WG_SB();
    load(std::integral_constant<int, 7>(), xw, dys, fb); WG_SB(); mm(fa); WG_SB();
    if (more) put_part(xw_n, 4, 5);
    WG_SB();
    mm(fb);
    WG_SB();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA of the next stage has landed (issued seven blocks ago)
    __syncthreads();
  };
#undef WG_SB

  // ---------------------------------------------------------------- main loop: one barrier per 16-tile stage
  if (g_begin < g_end) {
    int n = g_begin / p.spc, sidx = g_begin - n * p.spc;
    issue_dy(n, sidx, dy0);
    issue_x(n, sidx);
    put_part(xw0, 0, 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int g = g_begin; g < g_end; g += 2) {
      if (++sidx == p.spc) { sidx = 0; ++n; }
      compute(xw0, dy0, g + 1 < g_end, n, sidx, xw1, dy1);          // dy1 / xw1: last read at stage g - 1, behind the barrier
      if (g + 1 < g_end) {
        if (++sidx == p.spc) { sidx = 0; ++n; }
        compute(xw1, dy1, g + 2 < g_end, n, sidx, xw0, dy0);
      }
    }
  }

  // ---------------------------------------------------------------- slab: [z][position][ci][co]
  // Lane and wave coordinates are taken afresh here (lane id from mbcnt, wave index as a scalar): carried over from the top of
  // the kernel they would be live across the main loop, which has no register to spare (hipcc spilled `lane & 32`).
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int li_e = lane_e & 31, kh_e = lane_e >> 5;
  const int ah_e = wave_s >> 2;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int pos = XI * 4 + (q >> 1);                 // acc[2 nu + h]: horizontal position nu, output-channel half h
    float* out = p.slab + (((long)z * 16 + pos) * p.Cin + cib * 64 + ah_e * 32) * p.Cout + cob * 64 + (q & 1) * 32 + li_e;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh_e;
      out[(long)row * p.Cout] = acc[q][r];
    }
  }
}

template <int SS>
__device__ __forceinline__ void wino_wgrad_entry(const WinoWgradParams& p, float* xw0, float* xw1, float* dy0, float* dy1) {
  switch ((threadIdx.x >> 6) & 3) {      // (pairing a light frequency - 0 / 3: 6 packed adds - with a heavy one per SIMD: no difference)
    case 0: wino_wgrad_body<0, SS>(p, xw0, xw1, dy0, dy1); break;
    case 1: wino_wgrad_body<1, SS>(p, xw0, xw1, dy0, dy1); break;
    case 2: wino_wgrad_body<2, SS>(p, xw0, xw1, dy0, dy1); break;
    default: wino_wgrad_body<3, SS>(p, xw0, xw1, dy0, dy1); break;
  }
}
#define WINO_WGRAD_KERNEL(SS)                                                                                    \
  __global__ __launch_bounds__(512) void conv_wino_wgrad_kernel_s##SS(WinoWgradParams p) {                     \
    __shared__ float xw0[WG_XW_WORDS], xw1[WG_XW_WORDS];                                                         \
    __shared__ float dy0[WG_DY_WORDS], dy1[WG_DY_WORDS];                                                         \
    wino_wgrad_entry<SS>(p, xw0, xw1, dy0, dy1);                                                                 \
  }
WINO_WGRAD_KERNEL(1) WINO_WGRAD_KERNEL(2) WINO_WGRAD_KERNEL(3) WINO_WGRAD_KERNEL(4)
#undef WINO_WGRAD_KERNEL

// Slab reduction, one launch (round 4; it was two - the z sum and the fold - i.e. seven more kernel boundaries per step):
//   dU[pos][ci][co] = sum_z slab[z][pos][ci][co]      z in fixed order, four partial sums in double, rounded to fp32 once
//   dW[co][ci][a][b] = sum_{xi,nu} G[xi][a] G[nu][b] dU[xi*4+nu][ci][co]      in double, from the rounded dU
// - the same arithmetic as the two kernels it replaces, bit for bit.  A block of 256 threads owns PAIRS (ci, co) pairs and all
// 16 positions: thread = (position group, pair); the 16 z sums of a pair meet in LDS and one thread per pair folds them.
// PAIRS = 16: 64-byte runs per position, 16 threads per pair - the 64 x 64 layer (Z = 256, 67 MB of slabs) still spreads over
// 256 blocks; PAIRS = 64: 256-byte runs, four positions per thread, for the wide layers.
template <int PAIRS>
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, int Z, int Cout,
                                                                int Cin) {
  constexpr int PG = 256 / PAIRS;            // position groups among the threads: 16 or 4
  constexpr int PPT = 16 / PG;               // positions per thread: 1 or 4
  __shared__ float su[16][PAIRS];
  const long total = (long)Cin * Cout, per_z = 16 * total;
  const int lp = threadIdx.x % PAIRS, pg = threadIdx.x / PAIRS;
  const long pair = (long)blockIdx.x * PAIRS + lp;
  if (pair < total) {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int pos = pg * PPT + k;
      const long idx = (long)pos * total + pair;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int z = 0;
      for (; z + 4 <= Z; z += 4) {          // four loads in flight; the order of the additions is fixed all the same
        const float a = slab[(long)z * per_z + idx], b = slab[(long)(z + 1) * per_z + idx];
        const float c = slab[(long)(z + 2) * per_z + idx], d = slab[(long)(z + 3) * per_z + idx];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
      }
      for (; z < Z; ++z) s0 += (double)slab[(long)z * per_z + idx];
      su[pos][lp] = (float)((s0 + s1) + (s2 + s3));
    }
  }
  __syncthreads();
  if (pg != 0 || pair >= total) return;
  // pair index = ci * Cout + co (the slab's [ci][co] order)
  const int co = (int)(pair % Cout), ci = (int)(pair / Cout);
  double u[16];
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) u[pos] = (double)su[pos][lp];
  // t[a][nu] = sum_xi G[xi][a] u[xi][nu];  G^T rows: a=0: (1, .5, .5, 0)  a=1: (0, .5, -.5, 0)  a=2: (0, .5, .5, 1)
  double t[3][4];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    t[0][nu] = u[0 * 4 + nu] + 0.5 * (u[1 * 4 + nu] + u[2 * 4 + nu]);
    t[1][nu] = 0.5 * (u[1 * 4 + nu] - u[2 * 4 + nu]);
    t[2][nu] = 0.5 * (u[1 * 4 + nu] + u[2 * 4 + nu]) + u[3 * 4 + nu];
  }
  float* out = dW + ((long)co * Cin + ci) * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    out[a * 3 + 0] = (float)(t[a][0] + 0.5 * (t[a][1] + t[a][2]));
    out[a * 3 + 1] = (float)(0.5 * (t[a][1] - t[a][2]));
    out[a * 3 + 2] = (float)(0.5 * (t[a][1] + t[a][2]) + t[a][3]);
  }
}

struct WgradPlan { int st_shift, RS, segs, spc, total, Z, per; };
inline WgradPlan wino_wgrad_plan(int N, int H, int W, int Cin, int Cout) {
  WgradPlan g;
  const int TW = W / 2, TH = (H + 1) / 2;
  const int STW = TW < 16 ? TW : 16;
  g.st_shift = __builtin_ctz(STW);
  g.RS = 16 / STW;
  g.segs = TW / STW;
  g.spc = TW >= 16 ? TH * g.segs : (TH + g.RS - 1) / g.RS;
  g.total = N * g.spc;
  const int blocks = (Cin / 64) * (Cout / 64);
  // one workgroup per CU (100 KB of LDS, 8 wavefronts x 128 accumulators): K split so that the grid is one round of 256
  int Z = 256 / blocks;
  if (Z < 1) Z = 1;
  if (Z > g.total) Z = g.total;
  g.per = (g.total + Z - 1) / Z;
  g.Z = (g.total + g.per - 1) / g.per;
  return g;
}
}  // namespace

namespace acvae {
bool conv3x3_wino_wgrad_ok(int H, int W, int Cin, int Cout) {
  if (H < 1 || W < 4 || W > 64 || (W & (W - 1)) != 0) return false;
  return Cin % 64 == 0 && Cout % 64 == 0;
}
// the kernel addresses X and dY with 32-bit element offsets
static bool wino_wgrad_fits(int N, int H, int W, int Cin, int Cout) {
  return (long)N * H * W * (Cin > Cout ? Cin : Cout) < (1L << 31);
}
long conv3x3_wino_wgrad_slab_floats(int N, int H, int W, int Cin, int Cout) {
  if (!conv3x3_wino_wgrad_ok(H, W, Cin, Cout)) return 0;
  return (long)wino_wgrad_plan(N, H, W, Cin, Cout).Z * 16 * Cin * Cout;
}
// dW_oihw[co][ci][3][3] = sum_p dY[p][co] * act(X)[p + tap][ci]
int conv3x3_wino_wgrad(const float* dY, const float* X, const float* scale, const float* shift, float* dW_oihw, float* slab,
                       int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  if (!dY || !X || !dW_oihw || !slab) return ACVAE_EINVAL;
  if (!conv3x3_wino_wgrad_ok(H, W, Cin, Cout) || !wino_wgrad_fits(N, H, W, Cin, Cout)) return ACVAE_EUNSUPPORTED;
  if (!aligned16(dY) || !aligned16(X) || !aligned16(slab) || (scale && (!aligned16(scale) || !aligned16(shift)))) return ACVAE_EALIGN;
  const WgradPlan g = wino_wgrad_plan(N, H, W, Cin, Cout);
  WinoWgradParams p;
  p.dY = dY; p.X = X; p.scale = scale; p.shift = shift; p.slab = slab;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.st_shift = g.st_shift; p.RS = g.RS; p.segs = g.segs; p.spc = g.spc; p.total = g.total; p.per = g.per;
  prof_begin(ACVAE_PROF_CONV_WGRAD, st);
  const dim3 wgrid(g.Z, Cin / 64, Cout / 64);
  switch (g.st_shift) {
    case 1: hipLaunchKernelGGL(conv_wino_wgrad_kernel_s1, wgrid, dim3(512), 0, st, p); break;
    case 2: hipLaunchKernelGGL(conv_wino_wgrad_kernel_s2, wgrid, dim3(512), 0, st, p); break;
    case 3: hipLaunchKernelGGL(conv_wino_wgrad_kernel_s3, wgrid, dim3(512), 0, st, p); break;
    default: hipLaunchKernelGGL(conv_wino_wgrad_kernel_s4, wgrid, dim3(512), 0, st, p); break;
  }
  prof_end(ACVAE_PROF_CONV_WGRAD, st);
  const long total = (long)Cin * Cout;
  if (total < 16 * 1024)        // few pairs (64 x 64, 64 x 128): 16 per block so that the reduction still covers the chip
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel<16>, dim3((unsigned)cdiv(total, 16)), dim3(256), 0, st, slab, dW_oihw, g.Z, Cout, Cin);
  else
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel<64>, dim3((unsigned)cdiv(total, 64)), dim3(256), 0, st, slab, dW_oihw, g.Z, Cout, Cin);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
}  // namespace acvae
