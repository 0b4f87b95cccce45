// Composite driver for A1 Cnn10.forward (models/encoder.py:672-707) and its backward: sequences the
// kernels of conv.hip / gemm.hip on one stream with no host synchronisation.  All memory is supplied by
// the caller: `saved` keeps what the backward needs (raw conv outputs, pooled tensors, BN statistics,
// repacked weights), `scratch` is reusable within a call.
#include <algorithm>
#include "common.h"
#include "conv.h"
#include "../../include/acvae_hip.h"

namespace {

// arch 0: Cnn10 (models/encoder.py:651-707): 4 blocks, every block 2x2-pooled, 512-wide output, time / 16.
// arch 1: Cnn14_16k (models/encoder.py:871-964): 6 blocks up to 2048 channels, block 6 pooled (1,1), time / 32.
constexpr int kMaxBlocks = 6;
constexpr int kArchMask = 0xff;          // low byte: encoder family; ACVAE_ENC_BF16 (0x100): bf16 activation storage
constexpr int kChan[kMaxBlocks + 1] = {1, 64, 128, 256, 512, 1024, 2048};
struct Arch { int blocks; bool pool_last; };
inline bool arch_of(int arch, Arch& a) {
  if (arch & ~(kArchMask | ACVAE_ENC_BF16)) return false;
  if ((arch & kArchMask) == ACVAE_ARCH_CNN10) { a = {4, true}; return true; }
  if ((arch & kArchMask) == ACVAE_ARCH_CNN14_16K) { a = {6, false}; return true; }
  return false;
}
inline bool is_bf16(int arch) { return (arch & ACVAE_ENC_BF16) != 0; }
// fp32 convolutions (forward and data gradient) as Winograd F(2x2,3x3) wherever conv_wino.hip takes the shape
// (ACVAE_CONV_WINO=0: the implicit GEMM of conv.hip everywhere)
inline bool wino_on() {
  static const bool on = !getenv("ACVAE_CONV_WINO") || atoi(getenv("ACVAE_CONV_WINO")) != 0;
  return on;
}
template <class TA>
inline bool use_wino(int H, int W, int Cin, int Cout) {
  return sizeof(TA) == 4 && wino_on() && acvae::conv3x3_wino_ok(H, W, Cin, Cout);
}

struct EncLayout {
  int N, T, F;
  int nb, Cemb;                // blocks; channels of the last block (= width of audio_embeds)
  bool pool[kMaxBlocks + 1];   // block b ends in a 2x2 average pool
  int H[kMaxBlocks + 1], W[kMaxBlocks + 1];    // conv spatial dims of block b (1..nb); [0] = dims after the last block
  long y1[kMaxBlocks + 1], y2[kMaxBlocks + 1], p[kMaxBlocks + 1];     // float offsets into `saved`
  long wf1[kMaxBlocks + 1], wf2[kMaxBlocks + 1];
  long wd1[kMaxBlocks + 1], wd2[kMaxBlocks + 1];   // fp32: Winograd images of the data gradient's filters (built with the forward's)
  long bn[2 * kMaxBlocks + 1];  // each: 4*C floats (scale, shift, mean, invstd); 0 = bn0, 1+2*(b-1)+{0,1} = block b bn1/bn2
  long pooled_in;
  long total;
  // scratch
  long s_partials, s_dya, s_dyb, s_dpa, s_dpb, s_wd, s_slab, s_bnpart, s_c1w, s_c1b, s_dpart, s_total;
};

long align4(long x) { return (x + 63) & ~63L; }

// offsets are in 4-byte units; `esz` = bytes per activation / repacked-weight element (4: fp32, 2: bf16)
int make_layout(int arch, int N, int T, int F, EncLayout& L) {
  Arch A;
  if (!arch_of(arch, A)) return ACVAE_EINVAL;
  const long esz = is_bf16(arch) ? 2 : 4;
  auto units = [esz](long n) { return (n * esz + 3) / 4; };
  const int div = 1 << (A.pool_last ? A.blocks : A.blocks - 1);
  if (N <= 0 || F != 64 || T < div) return ACVAE_EINVAL;
  L.N = N; L.T = T; L.F = F; L.nb = A.blocks; L.Cemb = kChan[A.blocks];
  long off = 0;
  int h = T, w = F;
  long max_act = 0, max_pool = 0, max_part = 0, max_slab = 0, max_bnpart = 0;
  for (int b = 1; b <= L.nb; ++b) {
    L.H[b] = h; L.W[b] = w;
    L.pool[b] = b < L.nb || A.pool_last;
    const long act = (long)N * h * w * kChan[b];
    const long pool = L.pool[b] ? (long)N * (h / 2) * (w / 2) * kChan[b] : act;
    L.y1[b] = off; off = align4(off + units(act));
    L.y2[b] = off; off = align4(off + units(act));
    L.p[b] = off; off = align4(off + units(pool));
    const long wtaps = esz == 4 ? 16 : 9;      // fp32: room for the 16 Winograd positions of conv_wino.hip
    L.wf1[b] = off; off = align4(off + units((long)kChan[b] * wtaps * kChan[b - 1]));
    L.wf2[b] = off; off = align4(off + units((long)kChan[b] * wtaps * kChan[b]));
    // the data gradient's filter images, built with the forward's in one launch and kept for the backward (fp32: Winograd
    // images, 16 positions; bf16: the implicit GEMM's [cin][tap][cout] repack)
    L.wd1[b] = off; off = align4(off + units((long)kChan[b] * wtaps * kChan[b - 1]));
    L.wd2[b] = off; off = align4(off + units((long)kChan[b] * wtaps * kChan[b]));
    if (act > max_act) max_act = act;
    if (pool > max_pool) max_pool = pool;
    const long part = (long)acvae::conv_partials_rows(N, h, w) * 2 * kChan[b];
    if (part > max_part) max_part = part;
    if (w >= 4) {
      const long wpart = (long)acvae::conv_wino_partials_rows(N, h, w) * 2 * kChan[b];
      if (wpart > max_part) max_part = wpart;
    }
    long sl = esz == 2 ? acvae::conv3x3_wgrad_bf16_slab_floats(N, h, w, kChan[b], kChan[b])
                       : acvae::conv3x3_wgrad_slab_floats(N, h, w, kChan[b], kChan[b]);
    if (sl > max_slab) max_slab = sl;
    if (esz == 4) {
      sl = acvae::conv3x3_wino_wgrad_slab_floats(N, h, w, kChan[b], kChan[b]);
      if (sl > max_slab) max_slab = sl;
      if (b > 1) {
        sl = acvae::conv3x3_wino_wgrad_slab_floats(N, h, w, kChan[b - 1], kChan[b]);
        if (sl > max_slab) max_slab = sl;
      }
    }
    if (b > 1) {
      sl = esz == 2 ? acvae::conv3x3_wgrad_bf16_slab_floats(N, h, w, kChan[b - 1], kChan[b])
                    : acvae::conv3x3_wgrad_slab_floats(N, h, w, kChan[b - 1], kChan[b]);
      if (sl > max_slab) max_slab = sl;
    }
    long bp = (long)acvae::bn_bwd_blocks(N, h, w, kChan[b]) * 2 * kChan[b];
    if (esz == 4 && wino_on() && acvae::conv3x3_wino_ok(h, w, kChan[b], kChan[b]))      // the data gradient's fused reduction: its rows
      bp = std::max(bp, (long)acvae::conv_wino_partials_rows(N, h, w) * 2 * kChan[b]);
    if (bp > max_bnpart) max_bnpart = bp;
    if (L.pool[b]) { h /= 2; w /= 2; }
  }
  L.H[0] = h; L.W[0] = w;  // S and F' after the last block
  for (int i = 0; i < 2 * L.nb + 1; ++i) {
    const int C = i == 0 ? 64 : kChan[(i - 1) / 2 + 1];
    L.bn[i] = off; off = align4(off + 4L * C);
  }
  L.pooled_in = off; off = align4(off + (long)N * L.Cemb);
  L.total = off;
  // scratch
  const long c1 = (long)acvae::conv1_first_blocks(N, T) * 128;
  const long b0 = (long)acvae::bn0_partials_rows((long)N * T) * 128;
  if (c1 > max_part) max_part = c1;
  if (b0 > max_part) max_part = b0;
  long s = 0;
  L.s_dpart = s; s = align4(s + 2 * acvae::colsum_scratch_doubles(2 * L.Cemb > 1024 ? 2 * L.Cemb : 1024));
  L.s_partials = s; s = align4(s + max_part);
  L.s_bnpart = s; s = align4(s + max_bnpart);
  L.s_wd = s; s = align4(s + units((long)L.Cemb * (esz == 4 ? 16 : 9) * L.Cemb));
  L.s_slab = s; s = align4(s + max_slab);
  L.s_c1w = s; s = align4(s + (long)acvae::conv1_first_blocks(N, T) * 576);
  L.s_c1b = s; s = align4(s + (long)acvae::conv1_first_blocks(N, T) * 128);
  L.s_dpa = s; s = align4(s + units(max_pool));
  L.s_dpb = s; s = align4(s + units(max_pool));
  L.s_dya = s; s = align4(s + units(max_act));
  L.s_dyb = s; s = align4(s + units(max_act));
  L.s_total = s;
  return ACVAE_OK;
}

// parameter table order == state-dict order of the reference's Cnn10 (see include/acvae_hip.h)
inline int p_bn0(int k) { return k; }                                  // w, b, rm, rv, nbt
inline int p_conv(int b, int which) { return 5 + (b - 1) * 12 + (which - 1); }
inline int p_bn(int b, int which, int k) { return 5 + (b - 1) * 12 + 2 + (which - 1) * 5 + k; }
inline int p_fc_w(int nb) { return 5 + nb * 12; }       // embed_pooled (Cnn10) / fc1 (Cnn14_16k)
inline int p_fc_b(int nb) { return 6 + nb * 12; }

// Y = conv3x3(act(X), W): weights repacked / transformed into `wbuf`, BN partial rows returned in *nparts
template <class TA>
int conv_fwd(const TA* X, const float* scale, const float* shift, const float* W_oihw, float* wbuf, TA* Y, float* partials,
             int N, int H, int W, int Cin, int Cout, int* nparts, hipStream_t st, bool ready = false) {
  if constexpr (sizeof(TA) == 4) {
    if (use_wino<TA>(H, W, Cin, Cout)) {
      if (!ready) ACVAE_TRY(acvae::conv3x3_wino_weights(W_oihw, wbuf, Cout, Cin, false, st));
      *nparts = acvae::conv_wino_partials_rows(N, H, W);
      return acvae::conv3x3_wino(X, scale, shift, wbuf, Y, partials, N, H, W, Cin, Cout, st);
    }
  }
  if (!ready) ACVAE_TRY(acvae::repack_weights<TA>(W_oihw, (TA*)wbuf, nullptr, Cout, Cin, st));
  *nparts = acvae::conv_partials_rows(N, H, W);
  return acvae::conv3x3_igemm(X, scale, shift, (const TA*)wbuf, Y, partials, N, H, W, Cin, Cout, st);
}
// dX = conv3x3(dY, flipped / transposed W) for the layer Cin -> Cout
// red / redpart / red_rows: the Winograd launch also reduces the BatchNorm + ReLU backward that consumes dX (conv.h: WinoBnReduce);
// *red_rows = rows of sums written to redpart, 0 where the path taken does not do it
template <class TA>
int conv_dgrad(const TA* dY, const float* W_oihw, TA* wbuf, TA* dX, int N, int H, int W, int Cin, int Cout, hipStream_t st,
               bool ready = false, const acvae::WinoBnReduce* red = nullptr, float* redpart = nullptr, int* red_rows = nullptr) {
  if (red_rows) *red_rows = 0;
  if constexpr (sizeof(TA) == 4) {
    if (use_wino<TA>(H, W, Cout, Cin)) {
      if (!ready) ACVAE_TRY(acvae::conv3x3_wino_weights(W_oihw, (float*)wbuf, Cout, Cin, true, st));
      const bool fuse = red && redpart && red_rows;
      if (fuse) *red_rows = acvae::conv_wino_partials_rows(N, H, W);
      return acvae::conv3x3_wino(dY, nullptr, nullptr, (const float*)wbuf, dX, fuse ? redpart : nullptr, N, H, W, Cout, Cin, st,
                                 fuse ? red : nullptr);
    }
  }
  if (!ready) ACVAE_TRY(acvae::repack_weights<TA>(W_oihw, nullptr, wbuf, Cout, Cin, st));
  return acvae::conv3x3_igemm(dY, nullptr, nullptr, (const TA*)wbuf, dX, nullptr, N, H, W, Cout, Cin, st);
}

// dW = sum_p dY[p] (x) act(X)[p + tap]
template <class TA>
int conv_wgrad(const TA* dY, const TA* X, const float* scale, const float* shift, float* dW_oihw, float* slab, int N, int H,
               int W, int Cin, int Cout, hipStream_t st) {
  if constexpr (sizeof(TA) == 4) {
    if (wino_on() && acvae::conv3x3_wino_wgrad_ok(H, W, Cin, Cout)) {
      const int r = acvae::conv3x3_wino_wgrad(dY, X, scale, shift, dW_oihw, slab, N, H, W, Cin, Cout, st);
      if (r != ACVAE_EUNSUPPORTED) return r;          // tensors of 2^31 elements or more: the implicit GEMM below
    }
  }
  return acvae::conv3x3_wgrad(dY, X, scale, shift, dW_oihw, slab, N, H, W, Cin, Cout, st);
}

struct BnPtrs { float *scale, *shift, *mean, *invstd; };
inline BnPtrs bn_at(float* saved, const EncLayout& L, int i) {
  const int C = i == 0 ? 64 : kChan[(i - 1) / 2 + 1];
  float* b = saved + L.bn[i];
  return {b, b + C, b + 2 * C, b + 3 * C};
}

inline DropoutSpec dspec(float p, const uint8_t* const* masks, uint64_t seed, int site, int training) {
  DropoutSpec d;
  d.p = training ? p : 0.f;
  d.mask = (masks && training) ? masks[site] : nullptr;
  d.seed = seed; d.site = (uint32_t)site;
  return d;
}

}  // namespace

extern "C" int acvae_encoder_nparams(int arch) {
  Arch A;
  return arch_of(arch, A) ? 7 + A.blocks * 12 : -1;
}
extern "C" int acvae_encoder_out_dims(int arch, int T, int* S, int* C) {
  Arch A;
  if (!arch_of(arch, A) || !S || !C) return ACVAE_EINVAL;
  *S = T >> (A.pool_last ? A.blocks : A.blocks - 1);
  *C = kChan[A.blocks];
  return ACVAE_OK;
}
extern "C" int64_t acvae_encoder_saved_bytes(int arch, int N, int T, int F) {
  EncLayout L;
  if (make_layout(arch, N, T, F, L) != ACVAE_OK) return -1;
  return L.total * (int64_t)sizeof(float);
}
extern "C" int64_t acvae_encoder_scratch_bytes(int arch, int N, int T, int F) {
  EncLayout L;
  if (make_layout(arch, N, T, F, L) != ACVAE_OK) return -1;
  return L.s_total * (int64_t)sizeof(float);
}

namespace {
template <class TA>
int encoder_fwd_t(const void* const* params, const float* feats, float* audio_embeds, float* pooled,
                                 void* saved_v, int64_t saved_bytes, void* scratch_v, int64_t scratch_bytes, int arch,
                                 int N, int T, int F, int training, float p_block, float p_fc, uint64_t seed,
                                 const uint8_t* const* masks, void* stream) {
  EncLayout L;
  ACVAE_TRY(make_layout(arch, N, T, F, L));
  if (!params || !feats || !audio_embeds || !pooled || !saved_v || !scratch_v) return ACVAE_EINVAL;
  if (saved_bytes < L.total * (int64_t)sizeof(float) || scratch_bytes < L.s_total * (int64_t)sizeof(float))
    return ACVAE_EWORKSPACE;
  if (!aligned16(saved_v) || !aligned16(scratch_v) || !aligned16(feats)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* saved = (float*)saved_v;
  float* scratch = (float*)scratch_v;
  float* partials = scratch + L.s_partials;
  double* dpart = (double*)(scratch + L.s_dpart);
  auto P = [&](int i) { return (float*)params[i]; };

  // bn0 over the mel axis (encoder.py:679-681)
  BnPtrs b0 = bn_at(saved, L, 0);
  int nparts = 0;
  ACVAE_TRY(acvae::colsum_tickets_reset(dpart, st));        // the one memset of this call: every reduction's last-arriver tickets
  if (training) ACVAE_TRY(acvae::bn0_stats(feats, partials, (long)N * T, F, &nparts, st));
  ACVAE_TRY(acvae::bn_finalize(partials, nparts, 64, (double)N * T, P(p_bn0(0)), P(p_bn0(1)), P(p_bn0(2)), P(p_bn0(3)),
                               (int64_t*)params[p_bn0(4)], training, b0.scale, b0.shift, b0.mean, b0.invstd, dpart, st));
  // fp32: the Winograd images of every layer that takes that path - forward filters, and in training the data gradient's
  // too (kept in `saved` for the backward) - in ONE launch here instead of a launch in front of each convolution
  bool wready[kMaxBlocks + 1][2] = {};
  if constexpr (sizeof(TA) == 4) {
    acvae::WinoWeightsBatch wb;
    for (int b = 1; b <= L.nb; ++b) {
      const int H = L.H[b], W = L.W[b], C = kChan[b], Cin = kChan[b - 1];
      if (b > 1 && use_wino<TA>(H, W, Cin, C)) {
        wb.add(P(p_conv(b, 1)), saved + L.wf1[b], C, Cin, false);
        if (training && use_wino<TA>(H, W, C, Cin)) wb.add(P(p_conv(b, 1)), saved + L.wd1[b], C, Cin, true);
        wready[b][0] = true;
      }
      if (use_wino<TA>(H, W, C, C)) {
        wb.add(P(p_conv(b, 2)), saved + L.wf2[b], C, C, false);
        if (training) wb.add(P(p_conv(b, 2)), saved + L.wd2[b], C, C, true);
        wready[b][1] = true;
      }
    }
    ACVAE_TRY(acvae::conv3x3_wino_weights_batch(wb, st));
  }
  {   // the layers that run as implicit GEMMs (bf16 storage; fp32 shapes the Winograd kernels do not take): their repacks too
    acvae::RepackBatch rb;
    for (int b = 1; b <= L.nb; ++b) {
      const int C = kChan[b], Cin = kChan[b - 1];
      if (b > 1 && !wready[b][0]) {
        rb.add(P(p_conv(b, 1)), saved + L.wf1[b], C, Cin, false);
        if (training) rb.add(P(p_conv(b, 1)), saved + L.wd1[b], C, Cin, true);
        wready[b][0] = true;
      }
      if (!wready[b][1]) {
        rb.add(P(p_conv(b, 2)), saved + L.wf2[b], C, C, false);
        if (training) rb.add(P(p_conv(b, 2)), saved + L.wd2[b], C, C, true);
        wready[b][1] = true;
      }
    }
    ACVAE_TRY(acvae::repack_weights_batch<TA>(rb, st));
  }
  const TA* x_in = nullptr;
  for (int b = 1; b <= L.nb; ++b) {
    const int H = L.H[b], W = L.W[b], C = kChan[b], Cin = kChan[b - 1];
    const double cnt = (double)N * H * W;
    TA* Y1 = (TA*)(saved + L.y1[b]);
    TA* Y2 = (TA*)(saved + L.y2[b]);
    BnPtrs n1 = bn_at(saved, L, 1 + 2 * (b - 1)), n2 = bn_at(saved, L, 2 + 2 * (b - 1));
    int np1;
    if (b == 1) {
      ACVAE_TRY(acvae::conv1_first_fwd(feats, b0.scale, b0.shift, P(p_conv(1, 1)), Y1, training ? partials : nullptr,
                                       N, T, F, st));
      np1 = acvae::conv1_first_blocks(N, T);
    } else {
      ACVAE_TRY(conv_fwd<TA>(x_in, nullptr, nullptr, P(p_conv(b, 1)), saved + L.wf1[b], Y1, training ? partials : nullptr, N,
                             H, W, Cin, C, &np1, st, wready[b][0]));
    }
    ACVAE_TRY(acvae::bn_finalize(partials, np1, C, cnt, P(p_bn(b, 1, 0)), P(p_bn(b, 1, 1)), P(p_bn(b, 1, 2)),
                                 P(p_bn(b, 1, 3)), (int64_t*)params[p_bn(b, 1, 4)], training, n1.scale, n1.shift,
                                 n1.mean, n1.invstd, dpart, st));
    int np2;
    ACVAE_TRY(conv_fwd<TA>((const TA*)Y1, n1.scale, n1.shift, P(p_conv(b, 2)), saved + L.wf2[b], Y2,
                           training ? partials : nullptr, N, H, W, C, C, &np2, st, wready[b][1]));
    ACVAE_TRY(acvae::bn_finalize(partials, np2, C, cnt, P(p_bn(b, 2, 0)),
                                 P(p_bn(b, 2, 1)), P(p_bn(b, 2, 2)), P(p_bn(b, 2, 3)), (int64_t*)params[p_bn(b, 2, 4)],
                                 training, n2.scale, n2.shift, n2.mean, n2.invstd, dpart, st));
    ACVAE_TRY(acvae::bn_relu_pool<TA>(Y2, n2.scale, n2.shift, (TA*)(saved + L.p[b]), N, H, W, C,
                                      dspec(p_block, masks, seed, b - 1, training), st, L.pool[b]));
    x_in = (const TA*)(saved + L.p[b]);
  }
  const int S = L.H[0], Fp = L.W[0], Ce = L.Cemb;
  ACVAE_TRY(acvae::freq_mean<TA>((const TA*)(saved + L.p[L.nb]), audio_embeds, (long)N * S, Fp, Ce, st));
  // pooled branch (encoder.py:693-698 / :944-950): dropout sites nb, nb+1 after the nb block sites
  float* pin = saved + L.pooled_in;
  ACVAE_TRY(acvae::time_pool(audio_embeds, pin, N, S, Ce, dspec(p_fc, masks, seed, L.nb, training), st));
  ACVAE_TRY(acvae_gemm_nt_dual(pin, Ce, P(p_fc_w(L.nb)), Ce, Ce, nullptr, 0, nullptr, 0, 0, P(p_fc_b(L.nb)), pooled, Ce,
                               N, Ce, 0, st));
  ACVAE_TRY(acvae::relu_dropout(pooled, N * Ce, dspec(p_fc, masks, seed, L.nb + 1, training), st));
  return ACVAE_OK;
}
}  // namespace

extern "C" int acvae_encoder_fwd(const void* const* params, const float* feats, float* audio_embeds, float* pooled,
                                 void* saved_v, int64_t saved_bytes, void* scratch_v, int64_t scratch_bytes, int arch,
                                 int N, int T, int F, int training, float p_block, float p_fc, uint64_t seed,
                                 const uint8_t* const* masks, void* stream) {
  if (is_bf16(arch))
    return encoder_fwd_t<bf16_t>(params, feats, audio_embeds, pooled, saved_v, saved_bytes, scratch_v, scratch_bytes, arch,
                                 N, T, F, training, p_block, p_fc, seed, masks, stream);
  return encoder_fwd_t<float>(params, feats, audio_embeds, pooled, saved_v, saved_bytes, scratch_v, scratch_bytes, arch, N,
                              T, F, training, p_block, p_fc, seed, masks, stream);
}

// The ReLU decisions the backward will take at BN+ReLU site `site` (0 .. 2*blocks-1: block site/2 + 1, bn1 / bn2), read
// from the `saved` buffer of a forward call and written as uint8 [N,C,H,W] (the reference's layout).  Test aid: lets a
// checker evaluate the reference under exactly these decisions instead of tolerating rounding-dependent mask flips.
extern "C" int acvae_encoder_relu_mask(const void* saved_v, int64_t saved_bytes, int arch, int N, int T, int F, int site,
                                       uint8_t* mask_nchw, void* stream) {
  EncLayout L;
  ACVAE_TRY(make_layout(arch, N, T, F, L));
  if (!saved_v || !mask_nchw || site < 0 || site >= 2 * L.nb) return ACVAE_EINVAL;
  if (saved_bytes < L.total * (int64_t)sizeof(float)) return ACVAE_EWORKSPACE;
  float* saved = (float*)saved_v;
  const int b = site / 2 + 1, which = site % 2;
  BnPtrs bn = bn_at(saved, L, 1 + 2 * (b - 1) + which);
  const long off = which ? L.y2[b] : L.y1[b];
  if (is_bf16(arch))
    return acvae::relu_mask<bf16_t>((const bf16_t*)(saved + off), bn.scale, bn.shift, mask_nchw, N, L.H[b], L.W[b], kChan[b],
                                    (hipStream_t)stream);
  return acvae::relu_mask<float>(saved + off, bn.scale, bn.shift, mask_nchw, N, L.H[b], L.W[b], kChan[b], (hipStream_t)stream);
}

namespace {
template <class TA>
int encoder_bwd_t(const void* const* params, void* const* grads, const float* feats,
                                 const float* d_audio_embeds, void* saved_v, int64_t saved_bytes, void* scratch_v,
                                 int64_t scratch_bytes, int arch, int N, int T, int F, int training, float p_block,
                                 uint64_t seed, const uint8_t* const* masks, void* stream, void* block_done,
                                 void* user) {
  EncLayout L;
  ACVAE_TRY(make_layout(arch, N, T, F, L));
  if (!params || !grads || !feats || !d_audio_embeds || !saved_v || !scratch_v) return ACVAE_EINVAL;
  if (saved_bytes < L.total * (int64_t)sizeof(float) || scratch_bytes < L.s_total * (int64_t)sizeof(float))
    return ACVAE_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* saved = (float*)saved_v;
  float* scratch = (float*)scratch_v;
  auto P = [&](int i) { return (float*)params[i]; };
  auto G = [&](int i) { return (float*)grads[i]; };
  TA* dya = (TA*)(scratch + L.s_dya);
  TA* dyb = (TA*)(scratch + L.s_dyb);
  TA* dp_cur = (TA*)(scratch + L.s_dpa);
  TA* dp_nxt = (TA*)(scratch + L.s_dpb);
  TA* wd = (TA*)(scratch + L.s_wd);
  float* slab = scratch + L.s_slab;
  float* bnpart = scratch + L.s_bnpart;
  double* dpart = (double*)(scratch + L.s_dpart);
  const int S = L.H[0], Fp = L.W[0];
  ACVAE_TRY(acvae::colsum_tickets_reset(dpart, st));        // the one memset of this call: every reduction's last-arriver tickets
  ACVAE_TRY(acvae::freq_mean_bwd<TA>(d_audio_embeds, dp_cur, (long)N * S, Fp, L.Cemb, st));
  for (int b = L.nb; b >= 1; --b) {
    const int H = L.H[b], W = L.W[b], C = kChan[b], Cin = kChan[b - 1];
    const TA* Y1 = (const TA*)(saved + L.y1[b]);
    const TA* Y2 = (const TA*)(saved + L.y2[b]);
    BnPtrs n1 = bn_at(saved, L, 1 + 2 * (b - 1)), n2 = bn_at(saved, L, 2 + 2 * (b - 1));
    // conv2 / bn2 / pool / dropout
    ACVAE_TRY(acvae::bn_bwd<TA>(Y2, dp_cur, L.pool[b] ? UP_POOL : UP_DROP, n2.scale, n2.shift, n2.mean, n2.invstd, bnpart, G(p_bn(b, 2, 1)),
                            G(p_bn(b, 2, 0)), dya, dpart, N, H, W, C, dspec(p_block, masks, seed, b - 1, training), st,
                            training != 0));
    ACVAE_TRY(conv_wgrad<TA>((const TA*)dya, Y1, n1.scale, n1.shift, G(p_conv(b, 2)), slab, N, H, W, C, C, st));
    // the data gradient's Winograd images were built by the training forward (same parameters: the optimiser runs after us)
    // (Winograd images where that path runs, implicit-GEMM repacks elsewhere - the forward chose with the same predicate)
    const bool wd_ready = training != 0;
    // ... and the launch leaves the sums of bn1's backward reduction in bnpart (fp32 Winograd path)
    int red_rows = 0;
    acvae::WinoBnReduce red{(const float*)Y1, n1.scale, n1.shift, n1.mean, n1.invstd};
    ACVAE_TRY(conv_dgrad<TA>((const TA*)dya, P(p_conv(b, 2)), wd_ready ? (TA*)(saved + L.wd2[b]) : wd, dyb, N, H, W, C, C, st, wd_ready,
                             sizeof(TA) == 4 ? &red : nullptr, bnpart, &red_rows));
    // conv1 / bn1
    DropoutSpec none{0.f, nullptr, 0, 0};
    ACVAE_TRY(acvae::bn_bwd<TA>(Y1, dyb, UP_PLAIN, n1.scale, n1.shift, n1.mean, n1.invstd, bnpart, G(p_bn(b, 1, 1)),
                            G(p_bn(b, 1, 0)), dya, dpart, N, H, W, C, none, st, training != 0, red_rows));
    if (b > 1) {
      ACVAE_TRY(conv_wgrad<TA>((const TA*)dya, (const TA*)(saved + L.p[b - 1]), nullptr, nullptr, G(p_conv(b, 1)), slab, N, H,
                               W, Cin, C, st));
      // the forward built the data gradient's Winograd image only where BOTH directions take the Winograd path
      const bool fw = use_wino<TA>(H, W, Cin, C), dw = use_wino<TA>(H, W, C, Cin);
      const bool r1 = wd_ready && fw == dw;
      ACVAE_TRY(conv_dgrad<TA>((const TA*)dya, P(p_conv(b, 1)), r1 ? (TA*)(saved + L.wd1[b]) : wd, dp_nxt, N, H, W, Cin, C, st, r1));
      TA* t = dp_cur; dp_cur = dp_nxt; dp_nxt = t;
    } else {
      BnPtrs b0 = bn_at(saved, L, 0);
      ACVAE_TRY(acvae::conv1_first_bwd<TA>(feats, b0.scale, b0.shift, b0.mean, b0.invstd, P(p_conv(1, 1)), (const TA*)dya,
                                       scratch + L.s_c1w, scratch + L.s_c1b, G(p_conv(1, 1)), G(p_bn0(0)), G(p_bn0(1)),
                                       dpart, N, T, F, st));
    }
    if (block_done) ((void (*)(int, void*))block_done)(b, user);
  }
  return ACVAE_OK;
}
}  // namespace

extern "C" int acvae_encoder_bwd_hooked(const void* const* params, void* const* grads, const float* feats,
                                 const float* d_audio_embeds, void* saved_v, int64_t saved_bytes, void* scratch_v,
                                 int64_t scratch_bytes, int arch, int N, int T, int F, int training, float p_block,
                                 uint64_t seed, const uint8_t* const* masks, void* stream, void* block_done,
                                 void* user) {
  if (is_bf16(arch))
    return encoder_bwd_t<bf16_t>(params, grads, feats, d_audio_embeds, saved_v, saved_bytes, scratch_v, scratch_bytes, arch,
                                 N, T, F, training, p_block, seed, masks, stream, block_done, user);
  return encoder_bwd_t<float>(params, grads, feats, d_audio_embeds, saved_v, saved_bytes, scratch_v, scratch_bytes, arch, N,
                              T, F, training, p_block, seed, masks, stream, block_done, user);
}

extern "C" int acvae_encoder_bwd(const void* const* params, void* const* grads, const float* feats,
                                 const float* d_audio_embeds, void* saved_v, int64_t saved_bytes, void* scratch_v,
                                 int64_t scratch_bytes, int arch, int N, int T, int F, int training, float p_block,
                                 uint64_t seed, const uint8_t* const* masks, void* stream) {
  return acvae_encoder_bwd_hooked(params, grads, feats, d_audio_embeds, saved_v, saved_bytes, scratch_v, scratch_bytes,
                                  arch, N, T, F, training, p_block, seed, masks, stream, nullptr, nullptr);
}
