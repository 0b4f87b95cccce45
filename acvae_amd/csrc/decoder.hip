// Composite drivers for the text side of the path, forward and backward, each ONE library call:
//   A2  PosteriorRNN_hybrid.forward            models/text_encoder.py:182-216
//   A4  PriorRNN.forward (per step)            models/text_encoder.py:247-268
//   A5  VAERNNBahdanauAttnDecoder.forward      models/decoder.py:175-203
//   A6  sample_next_word (greedy)              models/word_model.py:173-207
//   A7  Hybrid_VAEModel.stepwise_forward / decode_step / prepare_decoder_input / stepwise_process_step
//                                              models/vae_model.py:700-730,792-869
//   A12 inference twin (greedy, z from the prior)  models/vae_model.py:880-894
// The reference runs ~45 small torch kernels and >=5 host<->device copies per decode step; here the
// host only enqueues kernels (no synchronisation inside a call), the loop-invariant halves of both
// attentions and of both recurrent input projections are hoisted out of the time loop as batched MFMA
// GEMMs whenever the words are known up front (teacher forcing), and the classifier / log-softmax /
// argmax run once over all N*Tc rows.  Buffers are batch-major [N,Tc,*] like the reference's outputs;
// "step t" addresses column t with row stride Tc*C.
#include <cstdlib>
#include "common.h"
#include "conv.h"
#include "rnn.h"
#include "decode_persist.h"
#include "../../include/acvae_hip.h"

namespace {

struct Bump {
  long off = 0;
  long take(long n) { const long o = off; off = (off + n + 63) & ~63L; return o; }
};

// call context: the stream plus the split-K workspace of the skinny GEMM (converts to hipStream_t for everything else)
struct Ctx {
  hipStream_t s;
  float* skws;
  operator hipStream_t() const { return s; }
};
inline int gemm(const float* A, long lda, const float* B, long ldb, const float* bias, float* C, long ldc, int M, int N,
                int K, int acc, const Ctx& st) {
  return acvae_gemm_nt_dual(A, lda, B, ldb, K, nullptr, 0, nullptr, 0, 0, bias, C, ldc, M, N, acc, st.s, st.skws);
}
inline int gemm2(const float* A1, long lda1, const float* B1, long ldb1, int K1, const float* A2, long lda2,
                 const float* B2, long ldb2, int K2, const float* bias, float* C, long ldc, int M, int N, int acc,
                 const Ctx& st) {
  if (M <= 64)
    return acvae_gemm_nt_dual(A1, lda1, B1, ldb1, K1, A2, lda2, B2, ldb2, K2, bias, C, ldc, M, N, acc, st.s, st.skws);
  ACVAE_TRY(acvae_gemm_nt_dual(A1, lda1, B1, ldb1, K1, nullptr, 0, nullptr, 0, 0, bias, C, ldc, M, N, acc, st.s, st.skws));
  return acvae_gemm_nt_dual(A2, lda2, B2, ldb2, K2, nullptr, 0, nullptr, 0, 0, nullptr, C, ldc, M, N, 1, st.s, st.skws);
}
struct TnWs { float* p; long bytes; };
inline int gemm_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K, TnWs ws,
                   hipStream_t st) {
  return acvae_gemm_tn_fused(A, lda, B, ldb, C, ldc, M, N, K, 0, ws.p, ws.bytes, st);     // ws: tickets (zeroed at the call's entry) | slabs
}
// the regions every composite call zeroes in front of its first kernel, in ONE launch (rnn.h)
inline void zb_skinny(acvae::ZeroBatch& zb, float* skws) { zb.add(skws, acvae_skinny_ticket_words()); }
inline void zb_colsum(acvae::ZeroBatch& zb, double* dpart) { zb.add(dpart, acvae::colsum_ticket_words()); }
inline void zb_tn(acvae::ZeroBatch& zb, TnWs ws) { zb.add(ws.p, TN_TICKETS); }
inline int transp(const float* in, long ld_in, float* out, long ld_out, int rows, int cols, hipStream_t st) {
  return acvae_transpose(in, ld_in, out, ld_out, rows, cols, st);
}
inline int zero(float* p, long n, hipStream_t st) {
  return hipMemsetAsync(p, 0, (size_t)n * sizeof(float), st) == hipSuccess ? ACVAE_OK : (int)hipGetLastError();
}
// fork/join of a call's work over two streams: begin() makes aux wait for everything queued on main so far, join() makes
// main wait for everything queued on aux; with aux == main both are no-ops.  Events are released when they complete.
struct Fork {
  hipStream_t main_s, aux;
  bool on() const { return aux != main_s; }
  // Events come from a small ring that is never destroyed: hipEventDestroy on an event that has not completed yet may
  // hold the host until it has, which would serialise the host with the GPU at every fork.  The ring belongs to the
  // CALLING THREAD and to its current device (thread_local, re-made when the thread switches devices), so concurrent
  // calls from several host threads - the autograd thread beside the main thread, SURVEY §8(b) - share nothing; an
  // event re-recorded while an older wait on it is still queued is legal (a wait captures the record that precedes it).
  static int edge(hipStream_t from, hipStream_t to) {
    constexpr int RING = 64;
    struct Ring { hipEvent_t ev[RING]; int next; int dev; };
    static thread_local Ring ring = {{}, 0, -1};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    if (dev != ring.dev) {                      // first use on this thread, or the thread moved to another device
      for (int i = 0; i < RING; ++i) ring.ev[i] = nullptr;   // the old device's events are left to its context
      ring.next = 0; ring.dev = dev;
    }
    hipEvent_t& e = ring.ev[ring.next];
    ring.next = (ring.next + 1) % RING;
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    hipError_t r = hipEventRecord(e, from);
    if (r == hipSuccess) r = hipStreamWaitEvent(to, e, 0);
    return r == hipSuccess ? ACVAE_OK : (int)r;
  }
  int begin() const { return on() ? edge(main_s, aux) : ACVAE_OK; }
  int join() const { return on() ? edge(aux, main_s) : ACVAE_OK; }
};
inline long tn_ws_floats(int M, int N, int K) { return acvae_gemm_tn_workspace_bytes(M, N, K) / 4 + 64 + TN_TICKETS; }

// text-parameter table (state-dict order after the encoder; see include/acvae_hip.h)
enum {
  TP_DEC_EMB, TP_DEC_WIH, TP_DEC_WHH, TP_DEC_BIH, TP_DEC_BHH, TP_DEC_CLS_W, TP_DEC_CLS_B, TP_DEC_ATT_V, TP_DEC_ATT_W,
  TP_DEC_ATT_B, TP_Q_EMB, TP_Q_WIH, TP_Q_WHH, TP_Q_BIH, TP_Q_BHH, TP_Q_WIH_R, TP_Q_WHH_R, TP_Q_BIH_R, TP_Q_BHH_R,
  TP_Q_TML_W, TP_Q_TML_B, TP_P_EMB, TP_P_ATT_V, TP_P_ATT_W, TP_P_ATT_B, TP_P_WIH, TP_P_WHH, TP_P_BIH, TP_P_BHH,
  TP_P_ML_W, TP_P_ML_B, TP_MLO_W, TP_MLO_B, TP_LN_W, TP_LN_B, TP_COUNT
};
static_assert(TP_COUNT == ACVAE_TEXT_NPARAMS, "text parameter table out of sync with the header");

// ------------------------------------------------------------------------------------------ posterior
struct PostLayout {
  // saved
  long words, x, hidden, save_f, save_r, hprev_f, hprev_r, argmax, saved_total;
  // scratch
  long gi_f, gi_r, gh, hf, ml, dml, dhid, dgi_f, dgi_r, dgh_f, dgh_r, dh_a, dh_b, dx, wt, wt2, wt_ih0, wt_ih1, wt_ml, pq_hbuf, pq_cnt, tn, dpart, skws,
       scratch_total;
  long tn_floats, dpart_doubles;
};
int post_layout(int N, int Tc, int E, int Hq, int V, PostLayout& L) {
  if (N <= 0 || Tc <= 0 || E <= 0 || Hq <= 0 || V <= 0) return ACVAE_EINVAL;
  const long R = (long)N * Tc;
  Bump s;
  L.words = s.take(R * 2);  // int64
  L.x = s.take(R * E);
  L.hidden = s.take(R * 2 * Hq);
  L.save_f = s.take(R * 4 * Hq); L.save_r = s.take(R * 4 * Hq);
  L.hprev_f = s.take(R * Hq); L.hprev_r = s.take(R * Hq);
  L.argmax = s.take((long)N * 2 * Hq);
  L.saved_total = s.off;
  Bump c;
  L.gi_f = c.take(R * 3 * Hq); L.gi_r = c.take(R * 3 * Hq);
  L.gh = c.take((long)N * 3 * Hq); L.hf = c.take((long)N * Hq);
  L.ml = c.take(R * 2 * E); L.dml = c.take(R * 2 * E); L.dhid = c.take(R * 2 * Hq);
  L.dgi_f = c.take(R * 3 * Hq); L.dgi_r = c.take(R * 3 * Hq); L.dgh_f = c.take(R * 3 * Hq); L.dgh_r = c.take(R * 3 * Hq);
  L.dh_a = c.take((long)N * Hq); L.dh_b = c.take((long)N * Hq);
  L.dx = c.take(R * E);
  long wt = (long)2 * E * 2 * Hq;
  if ((long)3 * Hq * Hq > wt) wt = (long)3 * Hq * Hq;
  if ((long)3 * Hq * E > wt) wt = (long)3 * Hq * E;
  L.wt = c.take(wt);
  L.wt2 = c.take((long)3 * Hq * Hq);                        // persistent BPTT: both directions' transposed weight_hh at once
  // the backward's other transposed weights (round 4: all five transposes of the call in one launch in front of it)
  L.wt_ih0 = c.take((long)3 * Hq * E); L.wt_ih1 = c.take((long)3 * Hq * E); L.wt_ml = c.take((long)2 * E * 2 * Hq);
  L.pq_hbuf = c.take((long)4 * N * Hq);                     // persistent forward: h in flight, [direction][parity][N][Hq]
  L.pq_cnt = c.take(acvae::posterior_persist_counter_words(Tc));
  long tn = tn_ws_floats(2 * E, 2 * Hq, (int)R);
  long t2 = tn_ws_floats(3 * Hq, E, (int)R); if (t2 > tn) tn = t2;
  t2 = tn_ws_floats(3 * Hq, Hq, (int)R); if (t2 > tn) tn = t2;
  L.tn_floats = tn;
  L.tn = c.take(tn);
  L.dpart_doubles = 4 * acvae::colsum_scratch_doubles(2 * E > 3 * Hq ? 2 * E : 3 * Hq);      // room for the batched column sums
  L.dpart = c.take(2 * L.dpart_doubles);
  L.skws = c.take(acvae_skinny_ws_floats());
  L.scratch_total = c.off;
  return ACVAE_OK;
}

// ------------------------------------------------------------------------------------------ decode
struct DecLayout {
  // saved
  long words, mem, encproj_d, encproj_p, qd, qp, attw_p, rnn_d, rnn_p, gru_save, hprev_d, lstm_save, c_all, hp_all,
      hpprev, lse, pool_arg, pool_hid, unfinished, saved_total;
  // saved as well (round 4): the transposed weights of the backward's dX = dY . W products, made by the FORWARD on its second
  // stream beside the persistent launch instead of in front of the BPTT on the backward's critical path
  long wt_cls, wt_dih, wt_dhh, wt_datt, wt_pih, wt_phh, wt_pml, wt_patt, wt_mlo;
  // fwd scratch
  long skws, skws_p, gi_d, gh_d, gates_p, ml, h0, pd_cnt, attfws_d, attfws_p, attfws_bytes, scratch_fwd;
  // bwd scratch
  long wt_ln;
  long d_out, dgi, dgh, dqd, dencproj, dvpart, dctx, dh_a, dh_b, dgates, dml_all, dml, dhp_a, dhp_b, dc_a, dc_b, dlz_a, pd_part,
      dlz_b, drnn, dz_dec, dqp, dmem, dhid, words_c, tn, dpart, attws, pd_cnt_b, pd_dqd, scratch_bwd;
  // bwd scratch private to the prior chain (it may run on the second stream)
  long dencproj_p, dvpart_p, dmem_p, drnn_p, tn_p, dpart_p, attws_p;
  long attws_bytes;
  long tn_floats, tn_p_floats, dpart_doubles;
};
int dec_layout(int N, int Tc, int S, int E, int H, int A, int V, int Eenc, DecLayout& L) {
  if (N <= 0 || Tc <= 0 || S <= 0 || E <= 0 || H <= 0 || A <= 0 || V <= 1 || Eenc <= 0) return ACVAE_EINVAL;
  const long R = (long)N * Tc;
  const int Hp = E;
  Bump s;
  L.words = s.take(R * 2);
  L.mem = s.take((long)N * S * E);
  L.encproj_d = s.take((long)N * S * A); L.encproj_p = s.take((long)N * S * E);
  L.qd = s.take(R * A); L.qp = s.take(R * E); L.attw_p = s.take(R * S);
  L.rnn_d = s.take(R * 3 * E); L.rnn_p = s.take(R * 3 * E);
  L.gru_save = s.take(R * 4 * H); L.hprev_d = s.take(R * H);
  L.lstm_save = s.take(R * 5 * Hp); L.c_all = s.take(R * Hp); L.hp_all = s.take(R * Hp); L.hpprev = s.take(R * Hp);
  L.lse = s.take(R); L.pool_arg = s.take((long)N * H); L.pool_hid = s.take((long)N * H);
  L.unfinished = s.take(N);
  L.wt_cls = s.take((long)H * V + 64); L.wt_dih = s.take((long)3 * E * 3 * H); L.wt_dhh = s.take((long)H * 3 * H);
  L.wt_datt = s.take((long)(E + H) * A); L.wt_pih = s.take((long)3 * E * 4 * Hp); L.wt_phh = s.take((long)Hp * 4 * Hp);
  L.wt_pml = s.take((long)Hp * 2 * E); L.wt_patt = s.take((long)2 * E * E); L.wt_mlo = s.take((long)H * 2 * E);
  L.saved_total = s.off;
  Bump f;
  L.skws = f.take(acvae_skinny_ws_floats());   // same offsets in the forward and backward scratch maps
  L.skws_p = f.take(acvae_skinny_ws_floats());
  L.gi_d = f.take(R * 3 * H); L.gh_d = f.take((long)N * 3 * H); L.gates_p = f.take(R * 4 * Hp);
  L.ml = f.take((long)N * 2 * E); L.h0 = f.take((long)N * (H > Hp ? H : Hp));
  L.pd_cnt = f.take(acvae::decode_persist_counter_words(Tc));      // arrival counters of the persistent decode loop
  // per-step path: workspaces of the split-over-frames attention, one per chain (the chains may run on two streams)
  L.attfws_bytes = acvae_attn_fwd_workspace_bytes(N, 1, S, A, E);
  { const long w2 = acvae_attn_fwd_workspace_bytes(N, 1, S, E, E); if (w2 > L.attfws_bytes) L.attfws_bytes = w2; }
  L.attfws_d = f.take(L.attfws_bytes / 4 + 64); L.attfws_p = f.take(L.attfws_bytes / 4 + 64);
  L.scratch_fwd = f.off;
  Bump b;
  b.take(acvae_skinny_ws_floats());            // skinny split-K workspaces first (L.skws, L.skws_p)
  b.take(acvae_skinny_ws_floats());
  L.wt_ln = b.take((long)Eenc * E);
  L.d_out = b.take(R * H); L.dgi = b.take(R * 3 * H); L.dgh = b.take(R * 3 * H); L.dqd = b.take(R * A);
  L.dencproj = b.take((long)N * S * (A > E ? A : E));
  L.dvpart = b.take((long)3 * N * (A > E ? A : E));             // up to three frame shares per clip (persistent BPTT, S > 64)
  L.dctx = b.take((long)N * E); L.dh_a = b.take((long)N * H); L.dh_b = b.take((long)N * H);
  L.dgates = b.take(R * 4 * Hp); L.dml_all = b.take(R * 2 * E); L.dml = b.take((long)N * 2 * E);
  L.dhp_a = b.take((long)N * Hp); L.dhp_b = b.take((long)N * Hp); L.dc_a = b.take((long)N * Hp);
  L.dc_b = b.take((long)N * Hp); L.dlz_a = b.take((long)N * E); L.dlz_b = b.take((long)N * E);
  L.drnn = b.take(R * 3 * E); L.dz_dec = b.take(R * E); L.dqp = b.take(R * E);
  L.dmem = b.take((long)N * S * E); L.dhid = b.take((long)N * H); L.words_c = b.take(R * 2);
  long tn = 0;
  auto mx = [&](int M, int Nn, int K) { const long t = tn_ws_floats(M, Nn, K); if (t > tn) tn = t; };
  mx(V, H, (int)R); mx(3 * H, 3 * E, (int)R); mx(3 * H, H, (int)R); mx(A, H, (int)R); mx(A, E, N * S); mx(E, E, N * S);
  mx(4 * Hp, 3 * E, (int)R); mx(4 * Hp, Hp, (int)R); mx(2 * E, Hp, (int)R); mx(E, E, (int)R); mx(2 * E, H, N);
  mx(E, Eenc, N * S);
  L.tn_floats = tn;
  L.tn = b.take(tn);
  {
    int w = V; if (4 * Hp > w) w = 4 * Hp; if (3 * H > w) w = 3 * H; if (2 * E > w) w = 2 * E; if (A > w) w = A;
    L.dpart_doubles = acvae::colsum_scratch_doubles(w);
    L.dpart = b.take(2 * L.dpart_doubles);
  }
  L.attws_bytes = acvae_attn_bwd_workspace_bytes(N, Tc, S, A > E ? A : E);
  L.attws = b.take(L.attws_bytes / 4 + 64);
  L.attws_p = b.take(L.attws_bytes / 4 + 64);
  L.dencproj_p = b.take((long)N * S * E); L.dvpart_p = b.take((long)N * E); L.dmem_p = b.take((long)N * S * E);
  L.drnn_p = b.take(R * 3 * E);
  // the second stream's own split-K slab and column-sum scratch: as large as the first's, because the deferred
  // parameter-gradient products of the decoder run there too (acvae_decode_bwd)
  L.tn_p_floats = L.tn_floats;
  L.tn_p = b.take(L.tn_p_floats);
  {
    int w = V; if (4 * Hp > w) w = 4 * Hp; if (3 * H > w) w = 3 * H; if (2 * E > w) w = 2 * E; if (A > w) w = A;
    L.dpart_p = b.take(2 * acvae::colsum_scratch_doubles(w));
  }
  L.pd_cnt_b = b.take(acvae::decode_persist_bwd_counter_words(Tc));   // arrival counters of the persistent BPTT launch
  L.pd_part = b.take(acvae::decode_persist_bwd_part_floats(N, E, H));  // its K-split partial tiles
  L.pd_dqd = b.take(S > 64 && S <= 192 ? (long)acvae::decode_persist_bwd_rc_splits(S) * R * A : 0);   // d qd shares of the split attention role
  L.scratch_bwd = b.off;
  return ACVAE_OK;
}

}  // namespace

// ==========================================================================================
// posterior
// ==========================================================================================
extern "C" int64_t acvae_posterior_saved_bytes(int N, int Tc, int E, int Hq, int V) {
  PostLayout L;
  return post_layout(N, Tc, E, Hq, V, L) == ACVAE_OK ? L.saved_total * 4 : -1;
}
extern "C" int64_t acvae_posterior_scratch_bytes(int N, int Tc, int E, int Hq, int V) {
  PostLayout L;
  return post_layout(N, Tc, E, Hq, V, L) == ACVAE_OK ? L.scratch_total * 4 : -1;
}

extern "C" int acvae_posterior_fwd(const void* const* params, const int64_t* caps, int64_t ld_caps, const int64_t* lens1,
                                   const float* eps_q, float* q_means, float* q_logs, float* q_z, float* q_means_utt,
                                   void* saved_v, int64_t saved_bytes, void* scratch_v, int64_t scratch_bytes, int N,
                                   int Tc, int E, int Hq, int V, void* stream, int flags) {
  PostLayout L;
  ACVAE_TRY(post_layout(N, Tc, E, Hq, V, L));
  if (!params || !caps || !lens1 || !eps_q || !q_means || !q_logs || !q_z || !q_means_utt || !saved_v || !scratch_v)
    return ACVAE_EINVAL;
  if (saved_bytes < L.saved_total * 4 || scratch_bytes < L.scratch_total * 4) return ACVAE_EWORKSPACE;
  float* sv = (float*)saved_v;
  float* sc = (float*)scratch_v;
  Ctx st{(hipStream_t)stream, sc + L.skws};
  const bool persist_q = !(flags & ACVAE_FLAG_NO_PERSIST) && acvae::posterior_persist_ok(N, Tc, Hq);
  {
    acvae::ZeroBatch zb;
    zb_skinny(zb, st.skws);
    if (persist_q) {
      zb.add(sc + L.pq_cnt, acvae::posterior_persist_counter_words(Tc));
      zb.add(sc + L.pq_hbuf, (long)4 * N * Hq);
      flags |= ACVAE_FLAG_INT_CNT_ZEROED;
    }
    ACVAE_TRY(acvae::zero_batch(zb, st.s));
  }
  auto P = [&](int i) { return (const float*)params[i]; };
  const int R = N * Tc;
  int64_t* words = (int64_t*)(sv + L.words);
  float* X = sv + L.x;
  float* hid = sv + L.hidden;
  ACVAE_TRY(acvae::gather_words(caps, ld_caps, 1, words, N, Tc, st));            // x[:, :-1] restricted to Tc steps
  ACVAE_TRY(acvae::embed_gather(words, 1, P(TP_Q_EMB), V, X, E, R, E, st));
  if (persist_q) {
    // both directions, all steps: one launch (decode_persist.hip); the hoisted input projections first
    PqParams pq;
    for (int dir = 0; dir < 2; ++dir) {
      const int o = dir * 4;
      float* gi = sc + (dir ? L.gi_r : L.gi_f);
      ACVAE_TRY(gemm(X, E, P(TP_Q_WIH + o), E, P(TP_Q_BIH + o), gi, 3 * Hq, R, 3 * Hq, E, 0, st));
      pq.w_hh[dir] = P(TP_Q_WHH + o); pq.b_hh[dir] = P(TP_Q_BHH + o); pq.gi[dir] = gi;
      pq.save[dir] = sv + (dir ? L.save_r : L.save_f);
      pq.hprev[dir] = sv + (dir ? L.hprev_r : L.hprev_f);
    }
    pq.lens1 = lens1; pq.hid = hid; pq.hbuf = sc + L.pq_hbuf; pq.cnt = (unsigned*)(sc + L.pq_cnt);
    pq.N = N; pq.Tc = Tc; pq.Hq = Hq;
    ACVAE_TRY(acvae::posterior_persist_fwd(pq, st.s, flags));          // hbuf and the counters: zeroed at the entry
  } else
  for (int dir = 0; dir < 2; ++dir) {
    const int o = dir * 4;
    float* gi = sc + (dir ? L.gi_r : L.gi_f);
    float* save = sv + (dir ? L.save_r : L.save_f);
    float* hprev = sv + (dir ? L.hprev_r : L.hprev_f);
    float* h = sc + L.hf;
    float* gh = sc + L.gh;
    ACVAE_TRY(gemm(X, E, P(TP_Q_WIH + o), E, P(TP_Q_BIH + o), gi, 3 * Hq, R, 3 * Hq, E, 0, st));
    ACVAE_TRY(acvae::copy_rows(h, Hq, nullptr, 0, N, Hq, st));
    for (int k = 0; k < Tc; ++k) {
      const int t = dir ? Tc - 1 - k : k;
      ACVAE_TRY(gemm(h, Hq, P(TP_Q_WHH + o), Hq, P(TP_Q_BHH + o), gh, 3 * Hq, N, 3 * Hq, Hq, 0, st));
      ACVAE_TRY(acvae::gru_fwd(gi + (long)t * 3 * Hq, (long)Tc * 3 * Hq, gh, 3 * Hq, h, Hq, h, Hq,
                               hid + (long)t * 2 * Hq + dir * Hq, (long)Tc * 2 * Hq, save + (long)t * 4 * Hq,
                               (long)Tc * 4 * Hq, hprev + (long)t * Hq, (long)Tc * Hq, lens1, t, N, Hq, st));
    }
  }
  float* ml = sc + L.ml;
  ACVAE_TRY(gemm(hid, 2 * Hq, P(TP_Q_TML_W), 2 * Hq, P(TP_Q_TML_B), ml, 2 * E, R, 2 * E, 2 * Hq, 0, st));
  ACVAE_TRY(acvae_reparam_fwd(ml, 2 * E, eps_q, E, q_means, q_logs, q_z, E, nullptr, 0, R, E, st));
  ACVAE_TRY(acvae::pool_fwd(hid, (long)Tc * 2 * Hq, 2 * Hq, lens1, q_means_utt, (int*)(sv + L.argmax), N, Tc, 2 * Hq, st));
  return ACVAE_OK;
}

extern "C" int acvae_posterior_bwd(const void* const* params, void* const* grads, const int64_t* lens1,
                                   const float* eps_q, const float* q_logs, const float* d_q_means,
                                   const float* d_q_logs, const float* d_q_z, const float* d_q_means_utt, void* saved_v,
                                   int64_t saved_bytes, void* scratch_v, int64_t scratch_bytes, int N, int Tc, int E,
                                   int Hq, int V, void* stream, int flags) {
  PostLayout L;
  ACVAE_TRY(post_layout(N, Tc, E, Hq, V, L));
  if (!params || !grads || !lens1 || !eps_q || !q_logs || !saved_v || !scratch_v) return ACVAE_EINVAL;
  if (saved_bytes < L.saved_total * 4 || scratch_bytes < L.scratch_total * 4) return ACVAE_EWORKSPACE;
  float* sv = (float*)saved_v;
  float* sc = (float*)scratch_v;
  Ctx st{(hipStream_t)stream, sc + L.skws};
  auto P = [&](int i) { return (const float*)params[i]; };
  auto G = [&](int i) { return (float*)grads[i]; };
  const int R = N * Tc;
  TnWs tn{sc + L.tn, L.tn_floats * 4};
  double* dpart = (double*)(sc + L.dpart);
  const bool persist = !(flags & ACVAE_FLAG_NO_PERSIST) && acvae::posterior_persist_ok(N, Tc, Hq);
  {
    acvae::ZeroBatch zb;
    zb_skinny(zb, st.skws); zb_colsum(zb, dpart); zb_tn(zb, tn);
    if (persist) { zb.add(sc + L.pq_cnt, acvae::posterior_persist_counter_words(Tc)); flags |= ACVAE_FLAG_INT_CNT_ZEROED; }
    zb.add(G(TP_Q_EMB), (long)V * E);          // the embedding-table gradient starts from zero (embed_scatter adds rows)
    ACVAE_TRY(acvae::zero_batch(zb, st.s));
  }
  const int64_t* words = (const int64_t*)(sv + L.words);
  float* X = sv + L.x;
  float* hid = sv + L.hidden;
  float* dml = sc + L.dml;
  float* dhid = sc + L.dhid;
  float* wt = sc + L.wt;
  float* wt_ml = sc + L.wt_ml;
  float* wt_ih[2] = {sc + L.wt_ih0, sc + L.wt_ih1};
  {   // the transposed weights of the dX = dY . W products: one launch (they depend on the parameters only)
    TransposeBatch tb;
    tb.add(P(TP_Q_TML_W), 2 * Hq, wt_ml, 2 * E, 2 * E, 2 * Hq);                             // [2Hq][2E]
    for (int dir = 0; dir < 2; ++dir) {
      tb.add(P(TP_Q_WIH + dir * 4), E, wt_ih[dir], 3 * Hq, 3 * Hq, E);                      // [E][3Hq]
      if (persist) tb.add(P(TP_Q_WHH + dir * 4), Hq, dir ? sc + L.wt2 : wt, 3 * Hq, 3 * Hq, Hq);   // [Hq][3Hq]
    }
    ACVAE_TRY(acvae_transpose_batch(tb, st.s));
  }
  ACVAE_TRY(acvae_reparam_bwd(d_q_z, E, d_q_means, d_q_logs, E, q_logs, E, eps_q, E, dml, 2 * E, R, E, st));
  ACVAE_TRY(gemm(dml, 2 * E, wt_ml, 2 * E, nullptr, dhid, 2 * Hq, R, 2 * Hq, 2 * E, 0, st));
  if (d_q_means_utt)
    ACVAE_TRY(acvae::pool_bwd(d_q_means_utt, lens1, (const int*)(sv + L.argmax), dhid, (long)Tc * 2 * Hq, 2 * Hq, 1, N,
                              Tc, 2 * Hq, st));
  ACVAE_TRY(gemm_tn(dml, 2 * E, hid, 2 * Hq, G(TP_Q_TML_W), 2 * Hq, 2 * E, 2 * Hq, R, tn, st));
  acvae::ColsumBatch cb;                     // the five bias gradients of the call: one launch at its end
  cb.add(dml, R, 2 * E, G(TP_Q_TML_B));
  float* dx = sc + L.dx;
  if (persist) {                 // BPTT of both directions in one launch; the parameter products below are unchanged
    PqbParams pb;
    for (int dir = 0; dir < 2; ++dir) {
      pb.wt[dir] = dir ? sc + L.wt2 : wt;                    // transposed weight_hh, made at the entry
      pb.save[dir] = sv + (dir ? L.save_r : L.save_f);
      pb.hprev[dir] = sv + (dir ? L.hprev_r : L.hprev_f);
      pb.dgi[dir] = sc + (dir ? L.dgi_r : L.dgi_f);
      pb.dgh[dir] = sc + (dir ? L.dgh_r : L.dgh_f);
    }
    pb.dhid = dhid; pb.lens1 = lens1; pb.cnt = (unsigned*)(sc + L.pq_cnt);
    pb.N = N; pb.Tc = Tc; pb.Hq = Hq;
    ACVAE_TRY(acvae::posterior_persist_bwd(pb, st.s, flags));
  }
  for (int dir = 0; dir < 2; ++dir) {
    const int o = dir * 4;
    float* save = sv + (dir ? L.save_r : L.save_f);
    float* hprev = sv + (dir ? L.hprev_r : L.hprev_f);
    float* dgi = sc + (dir ? L.dgi_r : L.dgi_f);
    float* dgh = sc + (dir ? L.dgh_r : L.dgh_f);
    float* dh = sc + L.dh_a;
    float* dh2 = sc + L.dh_b;
    if (!persist) {
      ACVAE_TRY(transp(P(TP_Q_WHH + o), Hq, wt, 3 * Hq, 3 * Hq, Hq, st));                   // [Hq][3Hq]
      ACVAE_TRY(acvae::copy_rows(dh, Hq, nullptr, 0, N, Hq, st));
      for (int k = 0; k < Tc; ++k) {
        const int t = dir ? k : Tc - 1 - k;  // reverse of the forward order
        ACVAE_TRY(acvae::gru_bwd(dh, Hq, dhid + (long)t * 2 * Hq + dir * Hq, (long)Tc * 2 * Hq, save + (long)t * 4 * Hq,
                                 (long)Tc * 4 * Hq, hprev + (long)t * Hq, (long)Tc * Hq, dgi + (long)t * 3 * Hq,
                                 (long)Tc * 3 * Hq, dgh + (long)t * 3 * Hq, (long)Tc * 3 * Hq, dh2, Hq, lens1, t, N, Hq,
                                 st));
        ACVAE_TRY(gemm(dgh + (long)t * 3 * Hq, (long)Tc * 3 * Hq, wt, 3 * Hq, nullptr, dh2, Hq, N, Hq, 3 * Hq, 1, st));
        float* tmp = dh; dh = dh2; dh2 = tmp;
      }
    }
    ACVAE_TRY(gemm_tn(dgi, 3 * Hq, X, E, G(TP_Q_WIH + o), E, 3 * Hq, E, R, tn, st));
    cb.add(dgi, R, 3 * Hq, G(TP_Q_BIH + o));
    ACVAE_TRY(gemm_tn(dgh, 3 * Hq, hprev, Hq, G(TP_Q_WHH + o), Hq, 3 * Hq, Hq, R, tn, st));
    cb.add(dgh, R, 3 * Hq, G(TP_Q_BHH + o));
    ACVAE_TRY(gemm(dgi, 3 * Hq, wt_ih[dir], 3 * Hq, nullptr, dx, E, R, E, 3 * Hq, dir, st));
  }
  ACVAE_TRY(acvae::colsum_batch(cb, dpart, L.dpart_doubles, st));
  ACVAE_TRY(acvae::embed_scatter(words, dx, E, G(TP_Q_EMB), V, R, E, st));
  return ACVAE_OK;
}

// ==========================================================================================
// prior + decoder loop
// ==========================================================================================
extern "C" int64_t acvae_decode_saved_bytes(int N, int Tc, int S, int E, int H, int A, int V, int Eenc) {
  DecLayout L;
  return dec_layout(N, Tc, S, E, H, A, V, Eenc, L) == ACVAE_OK ? L.saved_total * 4 : -1;
}
extern "C" int64_t acvae_decode_scratch_bytes(int N, int Tc, int S, int E, int H, int A, int V, int Eenc) {
  DecLayout L;
  if (dec_layout(N, Tc, S, E, H, A, V, Eenc, L) != ACVAE_OK) return -1;
  return (L.scratch_fwd > L.scratch_bwd ? L.scratch_fwd : L.scratch_bwd) * 4;
}

extern "C" int acvae_decode_fwd(const void* const* params, const float* mem_in, const int64_t* mem_lens,
                                const int64_t* caps, int64_t ld_caps, const int64_t* lens1, const float* q_z,
                                const float* eps_p, const int* ss_flags_host, const int* dis_flags_host, float* logits,
                                float* outputs, int64_t* seqs, float* sampled_logprobs, float* attn_w, float* p_means,
                                float* p_logs, float* p_z, float* p_means_utt, float* h_final, float* hp_final,
                                float* cp_final, void* saved_v, int64_t saved_bytes, void* scratch_v,
                                int64_t scratch_bytes, int N, int Tc, int S, int E, int H, int A, int V, int Eenc,
                                int start_idx, int end_idx, void* stream, void* aux_stream, int flags) {
  return acvae_decode_fwd_sampled(params, mem_in, mem_lens, caps, ld_caps, lens1, q_z, eps_p, ss_flags_host,
                                  dis_flags_host, logits, outputs, seqs, sampled_logprobs, attn_w, p_means, p_logs, p_z,
                                  p_means_utt, h_final, hp_final, cp_final, saved_v, saved_bytes, scratch_v,
                                  scratch_bytes, N, Tc, S, E, H, A, V, Eenc, start_idx, end_idx, stream, aux_stream,
                                  ACVAE_SAMPLE_GREEDY, 1.f, nullptr, nullptr, 0.f, flags);
}

extern "C" int acvae_decode_fwd_sampled(const void* const* params, const float* mem_in, const int64_t* mem_lens,
                                        const int64_t* caps, int64_t ld_caps, const int64_t* lens1, const float* q_z,
                                        const float* eps_p, const int* ss_flags_host, const int* dis_flags_host,
                                        float* logits, float* outputs, int64_t* seqs, float* sampled_logprobs,
                                        float* attn_w, float* p_means, float* p_logs, float* p_z, float* p_means_utt,
                                        float* h_final, float* hp_final, float* cp_final, void* saved_v,
                                        int64_t saved_bytes, void* scratch_v, int64_t scratch_bytes, int N, int Tc, int S,
                                        int E, int H, int A, int V, int Eenc, int start_idx, int end_idx, void* stream,
                                        void* aux_stream, int sample_method, float temp, const float* sample_noise,
                                        const uint8_t* emb_keep, float emb_drop_p, int flags) {
  if (emb_keep && !(emb_drop_p > 0.f && emb_drop_p <= 1.f)) return ACVAE_EINVAL;   // p = 1: nn.Dropout zeroes everything
  if (sample_method != ACVAE_SAMPLE_GREEDY &&
      ((sample_method != ACVAE_SAMPLE_GUMBEL && sample_method != ACVAE_SAMPLE_MULTINOMIAL) || !sample_noise ||
       !(temp > 0.f)))
    return ACVAE_EINVAL;
  DecLayout L;
  ACVAE_TRY(dec_layout(N, Tc, S, E, H, A, V, Eenc, L));
  if (!params || !mem_in || !mem_lens || !eps_p || !logits || !outputs || !seqs || !sampled_logprobs || !attn_w ||
      !p_means || !p_logs || !p_z || !saved_v || !scratch_v)
    return ACVAE_EINVAL;
  const bool train = caps != nullptr;
  if (train && (!lens1 || !q_z || !ss_flags_host || !dis_flags_host || !p_means_utt)) return ACVAE_EINVAL;
  if (train && H != E) return ACVAE_EUNSUPPORTED;  // mean_log_out = Linear(embed_size, .) is fed the GRU output
  if (saved_bytes < L.saved_total * 4 || scratch_bytes < L.scratch_fwd * 4) return ACVAE_EWORKSPACE;
  float* sv = (float*)saved_v;
  float* sc = (float*)scratch_v;
  auto P = [&](int i) { return (const float*)params[i]; };
  const int R = N * Tc, Hp = E;
  const bool has_ln = Eenc != E || params[TP_LN_W] != nullptr;
  bool teacher = train, prior_feeds_decoder = !train;
  if (train)
    for (int t = 0; t < Tc; ++t) {
      teacher = teacher && ss_flags_host[t] != 0;
      prior_feeds_decoder = prior_feeds_decoder || dis_flags_host[t] != 0;
    }
  // st: the decoder chain and everything batched; sp: the prior chain (second stream only when the chains are independent)
  Ctx st{(hipStream_t)stream, sc + L.skws};
  const Fork fork{st.s, (aux_stream && teacher) ? (hipStream_t)aux_stream : st.s};
  Ctx sp{fork.aux, sc + L.skws_p};
  // the step-by-step paths may use the split-over-frames attention: its arrival counters start at zero (ordered in front of
  // the fork); the persistent launch has its own attention
  const bool persist_fwd = teacher && !prior_feeds_decoder && !(flags & ACVAE_FLAG_NO_PERSIST) && H == E &&
                           acvae::decode_persist_ok(N, Tc, S, E, H, A);
  const bool attws_ready = !persist_fwd && L.attfws_bytes > 0;
  {   // every ticket / counter of the call in one launch, in front of the fork
    acvae::ZeroBatch zb;
    zb_skinny(zb, st.skws); zb_skinny(zb, sp.skws);
    if (attws_ready) { zb.add(sc + L.attfws_d, 256); zb.add(sc + L.attfws_p, 256); }
    if (persist_fwd) { zb.add(sc + L.pd_cnt, acvae::decode_persist_counter_words(Tc)); flags |= ACVAE_FLAG_INT_CNT_ZEROED; }
    ACVAE_TRY(acvae::zero_batch(zb, st.s));
  }

  int64_t* words = (int64_t*)(sv + L.words);
  // without the ln projection the attention memory IS the caller's tensor (it stays valid through the backward, which gets it
  // again): no copy into `saved` (round 4: one launch less in front of the persistent launch)
  const float* mem = has_ln ? sv + L.mem : mem_in;
  float* encproj_d = sv + L.encproj_d;
  float* encproj_p = sv + L.encproj_p;
  float* qd = sv + L.qd;
  float* qp = sv + L.qp;
  float* attw_p = sv + L.attw_p;
  float* rnn_d = sv + L.rnn_d;
  float* rnn_p = sv + L.rnn_p;
  float* gi_d = sc + L.gi_d;
  float* gh_d = sc + L.gh_d;
  float* gates_p = sc + L.gates_p;
  float* ml = sc + L.ml;
  float* zeros = sc + L.h0;
  uint8_t* unfinished = (uint8_t*)(sv + L.unfinished);
  const long ld3E = (long)Tc * 3 * E;

  // encoder memory (optional ln projection, vae_model.py:743-744) and the hoisted attention halves
  if (has_ln) ACVAE_TRY(gemm(mem_in, Eenc, P(TP_LN_W), Eenc, P(TP_LN_B), sv + L.mem, E, N * S, E, Eenc, 0, st));
  // the prior attention's half goes with the prior chain (second stream in the teacher-forced paths: the chain in front of the
  // persistent launch was 9 launches on the first stream and 5 behind them on the second; now 6 and 6 side by side)
  auto encproj_prior = [&](const Ctx& c) -> int {
    return gemm(mem, E, P(TP_P_ATT_W) + E, 2 * E, P(TP_P_ATT_B), encproj_p, E, N * S, E, E, 0, c);
  };
  auto encproj_dec = [&]() -> int {
    ACVAE_TRY(gemm(mem, E, P(TP_DEC_ATT_W) + H, E + H, P(TP_DEC_ATT_B), encproj_d, A, N * S, A, E, 0, st));
    return acvae::copy_rows(zeros, H > Hp ? H : Hp, nullptr, 0, N, H > Hp ? H : Hp, st);
  };

  // ---- per-range helpers; (t0,cnt) is either (0,Tc) [rows contiguous] or (t,1) [row stride Tc*C]
  auto rows_of = [&](int cnt) { return cnt == Tc ? R : N; };
  auto ldof = [&](int cnt, long C) { return cnt == Tc ? C : (long)Tc * C; };
  auto prior_pre = [&](int t0, int cnt) -> int {
    const int M = rows_of(cnt);
    // embedding -> rnn_p[:, t, 0:E]
    ACVAE_TRY(acvae::embed_gather(words + t0, cnt == Tc ? 1 : Tc, P(TP_P_EMB), V, rnn_p + (long)t0 * 3 * E,
                                  ldof(cnt, 3 * E), M, E, sp));
    // query projection and attention over the audio memory (text_encoder.py:251)
    ACVAE_TRY(gemm(rnn_p + (long)t0 * 3 * E, ldof(cnt, 3 * E), P(TP_P_ATT_W), 2 * E, nullptr, qp + (long)t0 * E,
                   ldof(cnt, E), M, E, E, 0, sp));
    ACVAE_TRY(acvae_attn_fwd(qp + (long)t0 * E, (long)Tc * E, E, encproj_p, mem, mem_lens, P(TP_P_ATT_V),
                             rnn_p + (long)t0 * 3 * E + E, ld3E, 3 * E, attw_p + (long)t0 * S, (long)Tc * S, S, N, cnt, S,
                             E, E, (cnt == 1 && attws_ready) ? sc + L.attfws_p : nullptr, L.attfws_bytes, sp, flags));
    // LSTM input projection of [emb; ctx] (+ both biases); the last_z / h parts are added per step
    ACVAE_TRY(gemm(rnn_p + (long)t0 * 3 * E, ldof(cnt, 3 * E), P(TP_P_WIH), 3 * E, P(TP_P_BIH),
                   gates_p + (long)t0 * 4 * Hp, ldof(cnt, 4 * Hp), M, 4 * Hp, 2 * E, 0, sp));
    return ACVAE_OK;
  };
  float* hp_all = sv + L.hp_all;
  float* c_all = sv + L.c_all;
  float* hpprev = sv + L.hpprev;
  float* lstm_save = sv + L.lstm_save;
  auto prior_step = [&](int t) -> int {
    const float* hprev = t ? hp_all + (long)(t - 1) * Hp : zeros;
    const long ldh = t ? (long)Tc * Hp : Hp;
    const float* cprev = t ? c_all + (long)(t - 1) * Hp : nullptr;
    if (t == 0) ACVAE_TRY(acvae::copy_rows(rnn_p + 2 * E, ld3E, nullptr, 0, N, E, sp));  // last_z = 0
    // gates += last_z . W_ih[:, 2E:3E]^T + h . W_hh^T + b_hh
    ACVAE_TRY(gemm2(rnn_p + (long)t * 3 * E + 2 * E, ld3E, P(TP_P_WIH) + 2 * E, 3 * E, E, hprev, ldh, P(TP_P_WHH), Hp,
                    Hp, P(TP_P_BHH), gates_p + (long)t * 4 * Hp, (long)Tc * 4 * Hp, N, 4 * Hp, 1, sp));
    ACVAE_TRY(acvae::lstm_fwd(gates_p + (long)t * 4 * Hp, (long)Tc * 4 * Hp, cprev, (long)Tc * Hp,
                              hp_all + (long)t * Hp, (long)Tc * Hp, c_all + (long)t * Hp, (long)Tc * Hp,
                              lstm_save + (long)t * 5 * Hp, (long)Tc * 5 * Hp, N, Hp, sp));
    ACVAE_TRY(gemm(hp_all + (long)t * Hp, (long)Tc * Hp, P(TP_P_ML_W), Hp, P(TP_P_ML_B), ml, 2 * E, N, 2 * E, Hp, 0, sp));
    float* z2 = (t + 1 < Tc) ? rnn_p + (long)(t + 1) * 3 * E + 2 * E : nullptr;
    ACVAE_TRY(acvae_reparam_fwd(ml, 2 * E, eps_p + (long)t * N * E, E, p_means + (long)t * E, p_logs + (long)t * E,
                                p_z + (long)t * E, (long)Tc * E, z2, ld3E, N, E, sp));
    return ACVAE_OK;
  };
  auto dec_pre = [&](int t0, int cnt) -> int {
    const int M = rows_of(cnt);
    ACVAE_TRY(acvae::embed_gather(words + t0, cnt == Tc ? 1 : Tc, P(TP_DEC_EMB), V, rnn_d + (long)t0 * 3 * E,
                                  ldof(cnt, 3 * E), M, E, st));
    if (emb_keep)          // the decoder's word-embedding dropout (models/decoder.py:33,184), mask [Tc][N][E]
      ACVAE_TRY(acvae::dropout_rows(rnn_d + (long)t0 * 3 * E, ld3E, 3 * E, emb_keep + (long)t0 * N * E, E, (long)N * E,
                                    emb_drop_p < 1.f ? 1.f / (1.f - emb_drop_p) : 0.f, N, cnt, E, st));
    // z: posterior sample unless this step drew the prior (vae_model.py:800-808); one copy when no step did
    if (cnt == Tc && train && !prior_feeds_decoder) {
      ACVAE_TRY(acvae::copy_rows(rnn_d + 2 * E, 3 * E, q_z, E, R, E, st));
    } else {
      for (int t = t0; t < t0 + cnt; ++t) {
        const float* zsrc = (train && !dis_flags_host[t]) ? q_z : p_z;
        ACVAE_TRY(acvae::copy_rows(rnn_d + (long)t * 3 * E + 2 * E, ld3E, zsrc + (long)t * E, (long)Tc * E, N, E, st));
      }
    }
    ACVAE_TRY(gemm2(rnn_d + (long)t0 * 3 * E, ldof(cnt, 3 * E), P(TP_DEC_WIH), 3 * E, E,
                    rnn_d + (long)t0 * 3 * E + 2 * E, ldof(cnt, 3 * E), P(TP_DEC_WIH) + 2 * E, 3 * E, E, P(TP_DEC_BIH),
                    gi_d + (long)t0 * 3 * H, ldof(cnt, 3 * H), M, 3 * H, 0, st));
    return ACVAE_OK;
  };
  float* gru_save = sv + L.gru_save;
  float* hprev_d = sv + L.hprev_d;
  static const bool pair_env = !(getenv("ACVAE_SKINNY_PAIR") && atoi(getenv("ACVAE_SKINNY_PAIR")) == 0);
  const bool pair_ok = pair_env && N <= 64;
  auto dec_step = [&](int t) -> int {
    const float* hprev = t ? outputs + (long)(t - 1) * H : zeros;
    const long ldh = t ? (long)Tc * H : H;
    // Both products of h_{t-1} (attention query, hidden-to-hidden gates) in one launch.  The kernels of this chain run
    // back to back and each is bound by its own load latency (7-20 us for microseconds of work), so what shortens a step
    // is one kernel fewer on the chain, not fewer launches as such: decode forward 1.33 -> 1.16 ms.
    if (pair_ok) {
      ACVAE_TRY(acvae_gemm_nt_pair(hprev, ldh, P(TP_DEC_ATT_W), E + H, H, nullptr, qd + (long)t * A, (long)Tc * A, A, 0,
                                   hprev, ldh, P(TP_DEC_WHH), H, H, P(TP_DEC_BHH), gh_d, 3 * H, 3 * H, 0, N, st));
    } else {
      ACVAE_TRY(gemm(hprev, ldh, P(TP_DEC_ATT_W), E + H, nullptr, qd + (long)t * A, (long)Tc * A, N, A, H, 0, st));
      ACVAE_TRY(gemm(hprev, ldh, P(TP_DEC_WHH), H, P(TP_DEC_BHH), gh_d, 3 * H, N, 3 * H, H, 0, st));
    }
    ACVAE_TRY(acvae_attn_fwd(qd + (long)t * A, (long)Tc * A, A, encproj_d, mem, mem_lens, P(TP_DEC_ATT_V),
                             rnn_d + (long)t * 3 * E + E, ld3E, 3 * E, attn_w + (long)t * S, (long)Tc * S, S, N, 1, S, A,
                             E, attws_ready ? sc + L.attfws_d : nullptr, L.attfws_bytes, st, flags));
    ACVAE_TRY(gemm(rnn_d + (long)t * 3 * E + E, ld3E, P(TP_DEC_WIH) + E, 3 * E, nullptr, gi_d + (long)t * 3 * H,
                   (long)Tc * 3 * H, N, 3 * H, E, 1, st));
    ACVAE_TRY(acvae::gru_fwd(gi_d + (long)t * 3 * H, (long)Tc * 3 * H, gh_d, 3 * H, hprev, ldh, outputs + (long)t * H,
                             (long)Tc * H, nullptr, 0, gru_save + (long)t * 4 * H, (long)Tc * 4 * H,
                             hprev_d + (long)t * H, (long)Tc * H, nullptr, t, N, H, st));
    return ACVAE_OK;
  };
  // the backward's transposed weights (one launch for the nine): queued on the prior chain's stream where that stream has
  // nothing else to do - beside the persistent launch / behind the prior chain - and joined with it before the call returns
  auto transposes = [&](hipStream_t ts) -> int {
    if (!train) return ACVAE_OK;
    TransposeBatch tb;
    tb.add(P(TP_DEC_CLS_W), H, sv + L.wt_cls, V, V, H);                     // [H][V]
    tb.add(P(TP_DEC_WIH), 3 * E, sv + L.wt_dih, 3 * H, 3 * H, 3 * E);       // [3E][3H]
    tb.add(P(TP_DEC_WHH), H, sv + L.wt_dhh, 3 * H, 3 * H, H);               // [H][3H]
    tb.add(P(TP_DEC_ATT_W), E + H, sv + L.wt_datt, A, A, E + H);            // [H+E][A]: rows 0:H query half, H: memory half
    tb.add(P(TP_P_WIH), 3 * E, sv + L.wt_pih, 4 * Hp, 4 * Hp, 3 * E);       // [3E][4Hp]
    tb.add(P(TP_P_WHH), Hp, sv + L.wt_phh, 4 * Hp, 4 * Hp, Hp);             // [Hp][4Hp]
    tb.add(P(TP_P_ML_W), Hp, sv + L.wt_pml, 2 * E, 2 * E, Hp);              // [Hp][2E]
    tb.add(P(TP_P_ATT_W), 2 * E, sv + L.wt_patt, E, E, 2 * E);              // [2E][E]
    tb.add(P(TP_MLO_W), H, sv + L.wt_mlo, 2 * E, 2 * E, H);                 // [H][2E]
    return acvae_transpose_batch(tb, ts);
  };
  float* lse = sv + L.lse;
  auto classify = [&](int t0, int cnt) -> int {
    const int M = rows_of(cnt);
    ACVAE_TRY(gemm(outputs + (long)t0 * H, ldof(cnt, H), P(TP_DEC_CLS_W), H, P(TP_DEC_CLS_B), logits + (long)t0 * V,
                   ldof(cnt, V), M, V, H, 0, st));
    ACVAE_TRY(acvae_row_logsoftmax_argmax(logits + (long)t0 * V, (long)Tc * V, V, seqs + t0, sampled_logprobs + t0,
                                          lse + t0, Tc, 1, N, cnt, V, st));
    if (sample_method != ACVAE_SAMPLE_GREEDY)      // word_model.py:188-203 overwrite the greedy choice
      ACVAE_TRY(acvae_sample_next_word(logits + (long)t0 * V, (long)Tc * V, V, sample_noise + (long)t0 * N * V, V,
                                       (long)N * V, sample_method, temp, seqs + t0, sampled_logprobs + t0, Tc, 1, N, cnt,
                                       V, st));
    return ACVAE_OK;
  };

  if (teacher) {
    ACVAE_TRY(acvae::gather_words(caps, ld_caps, 1, words, N, Tc, st));
    ACVAE_TRY(fork.begin());
    ACVAE_TRY(encproj_prior(sp));
    ACVAE_TRY(encproj_dec());
    ACVAE_TRY(prior_pre(0, Tc));
    if (prior_feeds_decoder) {            // dec_pre reads p_z: the chains run back to back
      for (int t = 0; t < Tc; ++t) ACVAE_TRY(prior_step(t));
      ACVAE_TRY(fork.join());
      ACVAE_TRY(dec_pre(0, Tc));
      for (int t = 0; t < Tc; ++t) ACVAE_TRY(dec_step(t));
    } else if (persist_fwd) {
      // independent chains, all words known: the Tc steps of both chains as ONE persistent launch (decode_persist.hip);
      // the hoisted halves are computed as before (the prior's on the second stream), then the streams join
      ACVAE_TRY(dec_pre(0, Tc));
      ACVAE_TRY(acvae::copy_rows(rnn_p + 2 * E, ld3E, nullptr, 0, N, E, sp));  // last_z = 0 at step 0
      ACVAE_TRY(fork.join());
      PdParams pd{};
      pd.w_att = P(TP_DEC_ATT_W); pd.w_hh = P(TP_DEC_WHH); pd.b_hh = P(TP_DEC_BHH); pd.w_ih = P(TP_DEC_WIH);
      pd.att_v = P(TP_DEC_ATT_V);
      pd.pw_ih = P(TP_P_WIH); pd.pw_hh = P(TP_P_WHH); pd.pb_hh = P(TP_P_BHH); pd.w_ml = P(TP_P_ML_W); pd.b_ml = P(TP_P_ML_B);
      pd.encproj = encproj_d; pd.mem = mem; pd.mem_lens = mem_lens; pd.gi = gi_d; pd.gates_p = gates_p;
      pd.eps_p = eps_p; pd.zeros = zeros;
      pd.qd = qd; pd.gh = gh_d; pd.rnn_d = rnn_d; pd.attn_w = attn_w; pd.outputs = outputs; pd.gru_save = sv + L.gru_save;
      pd.hprev_d = sv + L.hprev_d; pd.rnn_p = rnn_p; pd.hp_all = sv + L.hp_all; pd.c_all = sv + L.c_all;
      pd.lstm_save = sv + L.lstm_save; pd.p_means = p_means; pd.p_logs = p_logs; pd.p_z = p_z;
      pd.cnt = (unsigned*)(sc + L.pd_cnt);
      pd.N = N; pd.Tc = Tc; pd.S = S; pd.E = E; pd.H = H; pd.A = A;
      ACVAE_TRY(acvae::decode_persist_fwd(pd, st.s, flags));
    } else {                              // independent chains: feed both queues step by step
      ACVAE_TRY(dec_pre(0, Tc));
      for (int t = 0; t < Tc; ++t) {
        ACVAE_TRY(prior_step(t));
        ACVAE_TRY(dec_step(t));
      }
    }
    ACVAE_TRY(transposes(sp.s));          // persistent path: sp has been idle since the join in front of the launch
    ACVAE_TRY(classify(0, Tc));
    ACVAE_TRY(fork.join());
  } else {
    ACVAE_TRY(encproj_dec());
    ACVAE_TRY(encproj_prior(st));
    for (int t = 0; t < Tc; ++t) {
      ACVAE_TRY(acvae::select_word(caps, ld_caps, seqs, Tc, words, Tc, t, train && ss_flags_host[t], start_idx, N, st));
      ACVAE_TRY(prior_pre(t, 1));
      ACVAE_TRY(prior_step(t));
      ACVAE_TRY(dec_pre(t, 1));
      ACVAE_TRY(dec_step(t));
      ACVAE_TRY(classify(t, 1));
      if (!train) ACVAE_TRY(acvae::finish_rows(seqs, Tc, unfinished, t, end_idx, N, st));
    }
    ACVAE_TRY(transposes(st.s));
  }
  // h_{t-1} of the prior LSTM for its weight gradient (one shifted copy of hp_all after the loop instead of a copy per step)
  {                    // (and the final states) - five small copies in one launch
    acvae::CopyRowsBatch cb;
    cb.add(hpprev, (long)Tc * Hp, nullptr, 0, N, Hp);
    if (Tc > 1) cb.add(hpprev + Hp, (long)Tc * Hp, hp_all, (long)Tc * Hp, N, (Tc - 1) * Hp);
    if (h_final) cb.add(h_final, H, outputs + (long)(Tc - 1) * H, (long)Tc * H, N, H);
    if (hp_final) cb.add(hp_final, Hp, hp_all + (long)(Tc - 1) * Hp, (long)Tc * Hp, N, Hp);
    if (cp_final) cb.add(cp_final, Hp, c_all + (long)(Tc - 1) * Hp, (long)Tc * Hp, N, Hp);
    ACVAE_TRY(acvae::copy_rows_batch(cb, st));
  }
  if (train) {  // p_means_utt = mean_log_out(mean_with_lens + max_with_lens of the GRU outputs), vae_model.py:722-728
    float* hidp = sv + L.pool_hid;
    ACVAE_TRY(acvae::pool_fwd(outputs, (long)Tc * H, H, lens1, hidp, (int*)(sv + L.pool_arg), N, Tc, H, st));
    ACVAE_TRY(gemm(hidp, H, P(TP_MLO_W), H, P(TP_MLO_B), p_means_utt, 2 * E, N, 2 * E, H, 0, st));
  }
  return ACVAE_OK;
}

extern "C" int acvae_decode_bwd_defers(const int* dis_flags_host, int Tc, void* stream, void* aux_stream, int flags) {
  // Default: everything ordered on `stream` on return - the contract a plain C caller expects.  ACVAE_FLAG_DEFER_PARAM_GRADS
  // (Hybrid_VAEModel, which joins the second stream itself, passes it) leaves the parameter gradients trailing.
  // Round 2 (implicit-GEMM convolutions): decode backward -0.48 ms, encoder backward beside the trailing work +0.46 ms.
  // Round 3 (A/B inside one session, three runs each): 17.106 -> 17.034 ms per step.
  if (!(flags & ACVAE_FLAG_DEFER_PARAM_GRADS) || !aux_stream || aux_stream == stream || !dis_flags_host) return 0;
  for (int t = 0; t < Tc; ++t)
    if (dis_flags_host[t] != 0) return 0;      // the prior BPTT waits for the decoder's dz: nothing to overlap
  return 1;
}

extern "C" int acvae_decode_bwd(const void* const* params, void* const* grads, const float* mem_in,
                                const int64_t* mem_lens, const int64_t* lens1, const float* eps_p,
                                const int* dis_flags_host, const float* outputs, const float* attn_w,
                                const float* p_logs, const float* d_logits, const float* d_outputs_ext,
                                const float* d_p_means, const float* d_p_logs, const float* d_p_z,
                                const float* d_p_means_utt, float* d_mem_in, float* d_q_z, void* saved_v,
                                int64_t saved_bytes, void* scratch_v, int64_t scratch_bytes, int N, int Tc, int S, int E,
                                int H, int A, int V, int Eenc, void* stream, void* aux_stream, const uint8_t* emb_keep,
                                float emb_drop_p, int flags) {
  if (emb_keep && !(emb_drop_p > 0.f && emb_drop_p <= 1.f)) return ACVAE_EINVAL;   // p = 1: nn.Dropout zeroes everything
  DecLayout L;
  ACVAE_TRY(dec_layout(N, Tc, S, E, H, A, V, Eenc, L));
  if (!params || !grads || !mem_in || !mem_lens || !lens1 || !eps_p || !dis_flags_host || !outputs || !attn_w ||
      !p_logs || !d_mem_in || !d_q_z || !saved_v || !scratch_v)
    return ACVAE_EINVAL;
  if (saved_bytes < L.saved_total * 4 || scratch_bytes < L.scratch_bwd * 4) return ACVAE_EWORKSPACE;
  float* sv = (float*)saved_v;
  float* sc = (float*)scratch_v;
  bool prior_feeds_decoder = false;
  for (int t = 0; t < Tc; ++t) prior_feeds_decoder = prior_feeds_decoder || dis_flags_host[t] != 0;
  // st: decoder chain; sp: prior chain, on the second stream when given (it has its own accumulators; the two meet in
  // dmem at the end).  A step that fed the prior's z to the decoder makes the prior BPTT wait for the decoder's dz.
  Ctx st{(hipStream_t)stream, sc + L.skws};
  const Fork fork{st.s, aux_stream ? (hipStream_t)aux_stream : st.s};
  Ctx sp{fork.aux, sc + L.skws_p};
  auto P = [&](int i) { return (const float*)params[i]; };
  auto G = [&](int i) { return (float*)grads[i]; };
  const int R = N * Tc, Hp = E;
  // d_mem_in and d_q_z are on the step's critical path (the encoder backward waits for d_mem_in and, through the posterior's
  // backward, for d_q_z).  With a second stream and independent chains, everything else - the parameter-gradient products,
  // the embedding gradients - is queued on the second stream AFTER the call has released the first one, where it runs beside
  // the posterior's and the encoder's backward.  The caller then owns two obligations (include/acvae_hip.h): the second
  // stream must be joined before the parameter gradients are read on another stream, and saved / scratch / the incoming
  // gradients must stay untouched until it has drained.
  const bool defer = acvae_decode_bwd_defers(dis_flags_host, Tc, stream, aux_stream, flags) != 0;
  const bool has_ln = Eenc != E || params[TP_LN_W] != nullptr;
  TnWs tn{sc + L.tn, L.tn_floats * 4};
  TnWs tn_p{sc + L.tn_p, L.tn_p_floats * 4};
  double* dpart = (double*)(sc + L.dpart);
  double* dpart_p = (double*)(sc + L.dpart_p);
  const bool persist_bwd = !prior_feeds_decoder && !(flags & ACVAE_FLAG_NO_PERSIST) && acvae::decode_persist_bwd_ok(N, Tc, S, E, H, A);
  {   // in front of the fork: both chains' tickets (skinny split-K, column sums, gemm_tn tiles), the persistent launch's arrival
      // counters and the prior attention's memory gradient - one launch instead of eight memsets
    acvae::ZeroBatch zb;
    zb_skinny(zb, st.skws); zb_skinny(zb, sp.skws); zb_colsum(zb, dpart); zb_colsum(zb, dpart_p); zb_tn(zb, tn); zb_tn(zb, tn_p);
    if (persist_bwd) {
      zb.add(sc + L.pd_cnt_b, acvae::decode_persist_bwd_counter_words(Tc));
      zb.add(sc + L.dmem_p, (long)N * S * E);
      flags |= ACVAE_FLAG_INT_CNT_ZEROED;
    }
    // the two embedding-table gradients start from zero (embed_scatter adds rows): here instead of two 10 MB memsets in the
    // trailing part beside the encoder backward
    zb.add(G(TP_DEC_EMB), (long)V * E); zb.add(G(TP_P_EMB), (long)V * E);
    ACVAE_TRY(acvae::zero_batch(zb, st.s));
  }
  const int64_t* words = (const int64_t*)(sv + L.words);
  const float* mem = has_ln ? sv + L.mem : mem_in;          // as in the forward
  float* rnn_d = sv + L.rnn_d;
  float* rnn_p = sv + L.rnn_p;

  // transposed weights for the dX = dY . W products: made by the forward (acvae_decode_fwd, training mode) into `saved`
  float *wt_cls = sv + L.wt_cls, *wt_dih = sv + L.wt_dih, *wt_dhh = sv + L.wt_dhh, *wt_datt = sv + L.wt_datt;
  float *wt_pih = sv + L.wt_pih, *wt_phh = sv + L.wt_phh, *wt_pml = sv + L.wt_pml, *wt_patt = sv + L.wt_patt;
  float *wt_mlo = sv + L.wt_mlo, *wt_ln = sc + L.wt_ln;
  int64_t* words_c = (int64_t*)(sc + L.words_c);
  ACVAE_TRY(acvae::gather_words(words, Tc, 1, words_c, N, Tc, st));
  ACVAE_TRY(fork.begin());

  // ---- d_outputs = external + utterance head + classifier
  float* d_out = sc + L.d_out;
  if (d_outputs_ext) ACVAE_TRY(acvae::copy_rows(d_out, H, d_outputs_ext, H, R, H, st));
  else ACVAE_TRY(zero(d_out, (long)R * H, st));
  if (d_p_means_utt) {
    float* dhid = sc + L.dhid;
    ACVAE_TRY(gemm(d_p_means_utt, 2 * E, wt_mlo, 2 * E, nullptr, dhid, H, N, H, 2 * E, 0, st));
    ACVAE_TRY(acvae::pool_bwd(dhid, lens1, (const int*)(sv + L.pool_arg), d_out, (long)Tc * H, H, 1, N, Tc, H, st));
  }
  if (d_logits) ACVAE_TRY(gemm(d_logits, V, wt_cls, V, nullptr, d_out, H, R, H, V, 1, st));
  // parameter gradients of the two heads (c: the stream / workspaces they are queued with)
  auto heads_params = [&](const Ctx& c, TnWs ws, double* dp) -> int {
    acvae::ColsumBatch cb;
    if (d_p_means_utt) {
      ACVAE_TRY(gemm_tn(d_p_means_utt, 2 * E, sv + L.pool_hid, H, G(TP_MLO_W), H, 2 * E, H, N, ws, c));
      cb.add(d_p_means_utt, N, 2 * E, G(TP_MLO_B));
    } else {
      ACVAE_TRY(zero(G(TP_MLO_W), (long)2 * E * H, c));
      ACVAE_TRY(zero(G(TP_MLO_B), 2 * E, c));
    }
    if (d_logits) {
      ACVAE_TRY(gemm_tn(d_logits, V, outputs, H, G(TP_DEC_CLS_W), H, V, H, R, ws, c));
      cb.add(d_logits, R, V, G(TP_DEC_CLS_B));
    } else {
      ACVAE_TRY(zero(G(TP_DEC_CLS_W), (long)V * H, c));
      ACVAE_TRY(zero(G(TP_DEC_CLS_B), V, c));
    }
    return acvae::colsum_batch(cb, dp, L.dpart_doubles, c);
  };
  // They feed nothing downstream.  With a second stream they go there, in front of the prior's BPTT: the two serial
  // chains leave most of the GPU idle, so the 75 us of these products cost the first stream nothing.
  static const bool heads_aux = !(getenv("ACVAE_HEADS_AUX") && atoi(getenv("ACVAE_HEADS_AUX")) == 0);
  // (also in trailing-gradient mode, round 4: their 0.1 ms then run beside the persistent BPTT launch, which leaves most of the GPU
  // idle, instead of beside the encoder backward's first weight gradient)
  const bool heads_on_aux = heads_aux && fork.on();
  if (heads_on_aux) {
    ACVAE_TRY(heads_params(sp, tn_p, dpart_p));   // aux is behind fork.begin(): the upstream gradients are in place
  } else if (!defer) {
    ACVAE_TRY(heads_params(st, tn, dpart));
  }

  // ---- decoder BPTT
  float* dgi = sc + L.dgi;
  float* dgh = sc + L.dgh;
  float* dqd = sc + L.dqd;
  float* dencproj = sc + L.dencproj;
  float* dvpart = sc + L.dvpart;
  float* dctx = sc + L.dctx;
  float* dmem = sc + L.dmem;
  float* dh = sc + L.dh_a;
  float* dh2 = sc + L.dh_b;
  const float* gru_save = sv + L.gru_save;
  const float* hprev_d = sv + L.hprev_d;
  const float* qd = sv + L.qd;
  // accumulators of the per-step BPTT (the persistent launch writes all four itself: four memsets fewer in front of it)
  auto dec_begin = [&]() -> int {
    ACVAE_TRY(zero(dencproj, (long)N * S * A, st));
    ACVAE_TRY(zero(dvpart, (long)N * A, st));
    ACVAE_TRY(zero(dmem, (long)N * S * E, st));
    ACVAE_TRY(zero(dh, (long)N * H, st));
    return ACVAE_OK;
  };
  float* drnn = sc + L.drnn;
  float* dz_dec = sc + L.dz_dec;
  int dv_rows = N;                 // rows of dvpart (the persistent BPTT of long clips writes one row per frame share)
  auto dec_bptt = [&](int t) -> int {
    ACVAE_TRY(acvae::gru_bwd(dh, H, d_out + (long)t * H, (long)Tc * H, gru_save + (long)t * 4 * H, (long)Tc * 4 * H,
                             hprev_d + (long)t * H, (long)Tc * H, dgi + (long)t * 3 * H, (long)Tc * 3 * H,
                             dgh + (long)t * 3 * H, (long)Tc * 3 * H, dh2, H, nullptr, t, N, H, st));
    // dctx = dgi . W_ih[:, E:2E]
    ACVAE_TRY(gemm(dgi + (long)t * 3 * H, (long)Tc * 3 * H, wt_dih + (long)E * 3 * H, 3 * H, nullptr, dctx, E, N, E,
                   3 * H, 0, st));
    ACVAE_TRY(acvae_attn_bwd(dctx, E, 0, qd + (long)t * A, (long)Tc * A, A, sv + L.encproj_d, mem, mem_lens,
                             P(TP_DEC_ATT_V), attn_w + (long)t * S, (long)Tc * S, S, dqd + (long)t * A, (long)Tc * A, A,
                             dencproj, dmem, dvpart, sc + L.attws, L.attws_bytes, N, 1, S, A, E, st));
    // dh_prev = dh*z + dgh . W_hh + dqd . W_att[:, :H]
    ACVAE_TRY(gemm2(dgh + (long)t * 3 * H, (long)Tc * 3 * H, wt_dhh, 3 * H, 3 * H, dqd + (long)t * A, (long)Tc * A,
                    wt_datt, A, A, nullptr, dh2, H, N, H, 1, st));
    float* tmp = dh; dh = dh2; dh2 = tmp;
    return ACVAE_OK;
  };
  // the decoder attention's share of the memory gradient (critical)
  auto dec_memgrad = [&]() -> int {
    return gemm(dencproj, A, wt_datt + (long)H * A, A, nullptr, dmem, E, N * S, E, A, 1, st);
  };
  // d z of the decoder's input gates, routed to the posterior sample (here) or to the prior sample (prior chain, below), per
  // step.  The posterior's backward waits for d_q_z and the encoder's backward for the posterior's (its pooled audio
  // embedding): in deferred mode this product stays on the FIRST stream (round 4: at the tail of the trailing work it held the
  // encoder backward back by the whole 0.6 ms of parameter-gradient products in front of it).
  auto dec_dz = [&](const Ctx& c) -> int {
    ACVAE_TRY(gemm(dgi, 3 * H, wt_dih + (long)2 * E * 3 * H, 3 * H, nullptr, dz_dec, E, R, E, 3 * H, 0, c));  // d z
    if (!prior_feeds_decoder) {
      ACVAE_TRY(acvae::copy_rows(d_q_z, E, dz_dec, E, R, E, c));
    } else {
      for (int t = 0; t < Tc; ++t)
        ACVAE_TRY(acvae::copy_rows(d_q_z + (long)t * E, (long)Tc * E, dis_flags_host[t] ? nullptr : dz_dec + (long)t * E,
                                   (long)Tc * E, N, E, c));
    }
    return ACVAE_OK;
  };
  // batched parameter gradients of the decoder, the embedding gradient and (with_dz) d_q_z
  auto dec_params = [&](const Ctx& c, TnWs ws, double* dp, bool with_dz = true) -> int {
    ACVAE_TRY(gemm_tn(dgi, 3 * H, rnn_d, 3 * E, G(TP_DEC_WIH), 3 * E, 3 * H, 3 * E, R, ws, c));
    ACVAE_TRY(gemm_tn(dgh, 3 * H, hprev_d, H, G(TP_DEC_WHH), H, 3 * H, H, R, ws, c));
    {   // the four bias-shaped gradients of the decoder in one launch
      acvae::ColsumBatch cb;
      cb.add(dgi, R, 3 * H, G(TP_DEC_BIH)); cb.add(dgh, R, 3 * H, G(TP_DEC_BHH));
      cb.add(dencproj, N * S, A, G(TP_DEC_ATT_B)); cb.add(dvpart, dv_rows, A, G(TP_DEC_ATT_V));
      ACVAE_TRY(acvae::colsum_batch(cb, dp, L.dpart_doubles, c));
    }
    // attention parameters: W = [query half | memory half]
    ACVAE_TRY(gemm_tn(dqd, A, hprev_d, H, G(TP_DEC_ATT_W), E + H, A, H, R, ws, c));
    ACVAE_TRY(gemm_tn(dencproj, A, mem, E, G(TP_DEC_ATT_W) + H, E + H, A, E, N * S, ws, c));
    // d(rnn_input) for the embedding and z columns
    ACVAE_TRY(gemm(dgi, 3 * H, wt_dih, 3 * H, nullptr, drnn, E, R, E, 3 * H, 0, c));                    // d emb
    if (emb_keep)          // back through the word-embedding dropout
      ACVAE_TRY(acvae::dropout_rows(drnn, (long)Tc * E, E, emb_keep, E, (long)N * E, emb_drop_p < 1.f ? 1.f / (1.f - emb_drop_p) : 0.f, N, Tc, E, c));
    if (with_dz) ACVAE_TRY(dec_dz(c));
    ACVAE_TRY(acvae::embed_scatter(words_c, drnn, E, G(TP_DEC_EMB), V, R, E, c));        // (table zeroed at the call's entry)
    return ACVAE_OK;
  };

  // ---- prior BPTT
  float* dgates = sc + L.dgates;
  float* dml_all = sc + L.dml_all;
  float* dhp = sc + L.dhp_a;
  float* dhp2 = sc + L.dhp_b;
  float* dc = sc + L.dc_a;
  float* dc2 = sc + L.dc_b;
  float* dlz = sc + L.dlz_a;
  float* dlz2 = sc + L.dlz_b;
  const float* lstm_save = sv + L.lstm_save;
  const float* c_all = sv + L.c_all;
  const float* hp_all = sv + L.hp_all;
  const float* hpprev = sv + L.hpprev;
  float* dpz = sc + L.dqp;  // d p_z total [N,Tc,E]; the buffer becomes dqp once the prior attention needs it
  float* drnn_p = sc + L.drnn_p;
  float* dencproj_p = sc + L.dencproj_p;
  float* dvpart_p = sc + L.dvpart_p;
  float* dmem_p = sc + L.dmem_p;
  auto prior_begin = [&]() -> int {
  for (int t = 0; t < Tc; ++t) {
    ACVAE_TRY(acvae::copy_rows(dpz + (long)t * E, (long)Tc * E, d_p_z ? d_p_z + (long)t * E : nullptr, (long)Tc * E, N, E,
                               sp));
    if (dis_flags_host[t])
      ACVAE_TRY(acvae::add_rows(dpz + (long)t * E, (long)Tc * E, dz_dec + (long)t * E, (long)Tc * E, N, E, sp));
  }
  ACVAE_TRY(zero(dmem_p, (long)N * S * E, sp));
  ACVAE_TRY(zero(dhp, (long)N * Hp, sp));
  ACVAE_TRY(zero(dc, (long)N * Hp, sp));
  ACVAE_TRY(zero(dlz, (long)N * E, sp));
    return ACVAE_OK;
  };
  auto prior_bptt = [&](int t) -> int {
    // dz_t = d p_z[:,t] (+ decoder share) + d last_z from step t+1
    ACVAE_TRY(acvae::add_rows(dlz, E, dpz + (long)t * E, (long)Tc * E, N, E, sp));
    ACVAE_TRY(acvae_reparam_bwd(dlz, E, d_p_means ? d_p_means + (long)t * E : nullptr,
                                d_p_logs ? d_p_logs + (long)t * E : nullptr, (long)Tc * E, p_logs + (long)t * E,
                                (long)Tc * E, eps_p + (long)t * N * E, E, dml_all + (long)t * 2 * E, (long)Tc * 2 * E, N,
                                E, sp));
    // dh = dh_next + dml . W_ml
    ACVAE_TRY(gemm(dml_all + (long)t * 2 * E, (long)Tc * 2 * E, wt_pml, 2 * E, nullptr, dhp, Hp, N, Hp, 2 * E, 1, sp));
    ACVAE_TRY(acvae::lstm_bwd(dhp, Hp, dc, Hp, lstm_save + (long)t * 5 * Hp, (long)Tc * 5 * Hp,
                              t ? c_all + (long)(t - 1) * Hp : nullptr, (long)Tc * Hp, dgates + (long)t * 4 * Hp,
                              (long)Tc * 4 * Hp, dc2, Hp, N, Hp, sp));
    ACVAE_TRY(gemm(dgates + (long)t * 4 * Hp, (long)Tc * 4 * Hp, wt_phh, 4 * Hp, nullptr, dhp2, Hp, N, Hp, 4 * Hp, 0, sp));
    ACVAE_TRY(gemm(dgates + (long)t * 4 * Hp, (long)Tc * 4 * Hp, wt_pih + (long)2 * E * 4 * Hp, 4 * Hp, nullptr, dlz2, E,
                   N, E, 4 * Hp, 0, sp));
    float* tmp = dhp; dhp = dhp2; dhp2 = tmp;
    tmp = dc; dc = dc2; dc2 = tmp;
    tmp = dlz; dlz = dlz2; dlz2 = tmp;
    return ACVAE_OK;
  };
  float* dqp = sc + L.dqp;  // dpz is dead once the prior BPTT is through
  // the prior attention's share of the memory gradient (critical)
  auto prior_memgrad = [&]() -> int {
    // d[emb; ctx] of the prior
    ACVAE_TRY(gemm(dgates, 4 * Hp, wt_pih, 4 * Hp, nullptr, drnn_p, 3 * E, R, 2 * E, 4 * Hp, 0, sp));   // cols 0:2E of drnn_p[R,3E]
    // prior attention backward (all Tc queries of a clip inside one workgroup: deterministic accumulation)
    ACVAE_TRY(zero(dencproj_p, (long)N * S * E, sp));
    ACVAE_TRY(zero(dvpart_p, (long)N * E, sp));
    ACVAE_TRY(acvae_attn_bwd(drnn_p + E, (long)Tc * 3 * E, 3 * E, sv + L.qp, (long)Tc * E, E, sv + L.encproj_p, mem,
                             mem_lens, P(TP_P_ATT_V), sv + L.attw_p, (long)Tc * S, S, dqp, (long)Tc * E, E, dencproj_p,
                             dmem_p, dvpart_p, sc + L.attws_p, L.attws_bytes, N, Tc, S, E, E, sp));
    return gemm(dencproj_p, E, wt_patt + (long)E * E, E, nullptr, dmem_p, E, N * S, E, E, 1, sp);
  };
  // batched parameter gradients of the prior and its embedding gradient
  auto prior_params = [&]() -> int {
    ACVAE_TRY(gemm_tn(dml_all, 2 * E, hp_all, Hp, G(TP_P_ML_W), Hp, 2 * E, Hp, R, tn_p, sp));
    ACVAE_TRY(gemm_tn(dgates, 4 * Hp, rnn_p, 3 * E, G(TP_P_WIH), 3 * E, 4 * Hp, 3 * E, R, tn_p, sp));
    ACVAE_TRY(gemm_tn(dgates, 4 * Hp, hpprev, Hp, G(TP_P_WHH), Hp, 4 * Hp, Hp, R, tn_p, sp));
    {   // the prior's bias-shaped gradients in one launch (the LSTM's two biases share theirs)
      acvae::ColsumBatch cb;
      cb.add(dml_all, R, 2 * E, G(TP_P_ML_B)); cb.add(dgates, R, 4 * Hp, G(TP_P_BIH), G(TP_P_BHH));
      cb.add(dencproj_p, N * S, E, G(TP_P_ATT_B)); cb.add(dvpart_p, N, E, G(TP_P_ATT_V));
      ACVAE_TRY(acvae::colsum_batch(cb, dpart_p, L.dpart_doubles, sp));
    }
    // d emb_p = drnn[:, 0:E] + dqp . W_att[:, :E]
    ACVAE_TRY(gemm(dqp, E, wt_patt, E, nullptr, drnn_p, 3 * E, R, E, E, 1, sp));
    ACVAE_TRY(gemm_tn(dqp, E, rnn_p, 3 * E, G(TP_P_ATT_W), 2 * E, E, E, R, tn_p, sp));
    ACVAE_TRY(gemm_tn(dencproj_p, E, mem, E, G(TP_P_ATT_W) + E, 2 * E, E, E, N * S, tn_p, sp));
    return acvae::embed_scatter(words_c, drnn_p, 3 * E, G(TP_P_EMB), V, R, E, sp);       // (table zeroed at the call's entry)
  };
  if (prior_feeds_decoder) {   // the prior BPTT needs the decoder's dz: back to back
    ACVAE_TRY(dec_begin());
    for (int t = Tc - 1; t >= 0; --t) ACVAE_TRY(dec_bptt(t));
    ACVAE_TRY(dec_memgrad());
    ACVAE_TRY(dec_params(st, tn, dpart));
    if (fork.on()) ACVAE_TRY(Fork::edge(st.s, sp.s));
    ACVAE_TRY(prior_begin());
    for (int t = Tc - 1; t >= 0; --t) ACVAE_TRY(prior_bptt(t));
    ACVAE_TRY(prior_memgrad());
    ACVAE_TRY(prior_params());
  } else if (persist_bwd) {
    // independent chains: the Tc steps of both BPTT chains as ONE persistent launch on the first stream
    // (decode_persist.hip); everything batched behind it runs as before, the prior's share on the second stream.  Of
    // prior_begin() only the zeroing of the prior attention's memory gradient is needed (and its memsets of dhp / dc / dlz
    // on the second stream would race with the launch, which uses dhp as a hand-off buffer).
    PbParams pb{};                                   // (dmem_p: zeroed at the entry)
    pb.wt_dhh = wt_dhh; pb.wt_datt = wt_datt; pb.wt_dih = wt_dih; pb.wt_phh = wt_phh; pb.wt_pih = wt_pih; pb.wt_pml = wt_pml;
    pb.att_v = P(TP_DEC_ATT_V);
    pb.encproj = sv + L.encproj_d; pb.mem = mem; pb.qd = qd; pb.attn_w = attn_w; pb.gru_save = gru_save; pb.hprev_d = hprev_d;
    pb.lstm_save = lstm_save; pb.c_all = c_all; pb.p_logs = p_logs; pb.eps_p = eps_p; pb.mem_lens = mem_lens;
    pb.d_out = d_out; pb.d_p_z = d_p_z; pb.d_p_means = d_p_means; pb.d_p_logs = d_p_logs;
    pb.dgi = dgi; pb.dgh = dgh; pb.dqd = dqd; pb.dctx = drnn;      // [N, Tc, E] slots in the (later) d rnn_input buffer
    pb.dencproj = dencproj; pb.dmem = dmem; pb.dvpart = dvpart; pb.dgates = dgates; pb.dml_all = dml_all; pb.dhp = dhp;
    pb.dctx_part = sc + L.pd_part; pb.dhp_part = pb.dctx_part + 4L * N * E; pb.dml_part = pb.dhp_part + 4L * N * E;
    pb.cnt = (unsigned*)(sc + L.pd_cnt_b);
    pb.dqd_part = sc + L.pd_dqd; pb.ctx = rnn_d + E;
    pb.N = N; pb.Tc = Tc; pb.S = S; pb.E = E; pb.H = H; pb.A = A;
    ACVAE_TRY(acvae::decode_persist_bwd(pb, st.s, flags));
    dv_rows = N * acvae::decode_persist_bwd_rc_splits(S);
    ACVAE_TRY(dec_memgrad());
    if (fork.on()) ACVAE_TRY(Fork::edge(st.s, sp.s));     // the prior's batched work reads what the launch wrote
    ACVAE_TRY(prior_memgrad());
    if (!defer) {
      ACVAE_TRY(dec_params(st, tn, dpart));
      ACVAE_TRY(prior_params());
    }
  } else {                     // independent chains: feed both queues step by step
    ACVAE_TRY(dec_begin());
    ACVAE_TRY(prior_begin());
    for (int t = Tc - 1; t >= 0; --t) {
      ACVAE_TRY(dec_bptt(t));
      ACVAE_TRY(prior_bptt(t));
    }
    ACVAE_TRY(dec_memgrad());
    ACVAE_TRY(prior_memgrad());
    if (!defer) {
      ACVAE_TRY(dec_params(st, tn, dpart));
      ACVAE_TRY(prior_params());
    }
  }
  ACVAE_TRY(fork.join());      // deferred mode: the second stream holds only the critical part so far
  ACVAE_TRY(acvae::add_rows(dmem, E, dmem_p, E, N * S, E, st));
  // ---- memory gradient back through the optional ln projection
  if (has_ln) {
    ACVAE_TRY(transp(P(TP_LN_W), Eenc, wt_ln, E, E, Eenc, st));                          // [Eenc][E]
    ACVAE_TRY(gemm(dmem, E, wt_ln, E, nullptr, d_mem_in, Eenc, N * S, Eenc, E, 0, st));
    ACVAE_TRY(gemm_tn(dmem, E, mem_in, Eenc, G(TP_LN_W), Eenc, E, Eenc, N * S, tn, st));
    ACVAE_TRY(acvae::colsum2(dmem, N * S, E, dpart, G(TP_LN_B), nullptr, 0, st));
  } else {
    ACVAE_TRY(acvae::copy_rows(d_mem_in, E, dmem, E, N * S, E, st));
  }
  if (defer) {                 // d_q_z on the first stream; everything else behind the first stream's work so far, on the second
    ACVAE_TRY(dec_dz(st));
    ACVAE_TRY(Fork::edge(st.s, sp.s));
    if (!heads_on_aux) ACVAE_TRY(heads_params(sp, tn_p, dpart_p));
    ACVAE_TRY(dec_params(sp, tn_p, dpart_p, false));
    ACVAE_TRY(prior_params());
  }
  return ACVAE_OK;
}

extern "C" int acvae_caps_to_long(const float* caps, int64_t* out, int64_t n, void* stream) {
  if (!caps || !out || n <= 0) return ACVAE_EINVAL;
  return acvae::caps_to_long(caps, out, (long)n, (hipStream_t)stream);
}

// ==========================================================================================
// single decode steps (inference only): the per-step API of the reference's pnet / decoder modules, used by the
// beam search (models/vae_model.py:896-995) and by anyone calling the sub-modules directly
// ==========================================================================================
namespace {
struct StepLayout { long skws, encproj, rnn, q, gates, gh, ml, words, attws, attws_bytes, total; };
int step_layout(int N, int S, int E, int H, int A, int V, StepLayout& L) {
  if (N <= 0 || S <= 0 || E <= 0 || H <= 0 || A <= 0 || V <= 1) return ACVAE_EINVAL;
  Bump b;
  L.skws = b.take(acvae_skinny_ws_floats());
  L.encproj = b.take((long)N * S * (A > E ? A : E));
  L.rnn = b.take((long)N * 3 * E);
  L.q = b.take((long)N * (A > E ? A : E));
  L.gates = b.take((long)N * 4 * E);
  L.gh = b.take((long)N * 3 * H);
  L.ml = b.take((long)N * 2 * E);
  L.words = b.take((long)N * 2);
  // workspace of the split-over-frames attention (acvae_attn_fwd: few query rows); its counters are zeroed by every entry point
  L.attws_bytes = acvae_attn_fwd_workspace_bytes(N, 1, S, A, E);
  const long w2 = acvae_attn_fwd_workspace_bytes(N, 1, S, E, E);
  if (w2 > L.attws_bytes) L.attws_bytes = w2;
  L.attws = b.take(L.attws_bytes / 4 + 64);
  L.total = b.off;
  return ACVAE_OK;
}
inline int step_attws_reset(float* sc, const StepLayout& L, hipStream_t st) {
  return L.attws_bytes > 0 ? zero(sc + L.attws, 256, st) : ACVAE_OK;
}
}  // namespace

extern "C" int64_t acvae_step_scratch_bytes(int N, int S, int E, int H, int A, int V) {
  StepLayout L;
  return step_layout(N, S, E, H, A, V, L) == ACVAE_OK ? L.total * 4 : -1;
}

// encproj = mem . W[:, hs_dec:]^T + b for the decoder attention (which = 0) or the prior attention (which = 1)
extern "C" int acvae_attn_precompute(const void* const* params, int which, const float* mem, float* encproj, int N,
                                     int S, int E, int H, int A, void* stream) {
  if (!params || !mem || !encproj || N <= 0 || S <= 0) return ACVAE_EINVAL;
  Ctx st{(hipStream_t)stream, nullptr};
  auto P = [&](int i) { return (const float*)params[i]; };
  if (which == 0) return gemm(mem, E, P(TP_DEC_ATT_W) + H, E + H, P(TP_DEC_ATT_B), encproj, A, N * S, A, E, 0, st);
  return gemm(mem, E, P(TP_P_ATT_W) + E, 2 * E, P(TP_P_ATT_B), encproj, E, N * S, E, E, 0, st);
}

namespace {
// One prior / decoder step over R = Nm * Tq rows: row r = n * Tq + j attends over memory n (Tq = 1: one memory per row,
// the sub-module API; Tq = beam: the beams of a clip share its memory, no replicated copy).
int prior_step(const void* const* params, const int64_t* word, const float* mem, const int64_t* mem_lens, const float* ep,
               const float* h_prev, const float* c_prev, const float* last_z, const float* eps, float* mean, float* logv,
               float* z, float* h_out, float* c_out, float* attw, float* sc, const StepLayout& L, int Nm, int Tq, int S,
               int E, int V, const Ctx& st) {
  auto P = [&](int i) { return (const float*)params[i]; };
  const int N = Nm * Tq, Hp = E;
  float* rnn = sc + L.rnn;
  float* q = sc + L.q;
  float* gates = sc + L.gates;
  float* ml = sc + L.ml;
  ACVAE_TRY(acvae::embed_gather(word, 1, P(TP_P_EMB), V, rnn, 3 * E, N, E, st));
  ACVAE_TRY(gemm(rnn, 3 * E, P(TP_P_ATT_W), 2 * E, nullptr, q, E, N, E, E, 0, st));
  ACVAE_TRY(acvae_attn_fwd(q, (long)Tq * E, E, ep, mem, mem_lens, P(TP_P_ATT_V), rnn + E, (long)Tq * 3 * E, 3 * E, attw,
                           (long)Tq * S, S, Nm, Tq, S, E, E, L.attws_bytes > 0 ? sc + L.attws : nullptr, L.attws_bytes, st, 0));
  ACVAE_TRY(acvae::copy_rows(rnn + 2 * E, 3 * E, last_z, E, N, E, st));
  ACVAE_TRY(gemm(rnn, 3 * E, P(TP_P_WIH), 3 * E, P(TP_P_BIH), gates, 4 * Hp, N, 4 * Hp, 3 * E, 0, st));
  ACVAE_TRY(gemm(h_prev, Hp, P(TP_P_WHH), Hp, P(TP_P_BHH), gates, 4 * Hp, N, 4 * Hp, Hp, 1, st));
  ACVAE_TRY(acvae::lstm_fwd(gates, 4 * Hp, c_prev, Hp, h_out, Hp, c_out, Hp, nullptr, 0, N, Hp, st));
  ACVAE_TRY(gemm(h_out, Hp, P(TP_P_ML_W), Hp, P(TP_P_ML_B), ml, 2 * E, N, 2 * E, Hp, 0, st));
  return acvae_reparam_fwd(ml, 2 * E, eps, E, mean, logv, z, E, nullptr, 0, N, E, st);
}

int decoder_step(const void* const* params, const int64_t* word, const float* h_prev, const float* mem,
                 const int64_t* mem_lens, const float* ed, const float* z, float* logits, float* h_out, float* attw,
                 float* rnn_input, float* sc, const StepLayout& L, int Nm, int Tq, int S, int E, int H, int A, int V,
                 const Ctx& st) {
  auto P = [&](int i) { return (const float*)params[i]; };
  const int N = Nm * Tq;
  float* q = sc + L.q;
  float* gi = sc + L.gates;
  float* gh = sc + L.gh;
  ACVAE_TRY(acvae::embed_gather(word, 1, P(TP_DEC_EMB), V, rnn_input, 3 * E, N, E, st));
  ACVAE_TRY(gemm(h_prev, H, P(TP_DEC_ATT_W), E + H, nullptr, q, A, N, A, H, 0, st));
  ACVAE_TRY(acvae_attn_fwd(q, (long)Tq * A, A, ed, mem, mem_lens, P(TP_DEC_ATT_V), rnn_input + E, (long)Tq * 3 * E, 3 * E,
                           attw, (long)Tq * S, S, Nm, Tq, S, A, E, L.attws_bytes > 0 ? sc + L.attws : nullptr, L.attws_bytes, st, 0));
  ACVAE_TRY(acvae::copy_rows(rnn_input + 2 * E, 3 * E, z, E, N, E, st));
  ACVAE_TRY(gemm(rnn_input, 3 * E, P(TP_DEC_WIH), 3 * E, P(TP_DEC_BIH), gi, 3 * H, N, 3 * H, 3 * E, 0, st));
  ACVAE_TRY(gemm(h_prev, H, P(TP_DEC_WHH), H, P(TP_DEC_BHH), gh, 3 * H, N, 3 * H, H, 0, st));
  ACVAE_TRY(acvae::gru_fwd(gi, 3 * H, gh, 3 * H, h_prev, H, h_out, H, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, N,
                           H, st));
  return gemm(h_out, H, P(TP_DEC_CLS_W), H, P(TP_DEC_CLS_B), logits, V, N, V, H, 0, st);
}
}  // namespace

extern "C" int acvae_prior_step_fwd(const void* const* params, const int64_t* word, const float* mem,
                                    const int64_t* mem_lens, const float* encproj_p, const float* h_prev,
                                    const float* c_prev, const float* last_z, const float* eps, float* mean, float* logv,
                                    float* z, float* h_out, float* c_out, float* attw, void* scratch_v,
                                    int64_t scratch_bytes, int N, int S, int E, int V, void* stream) {
  StepLayout L;
  ACVAE_TRY(step_layout(N, S, E, E, E, V, L));
  if (!params || !word || !mem || !mem_lens || !h_prev || !c_prev || !last_z || !eps || !mean || !logv || !z ||
      !h_out || !c_out || !attw || !scratch_v)
    return ACVAE_EINVAL;
  if (scratch_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  float* sc = (float*)scratch_v;
  Ctx st{(hipStream_t)stream, sc + L.skws};
  ACVAE_TRY(acvae_skinny_ws_reset(st.skws, st.s));
  ACVAE_TRY(step_attws_reset(sc, L, st.s));
  const float* ep = encproj_p;
  if (!ep) {
    ACVAE_TRY(acvae_attn_precompute(params, 1, mem, sc + L.encproj, N, S, E, E, E, stream));
    ep = sc + L.encproj;
  }
  return prior_step(params, word, mem, mem_lens, ep, h_prev, c_prev, last_z, eps, mean, logv, z, h_out, c_out, attw, sc,
                    L, N, 1, S, E, V, st);
}

extern "C" int acvae_decoder_step_fwd(const void* const* params, const int64_t* word, const float* h_prev,
                                      const float* mem, const int64_t* mem_lens, const float* encproj_d, const float* z,
                                      float* logits, float* h_out, float* attw, float* rnn_input, void* scratch_v,
                                      int64_t scratch_bytes, int N, int S, int E, int H, int A, int V, void* stream) {
  StepLayout L;
  ACVAE_TRY(step_layout(N, S, E, H, A, V, L));
  if (!params || !word || !h_prev || !mem || !mem_lens || !z || !logits || !h_out || !attw || !rnn_input || !scratch_v)
    return ACVAE_EINVAL;
  if (scratch_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  float* sc = (float*)scratch_v;
  Ctx st{(hipStream_t)stream, sc + L.skws};
  ACVAE_TRY(acvae_skinny_ws_reset(st.skws, st.s));
  ACVAE_TRY(step_attws_reset(sc, L, st.s));
  const float* ed = encproj_d;
  if (!ed) {
    ACVAE_TRY(acvae_attn_precompute(params, 0, mem, sc + L.encproj, N, S, E, H, A, stream));
    ed = sc + L.encproj;
  }
  return decoder_step(params, word, h_prev, mem, mem_lens, ed, z, logits, h_out, attw, rnn_input, sc, L, N, 1, S, E, H,
                      A, V, st);
}

// ==========================================================================================
// validation beam search as ONE call (models/vae_model.py:896-995): all clips advance together, the
// beams of a clip share its memory, and nothing but kernels is enqueued per step.  Instead of re-gathering the word and
// attention-weight histories by prev_word_inds every step (:917-921, :979), each step's parent row, word and weights are
// kept and beam 0 is traced back once at the end, which yields the same seqs[0] / attn_weights[0] (:990-995).
// ==========================================================================================
namespace {
struct BeamLayout { long step, encd, encp, h, hp, cp, lz, mean, logv, z, h2, hp2, cp2, attp, logits, rnn, lse, topk, scores,
                    attw, words, total; };
int beam_layout(int N, int beam, int T, int S, int E, int H, int A, int V, BeamLayout& L) {
  if (N <= 0 || beam <= 0 || T <= 0) return ACVAE_EINVAL;
  StepLayout sl;
  const long R = (long)N * beam;
  if (R > (1L << 20)) return ACVAE_EUNSUPPORTED;
  ACVAE_TRY(step_layout((int)R, S, E, H, A, V, sl));
  Bump b;
  L.step = b.take(sl.total);
  L.encd = b.take((long)N * S * A);
  L.encp = b.take((long)N * S * E);
  L.h = b.take(R * H); L.hp = b.take(R * E); L.cp = b.take(R * E); L.lz = b.take(R * E);
  L.mean = b.take(R * E); L.logv = b.take(R * E); L.z = b.take(R * E);
  L.h2 = b.take(R * H); L.hp2 = b.take(R * E); L.cp2 = b.take(R * E);
  L.attp = b.take(R * S);
  L.logits = b.take(R * V);
  L.rnn = b.take(R * 3 * E);
  L.lse = b.take(R);
  L.topk = b.take(R);
  L.scores = b.take(R * V);
  L.attw = b.take((long)T * R * S);
  L.words = b.take(2 * ((long)(3 * T + 1) * R));        // int64: word [R], then per step idx / parent / word [T][R]
  L.total = b.off;
  return ACVAE_OK;
}

struct GatherJob { const float* src; float* dst; int width; };
__global__ __launch_bounds__(256) void beam_gather_kernel(GatherJob a, GatherJob b, GatherJob c, GatherJob d,
                                                          const int64_t* __restrict__ parent) {
  const GatherJob j = blockIdx.y == 0 ? a : blockIdx.y == 1 ? b : blockIdx.y == 2 ? c : d;
  const long r = blockIdx.x, p = parent[r];
  for (int i = threadIdx.x; i < j.width; i += blockDim.x) j.dst[r * j.width + i] = j.src[p * j.width + i];
}

// one workgroup per clip: follow beam 0's parents from the last step to the first
__global__ __launch_bounds__(256) void beam_trace_kernel(const int64_t* __restrict__ parent, const int64_t* __restrict__ word,
                                                         const float* __restrict__ attw, int64_t* __restrict__ seqs,
                                                         float* __restrict__ attw_out, long hist_stride, int R, int beam, int T,
                                                         int S) {
  const int n = blockIdx.x;
  long r = (long)n * beam;
  for (int t = T - 1; t >= 0; --t) {
    const long p = parent[t * hist_stride + r];
    if (threadIdx.x == 0) seqs[(long)n * T + t] = word[t * hist_stride + r];
    const float* w = attw + ((long)t * R + p) * S;
    for (int s = threadIdx.x; s < S; s += blockDim.x) attw_out[((long)n * S + s) * T + t] = w[s];
    r = p;
  }
}
__global__ void fill_words_kernel(int64_t* w, int64_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = v;
}
}  // namespace

extern "C" int64_t acvae_beam_search_scratch_bytes(int N, int beam, int max_length, int S, int E, int H, int A, int V) {
  BeamLayout L;
  return beam_layout(N, beam, max_length, S, E, H, A, V, L) == ACVAE_OK ? L.total * 4 : -1;
}

extern "C" int acvae_beam_search(const void* const* params, const float* mem, const int64_t* mem_lens, const float* eps,
                                 int64_t start_idx, int64_t* seqs, float* attw_out, void* scratch_v, int64_t scratch_bytes,
                                 int N, int beam, int max_length, int S, int E, int H, int A, int V, void* stream) {
  BeamLayout L;
  ACVAE_TRY(beam_layout(N, beam, max_length, S, E, H, A, V, L));
  if (!params || !mem || !mem_lens || !eps || !seqs || !attw_out || !scratch_v) return ACVAE_EINVAL;
  if (start_idx < 0 || start_idx >= V || beam > 64 || H != E) return ACVAE_EINVAL;   // the prior LSTM is E wide
  if (scratch_bytes < L.total * 4) return ACVAE_EWORKSPACE;
  StepLayout SL;
  const int R = N * beam, T = max_length;
  ACVAE_TRY(step_layout(R, S, E, H, A, V, SL));
  float* sc = (float*)scratch_v;
  float* ssc = sc + L.step;
  Ctx st{(hipStream_t)stream, ssc + SL.skws};
  ACVAE_TRY(acvae_skinny_ws_reset(st.skws, st.s));
  ACVAE_TRY(step_attws_reset(ssc, SL, st.s));
  ACVAE_TRY(acvae_attn_precompute(params, 0, mem, sc + L.encd, N, S, E, H, A, stream));
  ACVAE_TRY(acvae_attn_precompute(params, 1, mem, sc + L.encp, N, S, E, E, E, stream));
  float *h = sc + L.h, *hp = sc + L.hp, *cp = sc + L.cp, *lz = sc + L.lz;
  float *h2 = sc + L.h2, *hp2 = sc + L.hp2, *cp2 = sc + L.cp2, *z = sc + L.z;
  float* topk = sc + L.topk;
  int64_t* word = (int64_t*)(sc + L.words);
  int64_t* hist = word + R;                              // [T][3][R]: flat index, parent row, word
  ACVAE_TRY(zero(h, (long)R * H, st));
  ACVAE_TRY(zero(hp, (long)R * E, st));
  ACVAE_TRY(zero(cp, (long)R * E, st));
  ACVAE_TRY(zero(lz, (long)R * E, st));
  ACVAE_TRY(zero(topk, R, st));
  hipLaunchKernelGGL(fill_words_kernel, dim3((R + 255) / 256), dim3(256), 0, st.s, word, start_idx, R);
  const int64_t* w_t = word;
  for (int t = 0; t < T; ++t) {
    float* attw_t = sc + L.attw + (long)t * R * S;
    int64_t* idx_t = hist + (long)t * 3 * R;
    int64_t* par_t = idx_t + R;
    int64_t* nxt_t = par_t + R;
    ACVAE_TRY(prior_step(params, w_t, mem, mem_lens, sc + L.encp, hp, cp, lz, eps + (long)t * R * E, sc + L.mean,
                         sc + L.logv, z, hp2, cp2, sc + L.attp, ssc, SL, N, beam, S, E, V, st));
    ACVAE_TRY(decoder_step(params, w_t, h, mem, mem_lens, sc + L.encd, z, sc + L.logits, h2, attw_t, sc + L.rnn, ssc, SL,
                           N, beam, S, E, H, A, V, st));
    ACVAE_TRY(acvae_row_logsoftmax_argmax(sc + L.logits, V, V, nullptr, nullptr, sc + L.lse, 1, 1, R, 1, V, stream));
    ACVAE_TRY(acvae_logprob_add(sc + L.logits, V, sc + L.lse, topk, sc + L.scores, R, V, stream));
    ACVAE_TRY(acvae_topk_flat_batched(sc + L.scores, (int64_t)beam * V, (int64_t)beam * V, beam, V, topk, idx_t, par_t,
                                      nxt_t, N, beam, stream));
    if (t + 1 < T) {                                     // vae_model.py:961-968: next step's states follow their parents
      hipLaunchKernelGGL(beam_gather_kernel, dim3(R, 4), dim3(256), 0, st.s, GatherJob{h2, h, H}, GatherJob{hp2, hp, E},
                         GatherJob{cp2, cp, E}, GatherJob{z, lz, E}, par_t);
      w_t = nxt_t;
    }
  }
  // hist rows are [idx | parent | word] per step: strided views for the trace
  hipLaunchKernelGGL(beam_trace_kernel, dim3(N), dim3(256), 0, st.s, hist + R, hist + 2 * R, sc + L.attw, seqs, attw_out,
                     3L * R, R, beam, T, S);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
