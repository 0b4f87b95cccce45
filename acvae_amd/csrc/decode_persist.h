// decode_persist.hip: the teacher-forced decode loop as one persistent launch (internal C++ interface; the C ABI entry is
// acvae_decode_fwd, which takes this path when acvae::decode_persist_ok says so: the shape is one the kernel takes AND its
// whole grid is resident on the current device at once; `flags`: ACVAE_FLAG_* of include/acvae_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

enum { PD_C_D1Q, PD_C_D1H, PD_C_D2, PD_C_D3, PD_C_P1, PD_C_P2, PD_C_COUNT };   // arrival counters: [role][step]

struct PdParams {
  // decoder (models/decoder.py:175-203): attention h2attn [A, H + E] (query half = columns 0..H), GRU weight_hh [3H, H] /
  // bias_hh, weight_ih [3H, 3E] (context part = columns E..2E), attention vector v [A]
  const float *w_att, *w_hh, *b_hh, *w_ih, *att_v;
  // prior (models/text_encoder.py:247-268): LSTM weight_ih [4Hp, 3E] (last_z part = columns 2E..3E), weight_hh [4Hp, Hp] /
  // bias_hh, mean_log_out [2E, Hp] / bias
  const float *pw_ih, *pw_hh, *pb_hh, *w_ml, *b_ml;
  // produced in front of the launch
  const float *encproj, *mem;      // [N, S, A], [N, S, E]
  const int64_t* mem_lens;         // [N]
  const float *gi, *gates_p;       // hoisted input projections [N, Tc, 3H], [N, Tc, 4Hp]
  const float *eps_p, *zeros;      // [Tc, N, E]; N x max(H, Hp) zeros
  // written by the launch ([N, Tc, .] batch-major like the reference's outputs)
  float *qd, *gh, *rnn_d, *attn_w, *outputs, *gru_save, *hprev_d;
  float *rnn_p, *hp_all, *c_all, *lstm_save, *p_means, *p_logs, *p_z;
  unsigned* cnt;                   // decode_persist_counter_words(Tc) words, zeroed by the launcher
  unsigned* abort_word;            // set by the launcher (last counter word)
  unsigned spin_limit;             // set by the launcher: polls after which a wait gives up and raises abort_word
  int N, Tc, S, E, H, A;
  int n_d1, n_d3, n_p1, n_p2;      // set by the launcher
  int att_resident;                // set by the launcher: the attention keeps its clip's memory on the CU
};

enum { PB_C_RA, PB_C_RB, PB_C_RC, PB_C_PA, PB_C_PB, PB_C_COUNT };

// backward through time of the same loop (acvae_decode_bwd)
struct PbParams {
  // transposed weights (acvae_decode_bwd's scratch): [H][3H], [H+E][A], [3E][3H], [Hp][4Hp], [3E][4Hp], [Hp][2E]
  const float *wt_dhh, *wt_datt, *wt_dih, *wt_phh, *wt_pih, *wt_pml;
  const float* att_v;
  // saved by the forward
  const float *encproj, *mem, *qd, *attn_w, *gru_save, *hprev_d, *lstm_save, *c_all, *p_logs, *eps_p;
  const int64_t* mem_lens;
  // upstream gradients ([N, Tc, .]; the three of the prior may be NULL)
  const float *d_out, *d_p_z, *d_p_means, *d_p_logs;
  // written by the launch (dctx: [N, Tc, E], one slot per step)
  float *dgi, *dgh, *dqd, *dctx, *dencproj, *dmem, *dvpart, *dgates, *dml_all, *dhp;
  // clips of more than 64 frames: the attention role runs as rc_splits workgroups per clip (set by the launcher); their shares
  // of d qd [rc_splits][N][Tc][A] (RA adds them and writes the sum to dqd) and the forward's rnn_d ([N][Tc][3E]: ctx = columns E..2E)
  float* dqd_part;
  const float* ctx;
  int rc_splits;
  // K-split partials handed over inside the launch (scratch, decode_persist_bwd_part_floats): dctx [ks_rb][N][E],
  // dhp [ks_pa][N][Hp], dml [ks_pa][N][2E]
  float *dctx_part, *dhp_part, *dml_part;
  unsigned* cnt;
  unsigned* abort_word;
  unsigned spin_limit;
  int N, Tc, S, E, H, A;
  int n_ra, n_rb, n_pa, n_pb;
  int ks_rb, ks_pa;                // K-splits of the RB / PA products (set by the launcher)
};

namespace acvae {
bool decode_persist_bwd_ok(int N, int Tc, int S, int E, int H, int A);
// internal flag beside the public ACVAE_FLAG_* of a launch: the caller has zeroed p.cnt in this stream (with its other tickets,
// one launch: rnn.h ZeroBatch) - the launcher skips its own memset
#define ACVAE_FLAG_INT_CNT_ZEROED (1 << 16)
long decode_persist_bwd_counter_words(int Tc);
int decode_persist_bwd_rc_splits(int S);     // attention workgroups per clip (64 frames each)
int decode_persist_bwd(PbParams p, hipStream_t st, int flags);
long decode_persist_bwd_part_floats(int N, int E, int H);
bool decode_persist_ok(int N, int Tc, int S, int E, int H, int A);
long decode_persist_counter_words(int Tc);
int decode_persist_fwd(PdParams p, hipStream_t st, int flags);
}  // namespace acvae


// ---------------------------------------------------------------------------------------------------------------------
// The posterior's packed bidirectional GRU (PosteriorRNN_hybrid, models/text_encoder.py:189-191) as one persistent launch per
// pass: both directions, all Tc steps.  A workgroup owns 32 hidden units of one direction: forward it forms the three gate
// tiles gh = h . Whh^T + b of its units (weight fragments resident in registers), runs the GRU cell and hands its slice of h to
// the direction's other workgroups; backward it forms dh = dh . z + dgh . Whh for its units, runs the cell backward and
// hands over its slice of dgh.  One hand-off per step and direction (protocol of decode_persist.hip).
struct PqParams {
  const float* w_hh[2];        // [3Hq][Hq] per direction
  const float* b_hh[2];        // [3Hq]
  const float* gi[2];          // hoisted input projections [N][Tc][3Hq]
  const int64_t* lens1;        // [N]
  float* hid;                  // [N][Tc][2Hq]
  float* save[2];              // [N][Tc][4Hq] = r | z | n | gh_n
  float* hprev[2];             // [N][Tc][Hq]
  float* hbuf;                 // [2 directions][2 step parities][N][Hq]: the state in flight, ZEROED by the caller (step 0 reads parity 1)
  unsigned* cnt;               // posterior_persist_counter_words(Tc), zeroed by the launcher
  unsigned* abort_word;        // set by the launcher
  unsigned spin_limit;         // set by the launcher
  int N, Tc, Hq;
};
struct PqbParams {
  const float* wt[2];          // transposed weight_hh: [Hq][3Hq]
  const float* dhid;           // [N][Tc][2Hq]
  const float* save[2];
  const float* hprev[2];
  const int64_t* lens1;
  float* dgi[2];               // [N][Tc][3Hq]
  float* dgh[2];               // [N][Tc][3Hq] (handed over inside the launch)
  unsigned* cnt;
  unsigned* abort_word;
  unsigned spin_limit;
  int N, Tc, Hq;
};
namespace acvae {
bool posterior_persist_ok(int N, int Tc, int Hq);
long posterior_persist_counter_words(int Tc);
int posterior_persist_fwd(PqParams p, hipStream_t st, int flags);
int posterior_persist_bwd(PqbParams p, hipStream_t st, int flags);
}  // namespace acvae
