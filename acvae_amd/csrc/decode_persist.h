// decode_persist.hip: the teacher-forced decode loop as one persistent launch (internal C++ interface; the C ABI entry is
// acvae_decode_fwd, which takes this path when acvae::decode_persist_ok says so).
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
  int N, Tc, S, E, H, A;
  int n_d1, n_d3, n_p1, n_p2;      // set by the launcher
  int att_resident;                // set by the launcher: the attention keeps its clip's memory on the CU
};

namespace acvae {
bool decode_persist_ok(int N, int Tc, int S, int E, int H, int A);
long decode_persist_counter_words(int Tc);
int decode_persist_fwd(PdParams p, hipStream_t st);
}  // namespace acvae
