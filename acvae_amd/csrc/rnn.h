// Internal C++ launcher API of rnn.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace acvae {
int caps_to_long(const float* caps, int64_t* out, long n, hipStream_t st);
int select_word(const int64_t* caps, long ld_caps, const int64_t* seqs, long ld_seqs, int64_t* words, long ld_words,
                int t, int use_caps, int start_idx, int N, hipStream_t st);
int finish_rows(int64_t* seqs, long ld_seqs, uint8_t* unfinished, int t, int end_idx, int N, hipStream_t st);
int embed_gather(const int64_t* words, long w_stride, const float* table, int V, float* out, long ld_out, int rows,
                 int E, hipStream_t st);
int embed_scatter(const int64_t* words_contig, const float* d, long ld_d, float* dtable, int V, int rows, int E,
                  hipStream_t st);
int dropout_rows(float* x, long ld_n, long ld_t, const uint8_t* keep, long k_sn, long k_st, float scale, int N, int cnt,
                 int E, hipStream_t st);
int gather_words(const int64_t* src, long s_sn, long s_st, int64_t* dst, int N, int T, hipStream_t st);
int gru_fwd(const float* gi, long ld_gi, const float* gh, long ld_gh, const float* hp, long ld_hp, float* ho,
            long ld_ho, float* oseq, long ld_os, float* save, long ld_sv, float* hps, long ld_hps, const int64_t* lens,
            int t, int N, int H, hipStream_t st);
int gru_bwd(const float* dh_next, long ld_dn, const float* d_out, long ld_do, const float* save, long ld_sv,
            const float* hps, long ld_hps, float* dgi, long ld_dgi, float* dgh, long ld_dgh, float* dh_prev, long ld_dp,
            const int64_t* lens, int t, int N, int H, hipStream_t st);
int lstm_fwd(const float* gates, long ld_g, const float* cp, long ld_cp, float* ho, long ld_ho, float* co, long ld_co,
             float* save, long ld_sv, int N, int H, hipStream_t st);
int lstm_bwd(const float* dh, long ld_dh, const float* dc_next, long ld_dc, const float* save, long ld_sv,
             const float* cp, long ld_cp, float* dgates, long ld_dg, float* dc_prev, long ld_dcp, int N, int H,
             hipStream_t st);
int pool_fwd(const float* x, long ld_n, long ld_t, const int64_t* lens, float* out, int* argmax, int N, int T, int C,
             hipStream_t st);
int pool_bwd(const float* d, const int64_t* lens, const int* argmax, float* dx, long ld_n, long ld_t, int accumulate,
             int N, int T, int C, hipStream_t st);
int colsum_rows(const float* x, long ld, int rows, int cols, float* out, int accumulate, hipStream_t st);
int add_rows(float* dst, long ld_d, const float* src, long ld_s, int rows, int cols, hipStream_t st);
int copy_rows(float* dst, long ld_d, const float* src, long ld_s, int rows, int cols, hipStream_t st);
// up to 8 copy_rows jobs (src == nullptr: zero fill) in ONE launch; the jobs must not overlap
struct CopyRowsBatch {
  static constexpr int MAXJ = 8;
  float* dst[MAXJ]; const float* src[MAXJ]; long ld_d[MAXJ], ld_s[MAXJ]; int rows[MAXJ], cols[MAXJ]; int n = 0;
  void add(float* d, long ldd, const float* s, long lds, int r, int c) {
    dst[n] = d; src[n] = s; ld_d[n] = ldd; ld_s[n] = lds; rows[n] = r; cols[n] = c; ++n;
  }
};
int copy_rows_batch(const CopyRowsBatch& b, hipStream_t st);
// up to 10 regions of 32-bit words zeroed in ONE launch: the tickets / arrival counters / accumulators a composite call starts
// from (each was a 5-us hipMemsetAsync in front of the call's first kernel; words % 1 == 0, 4-byte aligned)
struct ZeroBatch {
  static constexpr int MAXJ = 10;
  void* p[MAXJ]; long words[MAXJ]; int n = 0;        // n > MAXJ: more regions were added than fit (zero_batch refuses)
  void add(void* ptr, long w) { if (ptr && w > 0) { if (n < MAXJ) { p[n] = ptr; words[n] = w; } ++n; } }
};
int zero_batch(const ZeroBatch& b, hipStream_t st);
}  // namespace acvae
