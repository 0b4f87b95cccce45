// Elementwise / gather kernels of the recurrent part of the path: word-embedding gather and its
// deterministic scatter-add, GRU and LSTM cells (torch.nn.GRU / nn.LSTM formulas, gate order r,z,n /
// i,f,g,o) forward and backward, masked mean+max pooling (utils/train_util.py:208-231), token helpers.
// References: models/decoder.py:183-190 (GRU step), models/text_encoder.py:187-191 (packed BiGRU),
// :253 (LSTM step), models/vae_model.py:722-725,818-848.
#include "common.h"
#include "rnn.h"

namespace {

constexpr int TH = 256;
inline int grid1(long n) {
  long b = (n + TH - 1) / TH;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------- tokens
__global__ void caps_to_long_kernel(const float* __restrict__ caps, int64_t* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < n; i += (long)gridDim.x * TH) out[i] = (int64_t)caps[i];
}
// word for step t: teacher (caps[n,t]) if use_caps, else <start> at t==0, else seqs[n,t-1]   (vae_model.py:826-833)
__global__ void select_word_kernel(const int64_t* __restrict__ caps, long ld_caps, const int64_t* __restrict__ seqs,
                                   long ld_seqs, int64_t* __restrict__ words, long ld_words, int t, int use_caps,
                                   int start_idx, int N) {
  const int n = blockIdx.x * TH + threadIdx.x;
  if (n >= N) return;
  int64_t w;
  if (use_caps) w = caps[n * ld_caps + t];
  else if (t == 0) w = start_idx;
  else w = seqs[n * ld_seqs + t - 1];
  words[n * ld_words + t] = w;
}
// inference bookkeeping (vae_model.py:711-716): unfinished &= (w != end); finished rows emit <end>
__global__ void finish_rows_kernel(int64_t* __restrict__ seqs, long ld_seqs, uint8_t* __restrict__ unfinished, int t,
                                   int end_idx, int N) {
  const int n = blockIdx.x * TH + threadIdx.x;
  if (n >= N) return;
  const bool ut = seqs[n * ld_seqs + t] != end_idx;
  const bool u = (t == 0) ? ut : (unfinished[n] && ut);
  unfinished[n] = u;
  if (!u) seqs[n * ld_seqs + t] = end_idx;
}

// ---------------------------------------------------------------- embedding
__global__ void embed_gather_kernel(const int64_t* __restrict__ words, long w_stride, const float* __restrict__ table,
                                    int V, float* __restrict__ out, long ld_out, int rows, int E) {
  const long total = (long)rows * E;
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < total; i += (long)gridDim.x * TH) {
    const int r = (int)(i / E), e = (int)(i % E);
    long w = words[r * w_stride];
    if (w < 0 || w >= V) w = 0;  // never read out of bounds
    out[r * ld_out + e] = table[w * E + e];
  }
}
// nn.Dropout on rows (n, t), t < cnt: x *= keep ? 1/(1-p) : 0   (models/decoder.py:33,184: the decoder's word-embedding
// dropout; the keep mask is drawn by the host on the CPU generator in the reference's call order)
__global__ void dropout_rows_kernel(float* __restrict__ x, long ld_n, long ld_t, const uint8_t* __restrict__ keep,
                                    long k_sn, long k_st, float scale, int N, int cnt, int E) {
  const long total = (long)N * cnt * E;
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < total; i += (long)gridDim.x * TH) {
    const int e = (int)(i % E);
    const long r = i / E;
    const int t = (int)(r % cnt), n = (int)(r / cnt);
    float* p = x + n * ld_n + t * ld_t + e;
    *p = keep[n * k_sn + t * k_st + e] ? *p * scale : 0.f;
  }
}
// dTable[w] += sum over rows r with words[r]==w of d[r] — one workgroup per row, the FIRST occurrence of a
// word sums all its duplicates in row order: deterministic, no atomics.  dTable must be zeroed by the caller.
__global__ void embed_scatter_kernel(const int64_t* __restrict__ words, const float* __restrict__ d, long ld_d,
                                     float* __restrict__ dtable, int V, int rows, int E) {
  extern __shared__ unsigned flags[];   // bitmap over rows: 1 where words[q] == w  (rows/32 + 1 words) + 1 word "earlier"
  const int r = blockIdx.x;
  const long w = words[r];
  if (w < 0 || w >= V) return;
  const int nw = (rows + 31) / 32;
  for (int i = threadIdx.x; i <= nw; i += TH) flags[i] = 0u;
  __syncthreads();
  for (int q = threadIdx.x; q < rows; q += TH)
    if (words[q] == w) {
      if (q < r) flags[nw] = 1u;                       // an earlier row owns this word (benign race: all write 1)
      else atomicOr(&flags[q >> 5], 1u << (q & 31));
    }
  __syncthreads();
  if (flags[nw]) return;                                // uniform across the block
  for (int e = threadIdx.x; e < E; e += TH) {
    float a = 0.f;
    for (int i = r >> 5; i < nw; ++i) {                 // ascending row order: deterministic
      unsigned m = flags[i];
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        a += d[(long)(i * 32 + b) * ld_d + e];
      }
    }
    dtable[w * E + e] += a;
  }
}
__global__ void gather_words_kernel(const int64_t* __restrict__ src, long s_sn, long s_st, int64_t* __restrict__ dst,
                                    int N, int T) {
  const int i = blockIdx.x * TH + threadIdx.x;
  if (i < N * T) dst[i] = src[(i / T) * s_sn + (i % T) * s_st];
}

// ---------------------------------------------------------------- GRU cell
// save: [N][4H] = r | z | n | gh_n
__global__ void gru_fwd_kernel(const float* __restrict__ gi, long ld_gi, const float* __restrict__ gh, long ld_gh,
                               const float* __restrict__ hp, long ld_hp, float* __restrict__ ho, long ld_ho,
                               float* __restrict__ oseq, long ld_os, float* __restrict__ save, long ld_sv,
                               float* __restrict__ hps, long ld_hps, const int64_t* __restrict__ lens, int t, int N,
                               int H) {
  const int total = N * H;
  for (int i = blockIdx.x * TH + threadIdx.x; i < total; i += gridDim.x * TH) {
    const int n = i / H, j = i % H;
    const float h = hp ? hp[n * ld_hp + j] : 0.f;
    if (hps) hps[n * ld_hps + j] = h;
    const bool valid = lens ? (t < (int)lens[n]) : true;
    float hn = h, os = 0.f;
    if (valid) {
      const float* a = gi + n * ld_gi;
      const float* b = gh + n * ld_gh;
      const float r = sigmoidf_(a[j] + b[j]);
      const float z = sigmoidf_(a[H + j] + b[H + j]);
      const float ghn = b[2 * H + j];
      const float nn = tanhf(a[2 * H + j] + r * ghn);
      hn = (1.f - z) * nn + z * h;
      os = hn;
      if (save) {
        float* s = save + n * ld_sv;
        s[j] = r; s[H + j] = z; s[2 * H + j] = nn; s[3 * H + j] = ghn;
      }
    }
    ho[n * ld_ho + j] = hn;
    if (oseq) oseq[n * ld_os + j] = os;
  }
}
// dh = dh_next + d_out(valid only).  Writes dgi, dgh ([N,3H]) and dh_prev.
__global__ void gru_bwd_kernel(const float* __restrict__ dh_next, long ld_dn, const float* __restrict__ d_out,
                               long ld_do, const float* __restrict__ save, long ld_sv, const float* __restrict__ hps,
                               long ld_hps, float* __restrict__ dgi, long ld_dgi, float* __restrict__ dgh, long ld_dgh,
                               float* __restrict__ dh_prev, long ld_dp, const int64_t* __restrict__ lens, int t, int N,
                               int H) {
  const int total = N * H;
  for (int i = blockIdx.x * TH + threadIdx.x; i < total; i += gridDim.x * TH) {
    const int n = i / H, j = i % H;
    const bool valid = lens ? (t < (int)lens[n]) : true;
    float dh = dh_next ? dh_next[n * ld_dn + j] : 0.f;
    float* a = dgi + n * ld_dgi;
    float* b = dgh + n * ld_dgh;
    if (!valid) {
      a[j] = a[H + j] = a[2 * H + j] = 0.f;
      b[j] = b[H + j] = b[2 * H + j] = 0.f;
      dh_prev[n * ld_dp + j] = dh;
      continue;
    }
    if (d_out) dh += d_out[n * ld_do + j];
    const float* s = save + n * ld_sv;
    const float r = s[j], z = s[H + j], nn = s[2 * H + j], ghn = s[3 * H + j];
    const float h = hps[n * ld_hps + j];
    const float dn = dh * (1.f - z);
    const float dz = dh * (h - nn);
    const float dnp = dn * (1.f - nn * nn);
    const float drp = dnp * ghn * r * (1.f - r);
    const float dzp = dz * z * (1.f - z);
    a[j] = drp; a[H + j] = dzp; a[2 * H + j] = dnp;
    b[j] = drp; b[H + j] = dzp; b[2 * H + j] = dnp * r;
    dh_prev[n * ld_dp + j] = dh * z;
  }
}

// ---------------------------------------------------------------- LSTM cell
// save: [N][5H] = i | f | g | o | tanh(c')
__global__ void lstm_fwd_kernel(const float* __restrict__ gates, long ld_g, const float* __restrict__ cp, long ld_cp,
                                float* __restrict__ ho, long ld_ho, float* __restrict__ co, long ld_co,
                                float* __restrict__ save, long ld_sv, int N, int H) {
  const int total = N * H;
  for (int i = blockIdx.x * TH + threadIdx.x; i < total; i += gridDim.x * TH) {
    const int n = i / H, j = i % H;
    const float* g = gates + n * ld_g;
    const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), og = sigmoidf_(g[3 * H + j]);
    const float c = cp ? cp[n * ld_cp + j] : 0.f;
    const float c2 = fg * c + ig * gg;
    const float tc = tanhf(c2);
    ho[n * ld_ho + j] = og * tc;
    co[n * ld_co + j] = c2;
    if (save) {
      float* s = save + n * ld_sv;
      s[j] = ig; s[H + j] = fg; s[2 * H + j] = gg; s[3 * H + j] = og; s[4 * H + j] = tc;
    }
  }
}
__global__ void lstm_bwd_kernel(const float* __restrict__ dh, long ld_dh, const float* __restrict__ dc_next, long ld_dc,
                                const float* __restrict__ save, long ld_sv, const float* __restrict__ cp, long ld_cp,
                                float* __restrict__ dgates, long ld_dg, float* __restrict__ dc_prev, long ld_dcp, int N,
                                int H) {
  const int total = N * H;
  for (int i = blockIdx.x * TH + threadIdx.x; i < total; i += gridDim.x * TH) {
    const int n = i / H, j = i % H;
    const float* s = save + n * ld_sv;
    const float ig = s[j], fg = s[H + j], gg = s[2 * H + j], og = s[3 * H + j], tc = s[4 * H + j];
    const float d = dh[n * ld_dh + j];
    float dc = d * og * (1.f - tc * tc);
    if (dc_next) dc += dc_next[n * ld_dc + j];
    const float c = cp ? cp[n * ld_cp + j] : 0.f;
    float* g = dgates + n * ld_dg;
    g[j] = dc * gg * ig * (1.f - ig);
    g[H + j] = dc * c * fg * (1.f - fg);
    g[2 * H + j] = dc * ig * (1.f - gg * gg);
    g[3 * H + j] = d * tc * og * (1.f - og);
    dc_prev[n * ld_dcp + j] = dc * fg;
  }
}

// ---------------------------------------------------------------- mean_with_lens + max_with_lens
__global__ void pool_fwd_kernel(const float* __restrict__ x, long ld_n, long ld_t, const int64_t* __restrict__ lens,
                                float* __restrict__ out, int* __restrict__ argmax, int N, int T, int C) {
  const int total = N * C;
  for (int i = blockIdx.x * TH + threadIdx.x; i < total; i += gridDim.x * TH) {
    const int n = i / C, c = i % C;
    int len = (int)lens[n];
    if (len > T) len = T;
    float sm = 0.f, mx = -INFINITY;
    int am = 0;
    for (int t = 0; t < len; ++t) {
      const float v = x[n * ld_n + t * ld_t + c];
      sm += v;
      if (v > mx) { mx = v; am = t; }
    }
    out[i] = sm / (float)lens[n] + mx;
    argmax[i] = am;
  }
}
// dx[n,t,c] (+)= d[n,c]/len (t<len) + d[n,c]*(t==argmax)
__global__ void pool_bwd_kernel(const float* __restrict__ d, const int64_t* __restrict__ lens,
                                const int* __restrict__ argmax, float* __restrict__ dx, long ld_n, long ld_t,
                                int accumulate, int N, int T, int C) {
  const long total = (long)N * T * C;
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < total; i += (long)gridDim.x * TH) {
    const int c = (int)(i % C);
    const int t = (int)((i / C) % T);
    const int n = (int)(i / ((long)C * T));
    const int len = (int)lens[n];
    float g = 0.f;
    if (t < len) {
      const float dv = d[n * C + c];
      g = dv / (float)len + (argmax[n * C + c] == t ? dv : 0.f);
    }
    float* p = dx + n * ld_n + t * ld_t + c;
    *p = accumulate ? *p + g : g;
  }
}

// out[c] (+)= sum_r x[r][c]: one thread per column, rows in order (deterministic); R is small (<= N*Tc)
__global__ void colsum_rows_kernel(const float* __restrict__ x, long ld, int rows, int cols, float* __restrict__ out,
                                   int accumulate) {
  const int c = blockIdx.x * TH + threadIdx.x;
  if (c >= cols) return;
  float a = 0.f;
  for (int r = 0; r < rows; ++r) a += x[r * ld + c];
  out[c] = accumulate ? out[c] + a : a;
}
__global__ void add_rows_kernel(float* __restrict__ dst, long ld_d, const float* __restrict__ src, long ld_s, int rows,
                                int cols) {
  const long total = (long)rows * cols;
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < total; i += (long)gridDim.x * TH) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[r * ld_d + c] += src[r * ld_s + c];
  }
}
__global__ void copy_rows_kernel(float* __restrict__ dst, long ld_d, const float* __restrict__ src, long ld_s, int rows,
                                 int cols) {
  const long total = (long)rows * cols;
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < total; i += (long)gridDim.x * TH) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[r * ld_d + c] = src ? src[r * ld_s + c] : 0.f;
  }
}

// several copy_rows jobs in one launch (blockIdx.y = job): the decode forward ended in five 5-us copies of a few KB each
__global__ void copy_rows_batch_kernel(acvae::CopyRowsBatch b) {
  const int j = blockIdx.y;
  const long total = (long)b.rows[j] * b.cols[j];
  const int cols = b.cols[j];
  float* dst = b.dst[j];
  const float* src = b.src[j];
  const long ld_d = b.ld_d[j], ld_s = b.ld_s[j];
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < total; i += (long)gridDim.x * TH) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[r * ld_d + c] = src ? src[r * ld_s + c] : 0.f;
  }
}

__global__ void zero_batch_kernel(acvae::ZeroBatch b) {
  const int j = blockIdx.y;
  unsigned* p = static_cast<unsigned*>(b.p[j]);
  const long n = b.words[j];
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < n; i += (long)gridDim.x * TH) p[i] = 0u;
}

}  // namespace

namespace acvae {
#define LAUNCH(k, g, ...) hipLaunchKernelGGL(k, dim3(g), dim3(TH), 0, st, __VA_ARGS__)
int zero_batch(const ZeroBatch& b, hipStream_t st) {
  if (b.n <= 0) return ACVAE_OK;
  if (b.n > ZeroBatch::MAXJ) return ACVAE_EINVAL;
  long most = 1;
  for (int j = 0; j < b.n; ++j) if (b.words[j] > most) most = b.words[j];
  hipLaunchKernelGGL(zero_batch_kernel, dim3(grid1(most), b.n), dim3(TH), 0, st, b);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int copy_rows_batch(const CopyRowsBatch& b, hipStream_t st) {
  if (b.n <= 0) return ACVAE_OK;
  long most = 1;
  for (int j = 0; j < b.n; ++j) {
    if (!b.dst[j] || b.rows[j] <= 0 || b.cols[j] <= 0) return ACVAE_EINVAL;
    const long t = (long)b.rows[j] * b.cols[j];
    if (t > most) most = t;
  }
  hipLaunchKernelGGL(copy_rows_batch_kernel, dim3(grid1(most), b.n), dim3(TH), 0, st, b);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}

int caps_to_long(const float* caps, int64_t* out, long n, hipStream_t st) {
  LAUNCH(caps_to_long_kernel, grid1(n), caps, out, n);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int select_word(const int64_t* caps, long ld_caps, const int64_t* seqs, long ld_seqs, int64_t* words, long ld_words,
                int t, int use_caps, int start_idx, int N, hipStream_t st) {
  LAUNCH(select_word_kernel, cdiv(N, TH), caps, ld_caps, seqs, ld_seqs, words, ld_words, t, use_caps, start_idx, N);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int finish_rows(int64_t* seqs, long ld_seqs, uint8_t* unfinished, int t, int end_idx, int N, hipStream_t st) {
  LAUNCH(finish_rows_kernel, cdiv(N, TH), seqs, ld_seqs, unfinished, t, end_idx, N);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int embed_gather(const int64_t* words, long w_stride, const float* table, int V, float* out, long ld_out, int rows,
                 int E, hipStream_t st) {
  LAUNCH(embed_gather_kernel, grid1((long)rows * E), words, w_stride, table, V, out, ld_out, rows, E);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int embed_scatter(const int64_t* words_contig, const float* d, long ld_d, float* dtable, int V, int rows, int E,
                  hipStream_t st) {
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(rows), dim3(TH), (size_t)((rows + 31) / 32 + 2) * sizeof(unsigned), st,
                     words_contig, d, ld_d, dtable, V, rows, E);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int dropout_rows(float* x, long ld_n, long ld_t, const uint8_t* keep, long k_sn, long k_st, float scale, int N, int cnt,
                 int E, hipStream_t st) {
  LAUNCH(dropout_rows_kernel, grid1((long)N * cnt * E), x, ld_n, ld_t, keep, k_sn, k_st, scale, N, cnt, E);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int gather_words(const int64_t* src, long s_sn, long s_st, int64_t* dst, int N, int T, hipStream_t st) {
  LAUNCH(gather_words_kernel, cdiv(N * T, TH), src, s_sn, s_st, dst, N, T);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int gru_fwd(const float* gi, long ld_gi, const float* gh, long ld_gh, const float* hp, long ld_hp, float* ho,
            long ld_ho, float* oseq, long ld_os, float* save, long ld_sv, float* hps, long ld_hps, const int64_t* lens,
            int t, int N, int H, hipStream_t st) {
  LAUNCH(gru_fwd_kernel, grid1((long)N * H), gi, ld_gi, gh, ld_gh, hp, ld_hp, ho, ld_ho, oseq, ld_os, save, ld_sv, hps,
         ld_hps, lens, t, N, H);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int gru_bwd(const float* dh_next, long ld_dn, const float* d_out, long ld_do, const float* save, long ld_sv,
            const float* hps, long ld_hps, float* dgi, long ld_dgi, float* dgh, long ld_dgh, float* dh_prev, long ld_dp,
            const int64_t* lens, int t, int N, int H, hipStream_t st) {
  LAUNCH(gru_bwd_kernel, grid1((long)N * H), dh_next, ld_dn, d_out, ld_do, save, ld_sv, hps, ld_hps, dgi, ld_dgi, dgh,
         ld_dgh, dh_prev, ld_dp, lens, t, N, H);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int lstm_fwd(const float* gates, long ld_g, const float* cp, long ld_cp, float* ho, long ld_ho, float* co, long ld_co,
             float* save, long ld_sv, int N, int H, hipStream_t st) {
  LAUNCH(lstm_fwd_kernel, grid1((long)N * H), gates, ld_g, cp, ld_cp, ho, ld_ho, co, ld_co, save, ld_sv, N, H);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int lstm_bwd(const float* dh, long ld_dh, const float* dc_next, long ld_dc, const float* save, long ld_sv,
             const float* cp, long ld_cp, float* dgates, long ld_dg, float* dc_prev, long ld_dcp, int N, int H,
             hipStream_t st) {
  LAUNCH(lstm_bwd_kernel, grid1((long)N * H), dh, ld_dh, dc_next, ld_dc, save, ld_sv, cp, ld_cp, dgates, ld_dg, dc_prev,
         ld_dcp, N, H);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int pool_fwd(const float* x, long ld_n, long ld_t, const int64_t* lens, float* out, int* argmax, int N, int T, int C,
             hipStream_t st) {
  LAUNCH(pool_fwd_kernel, grid1((long)N * C), x, ld_n, ld_t, lens, out, argmax, N, T, C);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int pool_bwd(const float* d, const int64_t* lens, const int* argmax, float* dx, long ld_n, long ld_t, int accumulate,
             int N, int T, int C, hipStream_t st) {
  LAUNCH(pool_bwd_kernel, grid1((long)N * T * C), d, lens, argmax, dx, ld_n, ld_t, accumulate, N, T, C);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int colsum_rows(const float* x, long ld, int rows, int cols, float* out, int accumulate, hipStream_t st) {
  LAUNCH(colsum_rows_kernel, cdiv(cols, TH), x, ld, rows, cols, out, accumulate);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int add_rows(float* dst, long ld_d, const float* src, long ld_s, int rows, int cols, hipStream_t st) {
  LAUNCH(add_rows_kernel, grid1((long)rows * cols), dst, ld_d, src, ld_s, rows, cols);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
int copy_rows(float* dst, long ld_d, const float* src, long ld_s, int rows, int cols, hipStream_t st) {
  LAUNCH(copy_rows_kernel, grid1((long)rows * cols), dst, ld_d, src, ld_s, rows, cols);
  ACVAE_LAUNCH_CHECK(); return ACVAE_OK;
}
#undef LAUNCH
}  // namespace acvae
