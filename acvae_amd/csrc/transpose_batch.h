// gemm.hip: several transposes in one launch (internal C++ interface)
#pragma once
#include <hip/hip_runtime.h>
// several transposes in ONE launch (out[c][r] = in[r][c]): the decode backward needs nine transposed weight matrices per step
struct TransposeBatch {
  static constexpr int MAXM = 12;
  const float* in[MAXM]; float* out[MAXM]; long ld_in[MAXM], ld_out[MAXM]; int rows[MAXM], cols[MAXM]; int tile0[MAXM + 1]; int n = 0;
  void add(const float* i, long li, float* o, long lo, int r, int c) { in[n] = i; ld_in[n] = li; out[n] = o; ld_out[n] = lo; rows[n] = r; cols[n] = c; ++n; }
};
int acvae_transpose_batch(TransposeBatch& b, hipStream_t st);
