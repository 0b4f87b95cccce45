#pragma once
#include <hip/hip_runtime.h>
#define ACVAE_PROF_NTAGS 4
#define ACVAE_PROF_CONV_IGEMM 0   /* conv3x3 implicit GEMM: forward and data-gradient launches */
#define ACVAE_PROF_CONV_WGRAD 1   /* conv3x3 weight-gradient launches */
namespace acvae {
void prof_begin(int tag, hipStream_t st);
void prof_end(int tag, hipStream_t st);
}
