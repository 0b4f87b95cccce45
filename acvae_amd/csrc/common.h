// Shared device/host helpers for the AC-VAE gfx950 kernels.  CDNA4 only: 64-lane wavefronts are
// hard-coded (cdna_hip_programming.md §1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ACVAE_OK 0
#define ACVAE_EINVAL (-1)      // bad dims / null pointer
#define ACVAE_EALIGN (-2)      // pointer or leading dimension not aligned as the kernel needs
#define ACVAE_EUNSUPPORTED (-3)
#define ACVAE_EWORKSPACE (-4)  // workspace too small

// Launch check: never throws, never exits; returns the hipError_t as a positive code.
#define ACVAE_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

#define ACVAE_TRY(x)                 \
  do {                               \
    int r__ = (x);                   \
    if (r__ != 0) return r__;        \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- bf16 storage (configs[2]: bf16 forward / fp32 loss).  Tensors are stored as bf16 and every kernel computes in fp32:
// load4 / store4 move 4 consecutive elements of either storage type (16 B of float, 8 B of bf16; round-to-nearest-even
// on the way out: hipcc emits v_cvt_pk_bf16_f32 for the cast).
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {     // a -> low half, b -> high half
  union { bf16_t h[2]; unsigned u; } x;
  x.h[0] = (bf16_t)a; x.h[1] = (bf16_t)b;
  return x.u;
}
__device__ __forceinline__ float round_bf16(float a) { return (float)(bf16_t)a; }
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const bf16_t* p) { return (float)*p; }

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), total in every lane of the row: four v_add_f32 with a DPP
// operand (xor 1, xor 2 inside the quad, then half-mirror / mirror: the partner quad / half holds one value in all its
// lanes by then, so the result is bit-identical to the xor butterfly) - VALU only, where __shfl_xor is a ds_bpermute_b32
// on the LDS pipe.
__device__ __forceinline__ float row16_sum(float v) {
#define ACVAE_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  ACVAE_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  ACVAE_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  ACVAE_DPP_ADD(0x141);   // row_half_mirror
  ACVAE_DPP_ADD(0x140);   // row_mirror
#undef ACVAE_DPP_ADD
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// tanh of the additive-attention scores (models/attn_model.py:33: tanh over [N, S, A] every decode step - 31 744 values per
// clip and step, the bulk of the attention's instructions): 1 - 2 / (e^{2x} + 1) on the hardware exp2 and reciprocal, five
// instructions instead of the ~50 of the library's tanhf; absolute error <= 2.5e-7 (measured 2.1e-7) (the library's is relative: near zero this
// form is less precise, which a sum of 512 terms weighted by v does not see: scores agree to ~1e-6, bound in
// tests/test_ops_gpu.py).  Forward and backward use the same function; saturates to +-1 for |x| > 44.
// A library built with -DACVAE_EXACT_TANH (ACVAE_EXACT_TANH=1 python -m acvae_amd.build --force) uses the library's tanhf
// instead: the switch for parity debugging against the reference's numerics; tests/test_ops_gpu.py bounds the fast form
// element-wise against fp64 over [-20, 20] including |x| < 1e-4 (acvae_tanh_att).
__device__ __forceinline__ float tanh_att(float x) {
#ifdef ACVAE_EXACT_TANH
  return tanhf(x);
#else
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
#endif
}

// Counter-based RNG for dropout (Philox-4x32-10).  One call -> 4 uniform 32-bit words.
__device__ __forceinline__ uint4 philox4x32(uint64_t seed, uint64_t ctr_lo, uint32_t ctr_hi) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = ctr_hi, c3 = 0x5eed5eedu;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
// keep-decision for element `idx` of dropout site `site` with keep-probability 1-p.
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, float p) {
  const uint4 r = philox4x32(seed, idx >> 2, site);
  const uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
  return (float)(w >> 8) * (1.0f / 16777216.0f) >= p;
}
#endif
