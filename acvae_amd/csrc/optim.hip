// A10: global-norm gradient clipping + Adam over ONE flat fp32 parameter/gradient buffer.
// Reference: runners/pytorch_runner_vae.py:321-324 (loss.backward -> clip_grad_norm_(max_grad_norm) ->
// optimizer.step with torch.optim.Adam).  HBM-bound streaming kernels: the norm is one read of the gradients
// (two-level fixed-order reduction, fp64 combine: deterministic), the update is one pass that reads p,g,m,v and
// writes p,m,v (28 B/param), with the clip coefficient taken from the device-side norm so no host sync is needed.
#include "common.h"
#include "../../include/acvae_hip.h"

namespace {
constexpr int TH = 256;
constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(TH) void sqsum_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
  __shared__ float red[16];
  float acc = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < n4; i += (long)gridDim.x * TH) {
    const float4 v = g4[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += TH) acc += g[i] * g[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// norm = scale * sqrt(sum partials)
__global__ void norm_final_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out) {
  __shared__ double redd[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)partials[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += redd[i];
    out[0] = (float)(sqrt(t) * (double)scale);
  }
}

__global__ __launch_bounds__(TH) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                  float beta1, float beta2, float eps, float weight_decay,
                                                  float bc1, float bc2_sqrt, float grad_scale, float max_norm,
                                                  const float* __restrict__ total_norm) {
  // clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max=1)
  float coef = grad_scale;
  if (total_norm && max_norm > 0.f) {
    float c = max_norm / (total_norm[0] + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  const float step_size = lr / bc1;
  for (long i = blockIdx.x * (long)TH + threadIdx.x; i < n; i += (long)gridDim.x * TH) {
    float gi = g[i] * coef;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
}  // namespace

extern "C" int64_t acvae_grad_norm_partials(void) { return NORM_BLOCKS; }

extern "C" int acvae_grad_norm(const float* grads, int64_t n, float grad_scale, float* partials, float* out_norm,
                               void* stream) {
  if (!grads || !partials || !out_norm || n <= 0) return ACVAE_EINVAL;
  if (!aligned16(grads)) return ACVAE_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  long nb = (n / 4 + TH - 1) / TH;
  if (nb < 1) nb = 1;
  if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
  hipLaunchKernelGGL(sqsum_kernel, dim3((int)nb), dim3(TH), 0, st, grads, (long)n, partials);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, partials, (int)nb, grad_scale, out_norm);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}

extern "C" int acvae_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                               float grad_scale, float max_grad_norm, const float* total_norm, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return ACVAE_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long nb = (n + TH - 1) / TH;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(TH), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     (long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale,
                     max_grad_norm, total_norm);
  ACVAE_LAUNCH_CHECK();
  return ACVAE_OK;
}
