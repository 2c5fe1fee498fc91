// Shared device/host helpers for the STonKGs MI355X (gfx950) HIP kernels.
// Everything here is CDNA4-only: 64-lane wavefronts, bf16 MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// C-ABI status codes (include/stonk_hip.h): 0 ok, <0 bad argument, >0 hipError_t.
#define STONK_OK 0
#define STONK_EINVAL (-1)
#define STONK_ESHAPE (-2)
#define STONK_EALIGN (-3)

#define STONK_CHECK_ARG(cond, code) \
  do {                              \
    if (!(cond)) return (code);     \
  } while (0)

static inline int stonk_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? STONK_OK : (int)e;
}

#define WAVE 64

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

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 GELU's own rounding and far below bf16):
// one v_rcp, one v_exp and a 5-term Horner chain instead of ocml's branchy erff - the GELU epilogues of the
// FFN GEMMs are VALU-bound on it otherwise. Also returns exp(-x^2) for the derivative.
__device__ __forceinline__ float erf_as(float x, float& e_mx2) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));  // v_rcp_f32 (1 ulp), not the IEEE division sequence
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e_mx2 = __expf(-ax * ax);
  const float r = 1.0f - p * t * e_mx2;
  return copysignf(r, x);
}
// Exact (erf) GELU, as hf:models/bert (hidden_act="gelu"), and its derivative.
__device__ __forceinline__ float gelu_erf(float x) {
  float e;
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float e;  // = exp(-x^2 / 2)
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e));
  return cdf + x * 0.39894228040143267794f * e;
}

// Counter-based dropout RNG: one 32-bit mix ("lowbias32") of (element index, seed).
// Forward and backward regenerate the same keep-mask from (seed, index); nothing is stored.
__device__ __forceinline__ uint32_t stonk_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// keep-probability threshold in 24-bit fixed point: keep iff (hash >> 8) >= thr
__device__ __forceinline__ bool stonk_keep(uint32_t idx, uint32_t seed, uint32_t thr24) {
  return (stonk_hash32(idx ^ (seed * 0x9E3779B9U + 0x85ebca6bU)) >> 8) >= thr24;
}
static inline uint32_t stonk_drop_thr24(float p) {
  double t = (double)p * 16777216.0;
  if (t < 0) t = 0;
  if (t > 16777215.0) t = 16777215.0;
  return (uint32_t)(t + 0.5);
}
