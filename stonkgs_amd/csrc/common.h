// Shared device/host helpers for the STonKGs MI355X (gfx950) HIP kernels.
// Everything here is CDNA4-only: 64-lane wavefronts, bf16 MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
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

// fp32 -> fp16 with the range clamped (STONK_EPI_OUT_F16: an overflowing logit must not become an infinity)
__device__ __forceinline__ _Float16 to_f16_sat(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }

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

// what SAVE_PREACT stores / what GELU_BWD multiplies by, with and without STONK_EPI_AUX_GRAD
__device__ __forceinline__ float gelu_saved(float pre, bool aux_grad) { return aux_grad ? gelu_erf_grad(pre) : pre; }
__device__ __forceinline__ float gelu_factor(float aux, bool aux_grad) { return aux_grad ? aux : gelu_erf_grad(aux); }

// Counter-based dropout RNG. Forward and backward regenerate the same keep-mask from (seed, row, column); nothing is
// stored. An element's 32-bit counter is  x = (row * G_ROW + mix(seed)) ^ (column * G_COL)  - both keys are linear, so a
// lane walking rows or columns advances them with one add - followed by two 24-bit multiply rounds (v_mad_u32_u24 /
// v_mul_u32_u24 issue at full rate; a 32-bit v_mul_lo_u32 costs four slots):  y = lo24(x) * C1 + x,  z = (y >> 8) * C2,
// keep iff z >= p * 2^32.  Five full-rate VALU ops per element where the previous two-multiply "lowbias32" mix cost the
// equivalent of ~18; keep-rate, row/column sums and pairwise joint drop rates are indistinguishable from an ideal
// generator at the sizes used here (checked offline over 4096 x 512 blocks).
constexpr uint32_t STONK_G_ROW = 0x9E3779B1u, STONK_G_COL = 0x85EBCA77u;
__host__ __device__ inline uint32_t stonk_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// once per launch, on the host: the seed word the kernels see
static inline uint32_t stonk_seed_mix(uint32_t seed) { return stonk_hash32(seed * 0x9E3779B9U + 0x85ebca6bU); }
__device__ __forceinline__ uint32_t stonk_rowkey(uint32_t row, uint32_t seedmix) { return row * STONK_G_ROW + seedmix; }
__device__ __forceinline__ uint32_t stonk_colkey(uint32_t col) { return col * STONK_G_COL; }
__device__ __forceinline__ bool stonk_keep_key(uint32_t rowkey, uint32_t colkey, uint32_t thr32) {
  const uint32_t x = rowkey ^ colkey;
  const uint32_t y = __umul24(x, 0xB5297Bu) + x;
  return __umul24(y >> 8, 0x68E31Du) >= thr32;
}
// The same generator for a 4 x 4 BLOCK of (row, column) neighbours (rows 4i .. 4i+3, columns 4j .. 4j+3) - the attention
// probabilities' dropout: the first round is computed once per block from the block's keys (rowkey(i), colkey(j)), the
// last round once per element with one of sixteen multipliers, STONK_C2_BLK[4 (row & 3) + (column & 3)]. Whichever way a
// lane walks the scores, four of its consecutive elements share a first round: a lane that walks the keys of one query
// (forward, dQ: four consecutive keys per accumulator group) keeps the four multipliers of its row in registers, a lane
// that owns one key and walks the queries (dK/dV: four consecutive query rows per accumulator group) the four of its
// column - 4 VALU operations per score in all three kernels (the dK/dV kernel paid 7 while only the columns were grouped).
// The multipliers were chosen by exhaustive enumeration of the 2^24 first-round values (tools/dropout_multipliers.py):
// every pairwise joint drop rate of two block members is within 0.3 % of p^2 and every triple within 1.3 % of p^3 at
// p = 0.1 and 0.25, and the number of drops in a block follows the binomial to 0.1 % up to 5 of 16.
constexpr uint32_t STONK_C2_BLK[16] = {0x45821Fu, 0xFAC0C5u, 0xD90A1Bu, 0xE6B85Bu, 0xE11EE7u, 0x6E41B9u, 0xB8115Du, 0xB1494Bu,
                                       0xBF440Fu, 0xA96C25u, 0x7B570Fu, 0xA2609Du, 0x98207Du, 0xB5677Du, 0x82C471u, 0xAC3C19u};
// first round of a block: rowkey = stonk_rowkey(row >> 2, seed), colkey = stonk_colkey(column >> 2)
__device__ __forceinline__ uint32_t stonk_blk_round1(uint32_t rowkey, uint32_t colkey) {
  const uint32_t x = rowkey ^ colkey;
  return (__umul24(x, 0xB5297Bu) + x) >> 8;
}
__device__ __forceinline__ bool stonk_blk_keep(uint32_t y8, uint32_t c2, uint32_t thr32) { return __umul24(y8, c2) >= thr32; }
__device__ __forceinline__ bool stonk_keep(uint32_t row, uint32_t col, uint32_t seedmix, uint32_t thr32) {
  return stonk_keep_key(stonk_rowkey(row, seedmix), stonk_colkey(col), thr32);
}
// drop-probability threshold in 32-bit fixed point: keep iff z >= thr
static inline uint32_t stonk_drop_thr32(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)(t + 0.5);
}
