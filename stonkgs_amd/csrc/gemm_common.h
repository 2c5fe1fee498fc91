// Argument block and helpers shared by the two bf16 GEMM kernels (gemm_bf16.hip: 128x128 tiles, gemm256.hip:
// persistent 256x256 tiles). See stonk_gemm_nt_bf16 in include/stonk_hip.h for the meaning of each field.
#pragma once
#include "common.h"
#include "stonk_flags.h"

namespace stonk_gemm {

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const bf16* resid;
  bf16* aux;
  const int* m_dev;
  const int* k_dev;
  long lda, ldb, ldc, ldr, ldaux;
  int M, N, K;
  int flags;
  float alpha;
  int split_k;
  uint32_t drop_thr32;
  float drop_scale;
  uint32_t seed;      // already mixed (stonk_seed_mix)
};

// blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles so
// neighbouring tiles (same A row panel / same B column panel) hit the same L2. Bijective for any n.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (b >> 3);
}


// Fused epilogue on 4 consecutive output columns n..n+3 of row m (values already scaled by alpha).
__device__ __forceinline__ f32x4 epilogue4(f32x4 v, const GemmArgs& p, int flags, int m, int n) {
  if (flags & STONK_EPI_BIAS) v += *(const f32x4*)(p.bias + n);
  if (flags & STONK_EPI_SAVE_PREACT) {
    const bool ag = (flags & STONK_EPI_AUX_GRAD) != 0;
    bf16x4 u = {(bf16)gelu_saved(v[0], ag), (bf16)gelu_saved(v[1], ag), (bf16)gelu_saved(v[2], ag),
                (bf16)gelu_saved(v[3], ag)};
    *(bf16x4*)(p.aux + (long)m * p.ldaux + n) = u;
  }
  if (flags & STONK_EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
  }
  if (flags & STONK_EPI_GELU_BWD) {
    const bf16x4 u = *(const bf16x4*)(p.aux + (long)m * p.ldaux + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= gelu_factor((float)u[r], (flags & STONK_EPI_AUX_GRAD) != 0);
  }
  if (flags & STONK_EPI_DROPOUT) {
    const uint32_t rk = stonk_rowkey((uint32_t)m, p.seed), ck = stonk_colkey((uint32_t)n);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = stonk_keep_key(rk, ck + (uint32_t)r * STONK_G_COL, p.drop_thr32) ? v[r] * p.drop_scale : 0.f;
  }
  if (flags & STONK_EPI_RESID) {
    const bf16x4 rr = *(const bf16x4*)(p.resid + (long)m * p.ldr + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
  }
  return v;
}

// Fused epilogue on 8 consecutive output columns n..n+7 of row m (fp32 values already scaled by alpha): every side
// operand (bias, saved pre-activation, residual) is read / written as one 16- or 32-byte vector.
__device__ __forceinline__ void epilogue8(float (&v)[8], const GemmArgs& p, int flags, int m, int n) {
  if (flags & STONK_EPI_BIAS) {
    const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] += b0[r];
      v[4 + r] += b1[r];
    }
  }
  if (flags & STONK_EPI_SAVE_PREACT) {
    bf16x8 u;
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = (bf16)gelu_saved(v[r], (flags & STONK_EPI_AUX_GRAD) != 0);
    *(bf16x8*)(p.aux + (long)m * p.ldaux + n) = u;
  }
  if (flags & STONK_EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
  }
  if (flags & STONK_EPI_GELU_BWD) {
    const bf16x8 u = *(const bf16x8*)(p.aux + (long)m * p.ldaux + n);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= gelu_factor((float)u[r], (flags & STONK_EPI_AUX_GRAD) != 0);
  }
  if (flags & STONK_EPI_DROPOUT) {
    const uint32_t rk = stonk_rowkey((uint32_t)m, p.seed), ck = stonk_colkey((uint32_t)n);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      v[r] = stonk_keep_key(rk, ck + (uint32_t)r * STONK_G_COL, p.drop_thr32) ? v[r] * p.drop_scale : 0.f;
  }
  if (flags & STONK_EPI_RESID) {
    const bf16x8 rr = *(const bf16x8*)(p.resid + (long)m * p.ldr + n);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += (float)rr[r];
  }
}

// Side operands of one epilogue8 call, fetched AHEAD of their use (one round earlier) so that the global-load latency
// of the residual / saved pre-activation is not paid once per 16-row round.
struct SideOps {
  bf16x8 aux, res;
};
__device__ __forceinline__ void side_prefetch(SideOps& s, const GemmArgs& p, int flags, int m, int n, bool ok) {
  if (!ok) return;
  if (flags & STONK_EPI_GELU_BWD) s.aux = *(const bf16x8*)(p.aux + (long)m * p.ldaux + n);
  if (flags & STONK_EPI_RESID) s.res = *(const bf16x8*)(p.resid + (long)m * p.ldr + n);
}
// single-vector form for kernels that carry exactly one side operand (GELU' input OR residual)
__device__ __forceinline__ bf16x8 side_load1(const GemmArgs& p, int flags, int m, int n, bool ok) {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
  if (!ok) return z;
  if (flags & STONK_EPI_GELU_BWD) return *(const bf16x8*)(p.aux + (long)m * p.ldaux + n);
  return *(const bf16x8*)(p.resid + (long)m * p.ldr + n);
}
// epilogue8 with preloaded bias (8 columns of this lane, fixed for the whole tile) and side operands
__device__ __forceinline__ void epilogue8_pre(float (&v)[8], const GemmArgs& p, int flags, int m, int n, const f32x4& b0,
                                              const f32x4& b1, const SideOps& s) {
  if (flags & STONK_EPI_BIAS) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] += b0[r];
      v[4 + r] += b1[r];
    }
  }
  if (flags & STONK_EPI_SAVE_PREACT) {
    bf16x8 u;
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = (bf16)gelu_saved(v[r], (flags & STONK_EPI_AUX_GRAD) != 0);
    *(bf16x8*)(p.aux + (long)m * p.ldaux + n) = u;
  }
  if (flags & STONK_EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
  }
  if (flags & STONK_EPI_GELU_BWD) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= gelu_factor((float)s.aux[r], (flags & STONK_EPI_AUX_GRAD) != 0);
  }
  if (flags & STONK_EPI_DROPOUT) {
    const uint32_t rk = stonk_rowkey((uint32_t)m, p.seed), ck = stonk_colkey((uint32_t)n);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      v[r] = stonk_keep_key(rk, ck + (uint32_t)r * STONK_G_COL, p.drop_thr32) ? v[r] * p.drop_scale : 0.f;
  }
  if (flags & STONK_EPI_RESID) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += (float)s.res[r];
  }
}

}  // namespace stonk_gemm
