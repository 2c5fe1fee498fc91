// LayerNorm (forward / backward) and the two embedding front-ends of the STonKGs step, as wavefront-
// reduction kernels: one 64-lane wave owns one row, each lane holds 16-byte chunks of it in registers,
// reductions are xor-shuffles across the wave, nothing goes through LDS except the dgamma/dbeta
// block reduction. HBM-bound by construction (one read, one write per element).
//
// Replaces: nn.LayerNorm(eps=1e-12) in hf:models/bert/modeling_bert.py BertEmbeddings :98-108,
// BertSelfOutput :289-293, BertOutput :347-351, BertPredictionHeadTransform :476-480; the KG gather +
// concat + cast of ref:src/stonkgs/models/stonkgs_model.py:182-200 (K2/K3 in SURVEY.md section 2.3).
#include <cstdlib>
#include "common.h"
#include "stonk_flags.h"

namespace {

template <int CPL>
struct RowRegs {
  float v[CPL][8];
};

template <int CPL>
__device__ __forceinline__ void zero_row(RowRegs<CPL>& r) {
#pragma unroll
  for (int i = 0; i < CPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
}

template <int CPL>
__device__ __forceinline__ void load_bf16_row(const bf16* row, int nch, int lane, RowRegs<CPL>& r) {
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const bf16x8 x = *(const bf16x8*)(row + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = (float)x[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
    }
  }
}
template <int CPL>
__device__ __forceinline__ void load_f32_row(const float* row, int nch, int lane, RowRegs<CPL>& r) {
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const f32x4 a = *(const f32x4*)(row + c * 8);
      const f32x4 b = *(const f32x4*)(row + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.v[i][j] = a[j];
        r.v[i][4 + j] = b[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
    }
  }
}
template <int CPL>
__device__ __forceinline__ void add_f32_row(const float* row, int nch, int lane, RowRegs<CPL>& r) {
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const f32x4 a = *(const f32x4*)(row + c * 8);
      const f32x4 b = *(const f32x4*)(row + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.v[i][j] += a[j];
        r.v[i][4 + j] += b[j];
      }
    }
  }
}
template <int CPL>
__device__ __forceinline__ void store_bf16_row(bf16* row, int nch, int lane, const RowRegs<CPL>& r) {
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)r.v[i][j];
      *(bf16x8*)(row + c * 8) = o;
    }
  }
}

// two-pass (mean, then centred variance) statistics in fp32, as torch's native_layer_norm computes them
template <int CPL>
__device__ __forceinline__ void row_stats(const RowRegs<CPL>& r, int nch, int lane, int H, float eps, float& mean,
                                          float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += r.v[i][j];
  mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = r.v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float var = wave_sum(q) / (float)H;
  rstd = rsqrtf(var + eps);
}

// y = (x - mean) * rstd * gamma + beta, optional inverted dropout on y
template <int CPL>
__device__ __forceinline__ void normalize_store(RowRegs<CPL>& r, int nch, int lane, float mean, float rstd,
                                                const float* gamma, const float* beta, bf16* yrow, long row_idx, int H,
                                                bool drop, uint32_t thr32, float dscale, uint32_t seed) {
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + c * 8), b1 = *(const f32x4*)(beta + c * 8 + 4);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = j < 4 ? g0[j] : g1[j - 4];
        const float b = j < 4 ? b0[j] : b1[j - 4];
        float y = (r.v[i][j] - mean) * rstd * g + b;
        if (drop) y = stonk_keep((uint32_t)row_idx, (uint32_t)(c * 8 + j), seed, thr32) ? y * dscale : 0.f;
        o[j] = (bf16)y;
      }
      *(bf16x8*)(yrow + c * 8) = o;
    }
  }
}

template <int CPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, int H, float eps, int flags, uint32_t thr32,
                                                            float dscale, uint32_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    RowRegs<CPL> r;
    load_bf16_row<CPL>(x + row * H, nch, lane, r);
    float mean, rstd;
    row_stats<CPL>(r, nch, lane, H, eps, mean, rstd);
    if (lane == 0 && mean_out) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    normalize_store<CPL>(r, nch, lane, mean, rstd, gamma, beta, y + row * H, row, H, flags & STONK_LN_DROPOUT, thr32,
                         dscale, seed);
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma += sum dy * xhat; dbeta += sum dy
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ gamma, bf16* __restrict__ dx,
    bf16* __restrict__ dx_drop, float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int H, int flags,
    uint32_t thr32_in, float dscale_in, uint32_t seed_in, uint32_t thr32_out, float dscale_out, uint32_t seed_out,
    float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float sred[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  float ag[CPL][8], ab[CPL][8];
#pragma unroll
  for (int i = 0; i < CPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[i][j] = ab[i][j] = 0.f;

  // software prefetch: the raw bf16 chunks of the NEXT row are requested before the current row is processed (16 extra
  // VGPRs), so each wave keeps two rows' loads in flight instead of stalling a full memory latency per row
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  bf16x8 nx[CPL], nd[CPL];
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](long r) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        nx[i] = *(const bf16x8*)(x + r * H + c * 8);
        nd[i] = *(const bf16x8*)(dy + r * H + c * 8);
      }
    }
    nmean = mean_in[r];
    nrstd = rstd_in[r];
  };
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    RowRegs<CPL> rx, rd;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = lane + 64 * i < nch;
        rx.v[i][j] = ok ? (float)nx[i][j] : 0.f;
        rd.v[i][j] = ok ? (float)nd[i][j] : 0.f;
      }
    const float mean = nmean, rstd = nrstd;
    if (row + stride < rows) fetch(row + stride);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float d = rd.v[i][j];
          if (flags & STONK_LN_DROPOUT)
            d = stonk_keep((uint32_t)row, (uint32_t)(c * 8 + j), seed_in, thr32_in) ? d * dscale_in : 0.f;
          const float xh = (rx.v[i][j] - mean) * rstd;
          ag[i][j] += d * xh;
          ab[i][j] += d;
          const float g = d * (j < 4 ? g0[j] : g1[j - 4]);
          rx.v[i][j] = xh;
          rd.v[i][j] = g;
          s1 += g;
          s2 += g * xh;
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        bf16x8 o, od;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = rstd * (rd.v[i][j] - c1 - rx.v[i][j] * c2);
          o[j] = (bf16)v;
          if (dx_drop)
            od[j] = (bf16)(stonk_keep((uint32_t)row, (uint32_t)(c * 8 + j), seed_out, thr32_out) ? v * dscale_out : 0.f);
        }
        *(bf16x8*)(dx + row * H + c * 8) = o;
        if (dx_drop) *(bf16x8*)(dx_drop + row * H + c * 8) = od;
      }
    }
  }
  // block reduction of the per-wave column partials, then one atomic per column per block
  if (dgamma) {
    float* sg = sred;          // [4][H]
    float* sb = sred + 4 * H;  // [4][H]
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sg[wave * H + c * 8 + j] = ag[i][j];
          sb[wave * H + c * 8 + j] = ab[i][j];
        }
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256) {
      const float g = sg[col] + sg[H + col] + sg[2 * H + col] + sg[3 * H + col];
      const float b = sb[col] + sb[H + col] + sb[2 * H + col] + sb[3 * H + col];
      if (ws) {  // per-workgroup partials, summed by ln_partial_reduce_kernel: ~10^3 workgroups adding atomically into the
                 // same 2H addresses run at the contended-atomic rate (the kernel spent more time there than streaming)
        ws[(long)blockIdx.x * 2 * H + col] = g;
        ws[(long)blockIdx.x * 2 * H + H + col] = b;
      } else {
        atomicAdd(dgamma + col, g);
        atomicAdd(dbeta + col, b);
      }
    }
  }
}

// ---- lane-layout backward for H = 64 * EPL (EPL = 8, 12, 16: H = 512, 768, 1024) ---------------------------------------
// The generic kernel above is VALU-bound (~470 VALU instructions per row at H = 768, 69 us for 32 768 rows where the
// 200 MB it moves would take 25-40 us): a quarter of its lanes idle on the second 8-element chunk (96 chunks over 64
// lanes), the dropout switches are run-time branches per element, both row sums go through ds_bpermute, and column keys
// and gamma are re-derived per row. Here every lane owns EPL columns for the whole launch (one 16-byte piece per 512
// columns plus an 8-byte piece when EPL % 8 == 4), so gamma and the dropout column keys live in registers, the switches are
// template parameters, and the two row sums use DPP adds + v_readlane (wave-uniform results in SGPRs).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {   // all 64 lanes must be active
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror: every lane of a 16-lane row now holds the row's sum
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

template <int EPL, bool DROP_IN, bool DROP_OUT>
__global__ __launch_bounds__(256) void layernorm_bwd_lane_kernel(
    const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ gamma, bf16* __restrict__ dx,
    bf16* __restrict__ dx_drop, float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, uint32_t thr32_in,
    float dscale_in, uint32_t seed_in, uint32_t thr32_out, float dscale_out, uint32_t seed_out, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float sred[];
  constexpr int H = EPL * 64, NQ = EPL / 8;
  constexpr bool TAIL = (EPL & 4) != 0;
  static_assert(EPL % 4 == 0 && EPL >= 4, "lane layout: 4-element granules");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto col_of = [&](int e) { return e < NQ * 8 ? (e >> 3) * 512 + lane * 8 + (e & 7) : NQ * 512 + lane * 4 + (e & 3); };

  float gam[EPL], ag[EPL], ab[EPL];
  uint32_t ck[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    gam[e] = gamma[col_of(e)];
    ck[e] = stonk_colkey((uint32_t)col_of(e));
    ag[e] = ab[e] = 0.f;
  }

  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  bf16x8 nx8[NQ > 0 ? NQ : 1], nd8[NQ > 0 ? NQ : 1];
  bf16x4 nx4, nd4;
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](long r) {
    const bf16* xr = x + r * H;
    const bf16* dr = dy + r * H;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      nx8[q] = *(const bf16x8*)(xr + q * 512 + lane * 8);
      nd8[q] = *(const bf16x8*)(dr + q * 512 + lane * 8);
    }
    if (TAIL) {
      nx4 = *(const bf16x4*)(xr + NQ * 512 + lane * 4);
      nd4 = *(const bf16x4*)(dr + NQ * 512 + lane * 4);
    }
    nmean = mean_in[r];
    nrstd = rstd_in[r];
  };
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    float xh[EPL], g[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      xh[e] = e < NQ * 8 ? (float)nx8[e >> 3][e & 7] : (float)nx4[e & 3];
      g[e] = e < NQ * 8 ? (float)nd8[e >> 3][e & 7] : (float)nd4[e & 3];
    }
    const float mean = nmean, rstd = nrstd;
    if (row + stride < rows) fetch(row + stride);
    const uint32_t rk_in = stonk_rowkey((uint32_t)row, seed_in), rk_out = stonk_rowkey((uint32_t)row, seed_out);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      float d = g[e];
      if (DROP_IN) d = stonk_keep_key(rk_in, ck[e], thr32_in) ? d * dscale_in : 0.f;
      const float h = (xh[e] - mean) * rstd;
      ag[e] += d * h;
      ab[e] += d;
      const float gg = d * gam[e];
      xh[e] = h;
      g[e] = gg;
      s1 += gg;
      s2 += gg * h;
    }
    const float c1 = wave_sum_uniform(s1) * (1.f / (float)H), c2 = wave_sum_uniform(s2) * (1.f / (float)H);
    bf16* dxr = dx + row * H;
    bf16* ddr = dx_drop + row * H;
    float v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = rstd * (g[e] - c1 - xh[e] * c2);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)v[q * 8 + j];
      *(bf16x8*)(dxr + q * 512 + lane * 8) = o;
    }
    if (TAIL) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16)v[NQ * 8 + j];
      *(bf16x4*)(dxr + NQ * 512 + lane * 4) = o;
    }
    if (DROP_OUT) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[e] = stonk_keep_key(rk_out, ck[e], thr32_out) ? v[e] * dscale_out : 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)v[q * 8 + j];
        *(bf16x8*)(ddr + q * 512 + lane * 8) = o;
      }
      if (TAIL) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[NQ * 8 + j];
        *(bf16x4*)(ddr + NQ * 512 + lane * 4) = o;
      }
    }
  }
  if (dgamma) {   // per-wave column partials -> LDS -> one partial row per workgroup (or atomics without a workspace)
    float* sg = sred;          // [4][H]
    float* sb = sred + 4 * H;  // [4][H]
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      sg[wave * H + col_of(e)] = ag[e];
      sb[wave * H + col_of(e)] = ab[e];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256) {
      const float gs = sg[col] + sg[H + col] + sg[2 * H + col] + sg[3 * H + col];
      const float bs = sb[col] + sb[H + col] + sb[2 * H + col] + sb[3 * H + col];
      if (ws) {
        ws[(long)blockIdx.x * 2 * H + col] = gs;
        ws[(long)blockIdx.x * 2 * H + H + col] = bs;
      } else {
        atomicAdd(dgamma + col, gs);
        atomicAdd(dbeta + col, bs);
      }
    }
  }
}

// Forward in the same lane-owned-column layout (H = 64 * EPL): gamma / beta in registers for the whole launch, the next row's
// loads in flight while the current one is reduced, both statistics by DPP adds + v_readlane.
template <int EPL, bool DROP>
__global__ __launch_bounds__(256) void layernorm_fwd_lane_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, bf16* __restrict__ y,
                                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                 long rows, float eps, uint32_t thr32, float dscale,
                                                                 uint32_t seed) {
  constexpr int H = EPL * 64, NQ = EPL / 8;
  constexpr bool TAIL = (EPL & 4) != 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto col_of = [&](int e) { return e < NQ * 8 ? (e >> 3) * 512 + lane * 8 + (e & 7) : NQ * 512 + lane * 4 + (e & 3); };
  float gam[EPL], bet[EPL];
  uint32_t ck[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    gam[e] = gamma[col_of(e)];
    bet[e] = beta[col_of(e)];
    ck[e] = stonk_colkey((uint32_t)col_of(e));
  }
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  bf16x8 nx8[NQ > 0 ? NQ : 1];
  bf16x4 nx4;
  auto fetch = [&](long r) {
    const bf16* xr = x + r * H;
#pragma unroll
    for (int q = 0; q < NQ; ++q) nx8[q] = *(const bf16x8*)(xr + q * 512 + lane * 8);
    if (TAIL) nx4 = *(const bf16x4*)(xr + NQ * 512 + lane * 4);
  };
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    float v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = e < NQ * 8 ? (float)nx8[e >> 3][e & 7] : (float)nx4[e & 3];
    if (row + stride < rows) fetch(row + stride);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += v[e];
    const float mean = wave_sum_uniform(s) * (1.f / (float)H);
    float q2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      v[e] -= mean;
      q2 += v[e] * v[e];
    }
    const float rstd = rsqrtf(wave_sum_uniform(q2) * (1.f / (float)H) + eps);
    if (lane == 0 && mean_out) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    const uint32_t rk = stonk_rowkey((uint32_t)row, seed);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      float o = v[e] * rstd * gam[e] + bet[e];
      if (DROP) o = stonk_keep_key(rk, ck[e], thr32) ? o * dscale : 0.f;
      v[e] = o;
    }
    bf16* yr = y + row * H;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)v[q * 8 + j];
      *(bf16x8*)(yr + q * 512 + lane * 8) = o;
    }
    if (TAIL) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16)v[NQ * 8 + j];
      *(bf16x4*)(yr + NQ * 512 + lane * 4) = o;
    }
  }
}

// dgamma[c] += sum_b ws[b][c], dbeta[c] += sum_b ws[b][H + c]: grid (2H/256, 32) - each workgroup sums 1/32 of the
// partial rows for 256 columns (coalesced 1 KiB row segments), then 32 adders per address finish with atomics.
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ ws, int nb, int H,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;   // 0 .. 2H-1
  if (c >= 2 * H) return;
  const int per = (nb + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per;
  const int b1 = b0 + per < nb ? b0 + per : nb;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    a0 += ws[(long)b * 2 * H + c];
    a1 += ws[(long)(b + 1) * 2 * H + c];
    a2 += ws[(long)(b + 2) * 2 * H + c];
    a3 += ws[(long)(b + 3) * 2 * H + c];
  }
  for (; b < b1; ++b) a0 += ws[(long)b * 2 * H + c];
  const float s = (a0 + a1) + (a2 + a3);
  if (b1 > b0) atomicAdd(c < H ? dgamma + c : dbeta + (c - H), s);
}

// K2+K3: gather / concat / position + token-type add / LayerNorm in one pass.
//  text half  (s <  half): source row = frozen-backbone hidden state (bf16)
//  entity half(s >= half): source row = kg_table[input_ids[b,s]] (fp32; rows 100/102/103 hold the LM
//                          special-token vectors, quirks Q1/Q2 of SURVEY.md section 8)
template <int CPL>
__global__ __launch_bounds__(256) void joint_embed_ln_kernel(
    const long* __restrict__ input_ids, const long* __restrict__ token_type_ids, const bf16* __restrict__ text_hidden,
    const float* __restrict__ kg_table, const float* __restrict__ pos_emb, const float* __restrict__ type_emb,
    const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ sum_out, bf16* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int B, int S, int half, int H, long kg_rows,
    int type_rows, float eps, int flags, uint32_t thr32, float dscale, uint32_t seed, int* __restrict__ err,
    const int* __restrict__ pos_of_row, long n_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  const long rows = pos_of_row ? n_rows : (long)B * S;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    // packed layout (stonk_unpad_plan): output row `row` is padded position pos_of_row[row]; a row that belongs to no
    // position (the tail up to the launch's row count) is written as zeros - finite for everything downstream, and with
    // zero gradient coming back
    const long pos = pos_of_row ? pos_of_row[row] : row;
    if (pos < 0) {
      RowRegs<CPL> z;
      zero_row<CPL>(z);
      if (sum_out) store_bf16_row<CPL>(sum_out + row * H, nch, lane, z);
      store_bf16_row<CPL>(y + row * H, nch, lane, z);
      if (lane == 0 && mean_out) {
        mean_out[row] = 0.f;
        rstd_out[row] = 1.f;
      }
      continue;
    }
    const int b = (int)(pos / S), s = (int)(pos - (long)b * S);
    RowRegs<CPL> r;
    if (s < half) {
      load_bf16_row<CPL>(text_hidden + ((long)b * half + s) * H, nch, lane, r);
    } else {
      long id = input_ids[pos];
      if (id < 0 || id >= kg_rows) {  // the reference raises KeyError here (stonkgs_model.py:185)
        if (lane == 0) atomicOr(err, 1);
        id = 0;
      }
      load_f32_row<CPL>(kg_table + id * H, nch, lane, r);
    }
    long tt = token_type_ids ? token_type_ids[pos] : 0;
    if (tt < 0 || tt >= type_rows) {
      if (lane == 0) atomicOr(err, 2);
      tt = 0;
    }
    add_f32_row<CPL>(pos_emb + (long)s * H, nch, lane, r);
    add_f32_row<CPL>(type_emb + tt * H, nch, lane, r);
    if (sum_out) store_bf16_row<CPL>(sum_out + row * H, nch, lane, r);
    float mean, rstd;
    row_stats<CPL>(r, nch, lane, H, eps, mean, rstd);
    if (lane == 0 && mean_out) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    normalize_store<CPL>(r, nch, lane, mean, rstd, gamma, beta, y + row * H, row, H, flags & STONK_LN_DROPOUT, thr32,
                         dscale, seed);
  }
}

// Frozen LM backbone front-end: word_emb[ids] + pos_emb + type_emb[0] -> LayerNorm (hf BertEmbeddings :98-108)
template <int CPL>
__global__ __launch_bounds__(256) void text_embed_ln_kernel(const long* __restrict__ input_ids, long ld_ids,
                                                            const float* __restrict__ word_emb,
                                                            const float* __restrict__ pos_emb,
                                                            const float* __restrict__ type_emb,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ y, int B,
                                                            int S, int H, long vocab, float eps, int flags,
                                                            uint32_t thr32, float dscale, uint32_t seed,
                                                            int* __restrict__ err) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  const long rows = (long)B * S;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / S), s = (int)(row - (long)b * S);
    long id = input_ids[(long)b * ld_ids + s];
    if (id < 0 || id >= vocab) {
      if (lane == 0) atomicOr(err, 4);
      id = 0;
    }
    RowRegs<CPL> r;
    load_f32_row<CPL>(word_emb + id * H, nch, lane, r);
    add_f32_row<CPL>(pos_emb + (long)s * H, nch, lane, r);
    add_f32_row<CPL>(type_emb, nch, lane, r);
    float mean, rstd;
    row_stats<CPL>(r, nch, lane, H, eps, mean, rstd);
    normalize_store<CPL>(r, nch, lane, mean, rstd, gamma, beta, y + row * H, row, H, flags & STONK_LN_DROPOUT, thr32,
                         dscale, seed);
  }
}

// d(position_embeddings)[s] += sum_b dx[b,s,:];  d(token_type_embeddings)[t] += sum_{(b,s): tt==t} dx[b,s,:]
// One block per position: a thread owns 8 consecutive columns (16-byte loads) of every EG-th sequence; the row indices
// of its sequences are fetched first and the rows then eight at a time (a thread walking the batch one dependent
// index -> row load after the other took 155 us for 64 x 512 x 768).
constexpr int EG = 8;
__global__ __launch_bounds__(1024) void embed_grad_kernel(const bf16* __restrict__ dx,
                                                          const long* __restrict__ token_type_ids,
                                                          float* __restrict__ dpos, float* __restrict__ dtype, int B,
                                                          int S, int H, int type_rows,
                                                          const int* __restrict__ row_of_pos) {
  const int s = blockIdx.x, nch = H >> 3;
  const int c8 = threadIdx.x % nch, bg = threadIdx.x / nch;   // blockDim = nch * EG
  float accp[8], t0[8], t1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) accp[j] = t0[j] = t1[j] = 0.f;
  for (int b0 = bg; b0 < B; b0 += EG * 8) {
    long row[8], tt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int b = b0 + i * EG;
      const long pos = (long)b * S + s;
      row[i] = b < B ? (row_of_pos ? (long)row_of_pos[pos] : pos) : -1;   // packed layout: a dropped position has no row
      tt[i] = (b < B && token_type_ids) ? token_type_ids[pos] : 0;
    }
    bf16x8 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (row[i] >= 0) v[i] = *(const bf16x8*)(dx + row[i] * H + 8 * c8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (row[i] < 0) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[i][j];
        accp[j] += f;
        if (tt[i] == 0) t0[j] += f;
        else if (tt[i] == 1) t1[j] += f;
      }
    }
  }
  // the EG partial sums of a column meet in LDS (one array, used three times); one thread per column then adds to the
  // outputs: dpos[s] is this block's alone, the two type rows take ONE atomic per block and column (every block adding
  // EG times into the same 2 x H addresses was 6x slower than the whole kernel)
  __shared__ float red[EG][1024];
  float* vals[3] = {accp, t0, t1};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[bg][8 * c8 + j] = vals[a][j];
    __syncthreads();
    if (a == 2 && type_rows <= 1) break;
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < EG; ++g) t += red[g][col];
      if (a == 0) dpos[(long)s * H + col] += t;
      else atomicAdd(dtype + (a - 1) * H + col, t);
    }
  }
}

inline int ln_grid(long rows) {
  long g = (rows + 3) / 4;
  return (int)(g < 2048 ? (g > 0 ? g : 1) : 2048);
}

}  // namespace

#define LN_DISPATCH(H, CALL)                     \
  if ((H) <= 1024) { constexpr int CPL = 2; CALL; } \
  else if ((H) <= 2048) { constexpr int CPL = 4; CALL; } \
  else { constexpr int CPL = 8; CALL; }

extern "C" int stonk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, int64_t rows, int H, float eps, int flags, float drop_p,
                                   uint32_t seed, void* stream) {
  STONK_CHECK_ARG(x && gamma && beta && y, STONK_EINVAL);
  STONK_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 4096, STONK_ESHAPE);
  STONK_CHECK_ARG((mean == nullptr) == (rstd == nullptr), STONK_EINVAL);
  if (rows == 0) return STONK_OK;
  const uint32_t thr = stonk_drop_thr32(drop_p);
  const float ds = 1.f / (1.f - drop_p);
  const bool generic_only = false;   // (the lane-owned-column kernels serve H = 512 / 768 / 1024; other widths take the generic ones)
  const bool drop = (flags & STONK_LN_DROPOUT) != 0;
#define LN_FWD_LANE(EPL)                                                                                                \
  do {                                                                                                                  \
    const int grid = ln_grid(rows) < 1024 ? ln_grid(rows) : 1024;                                                       \
    if (drop)                                                                                                           \
      hipLaunchKernelGGL((layernorm_fwd_lane_kernel<EPL, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,         \
                         (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, (long)rows, eps, thr, ds,                   \
                         stonk_seed_mix(seed));                                                                         \
    else                                                                                                                \
      hipLaunchKernelGGL((layernorm_fwd_lane_kernel<EPL, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,        \
                         (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, (long)rows, eps, thr, ds,                   \
                         stonk_seed_mix(seed));                                                                         \
  } while (0)
  if (H == 768 && !generic_only) LN_FWD_LANE(12);
  else if (H == 1024 && !generic_only) LN_FWD_LANE(16);
  else if (H == 512 && !generic_only) LN_FWD_LANE(8);
  else
  LN_DISPATCH(H, hipLaunchKernelGGL((layernorm_fwd_kernel<CPL>), dim3(ln_grid(rows)), dim3(256), 0,
                                    (hipStream_t)stream, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, (long)rows,
                                    H, eps, flags, thr, ds, stonk_seed_mix(seed)));
  return stonk_launch_status();
}

// Workspace query (launchers never allocate): floats of `partial_ws` with which stonk_layernorm_bwd sums
// dgamma / dbeta through per-workgroup partials instead of contended atomics.
extern "C" int64_t stonk_layernorm_bwd_workspace_floats(int64_t rows, int H) {
  if (rows <= 0 || H <= 0) return 0;
  const int grid = ln_grid(rows) < 1024 ? ln_grid(rows) : 1024;
  return (int64_t)grid * 2 * H;
}

extern "C" int stonk_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                   const float* gamma, void* dx, void* dx_drop, float* dgamma, float* dbeta,
                                   int64_t rows, int H, int flags, float drop_p_in, uint32_t seed_in,
                                   float drop_p_out, uint32_t seed_out, float* partial_ws, int64_t ws_floats,
                                   void* stream) {
  STONK_CHECK_ARG(dy && x && mean && rstd && gamma && dx, STONK_EINVAL);
  STONK_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 4096, STONK_ESHAPE);
  STONK_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), STONK_EINVAL);
  if (rows == 0) return STONK_OK;
  const size_t lds = dgamma ? (size_t)8 * H * sizeof(float) : 0;
  const int grid = ln_grid(rows) < 1024 ? ln_grid(rows) : 1024;
  // workspace (grid x 2H floats) given: partial sums + a reduce kernel instead of contended atomics
  float* ws = (dgamma && partial_ws && ws_floats >= (int64_t)grid * 2 * H) ? partial_ws : nullptr;
  if (flags & STONK_LN_DEFER_REDUCE) STONK_CHECK_ARG(ws != nullptr, STONK_EINVAL);   // (needs the workspace)
  const bool din = (flags & STONK_LN_DROPOUT) != 0, dout = dx_drop != nullptr;
#define LN_BWD_LANE(EPL, DI, DO)                                                                                       \
  hipLaunchKernelGGL((layernorm_bwd_lane_kernel<EPL, DI, DO>), dim3(grid), dim3(256), lds, (hipStream_t)stream,         \
                     (const bf16*)dy, (const bf16*)x, mean, rstd, gamma, (bf16*)dx, (bf16*)dx_drop, dgamma, dbeta,      \
                     (long)rows, stonk_drop_thr32(drop_p_in), 1.f / (1.f - drop_p_in), stonk_seed_mix(seed_in),         \
                     stonk_drop_thr32(drop_p_out), 1.f / (1.f - drop_p_out), stonk_seed_mix(seed_out), ws)
#define LN_BWD_LANE_H(EPL)                                  \
  do {                                                      \
    if (din && dout) LN_BWD_LANE(EPL, true, true);          \
    else if (din) LN_BWD_LANE(EPL, true, false);            \
    else if (dout) LN_BWD_LANE(EPL, false, true);           \
    else LN_BWD_LANE(EPL, false, false);                    \
  } while (0)
  const bool generic_only = false;
  if (H == 768 && !generic_only) LN_BWD_LANE_H(12);
  else if (H == 1024 && !generic_only) LN_BWD_LANE_H(16);
  else if (H == 512 && !generic_only) LN_BWD_LANE_H(8);
  else
  LN_DISPATCH(H, hipLaunchKernelGGL((layernorm_bwd_kernel<CPL>), dim3(grid), dim3(256), lds, (hipStream_t)stream,
                                    (const bf16*)dy, (const bf16*)x, mean, rstd, gamma, (bf16*)dx, (bf16*)dx_drop,
                                    dgamma, dbeta, (long)rows, H, flags, stonk_drop_thr32(drop_p_in),
                                    1.f / (1.f - drop_p_in), stonk_seed_mix(seed_in), stonk_drop_thr32(drop_p_out),
                                    1.f / (1.f - drop_p_out), stonk_seed_mix(seed_out), ws));
  // STONK_LN_DEFER_REDUCE: the partial sums stay in the workspace; the caller finishes with stonk_layernorm_bwd_reduce
  // (the training step puts that launch on its weight-gradient stream: nothing on the main chain waits for dgamma / dbeta)
  if (ws && !(flags & STONK_LN_DEFER_REDUCE))
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((2 * H + 255) / 256, 32), dim3(256), 0, (hipStream_t)stream, ws, grid,
                       H, dgamma, dbeta);
  return stonk_launch_status();
}

extern "C" int stonk_layernorm_bwd_reduce(const float* partial_ws, int64_t rows, int H, float* dgamma, float* dbeta,
                                          void* stream) {
  STONK_CHECK_ARG(partial_ws && dgamma && dbeta, STONK_EINVAL);
  STONK_CHECK_ARG(rows >= 0 && H > 0 && H % 8 == 0 && H <= 4096, STONK_ESHAPE);
  if (rows == 0) return STONK_OK;
  const int grid = ln_grid(rows) < 1024 ? ln_grid(rows) : 1024;   // (as stonk_layernorm_bwd launched for these rows)
  hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((2 * H + 255) / 256, 32), dim3(256), 0, (hipStream_t)stream, partial_ws,
                     grid, H, dgamma, dbeta);
  return stonk_launch_status();
}

extern "C" int stonk_joint_embed_ln_fwd(const int64_t* input_ids, const int64_t* token_type_ids,
                                        const void* text_hidden, const float* kg_table, const float* pos_emb,
                                        const float* type_emb, const float* gamma, const float* beta, void* sum_out,
                                        void* y, float* mean, float* rstd, int B, int S, int half, int H,
                                        int64_t kg_rows, int type_rows, float eps, int flags, float drop_p,
                                        uint32_t seed, int* err_flag, const int* pos_of_row, int64_t n_rows,
                                        void* stream) {
  STONK_CHECK_ARG(input_ids && text_hidden && kg_table && pos_emb && type_emb && gamma && beta && y && err_flag,
                  STONK_EINVAL);
  STONK_CHECK_ARG(B >= 0 && S > 0 && half >= 0 && half <= S && H > 0 && H % 8 == 0 && H <= 4096, STONK_ESHAPE);
  STONK_CHECK_ARG(kg_rows > 0 && type_rows > 0, STONK_ESHAPE);
  STONK_CHECK_ARG(!pos_of_row || (n_rows >= 0 && n_rows <= (long)B * S), STONK_ESHAPE);
  if (B == 0) return STONK_OK;
  const long rows = pos_of_row ? (long)n_rows : (long)B * S;
  if (rows == 0) return STONK_OK;
  LN_DISPATCH(H, hipLaunchKernelGGL((joint_embed_ln_kernel<CPL>), dim3(ln_grid(rows)), dim3(256), 0,
                                    (hipStream_t)stream, (const long*)input_ids, (const long*)token_type_ids,
                                    (const bf16*)text_hidden, kg_table, pos_emb, type_emb, gamma, beta,
                                    (bf16*)sum_out, (bf16*)y, mean, rstd, B, S, half, H, (long)kg_rows, type_rows, eps,
                                    flags, stonk_drop_thr32(drop_p), 1.f / (1.f - drop_p), stonk_seed_mix(seed), err_flag,
                                    pos_of_row, (long)n_rows));
  return stonk_launch_status();
}

extern "C" int stonk_text_embed_ln_fwd(const int64_t* input_ids, int64_t ld_ids, const float* word_emb,
                                       const float* pos_emb, const float* type_emb, const float* gamma,
                                       const float* beta, void* y, int B, int S, int H, int64_t vocab, float eps,
                                       int flags, float drop_p, uint32_t seed, int* err_flag, void* stream) {
  STONK_CHECK_ARG(input_ids && word_emb && pos_emb && type_emb && gamma && beta && y && err_flag, STONK_EINVAL);
  STONK_CHECK_ARG(B >= 0 && S > 0 && ld_ids >= S && H > 0 && H % 8 == 0 && H <= 4096 && vocab > 0, STONK_ESHAPE);
  if (B == 0) return STONK_OK;
  const long rows = (long)B * S;
  LN_DISPATCH(H, hipLaunchKernelGGL((text_embed_ln_kernel<CPL>), dim3(ln_grid(rows)), dim3(256), 0,
                                    (hipStream_t)stream, (const long*)input_ids, (long)ld_ids, word_emb, pos_emb,
                                    type_emb, gamma, beta, (bf16*)y, B, S, H, (long)vocab, eps, flags,
                                    stonk_drop_thr32(drop_p), 1.f / (1.f - drop_p), stonk_seed_mix(seed), err_flag));
  return stonk_launch_status();
}

extern "C" int stonk_embed_grad(const void* dx, const int64_t* token_type_ids, float* dpos, float* dtype, int B, int S,
                                int H, int type_rows, const int* row_of_pos, void* stream) {
  STONK_CHECK_ARG(dx && dpos && dtype, STONK_EINVAL);
  STONK_CHECK_ARG(B >= 0 && S > 0 && H > 0 && H % 8 == 0 && H <= 1024 && type_rows >= 1, STONK_ESHAPE);
  STONK_CHECK_ARG((uintptr_t)dx % 16 == 0, STONK_EALIGN);
  if (B == 0) return STONK_OK;
  hipLaunchKernelGGL(embed_grad_kernel, dim3(S), dim3((H >> 3) * EG), 0, (hipStream_t)stream, (const bf16*)dx,
                     (const long*)token_type_ids, dpos, dtype, B, S, H, type_rows, row_of_pos);
  return stonk_launch_status();
}
