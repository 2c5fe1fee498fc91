// HBM-bound elementwise / layout kernels of the optimizer step (K16 in SURVEY.md section 2.3):
// global grad-norm, fused clip + AdamW + bf16 weight refresh + grad zeroing over ONE flat fp32 buffer,
// and the tiled transposes that feed the NT GEMM (wgrad operands, W^T copies for dgrad).
//
// Semantics follow what the reference's driver calls (ref:src/stonkgs/models/stonkgs_pretraining.py:171-223):
// hf:trainer.py:1780-1796 -> clip_grad_norm_(max_norm=1.0) then torch.optim.AdamW(betas=(0.9,0.999),
// eps=1e-8, weight_decay=0.0).step(), then model.zero_grad().
#include "common.h"
#include "stonk_flags.h"

namespace {

// Sum of squares in a FIXED order: every block leaves its partial sum in a slot, the last block to finish (ticket) adds
// the slots up in index order. With one atomicAdd per block the result depended on arrival order in its last bits - and
// with it the clip coefficient and every parameter update, so data-parallel ranks holding identical summed gradients
// drifted apart by an ulp per step (found with tools/dp_check.py). Slots and ticket live in a CALLER workspace
// (stonk_sumsq_workspace_floats(): SUMSQ_MAX_BLOCKS slots + the ticket word, zeroed once by the caller; the kernel
// leaves the ticket at zero), so launches on different streams with different workspaces do not meet.
// Grid cap = number of partial-sum slots. Fewer, longer-running workgroups stream better: alone on 974 MB (tools/sumsq_probe.py,
// us) 256 blocks 181, 384: 179, 512: 193, 768: 213, 1024: 279 (3.5 TB/s - the round-3 value), 2048: 376, 4096: 569.
#ifndef STONK_SUMSQ_BLOCKS
#define STONK_SUMSQ_BLOCKS 256
#endif
constexpr int SUMSQ_MAX_BLOCKS = STONK_SUMSQ_BLOCKS;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out,
                                                    float* __restrict__ g_sumsq_part, unsigned* __restrict__ ticket) {
  float acc = 0.f;
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  // four 16-byte loads requested before the first is used (a pure read of 974 MB: with one load in flight per thread it ran
  // at 3.4 TB/s); the summation order stays a function of (n, grid) alone - the same bits on every launch and rank
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 v0 = *(const f32x4*)(x + 4 * i), v1 = *(const f32x4*)(x + 4 * (i + stride)),
                v2 = *(const f32x4*)(x + 4 * (i + 2 * stride)), v3 = *(const f32x4*)(x + 4 * (i + 3 * stride));
    acc += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3];
    acc += v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
    acc += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3];
    acc += v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
  }
  for (; i < n4; i += stride) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += x[i] * x[i];
  acc = wave_sum(acc);
  __shared__ float part[4];
  __shared__ bool last;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&g_sumsq_part[blockIdx.x], part[0] + part[1] + part[2] + part[3], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();   // the slot is visible device-wide before the ticket is taken
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float tot = 0.f;   // 256 threads: strided slots, then a fixed tree - the same order on every launch and every rank
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
    tot += __hip_atomic_load(&g_sumsq_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tot = wave_sum(tot);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
}

// One pass over the flat buffers: p, g, m, v read; p, m, v, bf16(p) written; g zeroed - 34 bytes per parameter, each touched
// once per step (8.3 GB at 243 M parameters: nothing a cache could keep). HBM-bound. Round 4 (tools/adamw_ab.py, 243 M
// elements, one box): 1, 2 or 4 groups requested per thread before the first use, default or non-temporal accesses -
// 4.30 / 4.41 / 4.27 TB/s and 4.25 / 4.39 / 4.40: the mix of four read and five write streams runs at 4.3-4.4 TB/s whatever
// a thread keeps in flight (a plain copy reaches 6.3 on this chip). Two groups, default policy.
#ifndef STONK_ADAMW_UNROLL
#define STONK_ADAMW_UNROLL 2
#endif
#ifndef STONK_ADAMW_NT
#define STONK_ADAMW_NT 0
#endif
#ifndef STONK_ADAMW_BLOCKS
// grid cap. Alone on 243 M elements (ADAMW_AB=blocks tools/adamw_ab.py, us): 256 blocks 1516, 512: 1588, 1024: 1715, 4096: 1735 -
// but IN THE STEP 256 blocks cost +1.6 ms (27.9-28.4 against 26.3-26.7 ms, tools/ab_libs.sh): the kernel runs on the optimizer
// stream beside the next step's first launches, and 256 long-lived workgroups are placed once, on whatever CUs are free at
// that moment, where 4096 short ones follow the CUs as they come free. (The gradient-norm pass before it does gain from
// its 256 blocks in the step: -0.15 ms.)
#define STONK_ADAMW_BLOCKS 4096
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T* q) {
#if STONK_ADAMW_NT
  return __builtin_nontemporal_load(q);
#else
  return *q;
#endif
}
template <typename T>
__device__ __forceinline__ void stream_store(T* q, const T& x) {
#if STONK_ADAMW_NT
  __builtin_nontemporal_store(x, q);
#else
  *q = x;
#endif
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ pb, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ gnorm_sq, float max_norm,
                                                    float grad_scale, const long* __restrict__ decay_spans,
                                                    int n_spans, long span_base) {
  constexpr int U = STONK_ADAMW_UNROLL;
  float coef = grad_scale;
  if (gnorm_sq && max_norm > 0.f) {
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float total = sqrtf(*gnorm_sq) * grad_scale;
    const float c = max_norm / (total + 1e-6f);
    coef *= c < 1.f ? c : 1.f;
  }
  const float step = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) {
        pp[u] = stream_load((const f32x4*)(p + 4 * i));
        gg[u] = stream_load((const f32x4*)(g + 4 * i));
        mm[u] = stream_load((const f32x4*)(m + 4 * i));
        vv[u] = stream_load((const f32x4*)(v + 4 * i));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i >= n4) break;
      bf16x4 o;
      // decoupled weight decay: everywhere (no table), or on the elements inside one of the sorted [lo, hi) spans of the
      // flat buffer (HF Trainer's grouping: weights yes, biases and LayerNorm no). Tensors start at multiples of 256
      // elements, so the four elements of a group share the answer.
      float keep = 1.f - lr * wd;
      if (decay_spans && wd != 0.f) {
        const long e = span_base + 4 * i;
        int lo = 0, hi = n_spans;
        while (lo < hi) {   // last span whose start <= e
          const int mid = (lo + hi) >> 1;
          if (decay_spans[2 * mid] <= e) lo = mid + 1;
          else hi = mid;
        }
        if (lo == 0 || e >= decay_spans[2 * (lo - 1) + 1]) keep = 1.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gg[u][j] * coef;
        float w = pp[u][j] * keep;
        mm[u][j] = b1 * mm[u][j] + (1.f - b1) * gr;
        vv[u][j] = b2 * vv[u][j] + (1.f - b2) * gr * gr;
        const float denom = sqrtf(vv[u][j]) * inv_sqrt_bc2 + eps;
        w -= step * (mm[u][j] / denom);
        pp[u][j] = w;
        o[j] = (bf16)w;
      }
      stream_store((f32x4*)(p + 4 * i), pp[u]);
      stream_store((f32x4*)(m + 4 * i), mm[u]);
      stream_store((f32x4*)(v + 4 * i), vv[u]);
      stream_store((f32x4*)(g + 4 * i), (f32x4){0.f, 0.f, 0.f, 0.f});
      if (pb) *(bf16x4*)(pb + 4 * i) = o;   // (read again by the next forward: default policy)
    }
  }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= s;
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    *(bf16x4*)(y + 4 * i) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) y[i] = (bf16)x[i];
}

// 64x64 tile transpose through LDS. SrcT = bf16 or float; output bf16. Rows at or past *rows_dev read as zero
// (label-sparse decoder operands); optional column sums of the source (bias gradients) via atomics.
constexpr int TT = 64;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
__device__ __forceinline__ unsigned short bf16_bits(float f) {
  const bf16 b = (bf16)f;
  unsigned short u;
  __builtin_memcpy(&u, &b, 2);
  return u;
}
__device__ __forceinline__ float bits_to_float(unsigned short u) {
  return __uint_as_float(((unsigned)u) << 16);
}
template <typename SrcT>
__global__ __launch_bounds__(256) void transpose_kernel(const SrcT* __restrict__ in, long ld_in, bf16* __restrict__ out,
                                                        long ld_out, long rows, int cols, float* __restrict__ colsum,
                                                        const int* __restrict__ rows_dev) {
  __shared__ unsigned short tile[TT][TT + 2];
  long live = rows;
  if (rows_dev) {
    const long rd = *rows_dev;
    live = rd < rows ? rd : rows;
  }
  // only tiles up to the 64-row round-up of the live count are ever consumed; blocks walk them with stride gridDim.y
  const long live_tiles = (live + TT - 1) / TT;
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * TT;
  for (long rt = blockIdx.y; rt < live_tiles; rt += gridDim.y) {
    const long r0 = rt * TT;
    // load: thread -> (row = t >> 3 (+32), 8 columns)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int rr = (t >> 3) + 32 * pass;
      const int cc = (t & 7) * 8;
      const long r = r0 + rr;
      unsigned short vals[8];
      if (r < live && c0 + cc < cols) {
        if constexpr (sizeof(SrcT) == 2) {
          const u16x8 v = *(const u16x8*)(in + r * ld_in + c0 + cc);
#pragma unroll
          for (int j = 0; j < 8; ++j) vals[j] = v[j];
        } else {
          const f32x4 a = *(const f32x4*)(in + r * ld_in + c0 + cc);
          const f32x4 b = *(const f32x4*)(in + r * ld_in + c0 + cc + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vals[j] = bf16_bits(a[j]);
            vals[4 + j] = bf16_bits(b[j]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[j] = 0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[rr][cc + j] = vals[j];
    }
    __syncthreads();
    // store: thread -> (out row = column c0 + (t >> 2), 16 source rows = 32 contiguous bytes)
    {
      const int oc = t >> 2;
      const int rb = (t & 3) * 16;
      if (c0 + oc < cols) {
        unsigned short vals[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) vals[j] = tile[rb + j][oc];
        bf16* dst = out + (long)(c0 + oc) * ld_out + r0 + rb;
        u16x8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o0[j] = vals[j];
          o1[j] = vals[8 + j];
        }
        *(u16x8*)dst = o0;
        *(u16x8*)(dst + 8) = o1;
      }
    }
    if (colsum && t < TT && c0 + t < cols) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < TT; ++r) s += bits_to_float(tile[r][t]);
      atomicAdd(colsum + c0 + t, s);
    }
    __syncthreads();
  }
}

// Batched form of the bf16 transpose: ONE launch refreshes every W^T copy of the model (57 tensors per optimizer step; as
// separate launches they were 1.2 ms of mostly launch gaps and small grids on the optimizer stream). `desc` is a
// device-side table of n entries {in, out, ld_in, ld_out, rows, cols, first_tile, tiles_per_row}; the grid is the total
// number of 64x64 tiles and a workgroup finds its entry by bisection on first_tile.
struct TransposeDesc {
  const bf16* in;
  bf16* out;
  long ld_in, ld_out, rows;
  int cols, first_tile, col_tiles, pad;
};
__global__ __launch_bounds__(256) void transpose_batched_kernel(const TransposeDesc* __restrict__ desc, int n) {
  __shared__ unsigned short tile[TT][TT + 2];
  const int bid = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {   // last entry whose first_tile <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].first_tile <= bid) lo = mid;
    else hi = mid - 1;
  }
  const TransposeDesc d = desc[lo];
  const int tl = bid - d.first_tile;
  const long r0 = (long)(tl / d.col_tiles) * TT;
  const int c0 = (tl % d.col_tiles) * TT;
  const int t = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rr = (t >> 3) + 32 * pass, cc = (t & 7) * 8;
    const long r = r0 + rr;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < d.rows && c0 + cc < d.cols) v = *(const u16x8*)(d.in + r * d.ld_in + c0 + cc);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[rr][cc + j] = v[j];
  }
  __syncthreads();
  const int oc = t >> 2, rb = (t & 3) * 16;
  if (c0 + oc < d.cols) {
    u16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = tile[rb + j][oc];
      o1[j] = tile[rb + 8 + j][oc];
    }
    bf16* dst = d.out + (long)(c0 + oc) * d.ld_out + r0 + rb;
    *(u16x8*)dst = o0;
    *(u16x8*)(dst + 8) = o1;
  }
}

inline int ew_grid(long n) {
  long g = (n / 4 + 255) / 256;
  if (g < 1) g = 1;
  return (int)(g < 4096 ? g : 4096);
}

}  // namespace

extern "C" int64_t stonk_sumsq_workspace_floats(void) { return SUMSQ_MAX_BLOCKS + 1; }

extern "C" int stonk_sumsq_f32(const float* x, int64_t n, float* out_accum, float* workspace, int64_t ws_floats,
                               void* stream) {
  STONK_CHECK_ARG(x && out_accum && workspace && n >= 0, STONK_EINVAL);
  STONK_CHECK_ARG(ws_floats >= SUMSQ_MAX_BLOCKS + 1, STONK_EINVAL);
  STONK_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)workspace % 4 == 0, STONK_EALIGN);
  if (n == 0) return STONK_OK;
  const int grid = ew_grid(n) < SUMSQ_MAX_BLOCKS ? ew_grid(n) : SUMSQ_MAX_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long)n, out_accum, workspace,
                     (unsigned*)(workspace + SUMSQ_MAX_BLOCKS));
  return stonk_launch_status();
}

extern "C" int stonk_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                                const float* gnorm_sq_dev, float max_grad_norm, float grad_scale,
                                const int64_t* decay_spans, int n_spans, int64_t span_base, void* stream) {
  STONK_CHECK_ARG(p && g && m && v && n >= 0 && n % 4 == 0, STONK_EINVAL);
  STONK_CHECK_ARG(n_spans >= 0 && (decay_spans || n_spans == 0) && span_base >= 0 && span_base % 4 == 0, STONK_EINVAL);
  STONK_CHECK_ARG(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, STONK_EALIGN);
  STONK_CHECK_ARG(bias_corr1 > 0.f && bias_corr2 > 0.f, STONK_EINVAL);
  if (n == 0) return STONK_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n) < STONK_ADAMW_BLOCKS ? ew_grid(n) : STONK_ADAMW_BLOCKS), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16,
                     (long)n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, gnorm_sq_dev, max_grad_norm,
                     grad_scale, (const long*)decay_spans, n_spans, (long)span_base);
  return stonk_launch_status();
}

extern "C" int stonk_scale_f32(float* x, int64_t n, float s, void* stream) {
  STONK_CHECK_ARG(x && n >= 0, STONK_EINVAL);
  if (n == 0) return STONK_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (long)n, s);
  return stonk_launch_status();
}

extern "C" int stonk_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  STONK_CHECK_ARG(x && y && n >= 0, STONK_EINVAL);
  STONK_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 8 == 0, STONK_EALIGN);
  if (n == 0) return STONK_OK;
  hipLaunchKernelGGL(cast_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, (long)n);
  return stonk_launch_status();
}

// out[c][r] = in[r][c] for r < rows (rows >= *rows_dev read as zero), c < cols. out needs ld_out >= roundup(rows, 64).
extern "C" int stonk_transpose_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int cols,
                                    float* colsum, const int* rows_dev, void* stream) {
  STONK_CHECK_ARG(in && out && rows >= 0 && cols > 0, STONK_EINVAL);
  STONK_CHECK_ARG(cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, STONK_EALIGN);
  STONK_CHECK_ARG(ld_out >= ((rows + TT - 1) / TT) * TT, STONK_ESHAPE);
  if (rows == 0) return STONK_OK;
  long rtiles = (rows + TT - 1) / TT;
  if (rows_dev && rtiles > 64) rtiles = 64;  // live count unknown on the host: bounded grid, blocks stride
  if (rtiles > 65535) rtiles = 65535;
  const dim3 grid((cols + TT - 1) / TT, (unsigned)rtiles);
  hipLaunchKernelGGL((transpose_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)in, (long)ld_in,
                     (bf16*)out, (long)ld_out, (long)rows, cols, colsum, rows_dev);
  return stonk_launch_status();
}

extern "C" int stonk_transpose_bf16_batched(const void* desc_dev, int n, int total_tiles, void* stream) {
  STONK_CHECK_ARG(desc_dev && n > 0 && total_tiles > 0, STONK_EINVAL);
  STONK_CHECK_ARG((uintptr_t)desc_dev % 8 == 0, STONK_EALIGN);
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const TransposeDesc*)desc_dev, n);
  return stonk_launch_status();
}

// bf16 W^T copy of a dense fp32 [rows, cols] master weight: out[c][r] = bf16(in[r][c])
extern "C" int stonk_transpose_f32_to_bf16(const float* in, void* out, int64_t rows, int cols, int64_t ld_out,
                                           void* stream) {
  STONK_CHECK_ARG(in && out && rows >= 0 && cols > 0, STONK_EINVAL);
  STONK_CHECK_ARG(cols % 8 == 0 && ld_out % 8 == 0, STONK_EALIGN);
  STONK_CHECK_ARG(ld_out >= ((rows + TT - 1) / TT) * TT, STONK_ESHAPE);
  if (rows == 0) return STONK_OK;
  long rtiles = (rows + TT - 1) / TT;
  if (rtiles > 65535) rtiles = 65535;
  const dim3 grid((cols + TT - 1) / TT, (unsigned)rtiles);
  hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, in, (long)cols, (bf16*)out,
                     (long)ld_out, (long)rows, cols, (float*)nullptr, (const int*)nullptr);
  return stonk_launch_status();
}
