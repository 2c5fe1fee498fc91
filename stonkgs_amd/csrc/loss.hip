// Loss-side kernels of the STonKGs step (K11-K14 in SURVEY.md section 2.3): label compaction, row
// gather/scatter, fused softmax + cross-entropy (forward value AND logits gradient in one sweep), NSP
// cross-entropy, loss finalisation.
//
// Reference semantics: three nn.CrossEntropyLoss() (mean over targets != -100) summed,
// ref:src/stonkgs/models/stonkgs_model.py:223-245. Only rows whose label is not -100 contribute to the
// loss or to any gradient, so the decoders run on the compacted labelled rows (identical loss and
// gradients, SURVEY.md section 8d "label-sparse"); the row count lives in device memory - no host sync.
#include "common.h"
#include "stonk_flags.h"

namespace {

// Stable compaction of labels != -100 (one block; n = B * half is a few 10^4 at most).
// rows_out[i] = token row (b*S + offset + pos) of the i-th labelled position, targets_out[i] = its label.
// In chunks of 16 384 labels: sixteen COALESCED loads per thread requested at once (label j * 1024 + t), their ballots to
// LDS, one exclusive scan over the 256 ballot words' popcounts, then every thread writes its labelled ones - three barriers
// per chunk. (The first version walked the labels 1024 at a time with three barriers and a dependent global load per round:
// 32 us for 16 384 labels, twice per step; a contiguous run per thread, uncoalesced: 21 us.)
__global__ __launch_bounds__(1024) void label_compact_kernel(const long* __restrict__ labels, long n, int half, int S,
                                                             int offset, int* __restrict__ rows_out,
                                                             int* __restrict__ targets_out, int* __restrict__ count_out,
                                                             const int* __restrict__ row_of_pos) {
  constexpr int R = 16;                       // rounds of 1024 labels per chunk
  __shared__ unsigned long long ball[R * 16];  // ballot of (round j, wave w)
  __shared__ int pre[R * 16];                  // labelled positions before (j, w) inside the chunk
  __shared__ int wtot[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int base = 0;
  for (long c0 = 0; c0 < n; c0 += (long)R * 1024) {
    long lab[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long i = c0 + j * 1024 + t;
      lab[j] = i < n ? labels[i] : -100;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const unsigned long long bal = __ballot(lab[j] != -100);
      if (lane == 0) ball[j * 16 + w] = bal;
    }
    __syncthreads();
    // exclusive scan of the 256 popcounts by the first four waves (word index = j * 16 + w: the labels' own order)
    int cnt = 0, inc = 0;
    if (t < R * 16) {
      cnt = __popcll(ball[t]);
      inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63) wtot[w] = inc;
    }
    __syncthreads();
    if (t < R * 16) {
      int woff = 0;
      for (int k = 0; k < w; ++k) woff += wtot[k];
      pre[t] = woff + inc - cnt;
    }
    __syncthreads();
    const unsigned long long below = (1ULL << lane) - 1ULL;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (lab[j] == -100) continue;
      const long i = c0 + j * 1024 + t;
      const int dst = base + pre[j * 16 + w] + __popcll(ball[j * 16 + w] & below);
      const long b = i / half;
      const long pos = b * S + offset + (i - b * half);
      rows_out[dst] = row_of_pos ? row_of_pos[pos] : (int)pos;   // packed layout: a labelled position always has a row
      targets_out[dst] = (int)lab[j];
    }
    base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();   // (the next chunk overwrites ball / pre / wtot)
  }
  if (t == 0) *count_out = base;
}

// dst[i] = src[rows[i]] for i < count; rows in [count, roundup(count,128)) are zero-filled so that a
// 128-row GEMM tile past the count reads zeros.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ src, long ld_src,
                                                          const int* __restrict__ rows, const int* __restrict__ count,
                                                          bf16* __restrict__ dst, long ld_dst, int cols, long cap) {
  const int cnt = *count;
  long lim = ((long)(cnt + 127) / 128) * 128;
  lim = lim < cap ? lim : cap;
  const int nch = cols >> 3;
  const long total = lim * nch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / nch;
    const int c = (int)(i - r * nch);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
    if (r < cnt) v = *(const bf16x8*)(src + (long)rows[r] * ld_src + c * 8);
    *(bf16x8*)(dst + r * ld_dst + c * 8) = v;
  }
}

// dst[rows[i]] = src[i] for i < count (dst rows not named keep their content; the caller zeroes dst first)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16* __restrict__ src, long ld_src,
                                                           const int* __restrict__ rows, const int* __restrict__ count,
                                                           bf16* __restrict__ dst, long ld_dst, int cols) {
  const int cnt = *count;
  const int nch = cols >> 3;
  const long total = (long)cnt * nch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / nch;
    const int c = (int)(i - r * nch);
    *(bf16x8*)(dst + (long)rows[r] * ld_dst + c * 8) = *(const bf16x8*)(src + r * ld_src + c * 8);
  }
}

// fp32 source variant (split-K dgrad of the decoders accumulates in fp32): dst[rows[i]] = bf16(src[i])
__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(const float* __restrict__ src, long ld_src,
                                                               const int* __restrict__ rows,
                                                               const int* __restrict__ count, bf16* __restrict__ dst,
                                                               long ld_dst, int cols) {
  const int cnt = *count;
  const int nch = cols >> 3;
  const long total = (long)cnt * nch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / nch;
    const int c = (int)(i - r * nch);
    const f32x4 a = *(const f32x4*)(src + r * ld_src + c * 8), b = *(const f32x4*)(src + r * ld_src + c * 8 + 4);
    bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
    *(bf16x8*)(dst + (long)rows[r] * ld_dst + c * 8) = o;
  }
}

// One block per labelled row: online (max, sum-exp) sweep of the logits (fp32, or the fp16 the decoder GEMM can leave:
// LT), then a second sweep writes dlogits = (softmax - onehot) * gscale / count as bf16 (columns [ncols, npad) = 0).
// loss_sum += lse - x[target]. All arithmetic in fp32. HBM-bound: 10 bytes per logit with fp32 logits, 6 with fp16.
#ifndef STONK_XENT_BLOCKS
#define STONK_XENT_BLOCKS 2048   // workgroups, one labelled row at a time each; alone at 2 432 rows x 175 104 (tools/bench_xent.py, us): 256: 942, 512: 714, 1024: 547, 2048: 529
#endif
template <typename LT> struct LogitVec;
template <> struct LogitVec<float> {
  static __device__ __forceinline__ void load8(const float* x, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)x, b = *(const f32x4*)(x + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  }
};
template <> struct LogitVec<_Float16> {
  static __device__ __forceinline__ void load8(const _Float16* x, float (&v)[8]) {
    const f16x8 a = *(const f16x8*)x;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
  }
};

template <typename LT>
__global__ __launch_bounds__(256) void softmax_xent_kernel(const LT* __restrict__ logits, long ld, int ncols, int npad,
                                                           const int* __restrict__ targets, const int* __restrict__ count,
                                                           float* __restrict__ loss_sum, bf16* __restrict__ dlogits,
                                                           long ld_d, float gscale, int* __restrict__ err,
                                                           int cap_rows) {
  __shared__ float red_m[4], red_s[4];
  const int cnt = *count;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float g = gscale / (float)cnt;
  for (int row = blockIdx.x; row < cnt; row += gridDim.x) {
    const LT* x = logits + (long)row * ld;
    float m = -3.0e38f, s = 0.f;
    const int n8 = ncols >> 3;
    // four 16-byte (fp32: 32-byte) pieces per thread and iteration, all requested before the first is used: with one
    // load in flight per thread the sweep ran at the memory LATENCY (5 rows per CU x 4 KB: 2.5 TB/s on the entity
    // decoder's 350-KB rows); a piece past the row's end counts as -inf (exp = 0)
    for (int i0 = 0; i0 < n8; i0 += 1024) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u + t;
        if (i < n8) {
          LogitVec<LT>::load8(x + 8 * i, v[u]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = -__builtin_inff();
        }
      }
      float vm = v[0][0];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) vm = fmaxf(vm, v[u][j]);
      if (vm > m) {
        s *= __expf(m - vm);
        m = vm;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += __expf(v[u][j] - m);
    }
    for (int i = (n8 << 3) + t; i < ncols; i += 256) {
      const float v = (float)x[i];
      if (v > m) {
        s *= __expf(m - v);
        m = v;
      }
      s += __expf(v - m);
    }
    // combine (m, s) pairs across the wave, then across the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
      const float mn = fmaxf(m, m2);
      s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
      m = mn;
    }
    if (lane == 0) {
      red_m[w] = m;
      red_s[w] = s;
    }
    __syncthreads();
    float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float Ssum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) Ssum += red_s[j] * __expf(red_m[j] - M);
    const float lse = M + __logf(Ssum);
    int tgt = targets[row];
    if (tgt < 0 || tgt >= ncols) {  // torch raises "Target out of bounds"
      if (t == 0) atomicOr(err, 8);
      tgt = 0;
    }
    if (t == 0) atomicAdd(loss_sum, lse - (float)x[tgt]);
    if (dlogits) {
      bf16* d = dlogits + (long)row * ld_d;
      const int p8 = npad >> 3, f8 = ncols >> 3;   // whole pieces of the row: [0, f8); [f8, p8) holds the row's end and the padding
      const float nlse = -lse;
      for (int i0 = 0; i0 < f8; i0 += 1024) {   // (four pieces in flight per thread, as above)
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 256 * u + t;
          if (i < f8) LogitVec<LT>::load8(x + 8 * i, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 256 * u + t;
          if (i >= f8) continue;
          const int c0 = i * 8;
          float pr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pr[j] = __expf(v[u][j] + nlse);
          if (tgt >= c0 && tgt < c0 + 8) {   // (one piece of the row)
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (c0 + j == tgt) pr[j] -= 1.f;
          }
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16)(pr[j] * g);
          *(bf16x8*)(d + c0) = o;
        }
      }
      for (int i = f8 + t; i < p8; i += 256) {
        const int c0 = i * 8;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float pr = 0.f;
          if (c0 + j < ncols) {
            pr = __expf((float)x[c0 + j] - lse);
            if (c0 + j == tgt) pr -= 1.f;
          }
          o[j] = (bf16)(pr * g);
        }
        *(bf16x8*)(d + c0) = o;
      }
    }
    __syncthreads();
  }
  // rows [count, roundup64(count)) are read (as zeros) by the 64-token K step of the weight-gradient GEMM
  if (dlogits) {
    const int lim = (cnt + 63) / 64 * 64;
    for (int row = cnt + blockIdx.x; row < lim && row < cap_rows; row += gridDim.x) {
      bf16* d = dlogits + (long)row * ld_d;
      bf16x8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
      for (int i = t; i < (npad >> 3); i += 256) *(bf16x8*)(d + i * 8) = z;
    }
  }
}

// NSP head loss: B rows, C (= 2) classes, fp32. dlogits = (softmax - onehot) * gscale / B.
__global__ __launch_bounds__(256) void nsp_xent_kernel(const float* __restrict__ logits, const long* __restrict__ labels,
                                                       int B, int C, float* __restrict__ loss_sum,
                                                       float* __restrict__ dlogits, float gscale, int* __restrict__ err) {
  float local = 0.f;
  int cnt = 0;
  // count of non-ignored labels (CrossEntropyLoss ignore_index = -100 applies here too)
  for (int i = 0; i < B; ++i) cnt += labels[i] != -100;
  for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
    const float* x = logits + (long)b * C;
    long tgt = labels[b];
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += __expf(x[c] - m);
    const float lse = m + __logf(s);
    const bool ignored = tgt == -100;
    if (!ignored && (tgt < 0 || tgt >= C)) {
      atomicOr(err, 16);
      tgt = 0;
    }
    if (!ignored) local += lse - x[tgt];
    if (dlogits)
      for (int c = 0; c < C; ++c)
        dlogits[(long)b * C + c] = ignored ? 0.f : (__expf(x[c] - lse) - (c == tgt ? 1.f : 0.f)) * gscale / (float)cnt;
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0 && local != 0.f) atomicAdd(loss_sum, local);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(loss_sum + 1, (float)cnt);  // loss_sum[1] = label count
}

// loss_out[0] = total, [1] = text MLM, [2] = entity MLM, [3] = NSP (each the mean over its own labels)
__global__ void loss_finalize_kernel(const float* text_sum, const int* text_cnt, const float* ent_sum, const int* ent_cnt,
                                     const float* nsp_sum_cnt, float* loss_out) {
  const float lt = *text_sum / (float)*text_cnt;
  const float le = *ent_sum / (float)*ent_cnt;
  const float ln = nsp_sum_cnt[0] / nsp_sum_cnt[1];
  loss_out[0] = lt + le + ln;
  loss_out[1] = lt;
  loss_out[2] = le;
  loss_out[3] = ln;
}

}  // namespace

extern "C" int stonk_label_compact(const int64_t* labels, int64_t n, int half, int S, int offset, int* rows_out,
                                   int* targets_out, int* count_out, const int* row_of_pos, void* stream) {
  STONK_CHECK_ARG(labels && rows_out && targets_out && count_out, STONK_EINVAL);
  STONK_CHECK_ARG(n >= 0 && half > 0 && S >= half && offset >= 0, STONK_ESHAPE);
  hipLaunchKernelGGL(label_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long*)labels, (long)n,
                     half, S, offset, rows_out, targets_out, count_out, row_of_pos);
  return stonk_launch_status();
}

extern "C" int stonk_gather_rows_bf16(const void* src, int64_t ld_src, const int* rows, const int* count_dev, void* dst,
                                      int64_t ld_dst, int cols, int64_t cap, void* stream) {
  STONK_CHECK_ARG(src && rows && count_dev && dst, STONK_EINVAL);
  STONK_CHECK_ARG(cols > 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && cap >= 0, STONK_EALIGN);
  if (cap == 0) return STONK_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (long)ld_src,
                     rows, count_dev, (bf16*)dst, (long)ld_dst, cols, (long)cap);
  return stonk_launch_status();
}

extern "C" int stonk_scatter_rows_bf16(const void* src, int64_t ld_src, const int* rows, const int* count_dev,
                                       void* dst, int64_t ld_dst, int cols, void* stream) {
  STONK_CHECK_ARG(src && rows && count_dev && dst, STONK_EINVAL);
  STONK_CHECK_ARG(cols > 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, STONK_EALIGN);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (long)ld_src,
                     rows, count_dev, (bf16*)dst, (long)ld_dst, cols);
  return stonk_launch_status();
}

extern "C" int stonk_scatter_rows_f32_to_bf16(const float* src, int64_t ld_src, const int* rows, const int* count_dev,
                                             void* dst, int64_t ld_dst, int cols, void* stream) {
  STONK_CHECK_ARG(src && rows && count_dev && dst, STONK_EINVAL);
  STONK_CHECK_ARG(cols > 0 && cols % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0, STONK_EALIGN);
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, rows,
                     count_dev, (bf16*)dst, (long)ld_dst, cols);
  return stonk_launch_status();
}

extern "C" int stonk_softmax_xent_fwd_bwd(const float* logits, int64_t ld, int ncols, int npad, const int* targets,
                                          const int* count_dev, float* loss_sum, void* dlogits, int64_t ld_d,
                                          float grad_scale, int cap_rows, int* err_flag, void* stream) {
  STONK_CHECK_ARG(logits && targets && count_dev && loss_sum && err_flag && cap_rows >= 0, STONK_EINVAL);
  STONK_CHECK_ARG(ncols > 0 && npad >= ncols && npad % 8 == 0 && ld >= npad && ld % 4 == 0, STONK_ESHAPE);
  STONK_CHECK_ARG(!dlogits || (ld_d >= npad && ld_d % 8 == 0), STONK_ESHAPE);
  hipLaunchKernelGGL(softmax_xent_kernel<float>, dim3(STONK_XENT_BLOCKS), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, ncols,
                     npad, targets, count_dev, loss_sum, (bf16*)dlogits, (long)ld_d, grad_scale, err_flag, cap_rows);
  return stonk_launch_status();
}

extern "C" int stonk_softmax_xent_f16_fwd_bwd(const void* logits_f16, int64_t ld, int ncols, int npad, const int* targets,
                                              const int* count_dev, float* loss_sum, void* dlogits, int64_t ld_d,
                                              float grad_scale, int cap_rows, int* err_flag, void* stream) {
  STONK_CHECK_ARG(logits_f16 && targets && count_dev && loss_sum && err_flag && cap_rows >= 0, STONK_EINVAL);
  STONK_CHECK_ARG(ncols > 0 && npad >= ncols && npad % 8 == 0 && ld >= npad && ld % 8 == 0, STONK_ESHAPE);
  STONK_CHECK_ARG(!dlogits || (ld_d >= npad && ld_d % 8 == 0), STONK_ESHAPE);
  STONK_CHECK_ARG((uintptr_t)logits_f16 % 16 == 0, STONK_EALIGN);
  hipLaunchKernelGGL(softmax_xent_kernel<_Float16>, dim3(STONK_XENT_BLOCKS), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)logits_f16, (long)ld, ncols, npad, targets, count_dev, loss_sum, (bf16*)dlogits,
                     (long)ld_d, grad_scale, err_flag, cap_rows);
  return stonk_launch_status();
}

extern "C" int stonk_nsp_xent_fwd_bwd(const float* logits, const int64_t* labels, int B, int C, float* loss_sum_cnt,
                                      float* dlogits, float grad_scale, int* err_flag, void* stream) {
  STONK_CHECK_ARG(logits && labels && loss_sum_cnt && err_flag, STONK_EINVAL);
  STONK_CHECK_ARG(B > 0 && C > 0 && C <= 64, STONK_ESHAPE);
  hipLaunchKernelGGL(nsp_xent_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits,
                     (const long*)labels, B, C, loss_sum_cnt, dlogits, grad_scale, err_flag);
  return stonk_launch_status();
}

extern "C" int stonk_loss_finalize(const float* text_sum, const int* text_cnt, const float* ent_sum, const int* ent_cnt,
                                   const float* nsp_sum_cnt, float* loss_out, void* stream) {
  STONK_CHECK_ARG(text_sum && text_cnt && ent_sum && ent_cnt && nsp_sum_cnt && loss_out, STONK_EINVAL);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, text_sum, text_cnt, ent_sum, ent_cnt,
                     nsp_sum_cnt, loss_out);
  return stonk_launch_status();
}
