// Tiny fp32 heads of the STonKGs step (K9/K14 in SURVEY.md section 2.3): BertPooler tanh(W h[:,0] + b)
// (hf:models/bert/modeling_bert.py:457-463) and the NSP classifier Linear(H, 2) (:523,527). M = batch rows
// only (64 at the benchmark config), so these are latency-trivial wave-per-output dot products on the fp32
// master weights - no bf16 rounding on the pooled path.
#include "common.h"
#include "stonk_flags.h"

namespace {

__device__ __forceinline__ float load_x(const void* x, long idx, bool f32) {
  return f32 ? ((const float*)x)[idx] : (float)((const bf16*)x)[idx];
}

// y[m][n] = act(sum_k x[m*ldx + k] * W[n][k] + bias[n]);  one wave per (m, n)
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const void* __restrict__ x, long ldx,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int M, int N, int K, int act) {
  const int lane = threadIdx.x & 63;
  const long wave_id = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * 256) >> 6;
  const bool xf32 = act & STONK_SMALL_X_F32;
  for (long o = wave_id; o < (long)M * N; o += nw) {
    const int m = (int)(o / N), n = (int)(o - (long)m * N);
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += load_x(x, (long)m * ldx + k, xf32) * W[(long)n * K + k];
    acc = wave_sum(acc);
    if (lane == 0) {
      float v = acc + (bias ? bias[n] : 0.f);
      if (act & STONK_SMALL_TANH) v = tanhf(v);
      y[(long)m * N + n] = v;
    }
  }
}

// dpre = dy * (1 - y^2) for tanh, else dy.
// dW[n][k] += sum_m dpre[m][n] x[m][k];  db[n] += sum_m dpre[m][n]     (one thread per (n, k))
__global__ __launch_bounds__(256) void small_linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const void* __restrict__ x, long ldx,
                                                                 float* __restrict__ dW, float* __restrict__ db, int M,
                                                                 int N, int K, int act) {
  const bool xf32 = act & STONK_SMALL_X_F32, th = act & STONK_SMALL_TANH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)N * K; i += (long)gridDim.x * 256) {
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    float acc = 0.f, accb = 0.f;
    for (int m = 0; m < M; ++m) {
      float d = dy[(long)m * N + n];
      if (th) {
        const float yy = y[(long)m * N + n];
        d *= 1.f - yy * yy;
      }
      acc += d * load_x(x, (long)m * ldx + k, xf32);
      accb += d;
    }
    dW[i] += acc;
    if (k == 0 && db) db[n] += accb;
  }
}

// The same for M * 8 * 4 bytes of LDS or less (a batch of rows: the pooler's 64): a block owns 8 output features x 256
// input columns, stages dpre[M][8] in LDS once and reads every x[m][k] once for the 8 features - the one-thread-per-(n, k)
// form above re-reads dy / y per column and x per feature (83 us for the pooler's 768 x 768 from 64 rows; this one ~15).
constexpr int WG_NT = 8;
__global__ __launch_bounds__(256) void small_linear_wgrad_tiled_kernel(const float* __restrict__ dy,
                                                                       const float* __restrict__ y,
                                                                       const void* __restrict__ x, long ldx,
                                                                       float* __restrict__ dW, float* __restrict__ db,
                                                                       int M, int N, int K, int act) {
  extern __shared__ __attribute__((aligned(16))) float dpre_t[];   // [M][WG_NT]
  const bool xf32 = act & STONK_SMALL_X_F32, th = act & STONK_SMALL_TANH;
  const int n0 = blockIdx.y * WG_NT, k = blockIdx.x * 256 + threadIdx.x;
  for (int idx = threadIdx.x; idx < M * WG_NT; idx += 256) {
    const int m = idx / WG_NT, n = n0 + idx % WG_NT;
    float d = 0.f;
    if (n < N) {
      d = dy[(long)m * N + n];
      if (th) {
        const float yy = y[(long)m * N + n];
        d *= 1.f - yy * yy;
      }
    }
    dpre_t[idx] = d;
  }
  __syncthreads();
  if (k < K) {
    float acc[WG_NT];
#pragma unroll
    for (int j = 0; j < WG_NT; ++j) acc[j] = 0.f;
    for (int m = 0; m < M; ++m) {
      const float xv = load_x(x, (long)m * ldx + k, xf32);
#pragma unroll
      for (int j = 0; j < WG_NT; ++j) acc[j] += dpre_t[m * WG_NT + j] * xv;
    }
#pragma unroll
    for (int j = 0; j < WG_NT; ++j)
      if (n0 + j < N) dW[(long)(n0 + j) * K + k] += acc[j];
  }
  if (blockIdx.x == 0 && threadIdx.x < WG_NT && n0 + threadIdx.x < N && db) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dpre_t[m * WG_NT + threadIdx.x];
    db[n0 + threadIdx.x] += s;
  }
}

// dx[m][k] = sum_n dpre[m][n] W[n][k]; written as fp32 (dx_f32[m*K + k]) and/or ADDED into a bf16 row
// (dx_bf16[m*ld_dxb + k] += ...: the pooler's gradient lands on position 0 of d(sequence_output)).
// grid = (ceil(K/256), M): dpre[m][:] is staged in LDS once per block, W is read coalesced along k.
__global__ __launch_bounds__(256) void small_linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ W, float* __restrict__ dx_f32,
                                                                 bf16* __restrict__ dx_bf16, long ld_dxb, int M, int N,
                                                                 int K, int act) {
  extern __shared__ __attribute__((aligned(16))) float dpre[];
  const bool th = act & STONK_SMALL_TANH;
  const int m = blockIdx.y;
  for (int n = threadIdx.x; n < N; n += 256) {
    float d = dy[(long)m * N + n];
    if (th) {
      const float yy = y[(long)m * N + n];
      d *= 1.f - yy * yy;
    }
    dpre[n] = d;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int n = 0;
  for (; n + 4 <= N; n += 4) {
    a0 += dpre[n] * W[(long)n * K + k];
    a1 += dpre[n + 1] * W[(long)(n + 1) * K + k];
    a2 += dpre[n + 2] * W[(long)(n + 2) * K + k];
    a3 += dpre[n + 3] * W[(long)(n + 3) * K + k];
  }
  for (; n < N; ++n) a0 += dpre[n] * W[(long)n * K + k];
  const float acc = (a0 + a1) + (a2 + a3);
  const long i = (long)m * K + k;
  if (dx_f32) dx_f32[i] = acc;
  if (dx_bf16) {
    bf16* p = dx_bf16 + (long)m * ld_dxb + k;
    *p = (bf16)((float)*p + acc);
  }
}

}  // namespace

extern "C" int stonk_small_linear_fwd(const void* x, int64_t ldx, const float* W, const float* bias, float* y, int M,
                                      int N, int K, int act, void* stream) {
  STONK_CHECK_ARG(x && W && y, STONK_EINVAL);
  STONK_CHECK_ARG(M >= 0 && N > 0 && K > 0 && ldx >= K, STONK_ESHAPE);
  if (M == 0) return STONK_OK;
  long waves = (long)M * N;
  int grid = (int)((waves + 3) / 4 < 2048 ? (waves + 3) / 4 : 2048);
  hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, W, bias, y, M,
                     N, K, act);
  return stonk_launch_status();
}

extern "C" int stonk_small_linear_bwd(const float* dy, const float* y, const void* x, int64_t ldx, const float* W,
                                      float* dW, float* db, float* dx_f32, void* dx_bf16_accum, int64_t ld_dxb, int M,
                                      int N, int K, int act, void* stream) {
  STONK_CHECK_ARG(dy && x && W && dW, STONK_EINVAL);
  STONK_CHECK_ARG(!(act & STONK_SMALL_TANH) || y, STONK_EINVAL);
  STONK_CHECK_ARG(M >= 0 && N > 0 && K > 0 && ldx >= K, STONK_ESHAPE);
  if (M == 0) return STONK_OK;
  hipStream_t st = (hipStream_t)stream;
  long nk = (long)N * K;
  if ((size_t)M * WG_NT * sizeof(float) <= 48 * 1024)
    hipLaunchKernelGGL(small_linear_wgrad_tiled_kernel, dim3((K + 255) / 256, (N + WG_NT - 1) / WG_NT), dim3(256),
                       (size_t)M * WG_NT * sizeof(float), st, dy, y, x, (long)ldx, dW, db, M, N, K, act);
  else
    hipLaunchKernelGGL(small_linear_wgrad_kernel, dim3((unsigned)((nk + 255) / 256 < 4096 ? (nk + 255) / 256 : 4096)),
                       dim3(256), 0, st, dy, y, x, (long)ldx, dW, db, M, N, K, act);
  if (dx_f32 || dx_bf16_accum) {
    hipLaunchKernelGGL(small_linear_dgrad_kernel, dim3((K + 255) / 256, M), dim3(256), (size_t)N * sizeof(float), st, dy,
                       y, W, dx_f32, (bf16*)dx_bf16_accum, (long)ld_dxb, M, N, K, act);
  }
  return stonk_launch_status();
}

// du = dg * gelu'(u)  (bf16, elementwise): backward of the GELU inside BertPredictionHeadTransform
// (hf:models/bert/modeling_bert.py:476-480), where no GEMM epilogue is available to carry it.
namespace {
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ dg, const bf16* __restrict__ u,
                                                       bf16* __restrict__ du, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 a = *(const bf16x8*)(dg + 8 * i), b = *(const bf16x8*)(u + 8 * i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] * gelu_erf_grad((float)b[j]));
    *(bf16x8*)(du + 8 * i) = o;
  }
}
}  // namespace

extern "C" int stonk_gelu_bwd_bf16(const void* dg, const void* u, void* du, int64_t n, void* stream) {
  STONK_CHECK_ARG(dg && u && du && n >= 0 && n % 8 == 0, STONK_EINVAL);
  if (n == 0) return STONK_OK;
  const long n8 = n / 8;
  const long g = (n8 + 255) / 256;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)(g < 4096 ? g : 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dg, (const bf16*)u, (bf16*)du, n8);
  return stonk_launch_status();
}

// Inverted dropout on a small fp32 tensor (the pooled vector in front of the classification head,
// ref:src/stonkgs/models/stonkgs_finetuning.py:313); the same call with the same (seed) replays the mask on the gradient.
namespace {
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                          uint32_t thr32, float scale, uint32_t seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = stonk_keep(0u, (uint32_t)i, seed, thr32) ? x[i] * scale : 0.f;
}
__global__ void ratio_kernel(const float* num, const float* den, float* out) { *out = *num / *den; }
// Classification-head losses other than cross-entropy (ref:src/stonkgs/models/stonkgs_finetuning.py:328-338), mean over
// all B x C elements, one workgroup (B is a batch, C a handful of labels):
//   STONK_LOSS_MSE            torch.nn.MSELoss()(logits [B,C], targets [B,C])
//   STONK_LOSS_MSE_BROADCAST  the same call with C = 1 and targets of shape [B]: torch broadcasts [B,1] against [B] to
//                             [B,B] (with a warning) - what the reference computes for num_labels = 1 and 1-D labels:
//                             mean_ij (x_i - y_j)^2, d/dx_i = 2 (x_i - mean(y)) / B
//   STONK_LOSS_BCE            torch.nn.BCEWithLogitsLoss()(logits, targets): max(x,0) - x y + log(1 + exp(-|x|))
__global__ __launch_bounds__(256) void elementwise_loss_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               int B, int C, int mode, float* __restrict__ loss_out,
                                                               float* __restrict__ dx, float gscale) {
  __shared__ float red[4];
  __shared__ float s_mean_y, s_mean_y2;
  const int n = B * C;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  if (mode == STONK_LOSS_MSE_BROADCAST) {
    float sy = 0.f, sy2 = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
      sy += y[j];
      sy2 += y[j] * y[j];
    }
    sy = block_sum(sy);
    sy2 = block_sum(sy2);
    if (threadIdx.x == 0) {
      s_mean_y = sy / (float)B;
      s_mean_y2 = sy2 / (float)B;
    }
    __syncthreads();
  }
  float local = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float xi = x[i];
    float l, g;
    if (mode == STONK_LOSS_MSE) {
      const float d = xi - y[i];
      l = d * d;
      g = 2.f * d;
    } else if (mode == STONK_LOSS_MSE_BROADCAST) {
      l = xi * xi - 2.f * xi * s_mean_y + s_mean_y2;   // mean_j (x_i - y_j)^2
      g = 2.f * (xi - s_mean_y);
    } else {
      const float yi = y[i];
      const float e = __expf(-fabsf(xi));
      l = fmaxf(xi, 0.f) - xi * yi + log1pf(e);
      const float sig = xi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      g = sig - yi;
    }
    local += l;
    if (dx) dx[i] = g * gscale / (float)n;
  }
  local = block_sum(local);
  if (threadIdx.x == 0) *loss_out = local / (float)n;
}

}  // namespace

extern "C" int stonk_elementwise_loss_fwd_bwd(const float* logits, const float* targets, int B, int C, int mode,
                                              float* loss_out, float* dlogits, float grad_scale, void* stream) {
  STONK_CHECK_ARG(logits && targets && loss_out, STONK_EINVAL);
  STONK_CHECK_ARG(B > 0 && C > 0 && (long)B * C < (1L << 24), STONK_ESHAPE);
  STONK_CHECK_ARG(mode == STONK_LOSS_MSE || mode == STONK_LOSS_BCE || (mode == STONK_LOSS_MSE_BROADCAST && C == 1),
                  STONK_EINVAL);
  hipLaunchKernelGGL(elementwise_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, targets, B, C, mode,
                     loss_out, dlogits, grad_scale);
  return stonk_launch_status();
}

extern "C" int stonk_dropout_f32(const float* x, float* y, int64_t n, float p, uint32_t seed, void* stream) {
  STONK_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, STONK_EINVAL);
  if (n == 0) return STONK_OK;
  const long g = (n + 255) / 256;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(256), 0, (hipStream_t)stream, x, y,
                     (long)n, stonk_drop_thr32(p), 1.f / (1.f - p), stonk_seed_mix(seed));
  return stonk_launch_status();
}

// *out = *num / *den on the device (mean of a cross-entropy sum without a host round trip)
extern "C" int stonk_ratio_f32(const float* num, const float* den, float* out, void* stream) {
  STONK_CHECK_ARG(num && den && out, STONK_EINVAL);
  hipLaunchKernelGGL(ratio_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, num, den, out);
  return stonk_launch_status();
}
