// Weight-gradient GEMM straight from row-major activations ("TN" form), bf16 MFMA, fp32 atomic accumulate:
//
//     dW[M', N'] += alpha * sum_t dY[t][m'] * X[t][n']          db[m'] += alpha * sum_t dY[t][m']
//
// dY = [T, M'] and X = [T, N'] are the tensors backward already holds ([token][feature], feature-contiguous); the
// contraction runs over the token index, i.e. over the ROW index of both operands. Instead of materialising dY^T and
// X^T for the NT kernel (two extra HBM passes per weight), the [64 tokens][128 features] tiles are DMA'd as they lie
// and the MFMA fragments (8 consecutive tokens of one feature per lane) are gathered by the hardware transpose read
// ds_read_b64_tr_b16 (two per fragment). Autograd's mm for nn.Linear weights/biases
// (hf:models/bert/modeling_bert.py:154-156,289,335,348,477; ref:stonkgs_model.py:70-71).
//
// Bias gradient for free: workgroups of the first column tile issue one extra MFMA per (feature tile, k-step) against
// an all-ones B fragment - column sums of dY on the matrix pipe, no second pass over dY.
//
// Tile 128x128 over 64-token K steps, 4 waves (2x2), 2 workgroups per CU, LDS-DMA double buffered as gemm_bf16.hip.
// LDS image: 256-byte rows (128 features), 16-byte chunk c of row r stored at c ^ (((r&3)<<2) | ((r>>2)&3))
// (conflict-free for the transposed reads); the swizzle is applied on the DMA source address.
#include <cstdlib>

#include "gemm_common.h"

// persistent 256x256 variant (gemm256.hip, TN = true)
int stonk_gemm256_tn_launch(const stonk_gemm::GemmArgs& a, hipStream_t st);
int stonk_gemm_tn_w4_launch(const stonk_gemm::GemmArgs& a, hipStream_t st);
int stonk_gemm_tn_a4_launch(const stonk_gemm::GemmArgs& a, hipStream_t st);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BK * 256;          // 16 KiB: 64 token rows x 128 features
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int TN_LDS = 2 * STAGE_BYTES;       // 64 KiB

struct TnArgs {
  const bf16* A;   // dY [T, M']
  const bf16* B;   // X  [T, N']
  float* C;        // dW [M', N'] fp32, accumulated
  float* bias;     // db [M'] fp32, accumulated (nullable)
  const int* k_dev;
  long lda, ldb, ldc;
  int M, N, K;     // M', N', token count (capacity)
  float alpha;
  int split_k;
};

__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (b >> 3);
}

__device__ __forceinline__ int row_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int K = p.K;
  if (p.k_dev) {
    const int kd = *p.k_dev;
    K = kd < K ? kd : K;
  }
  const int nk_total = (K + BK - 1) / BK;   // rows in [K, roundup) must read as zero (caller guarantees)
  const int ntm = p.M / BM, ntn = p.N / BN;
  const int nk_per = (nk_total + p.split_k - 1) / p.split_k;
  const int per_split = ntm * ntn;
  const int total = per_split * p.split_k;
  const int t = ((int)gridDim.x == total) ? xcd_remap(blockIdx.x, total) : blockIdx.x;
  const int ks = t / per_split;
  const int tt = t - ks * per_split;
  int rt, ct;
  if (ntm >= ntn) {
    rt = tt / ntn;
    ct = tt - rt * ntn;
  } else {
    ct = tt / ntm;
    rt = tt - ct * ntm;
  }
  const int m0 = rt * BM, n0 = ct * BN;
  int nk = nk_total - ks * nk_per;
  nk = nk < nk_per ? nk : nk_per;
  if (nk <= 0) return;
  const long tok_begin = (long)ks * nk_per * BK;
  const bool want_bias = p.bias != nullptr && ct == 0;

  // ---- DMA sources: a wave moves 16 token rows of each operand per K step (4 + 4 pieces of 4 rows x 256 B)
  const bf16* ga[4];
  const bf16* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 4 + (lane >> 4);     // token row inside the K step
    const int c = (lane & 15) ^ row_swz(r);             // logical chunk for physical slot (lane & 15)
    ga[i] = p.A + (tok_begin + r) * p.lda + m0 + c * 8;
    gb[i] = p.B + (tok_begin + r) * p.ldb + n0 + c * 8;
  }
  auto stage = [&](int kt, int buf) {
    char* sa = smem + buf * STAGE_BYTES + wave * 4096;
    char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + (long)kt * BK * p.lda),
                                       (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + (long)kt * BK * p.ldb),
                                       (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
    }
  };

  // ---- transposed fragment offsets: lane (group g = lane>>4, q = (lane&15)>>2, pq = lane&3) addresses token row
  // 8g + 4*hi + q, features f0 + 4*pq .. +3 of the 16-feature tile; it receives feature f0 + (lane&15), 4 tokens.
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  int offa[2][4], offb[2][4];
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int row = 8 * g + 4 * hi + q;                 // + 32 per k-step: swizzle unchanged (32 = 0 mod 16 rows)
    const int sw = row_swz(row);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ca = wm * 64 + i * 16 + 4 * pq, cb = wn * 64 + i * 16 + 4 * pq;
      offa[hi][i] = row * 256 + (((ca >> 3) ^ sw) << 4) + ((ca & 7) << 1);
      offb[hi][i] = row * 256 + (((cb >> 3) ^ sw) << 4) + ((cb & 7) << 1);
    }
  }

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sa + s * 8192 + offa[0][i]));
        const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sa + s * 8192 + offa[1][i]));
        a[i] = (bf16x8){alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + s * 8192 + offb[0][i]));
        const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sb + s * 8192 + offb[1][i]));
        b[i] = (bf16x8){blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (want_bias && wn == 0) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    }
  };

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    compute(kt & 1);
  }

  // ---- epilogue: acc[i][j][r] = dW[m0 + wm*64 + i*16 + (lane>>4)*4 + r][n0 + wn*64 + j*16 + (lane&15)]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        atomicAdd(p.C + (long)m * p.ldc + n, acc[i][j][r] * p.alpha);
      }
      if (want_bias && wn == 0 && (lane & 15) == 0) atomicAdd(p.bias + m, accb[i][r] * p.alpha);
    }
}

}  // namespace

extern "C" int stonk_gemm_tn_bf16(const void* dY, int64_t lda, const void* X, int64_t ldb, float* dW, int64_t ldc,
                                  float* dbias, int M, int N, int K, float alpha, int split_k, const int* k_dev,
                                  void* stream) {
  STONK_CHECK_ARG(dY && X && dW, STONK_EINVAL);
  STONK_CHECK_ARG(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0, STONK_ESHAPE);
  STONK_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, STONK_EALIGN);
  STONK_CHECK_ARG((uintptr_t)dY % 16 == 0 && (uintptr_t)X % 16 == 0, STONK_EALIGN);
  // split_k == 0 selects the four-wave 256x256 kernel (gemm_tn_w4.hip) with an automatic split: work items = tiles x
  // splits aimed at one full round of the CUs. split_k <= -16: the same kernel limited to -split_k CUs' worth of
  // workgroups - for launches that run on a second stream beside other work: a persistent one-workgroup-per-CU kernel that
  // takes every CU stalls the other stream until its workgroups retire (the step is 1.5 ms faster with the weight
  // gradients on 160 of the 256 CUs). split_k == -1 keeps the older eight-wave 256x256 TN form of gemm256.hip.
  if (split_k <= 0) {
    STONK_CHECK_ARG(M >= 256 && N >= 256, STONK_ESHAPE);
    // (operands are addressed per 64-token K tile: 64 rows of either operand must stay inside 32-bit byte offsets)
    STONK_CHECK_ARG(lda % 64 == 0 && ldb % 64 == 0 && lda < (1L << 23) && ldb < (1L << 23), STONK_ESHAPE);
    // (split_k == -2, or <= -2016 = held to -split_k - 2000 CUs: the compiled four-wave kernel, gemm_tn_w4.hip, which the
    // written-out one replaced in round 4 - kept reachable for A/B runs)
    const bool old_w4 = split_k == -2 || split_k <= -2016;
    if (old_w4) split_k = split_k == -2 ? 0 : split_k + 2000;
    STONK_CHECK_ARG(split_k >= -1024, STONK_ESHAPE);
    stonk_gemm::GemmArgs g = {};
    g.A = (const bf16*)dY; g.B = (const bf16*)X; g.C = dW; g.bias = dbias; g.k_dev = k_dev;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha;
    const int nk = (K + BK - 1) / BK;
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long cus = split_k <= -16 ? -split_k : 256;
    long sk = tiles >= cus ? 1 : cus / tiles;   // floor: tiles x splits must not spill into a second, mostly empty round
    if (sk > nk / 8) sk = nk / 8 > 0 ? nk / 8 : 1;
    g.split_k = (int)sk;
    g.flags = (int)cus;   // grid cap
    if (split_k != -1 && !old_w4) return stonk_gemm_tn_a4_launch(g, (hipStream_t)stream);
    if (split_k != -1) return stonk_gemm_tn_w4_launch(g, (hipStream_t)stream);
    return stonk_gemm256_tn_launch(g, (hipStream_t)stream);
  }
  TnArgs a;
  a.A = (const bf16*)dY; a.B = (const bf16*)X; a.C = dW; a.bias = dbias; a.k_dev = k_dev;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.alpha = alpha;
  const int nk = (K + BK - 1) / BK;
  a.split_k = split_k < nk ? split_k : nk;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);
    attr_done = true;
  }
  const long tiles = (long)(M / BM) * (N / BN) * a.split_k;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)tiles), dim3(256), TN_LDS, (hipStream_t)stream, a);
  return stonk_launch_status();
}
