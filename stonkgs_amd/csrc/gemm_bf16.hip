// bf16 MFMA GEMM for gfx950, "NT" form:  C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T)
//
// Every contraction on the STonKGs hot path is expressed in this one form (both operands
// contraction-contiguous), which is the natural layout for nn.Linear (weight = [out, in]):
//   forward   y  = x . W^T             A = x [T,in]        B = W [out,in]
//   dgrad     dx = dy . W              A = dy [T,out]      B = W^T [in,out] (bf16 copy kept by the optimizer)
//   wgrad     dW += dy^T . x           A = dy^T [out,T]    B = x^T [in,T]   (fp32 atomic accumulate, split-K)
// Replaces torch addmm/mm issued by hf:models/bert/modeling_bert.py (BertSelfAttention :154-156,
// BertSelfOutput :289-293, BertIntermediate :334-337, BertOutput :347-351, transform :476-480) and the
// decoders of ref:src/stonkgs/models/stonkgs_model.py:47-49,70-71.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16.
// Operand tiles are staged global->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered;
// the LDS image is lane-linear, the XOR swizzle (chunk ^ (row & 7)) is applied on the per-lane SOURCE
// address and again on the ds_read_b128 fragment reads (conflict-free for 128-byte rows).
#include <cstdlib>

#include "gemm_common.h"

using namespace stonk_gemm;

// defined in gemm256.hip
int stonk_gemm256_launch(const GemmArgs& a, int out_mode, hipStream_t st);
int stonk_gemm_w4_launch(const GemmArgs& a, int out_mode, int tile_n, int items_per_wg, hipStream_t st);
int stonk_gemm_a4_launch(const GemmArgs& a, int tile_n, int items_per_wg, hipStream_t st);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // double buffered: 64 KiB

// EPI >= 0: epilogue flags known at compile time (the combinations the STonKGs step launches): no dead branches, and
// the side operands of a whole 64x64 wave tile are fetched in one batch BEFORE any of them is used, so their latency is
// paid once per tile instead of once per fragment. EPI < 0: flags read at run time.
template <int OUT_MODE, bool GLDS, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  if (p.m_dev) {
    const int md = *p.m_dev;
    M = md < M ? md : M;
  }
  const int ntm = (M + BM - 1) / BM, ntn = p.N / BN;
  // contraction length may also live in device memory (label-sparse wgrad): tiles past it are skipped
  int nk_total = p.K / BK;
  if (p.k_dev) {
    const int kd = (*p.k_dev + BK - 1) / BK;
    nk_total = kd < nk_total ? kd : nk_total;
  }
  const int nk_per = (nk_total + p.split_k - 1) / p.split_k;
  const int per_split = ntm * ntn;
  const int total = per_split * p.split_k;
  // one tile per workgroup (a grid sized for the capacity M may exceed the live tile count of a device-side M: the
  // surplus workgroups leave at once, the live ones are workgroups 0 .. total-1 and are remapped among themselves)
  const bool one_shot = ((int)gridDim.x >= total);

  // per-lane fragment read offsets (bytes inside a tile): row = base + (lane & 15), swizzle = lane & 7
  const int frag_row = lane & 15;
  const int frag_kc = lane >> 4;
  const int swz = lane & 7;

  for (int t0 = blockIdx.x; t0 < total; t0 += gridDim.x) {
    const int t = one_shot ? xcd_remap(t0, total) : t0;
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
    const long k_begin = (long)ks * nk_per * BK;
    int nk = nk_total - ks * nk_per;
    nk = nk < nk_per ? nk : nk_per;
    if (nk <= 0) continue;  // uniform per block

    // ---- staging addresses: each wave moves 32 rows of A and 32 rows of B per K tile (4 + 4 DMA ops)
    const bf16* ga[4];
    const bf16* gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);  // logical 16-byte chunk fetched into physical slot (lane & 7)
      int ra = m0 + r;
      ra = ra < M ? ra : M - 1;  // clamp: rows past M are computed on a duplicate row and never stored
      ga[i] = p.A + (long)ra * p.lda + k_begin + c * 8;
      gb[i] = p.B + (long)(n0 + r) * p.ldb + k_begin + c * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage_glds = [&](int kt, int buf) {
      char* sa = smem + buf * STAGE_BYTES + wave * 32 * 128;
      char* sb = sa + TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + (long)kt * BK),
                                         (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + (long)kt * BK),
                                         (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
      }
    };
    bf16x8 ra_[4], rb_[4];
    auto stage_load_regs = [&](int kt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra_[i] = *(const bf16x8*)(ga[i] + (long)kt * BK);
        rb_[i] = *(const bf16x8*)(gb[i] + (long)kt * BK);
      }
    };
    auto stage_store_regs = [&](int buf) {
      char* sa = smem + buf * STAGE_BYTES + wave * 32 * 128 + lane * 16;
      char* sb = sa + TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(bf16x8*)(sa + i * 1024) = ra_[i];
        *(bf16x8*)(sb + i * 1024) = rb_[i];
      }
    };

    auto compute = [&](int buf) {
      const char* sa = smem + buf * STAGE_BYTES + (wm * 64 + frag_row) * 128;
      const char* sb = smem + buf * STAGE_BYTES + TILE_BYTES + (wn * 64 + frag_row) * 128;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int coff = ((s * 4 + frag_kc) ^ swz) * 16;
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(sa + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(sb + j * 16 * 128 + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (OUT_MODE == 2)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            else  // swapped: D[n][m], so a lane ends up with 4 consecutive output columns of one row
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          }
      }
    };

    __syncthreads();  // previous tile's epilogue / fragment reads are done before restaging
    if (GLDS) {
      stage_glds(0, 0);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_glds(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
      }
    } else {
      stage_load_regs(0);
      stage_store_regs(0);
      for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        if (kt + 1 < nk) stage_load_regs(kt + 1);
        compute(kt & 1);
        if (kt + 1 < nk) stage_store_regs((kt + 1) & 1);
      }
    }

    // ---------------- epilogue (straight from the accumulators) ----------------
    const int flags = EPI >= 0 ? EPI : p.flags;
    if (OUT_MODE == 2) {
      // standard orientation: acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane>>4)*4 + r][n0 + wn*64 + j*16 + (lane&15)]
      float* C = (float*)p.C;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
          if (m < M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int n = n0 + wn * 64 + j * 16 + (lane & 15);
              atomicAdd(C + (long)m * p.ldc + n, acc[i][j][r] * p.alpha);
            }
          }
        }
    } else if (EPI >= 0) {
      // swapped: acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane&15)][n0 + wn*64 + j*16 + (lane>>4)*4 + r]
      const int nq = n0 + wn * 64 + (lane >> 4) * 4;
      const int mq = m0 + wm * 64 + (lane & 15);
      f32x4 bq[4];
      bf16x4 sa[4][4], sr[4][4];
      if (EPI & STONK_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = *(const f32x4*)(p.bias + nq + j * 16);
      }
      if (EPI & (STONK_EPI_GELU_BWD | STONK_EPI_RESID)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int m = mq + i * 16;
          m = m < M ? m : M - 1;   // clamped rows are loaded but never stored
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (EPI & STONK_EPI_GELU_BWD) sa[i][j] = *(const bf16x4*)(p.aux + (long)m * p.ldaux + nq + j * 16);
            if (EPI & STONK_EPI_RESID) sr[i][j] = *(const bf16x4*)(p.resid + (long)m * p.ldr + nq + j * 16);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mq + i * 16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nq + j * 16;
          f32x4 v = acc[i][j] * p.alpha;
          if (EPI & STONK_EPI_BIAS) v += bq[j];
          if (EPI & STONK_EPI_SAVE_PREACT) {
            constexpr bool AG = (EPI & STONK_EPI_AUX_GRAD) != 0;
            bf16x4 u = {(bf16)gelu_saved(v[0], AG), (bf16)gelu_saved(v[1], AG), (bf16)gelu_saved(v[2], AG),
                        (bf16)gelu_saved(v[3], AG)};
            *(bf16x4*)(p.aux + (long)m * p.ldaux + n) = u;
          }
          if (EPI & STONK_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
          }
          if (EPI & STONK_EPI_GELU_BWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_factor((float)sa[i][j][r], (EPI & STONK_EPI_AUX_GRAD) != 0);
          }
          if (EPI & STONK_EPI_DROPOUT) {
            const uint32_t rk = stonk_rowkey((uint32_t)m, p.seed), ck = stonk_colkey((uint32_t)n);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              v[r] = stonk_keep_key(rk, ck + (uint32_t)r * STONK_G_COL, p.drop_thr32) ? v[r] * p.drop_scale : 0.f;
          }
          if (EPI & STONK_EPI_RESID) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)sr[i][j][r];
          }
          if (OUT_MODE == 0) {
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = o;
          } else if (OUT_MODE == 3) {
            f16x4 o = {to_f16_sat(v[0]), to_f16_sat(v[1]), to_f16_sat(v[2]), to_f16_sat(v[3])};
            *(f16x4*)((_Float16*)p.C + (long)m * p.ldc + n) = o;
          } else {
            *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
          const f32x4 v = epilogue4(acc[i][j] * p.alpha, p, flags, m, n);
          if (OUT_MODE == 0) {
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = o;
          } else if (OUT_MODE == 3) {
            f16x4 o = {to_f16_sat(v[0]), to_f16_sat(v[1]), to_f16_sat(v[2]), to_f16_sat(v[3])};
            *(f16x4*)((_Float16*)p.C + (long)m * p.ldc + n) = o;
          } else {
            *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
          }
        }
      }
    }
  }
}

template <int OUT_MODE, bool GLDS, int EPI>
int launch(const GemmArgs& a, int grid, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<OUT_MODE, GLDS, EPI>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<OUT_MODE, GLDS, EPI>), dim3(grid), dim3(256), GEMM_LDS, st, a);
  return stonk_launch_status();
}

}  // namespace

extern "C" int stonk_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                  int M, int N, int K, int flags, const float* bias, const void* resid,
                                  int64_t ldr, void* aux, int64_t ldaux, float alpha, int split_k,
                                  const int* m_dev, const int* k_dev, float drop_p, uint32_t seed, int kernel,
                                  void* stream) {
  STONK_CHECK_ARG(A && B && C, STONK_EINVAL);
  STONK_CHECK_ARG(kernel >= STONK_GEMM_AUTO && kernel <= STONK_GEMM_ASM4_192, STONK_EINVAL);
  STONK_CHECK_ARG(M >= 0 && N > 0 && K > 0, STONK_ESHAPE);
  STONK_CHECK_ARG(N % BN == 0 && K % BK == 0, STONK_ESHAPE);
  STONK_CHECK_ARG(split_k >= 1 && split_k <= K / BK, STONK_ESHAPE);
  STONK_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, STONK_EALIGN);
  STONK_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), STONK_EALIGN);
  const int out_mode = flags & STONK_EPI_OUT_MASK;
  if (out_mode == STONK_EPI_OUT_F16)   // plain conversion of the product: no other epilogue, no split
    STONK_CHECK_ARG((flags & 0x1FC) == 0 && split_k == 1, STONK_EINVAL);
  STONK_CHECK_ARG(split_k == 1 || out_mode == STONK_EPI_OUT_F32_ATOMIC, STONK_EINVAL);
  if (flags & STONK_EPI_BIAS) STONK_CHECK_ARG(bias && ((uintptr_t)bias % 16 == 0), STONK_EINVAL);
  if (flags & STONK_EPI_RESID) STONK_CHECK_ARG(resid && ldr % 4 == 0, STONK_EINVAL);
  if (flags & (STONK_EPI_SAVE_PREACT | STONK_EPI_GELU_BWD)) STONK_CHECK_ARG(aux && ldaux % 4 == 0, STONK_EINVAL);
  if (flags & STONK_EPI_DROPOUT) STONK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, STONK_EINVAL);
  if (flags & STONK_EPI_AUX_GRAD)   // a modifier of the two aux users; storing gelu' only makes sense next to the GELU
    STONK_CHECK_ARG((flags & STONK_EPI_GELU_BWD) || ((flags & STONK_EPI_SAVE_PREACT) && (flags & STONK_EPI_GELU)), STONK_EINVAL);
  if (M == 0) return STONK_OK;

  GemmArgs a;
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C;
  a.bias = bias; a.resid = (const bf16*)resid; a.aux = (bf16*)aux; a.m_dev = m_dev; a.k_dev = k_dev;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.ldaux = ldaux;
  a.M = M; a.N = N; a.K = K; a.flags = flags; a.alpha = alpha; a.split_k = split_k;
  a.drop_thr32 = stonk_drop_thr32(drop_p);
  a.drop_scale = 1.0f / (1.0f - drop_p);
  a.seed = stonk_seed_mix(seed);

  hipStream_t st = (hipStream_t)stream;
  // large launches go to the persistent 256x256 kernel (gemm256.hip); small ones keep the 128x128 tiles
  // measured on MI355X (tools/bench_kernels.py): the 256x256 kernel wins for wide outputs (N >= 2304, where its
  // halved operand traffic per flop outweighs one-workgroup-per-CU epilogues); N = 768 quantises badly (3 column tiles)
  // ... and epilogues that READ a second [M,N] operand (residual, saved pre-activation) still favour two co-resident
  // workgroups per CU hiding each other's load latency (tools/bench_epilogue.py)
  const bool both_sides = (flags & STONK_EPI_GELU_BWD) && (flags & STONK_EPI_RESID);
  // (bias + dropout + residual has no constant-flag instance on the 256x256 kernel: it would spill)
  const bool bdr = (flags & STONK_EPI_DROPOUT) && (flags & STONK_EPI_RESID);
  const bool big = M >= 1024 && N >= 1536 && out_mode != STONK_EPI_OUT_F32_ATOMIC && !both_sides && !bdr;
  // its epilogue moves 16-byte row segments: strides of every side operand must keep them aligned
  const bool v2_ok = ldc % 8 == 0 && (!(flags & STONK_EPI_BIAS) || alpha == 1.0f) &&  // bias rides in the accumulators
                      (!(flags & STONK_EPI_RESID) || (ldr % 8 == 0 && (uintptr_t)resid % 16 == 0)) &&
                     (!(flags & (STONK_EPI_SAVE_PREACT | STONK_EPI_GELU_BWD)) ||
                      (ldaux % 8 == 0 && (uintptr_t)aux % 16 == 0));
  // the four-wave kernel walks K tiles in pairs: an even number per work item
  // ... and addresses its operands with 32-bit byte offsets from a per-K-tile base, chunk-swizzled by XOR (ld % 64)
  const bool w4_ok = ldc % 8 == 0 && !both_sides && out_mode != STONK_EPI_OUT_F16 && (K / BK) % (2 * split_k) == 0 && !k_dev && lda % 64 == 0 &&
                     ldb % 64 == 0 && (long)M * lda < (1L << 30) && (long)N * ldb < (1L << 30) && (long)M * ldc * (out_mode == 0 ? 2 : 4) < (1L << 31) &&
                     (!(flags & STONK_EPI_RESID) || (ldr % 8 == 0 && (uintptr_t)resid % 16 == 0)) &&
                     (!(flags & (STONK_EPI_SAVE_PREACT | STONK_EPI_GELU_BWD)) ||
                      (ldaux % 8 == 0 && (uintptr_t)aux % 16 == 0));
  // AUTO, as measured in the training step (interleaved A/B, tools/ab_step.py):
  //  * launches whose epilogue reads a second [M,N] operand (residual, saved GELU') go to the four-wave kernel: its
  //    128x128 wave tiles halve the LDS bytes per flop and its epilogue requests the side operand a round ahead
  //    (FFN-down / attention-output forward 157 against 174 us, dgrad + residual 152 against 177, dgrad through GELU' 172
  //    against 200; -1.4 ms per step together) - on 256x192 tiles where those quantise better (N = 768: two full rounds
  //    of the CUs instead of one and a half; another 8-21 % per launch);
  //  * the label-sparse decoders (fp16 / fp32 output) and wide launches that tile evenly by 256 stay on the eight-wave kernel;
  //  * everything else (N = 768 without a side operand, split-K atomics, small M) keeps the 128x128 tiles.
  const bool w4_side = (flags & (STONK_EPI_RESID | STONK_EPI_GELU_BWD)) != 0;
  const bool dispatched = kernel == STONK_GEMM_DISPATCHED || kernel == STONK_GEMM_DISPATCHED2;
  int k = dispatched ? STONK_GEMM_AUTO : kernel;
  if (k == STONK_GEMM_AUTO)
    k = (w4_ok && w4_side && M >= 1024 && out_mode == STONK_EPI_OUT_BF16) ? STONK_GEMM_WAVE4
        // FFN-up (bias + GELU, with or without the saved GELU'): the four-wave kernel, 36.20 against 36.58 ms per step in
        // round 2's interleaved A/B (its GELU epilogue is the longest of the step; fused QKV on it changes nothing: 36.47 / 36.50)
        : (w4_ok && big && (flags & STONK_EPI_GELU) && out_mode == STONK_EPI_OUT_BF16) ? STONK_GEMM_WAVE4
        // fused QKV (bias only, N = 2304 = 12 x 192: six full rounds of the CUs instead of four and a half): 35.24 against
        // 35.70 ms per step on 256x192 tiles, where the same kernel on 256x256 tiles changed nothing
        : (w4_ok && big && (flags & 0x1FC) == STONK_EPI_BIAS && N % 192 == 0 && out_mode == STONK_EPI_OUT_BF16)
              ? STONK_GEMM_WAVE4
        : (v2_ok && big)                                                     ? STONK_GEMM_WAVE8
        // ... and the plain N = 768 launches (attention-output dgrad, the head transform's dgrad) since the four-wave kernel
        // has 256x192 tiles: 42.6 against 52.3 us (tools/bench_w4_tiles.py)
        : (w4_ok && M >= 1024 && N % 192 == 0 && out_mode == STONK_EPI_OUT_BF16 && (flags & 0x1FC) == 0) ? STONK_GEMM_WAVE4
                                                                             : STONK_GEMM_TILE128;
  // the written-out four-wave kernel (gemm_a4.hip): no device-side K, one side operand at most; bf16 output with the fused
  // epilogues (no K split, alpha = 1), or the plain product as fp16, or as fp32 added atomically over a K split. Its
  // buffers start at a tile's first row: 256 rows of an operand must fit 2^30 bytes, the whole operand need not (16 384
  // x 175 104 logits)
  const bool a4_plain = (flags & 0x1FC) == 0;
  const bool a4_ok = ldc % 8 == 0 && !both_sides && !k_dev && (K / BK) % 2 == 0 && lda % 64 == 0 && ldb % 64 == 0 &&
                     lda < (1L << 20) && ldb < (1L << 20) && ldc < (1L << 20) &&
                     (out_mode == STONK_EPI_OUT_BF16 ? (w4_ok && split_k == 1 && alpha == 1.0f)
                      : out_mode == STONK_EPI_OUT_F16 ? (a4_plain && alpha == 1.0f)
                      : out_mode == STONK_EPI_OUT_F32_ATOMIC ? a4_plain : false);
  if (k == STONK_GEMM_ASM4 || k == STONK_GEMM_ASM4_192) {
    STONK_CHECK_ARG(a4_ok, STONK_ESHAPE);
    return stonk_gemm_a4_launch(a, k == STONK_GEMM_ASM4 ? 256 : 192, 0, st);
  }
  // AUTO (and DISPATCHED): every bf16-output launch of at least a thousand rows whose epilogue the written-out kernel has an
  // instance of goes there (round 4, tools/a4_probe.py at 26 432 rows, alone, us: QKV 89 against 106 on the compiled
  // four-wave kernel, attention-output 41 / 49, FFN-up 166 -> see profiles/ / 186, FFN-down 107 / 124, dgrad through GELU'
  // 152 / 157, dgrad + residual 105 / 119 and 82 / 92, plain 768 x 768 36 / 39); it picks its tile width itself
  // (fp16 logits too: round 4, the entity decoder 830 -> see profiles/ us; the atomic form only where the caller names it)
  if ((kernel == STONK_GEMM_AUTO || dispatched) && a4_ok && M >= 1024 && out_mode != STONK_EPI_OUT_F32_ATOMIC) {
    const int rc = stonk_gemm_a4_launch(a, 0, !dispatched ? 0 : (kernel == STONK_GEMM_DISPATCHED2 ? 2 : 1), st);
    if (rc != STONK_ESHAPE) return rc;   // (an epilogue it has no instance of: the older kernels below)
  }
  if (k == STONK_GEMM_WAVE4 || k == STONK_GEMM_WAVE4_192) {
    STONK_CHECK_ARG(w4_ok, STONK_ESHAPE);
    // chosen by AUTO: the launcher also picks the tile width (256x192 where N = 768 / 2304 quantise better on 256 CUs)
    const bool chosen = kernel == STONK_GEMM_AUTO || dispatched;
    return stonk_gemm_w4_launch(a, out_mode, chosen ? 0 : (k == STONK_GEMM_WAVE4 ? 256 : 192),
                                !dispatched ? 0 : (kernel == STONK_GEMM_DISPATCHED2 ? 2 : 1), st);
  }
  if (k == STONK_GEMM_WAVE8 && dispatched) k = STONK_GEMM_TILE128;
  if (k == STONK_GEMM_WAVE8) {
    STONK_CHECK_ARG(v2_ok, STONK_ESHAPE);
    return stonk_gemm256_launch(a, out_mode, st);
  }
  const long tiles = (long)((M + BM - 1) / BM) * (N / BN) * split_k;
  // with a device-side row count the grid is capped and blocks walk the tiles that exist at run time
  const long cap = m_dev ? 4096 : tiles;
  const int grid = (int)(tiles < cap ? tiles : cap);
  constexpr int Bi = STONK_EPI_BIAS, G = STONK_EPI_GELU, SV = STONK_EPI_SAVE_PREACT, GB = STONK_EPI_GELU_BWD,
                R = STONK_EPI_RESID, D = STONK_EPI_DROPOUT, AG = STONK_EPI_AUX_GRAD;
  const int epi = flags & (Bi | G | SV | GB | R | D | AG);
  if (out_mode == STONK_EPI_OUT_F32_ATOMIC) return launch<2, true, -1>(a, grid, st);
  if (out_mode == STONK_EPI_OUT_F16) return launch<3, true, 0>(a, grid, st);
  if (out_mode == STONK_EPI_OUT_F32) return epi == 0 ? launch<1, true, 0>(a, grid, st) : launch<1, true, -1>(a, grid, st);
  switch (epi) {
    case 0: return launch<0, true, 0>(a, grid, st);
    case Bi: return launch<0, true, Bi>(a, grid, st);
    case Bi | G | SV: return launch<0, true, Bi | G | SV>(a, grid, st);
    case GB: return launch<0, true, GB>(a, grid, st);
    case Bi | G | SV | AG: return launch<0, true, Bi | G | SV | AG>(a, grid, st);
    case GB | AG: return launch<0, true, GB | AG>(a, grid, st);
    case R: return launch<0, true, R>(a, grid, st);
    case Bi | R: return launch<0, true, Bi | R>(a, grid, st);
    case Bi | R | D: return launch<0, true, Bi | R | D>(a, grid, st);
    default: return launch<0, true, -1>(a, grid, st);
  }
}

// 4: stonk_comm_*, STONK_GEMM_ASM4*, written-out TN kernel behind split_k <= 0; 5: stonk_layernorm_bwd_reduce + STONK_LN_DEFER_REDUCE,
// the ASM4 kernels' fp16 / split-K atomic forms, 256 gradient-norm slots (stonk_sumsq_workspace_floats)
extern "C" int stonk_abi_version(void) { return 5; }
