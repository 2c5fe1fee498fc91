// Weight gradient on the four-wave schedule of gemm_w4.hip (256x256 tiles, one wave per SIMD, 128x128 wave tiles, operands
// global -> registers -> LDS -> fragments, one memory instruction per MFMA gap):
//   C[M', N'] (fp32) += alpha * sum_t A[t][m] * B[t][n]        A = dY [T, M'], B = X [T, N'], both row-major by token
//   bias[m]          += alpha * sum_t A[t][m]
// Same contract as gemm_tn.hip (stonk_gemm_tn_bf16 with split_k == 0 lands here). Why it pays here when the four-wave
// NT kernel does not pay in the step: a weight gradient has 9-36 output tiles and a contraction of 32 768 tokens, so a
// workgroup runs ONE long K loop (70+ K tiles) and one atomic epilogue - it is all main loop, which is where the
// schedule is strongest (1.3 PFLOP/s at long K against 0.75 for the 128x128 kernel).
//
// Differences from the NT kernel:
//  * a K tile is 64 TOKENS; an operand image is EIGHT sub-tiles of [64 tokens][32 features] - one per MFMA feature block,
//    64-byte rows, no swizzle: a transposed read touches 4 consecutive rows = 256 contiguous bytes, every bank once - so
//    every fragment address is one of 16 per-lane bases plus an immediate (the first version, 128-byte rows with an XOR
//    swizzle, had the compiler precompute and spill dozens of address variants); wave w loads the sub-tile pair
//    (2w, 2w+1) of A and of B: 8 pieces of 16 token rows x 64 B, the two halves of a 128-byte line back to back;
//  * fragments are transposed reads (ds_read_b64_tr_b16 pairs, 16 per k step): lane (feature, hh) receives 8 tokens;
//    both operands use the same token order, so the contraction is consistent;
//  * tokens past the end (device-side counts, K not a multiple of 128) read as zeros through the buffer range check, so
//    work items always have an even number of K tiles and need no tail code;
//  * the bias gradient (column sums of dY) comes off the A fragments the lane already holds: 4 v_dot2c_f32_bf16 against
//    ones per fragment into one float per feature block; done by the wave-column-0 waves of EVERY column tile for every
//    ntn-th K tile, so no workgroup carries more than 1/ntn of it (a fifth accumulator tile would not fit the registers);
//  * output by fp32 atomics (split-K), through the same LDS slab as the NT kernel's epilogue. Measured (3072 x 768,
//    32 768 tokens, 7 splits): 142 us for the K loops alone = 1.09 PFLOP/s, 195 us with the epilogue - the 64 MB of
//    float atomics cost about 50 us whatever their scope; the next step is to group a layer's four weight gradients
//    into one launch (2 splits instead of 7).
#include "gemm_common.h"

using namespace stonk_gemm;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int IMG_BYTES = 256 * BK * 2;             // one operand of one K tile: 32 KiB = 4 sub-tiles of 8 KiB
constexpr int STAGE_BYTES = 2 * IMG_BYTES;          // 64 KiB
constexpr int LDS_RING = 2 * STAGE_BYTES;           // 128 KiB
constexpr int SLAB_BYTES = 32 * 64 * 4;             // per wave: 32 rows x 64 fp32 columns
constexpr int LDS_BYTES = LDS_RING + 4 * SLAB_BYTES;  // 160 KiB

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct Work {
  int m0, n0;       // tile origin
  long k_begin;     // first token
  int nk;           // K tiles in this work item (even)
};

__device__ __forceinline__ void barrier() { __builtin_amdgcn_s_barrier(); }
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(256, 1) void gemm_tn_w4_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int M = p.M, N = p.N;
  int Kact = p.K;                                   // tokens that exist
  if (p.k_dev) {
    const int kd = *p.k_dev;
    Kact = kd < Kact ? kd : Kact;
  }
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  const int nk_total = (Kact + BK - 1) / BK;
  int nk_per = (nk_total + p.split_k - 1) / p.split_k;
  nk_per += nk_per & 1;                             // even: tokens past Kact read as zeros, extra K tiles are harmless
  const int per_split = ntm * ntn;
  const int total = per_split * p.split_k;
  const int G = gridDim.x;

  // work item -> (K split, tile); the workgroups of one XCD (blockIdx & 7) take neighbouring items: same K range
  auto get_work = [&](int w, Work& o) -> bool {
    if (w >= total) return false;
    int idx = w;
    {   // any grid size: XCD x = b & 7 hosts ceil((G - x) / 8) workgroups; give them one contiguous run of items
      const int r = w / G, b = w - r * G;
      const int x = b & 7, base = G >> 3, rem = G & 7;
      const int cand = r * G + x * base + (x < rem ? x : rem) + (b >> 3);
      if ((r + 1) * G <= total) idx = cand;
    }
    const int ks = idx / per_split;
    const int tt = idx - ks * per_split;
    int rt, ct;
    if (ntm >= ntn) {
      rt = tt / ntn;
      ct = tt - rt * ntn;
    } else {
      ct = tt / ntm;
      rt = tt - ct * ntm;
    }
    o.m0 = rt * BM;
    o.n0 = ct * BN;
    o.k_begin = (long)ks * nk_per * BK;
    int nk = nk_total - ks * nk_per;
    nk = nk < nk_per ? nk : nk_per;
    o.nk = nk <= 0 ? 0 : nk + (nk & 1);
    return true;
  };

  // ---- global -> staging registers: piece q (0-7: A, 8-15: B) = token rows 16 (q >> 1) .. +15 of sub-tile 2 wave + (q & 1)
  // (features 64 wave + 32 (q & 1) .. +31 of the tile): lane -> (row lane >> 2, 16-byte chunk lane & 3). Address = one
  // per-lane offset per operand + a scalar (first token of the K tile * ld + tile column + 16 (q >> 1) * ld + 64 (q & 1));
  // tokens past Kact are out of the buffer's range and read as zeros.
  const int lda2 = (int)p.lda * 2, ldb2 = (int)p.ldb * 2;
  const int voffA = (lane >> 2) * lda2 + wave * 128 + (lane & 3) * 16;
  const int voffB = (lane >> 2) * ldb2 + wave * 128 + (lane & 3) * 16;
  const int pstepA = 16 * lda2, pstepB = 16 * ldb2;
  // The buffer resources are re-based on the first token of every K tile (64-bit scalar adds), so operand extents beyond
  // 4 GB work - the entity decoder's dlogits are [16 384 x 175 104] bf16 = 5.7 GB - and `num_records` = the bytes from
  // there to the end of the live tokens (clamped to 32 bits; a K tile spans 64 rows, far less than the clamp).
  const bf16* curA = p.A;
  const bf16* curB = p.B;
  unsigned recA = 0, recB = 0;
  long ptok = 0;              // first token of the K tile the prefetch cursor points at
  int soffA = 0, soffB = 0;   // tile column, bytes
  auto set_cursor = [&]() {
    curA = p.A + ptok * p.lda;
    curB = p.B + ptok * p.ldb;
    const long left = (long)Kact - ptok;
    long ra = left * lda2, rb = left * ldb2;
    ra = ra < 0 ? 0 : (ra > 0xFFFFF000L ? 0xFFFFF000L : ra);
    rb = rb < 0 ? 0 : (rb > 0xFFFFF000L ? 0xFFFFF000L : rb);
    recA = (unsigned)ra;
    recB = (unsigned)rb;
  };
  const int wofs = wave * 8192 + lane * 16;   // sub-tile pair of this wave: 2 x 4 KiB
  auto set_sources = [&](const Work& w) {
    ptok = w.k_begin;
    soffA = w.m0 * 2;
    soffB = w.n0 * 2;
    set_cursor();
  };
  auto load_piece = [&](const int q, bf16x8& d) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int qq = q & 7;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)curA, 0, recA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)curB, 0, recB, 0x00020000);
    const u32x4 v = (q < 8) ? __builtin_amdgcn_raw_buffer_load_b128(rsrcA, voffA, soffA + (qq >> 1) * pstepA + (qq & 1) * 64, 0)
                            : __builtin_amdgcn_raw_buffer_load_b128(rsrcB, voffB, soffB + (qq >> 1) * pstepB + (qq & 1) * 64, 0);
    d = __builtin_bit_cast(bf16x8, v);
  };
  auto write_piece = [&](const int stage, const int q, const bf16x8& d) {
    const int qq = q & 7;
    *(bf16x8*)(smem + stage * IMG_BYTES + (q < 8 ? 0 : 2 * IMG_BYTES) + (qq & 1) * 4096 + (qq >> 1) * 1024 + wofs) = d;
  };
  bf16x8 stg[4][4];   // [k step][piece]

  // ---- transposed fragment reads. Feature block b of this wave's 128 is sub-tile 4 half + b; a k step is 16 token rows.
  // Lane (g = lane >> 4, i = lane & 15) addresses row 16 ks + 4 (g >> 1) + (i >> 2) (+8 for the second read), 4 features
  // at 16 (g & 1) + 4 (i & 3); it receives feature (lane & 31) x 8 tokens. 8 bases per operand, all else immediates.
  int tofsA[4][2], tofsB[4][2];
  {
    const int g = lane >> 4, i = lane & 15;
    const int col = 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 16 * ks + 4 * (g >> 1) + (i >> 2) + 8 * h;
        tofsA[ks][h] = row * 64 + col * 2 + wr * 16384;
        tofsB[ks][h] = row * 64 + col * 2 + wc * 16384 + 2 * IMG_BYTES;
      }
  }
  // B fragments (all four used by every sub-block) are double-buffered across k steps; an A fragment is used by one
  // sub-block only, so the NEXT k step's copy is read into the same registers right after that sub-block's MFMAs have
  // issued (16 registers instead of 32 - the kernel sits at the 256-VGPR limit, and a spill inside the K loop costs a
  // compiler-placed vmcnt(0), i.e. the whole prefetch distance)
  bf16x8 fra[4], frb[2][4];
  auto tr_read = [&](const int (&tofs)[4][2], const int stage, const int blk, const int ks) -> bf16x8 {
    const char* s = smem + stage * IMG_BYTES + blk * 4096;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s + tofs[ks][0]));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s + tofs[ks][1]));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  f32x16 acc[4][4];    // [A feature block][B feature block]
  float accb[4];       // bias gradient: this lane's share (8 of every 16 tokens) of the column sum of feature lane & 31
  f32x16 fzero;
#pragma unroll
  for (int e = 0; e < 16; ++e) fzero[e] = 0.f;
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
  const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};

  // ------------------------------------------------------------------ stream state
  Work cw, pw;
  int cwi = blockIdx.x;
  if (!get_work(cwi, cw)) return;
  while (cw.nk <= 0) {
    cwi += G;
    if (!get_work(cwi, cw)) return;
  }
  int pwi = cwi;
  pw = cw;
  int pk = 0;
  bool p_valid = true;
  set_sources(pw);
  auto advance_prefetch = [&]() {
    if (!p_valid) return;   // out of work: the cursor keeps re-reading its last K tile (never consumed)
    if (pk + 1 < pw.nk) {
      ++pk;
      ptok += BK;
      set_cursor();
      return;
    }
    Work nw;
    int nwi = pwi;
    bool ok;
    do {
      nwi += G;
      ok = get_work(nwi, nw);
    } while (ok && nw.nk <= 0);
    if (!ok) {
      p_valid = false;
      return;
    }
    pwi = nwi;
    pw = nw;
    pk = 0;
    set_sources(pw);
  };

  // One k step (see gemm_w4.hip): ST = stage of the K tile being multiplied, KS = k step, FIRST = first k step of a work
  // item (accumulators start from the MFMA's zero constant), BIAS = this K tile contributes to the bias gradient.
#define STONK_TNW4_KSTEP(ST, KS, FIRST)                                                                \
  do {                                                                                                 \
    constexpr int WST = ((KS) == 3) ? (ST) : ((ST) ^ 1);                                               \
    constexpr int Q0 = 4 * (((KS) + 1) & 3);                                                           \
    constexpr int RST = ((KS) == 3) ? ((ST) ^ 1) : (ST);                                               \
    constexpr int RKS = ((KS) + 1) & 3;                                                                \
    wait_vm<12>();                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
      /* A block 3 of THIS k step arrives late (sub-block 0), blocks 0-2 of the next k step after their last use */ \
      if (g == 0) fra[3] = tr_read(tofsA, ST, 3, KS);                                                  \
      else fra[g - 1] = tr_read(tofsA, RST, g - 1, RKS);                                               \
      frb[RKS & 1][g] = tr_read(tofsB, RST, g, RKS);                                                   \
      write_piece(WST, Q0 + g, stg[KS][g]);                                                            \
      load_piece(Q0 + g, stg[KS][g]);                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frb[(KS) & 1][j], fra[g],                   \
                                                            (FIRST) ? fzero : acc[g][j], 0, 0, 0);       \
      }                                                                                                \
      if (do_bias) {                                                                                   \
        const bf16x8 f = fra[g];                                                                       \
        accb[g] = __builtin_amdgcn_fdot2_f32_bf16((bf16x2){f[0], f[1]}, one2, accb[g], false);          \
        accb[g] = __builtin_amdgcn_fdot2_f32_bf16((bf16x2){f[2], f[3]}, one2, accb[g], false);          \
        accb[g] = __builtin_amdgcn_fdot2_f32_bf16((bf16x2){f[4], f[5]}, one2, accb[g], false);          \
        accb[g] = __builtin_amdgcn_fdot2_f32_bf16((bf16x2){f[6], f[7]}, one2, accb[g], false);          \
      }                                                                                                \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* the late A fragment first (g = 0: needed by g = 3) */ \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* MFMA */                                    \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* DS reads */                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); /* DS write */                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* buffer load */                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
    if ((KS) == 2) {                                                                                   \
      advance_prefetch();                                                                              \
      wait_lgkm0();                                                                                    \
      barrier();                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
  } while (0)
#define STONK_TNW4_KTILE(ST, FIRST)    \
  do {                                 \
    STONK_TNW4_KSTEP(ST, 0, FIRST);    \
    STONK_TNW4_KSTEP(ST, 1, false);    \
    STONK_TNW4_KSTEP(ST, 2, false);    \
    STONK_TNW4_KSTEP(ST, 3, false);    \
  } while (0)

  // ---- prologue (not pipelined): K tile 0 complete in stage 0, pieces 0-3 of K tile 1 in stage 1, the rest of K tile 1
  // and pieces 0-3 of K tile 2 requested into the staging registers, first fragments read
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(4 * g + i, stg[g][i]);
  advance_prefetch();
  wait_vm<0>();
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) write_piece(0, 4 * g + i, stg[g][i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) load_piece(i, stg[3][i]);
  wait_vm<0>();
#pragma unroll
  for (int i = 0; i < 4; ++i) write_piece(1, i, stg[3][i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(4 * (g + 1) + i, stg[g][i]);
  advance_prefetch();
#pragma unroll
  for (int i = 0; i < 4; ++i) load_piece(i, stg[3][i]);
  wait_lgkm0();
  barrier();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < 3) fra[g] = tr_read(tofsA, 0, g, 0);
    frb[0][g] = tr_read(tofsB, 0, g, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: the 128x128 corner leaves in 8 rounds of 32 rows x 64 columns through a private 8 KiB slab; one
  // atomic wave-instruction = 64 consecutive floats of one row of dW
  auto store_tile = [&](const Work& w, const bool with_bias) {
    int lv = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wv = wave;
    asm volatile("" : "+v"(lv), "+s"(wv));
    const int r = lv & 31, hh = lv >> 5;
    char* ep = smem + LDS_RING + wv * SLAB_BYTES;
    const int wm0 = w.m0 + (wv >> 1) * 128, wn0 = w.n0 + (wv & 1) * 128;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int mrow0 = wm0 + bi * 32;
        const int n0 = wn0 + qb * 64;
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = bj * 8 + 2 * g + hh;
            const f32x16& c = acc[bi][2 * qb + bj];
            f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
            *(f32x4*)(ep + r * 256 + ((ch ^ (r & 15)) << 4)) = v * p.alpha;
          }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int rr = 0; rr < 32; ++rr) {
          const float x = *(const float*)(ep + rr * 256 + (((lv >> 2) ^ (rr & 15)) << 4) + (lv & 3) * 4);
          if (mrow0 + rr < M && n0 + lv < N) atomicAdd((float*)p.C + (long)(mrow0 + rr) * p.ldc + n0 + lv, x);
        }
        __builtin_amdgcn_wave_barrier();
      }
      // bias gradient of this block's 32 features: the two lanes (hh = 0, 1) of a feature each add their half
      if (with_bias && wm0 + bi * 32 + r < M) atomicAdd((float*)p.bias + wm0 + bi * 32 + r, accb[bi] * p.alpha);
    }
  };

  // ------------------------------------------------------------------ stream of K tiles
  for (;;) {
    // bias gradient: wave column 0 of every column tile sums dY over the K tiles kt = ct (mod ntn) of this item
    const bool bias_wave = p.bias != nullptr && wc == 0;
    const int ct = cw.n0 / BN;
    const int kt0 = (int)(cw.k_begin / BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = 0.f;
    {
      const bool do_bias = bias_wave && ((kt0 + 0) % ntn == ct);
      STONK_TNW4_KTILE(0, true);
    }
    {
      const bool do_bias = bias_wave && ((kt0 + 1) % ntn == ct);
      STONK_TNW4_KTILE(1, false);
    }
    for (int ck = 2; ck < cw.nk; ck += 2) {
      {
        const bool do_bias = bias_wave && ((kt0 + ck) % ntn == ct);
        STONK_TNW4_KTILE(0, false);
      }
      {
        const bool do_bias = bias_wave && ((kt0 + ck + 1) % ntn == ct);
        STONK_TNW4_KTILE(1, false);
      }
    }
    store_tile(cw, bias_wave);
    Work nw;
    bool more;
    do {
      cwi += G;
      more = get_work(cwi, nw);
    } while (more && nw.nk <= 0);
    if (!more) break;
    cw = nw;
  }
  wait_vm<0>();
#undef STONK_TNW4_KTILE
#undef STONK_TNW4_KSTEP
}

}  // namespace

// weight-gradient form on the four-wave schedule; a.split_k must already be chosen
int stonk_gemm_tn_w4_launch(const GemmArgs& a, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    n_cu = prop.multiProcessorCount;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.split_k;
  const int cap = (a.flags > 0 && a.flags < n_cu) ? a.flags : n_cu;
  const int grid = (int)(tiles < cap ? tiles : cap);
  hipLaunchKernelGGL(gemm_tn_w4_kernel, dim3(grid), dim3(256), LDS_BYTES, st, a);
  return stonk_launch_status();
}
