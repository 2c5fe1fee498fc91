// Weight gradient on the written-out four-wave loop (gemm_a4.hip's pipeline; the K loop is generated text,
// tools/gen_gemm_a4.py -> gemm_a4_loop.inc, STONK_TN_A4_*):
//   C[M', N'] (fp32) += alpha * sum_t A[t][m] * B[t][n]        A = dY [T, M'], B = X [T, N'], both row-major by token
//   bias[m]          += alpha * sum_t A[t][m]
// Same contract as gemm_tn_w4.hip (stonk_gemm_tn_bf16 with split_k == 0 / <= -16 lands here).
//
// What carries over from the NT kernel: 256 x 256 tiles, four waves (one per SIMD, 128 x 128 wave tiles),
// v_mfma_f32_16x16x32_bf16 into a[0:255], operands by LDS-DMA two K tiles ahead and across work items, two barriers per
// K tile, every wait "all but this K tile's own pieces". What is specific to the contraction running over ROWS:
//  * a K tile = 64 tokens x 512 B per operand; a 1-KiB piece = two token rows = eight full 128-byte lines; rows sit 512 B
//    apart in LDS (every row on the same banks), so the sixteen 32-byte feature-block segments of a row are XOR-permuted by
//    f(row) = (row & 3) | ((row >> 3) & 1) << 2 - applied to the SOURCE address, the image stays lane-linear;
//  * fragments by ds_read_b64_tr_b16 pairs (4 token rows x 16 features per 16-lane group): the half-wave's eight rows fall
//    into eight different 32-byte bank slots; one per-lane address per feature block, the rest immediates;
//  * srcA = the dY fragment, so a lane holds one output COLUMN: after one v_permlane16_swap per register pair a register
//    covers two rows x 32 consecutive columns = two 128-byte segments, the shape float atomics run at full rate with - no
//    LDS slab;
//  * bias gradient: one extra MFMA per dY block against an all-ones operand on every ntn-th K tile, into VGPR accumulators;
//  * tokens past the (device-side) count are out of the buffers' range and arrive as zeros; the cursors carry 64-bit
//    byte counts (the entity decoder's dlogits are 5.7 GB).
#include "gemm_common.h"
#ifndef STONK_A4_LOOP_INC
#define STONK_A4_LOOP_INC "gemm_a4_loop.inc"
#endif
#include STONK_A4_LOOP_INC

using namespace stonk_gemm;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int IMG_BYTES = BK * 512;            // one operand of one K tile: 64 token rows x 512 B = 32 KiB
constexpr int LDS_BYTES = 4 * IMG_BYTES;       // two stages x (A image + B image)

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Work {
  int m0, n0;
  long k_begin;     // first token
  int nk;           // K tiles (even; tokens past the end read as zeros)
};

struct Cursor {     // where an operand stream stands: buffer words + bytes left from the base to the end of the live tokens
  i32x4 srd;
  int rem_lo, rem_hi;
};

__global__ __launch_bounds__(256, 1) void gemm_tn_a4_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int M = p.M, N = p.N;
  int Kact = p.K;
  if (p.k_dev) {
    const int kd = *p.k_dev;
    Kact = kd < Kact ? kd : Kact;
  }
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  const int nk_total = (Kact + BK - 1) / BK;
  int nk_per = (nk_total + p.split_k - 1) / p.split_k;
  nk_per += nk_per & 1;
  const int per_split = ntm * ntn;
  const int total = per_split * p.split_k;
  const int G = gridDim.x;

  auto get_work = [&](int w, Work& o) -> bool {
    if (w >= total) return false;
    int idx = w;
    {   // any grid size: the workgroups of one XCD (blockIdx & 7) take one contiguous run of items per round
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
  auto next_work = [&](int& wi, Work& o) -> bool {   // (k_dev may leave a K split empty)
    do {
      wi += G;
      if (!get_work(wi, o)) return false;
    } while (o.nk <= 0);
    return true;
  };

  // ---- LDS-DMA sources: piece q of this wave = token rows 16 wave + 2 q, + 1 of the K tile (two rows x 512 B); a lane moves
  // 16 bytes: row lane >> 5, PHYSICAL 16-byte chunk lane & 31 of the image row = 32-byte segment (lane & 31) >> 1, i.e.
  // LOGICAL segment ((lane & 31) >> 1) ^ f(row)
  const int lda2 = (int)p.lda * 2, ldb2 = (int)p.ldb * 2;
  int voffA[8], voffB[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = 16 * wave + 2 * q + (lane >> 5);
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int c = (((((lane & 31) >> 1) ^ f) << 1) | (lane & 1)) << 4;
    voffA[q] = row * lda2 + c;
    voffB[q] = row * ldb2 + c;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int m0a = (int)lds0 + wave * 8192;
  const int m0b = (int)lds0 + 2 * IMG_BYTES + wave * 8192;
  // ---- transposed fragment reads: lane (g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3) addresses token row
  // 32 h + 8 g + 4 hi + q, features 4 pq .. + 3 of the 16-feature block; it receives feature lane & 15, four tokens
  int taA[8], taB[8];
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int f = q | ((g & 1) << 2);
    const int rowb = (8 * g + q) * 512 + pq * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      taA[i] = (int)lds0 + rowb + ((8 * wr + (i ^ f)) << 5);
      taB[i] = (int)lds0 + 2 * IMG_BYTES + rowb + ((8 * wc + (i ^ f)) << 5);
    }
  }
  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // eight bf16 ones

  auto cursor_of = [&](const Work& w, Cursor& a, Cursor& b) {
    const unsigned long pa = (unsigned long)p.A + (unsigned long)((w.k_begin * p.lda + w.m0) * 2);
    const unsigned long pb = (unsigned long)p.B + (unsigned long)((w.k_begin * p.ldb + w.n0) * 2);
    const long ra = ((long)Kact - w.k_begin) * lda2 - (long)w.m0 * 2, rb = ((long)Kact - w.k_begin) * ldb2 - (long)w.n0 * 2;
    auto clamp = [](long r) -> int { return (int)(unsigned)(r < 0 ? 0 : (r > 0xFFFFF000L ? 0xFFFFF000L : r)); };
    a.srd = (i32x4){(int)(unsigned)pa, (int)((pa >> 32) & 0xffff), clamp(ra), 0x00020000};
    b.srd = (i32x4){(int)(unsigned)pb, (int)((pb >> 32) & 0xffff), clamp(rb), 0x00020000};
    a.rem_lo = (int)(unsigned)ra;
    a.rem_hi = (int)(ra >> 32);
    b.rem_lo = (int)(unsigned)rb;
    b.rem_hi = (int)(rb >> 32);
  };

  Work cw, nw;
  int cwi = blockIdx.x;
  if (!get_work(cwi, cw)) return;
  if (cw.nk <= 0 && !next_work(cwi, cw)) return;
  Cursor ca, cb;
  cursor_of(cw, ca, cb);
  const int stepa = BK * lda2, stepb = BK * ldb2;

#define STONK_TN_A4_DMA_OPERANDS                                                                                        \
  [voffA0] "v"(voffA[0]), [voffA1] "v"(voffA[1]), [voffA2] "v"(voffA[2]), [voffA3] "v"(voffA[3]), [voffA4] "v"(voffA[4]), \
      [voffA5] "v"(voffA[5]), [voffA6] "v"(voffA[6]), [voffA7] "v"(voffA[7]), [voffB0] "v"(voffB[0]),                   \
      [voffB1] "v"(voffB[1]), [voffB2] "v"(voffB[2]), [voffB3] "v"(voffB[3]), [voffB4] "v"(voffB[4]),                   \
      [voffB5] "v"(voffB[5]), [voffB6] "v"(voffB[6]), [voffB7] "v"(voffB[7]), [m0a] "s"(m0a), [m0b] "s"(m0b),           \
      [stepa] "s"(stepa), [stepb] "s"(stepb)

  // K tiles 0 and 1 of the first work item
  asm volatile(STONK_TN_A4_PROLOGUE
               : "+{s[36:39]}"(ca.srd), "+{s[40:43]}"(cb.srd), [ralo] "+s"(ca.rem_lo), [rahi] "+s"(ca.rem_hi),
                 [rblo] "+s"(cb.rem_lo), [rbhi] "+s"(cb.rem_hi)
               : STONK_TN_A4_DMA_OPERANDS
               : "m0", "scc", "s48", "memory");

  const bool want_bias = p.bias != nullptr;
  const int r16 = lane & 15, qq = lane >> 4;

  for (;;) {
    int nwi = cwi;
    const bool more = next_work(nwi, nw);
    Cursor na, nb;
    cursor_of(more ? nw : cw, na, nb);   // (no next item: the cursor re-reads this item's first K tiles, never consumed)
    f32x16 acc[16];
    f32x4 bacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int rem = cw.nk >> 1;
    // bias duty: column tile ct sums dY over the K tiles kt = ct (mod ntn) of its items (every workgroup carries 1/ntn of it)
    const int ct = cw.n0 / BN;
    const int kt0 = (int)(cw.k_begin / BK);
    int bph = want_bias ? ((kt0 - ct) % ntn + ntn) % ntn : 1;
    const int ntn_s = want_bias ? ntn : 0x7fffffff;   // (without a bias the phase never returns to 0)
#define STONK_A4_CLOBBERS_TN                                                                                           \
  "m0", "scc", "s48", "memory", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", \
      "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152",  \
      "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166",  \
      "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180",  \
      "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194",  \
      "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208",  \
      "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222",  \
      "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236",  \
      "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250",  \
      "v251", "v252", "v253", "v254", "v255"
    asm volatile(STONK_TN_A4_TILE
                 : "={a[0:15]}"(acc[0]), "={a[16:31]}"(acc[1]), "={a[32:47]}"(acc[2]), "={a[48:63]}"(acc[3]),
                   "={a[64:79]}"(acc[4]), "={a[80:95]}"(acc[5]), "={a[96:111]}"(acc[6]), "={a[112:127]}"(acc[7]),
                   "={a[128:143]}"(acc[8]), "={a[144:159]}"(acc[9]), "={a[160:175]}"(acc[10]), "={a[176:191]}"(acc[11]),
                   "={a[192:207]}"(acc[12]), "={a[208:223]}"(acc[13]), "={a[224:239]}"(acc[14]), "={a[240:255]}"(acc[15]),
                   "+{s[36:39]}"(ca.srd), "+{s[40:43]}"(cb.srd), [ralo] "+s"(ca.rem_lo), [rahi] "+s"(ca.rem_hi),
                   [rblo] "+s"(cb.rem_lo), [rbhi] "+s"(cb.rem_hi), [rem] "+s"(rem), [bph] "+s"(bph),
                   [bacc0] "+v"(bacc[0]), [bacc1] "+v"(bacc[1]), [bacc2] "+v"(bacc[2]), [bacc3] "+v"(bacc[3]),
                   [bacc4] "+v"(bacc[4]), [bacc5] "+v"(bacc[5]), [bacc6] "+v"(bacc[6]), [bacc7] "+v"(bacc[7])
                 : STONK_TN_A4_DMA_OPERANDS, [taA0] "v"(taA[0]), [taA1] "v"(taA[1]), [taA2] "v"(taA[2]), [taA3] "v"(taA[3]),
                   [taA4] "v"(taA[4]), [taA5] "v"(taA[5]), [taA6] "v"(taA[6]), [taA7] "v"(taA[7]), [taB0] "v"(taB[0]),
                   [taB1] "v"(taB[1]), [taB2] "v"(taB[2]), [taB3] "v"(taB[3]), [taB4] "v"(taB[4]), [taB5] "v"(taB[5]),
                   [taB6] "v"(taB[6]), [taB7] "v"(taB[7]), [ones] "v"(ones), [ntn] "s"(ntn_s),
                   [nal] "s"(na.srd[0]), [nah] "s"(na.srd[1]), [nralo] "s"(na.rem_lo), [nrahi] "s"(na.rem_hi),
                   [nbl] "s"(nb.srd[0]), [nbh] "s"(nb.srd[1]), [nrblo] "s"(nb.rem_lo), [nrbhi] "s"(nb.rem_hi)
                 : STONK_A4_CLOBBERS_TN);

    // ---- epilogue: accumulator block (i, j) = a[4 (8 i + j) ..]: lane (n = lane & 15, qq) holds rows 16 i + 4 qq + e, column
    // 16 j + n. A v_permlane16_swap of the blocks (i, 2 jp) = x and (i, 2 jp + 1) = y leaves, per register e,
    //   x: rows 16 i + 8 (lane >> 5) + e,     columns 32 jp + (lane & 31)
    //   y: rows 16 i + 8 (lane >> 5) + 4 + e, columns 32 jp + (lane & 31)
    // i.e. one atomic wave-instruction = two rows x 128 contiguous bytes.
    // Buffer atomics: a row past M' falls off the end of the buffer, a column past N' is pushed out of its range - no
    // predicate, no branch, 32-bit offsets (M' ldc * 4 < 2^31: the launcher checks).
    const int wm0 = cw.m0 + wr * 128, wn0 = cw.n0 + wc * 128;
    const int ldc_b = (int)p.ldc * 4;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, M * ldc_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, want_bias ? M * 4 : 0, 0x00020000);
    int coff[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const int n = wn0 + 32 * jp + (lane & 31);
      coff[jp] = n * 4 + (n < N ? 0 : 0x40000000);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float xs[4][4], ys[4][4];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const int bx = 8 * i + 2 * jp, by = bx + 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xs[jp][e] = acc[bx >> 2][4 * (bx & 3) + e];
          ys[jp][e] = acc[by >> 2][4 * (by & 3) + e];
        }
      }
#define STONK_A4_SWAP4(a, b) "v_permlane16_swap_b32 %" #a ", %" #b "\n"
      asm volatile("s_nop 1\n" STONK_A4_SWAP4(0, 16) STONK_A4_SWAP4(1, 17) STONK_A4_SWAP4(2, 18) STONK_A4_SWAP4(3, 19)
                   STONK_A4_SWAP4(4, 20) STONK_A4_SWAP4(5, 21) STONK_A4_SWAP4(6, 22) STONK_A4_SWAP4(7, 23)
                   STONK_A4_SWAP4(8, 24) STONK_A4_SWAP4(9, 25) STONK_A4_SWAP4(10, 26) STONK_A4_SWAP4(11, 27)
                   STONK_A4_SWAP4(12, 28) STONK_A4_SWAP4(13, 29) STONK_A4_SWAP4(14, 30) STONK_A4_SWAP4(15, 31) "s_nop 1"
                   : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[0][2]), "+v"(xs[0][3]), "+v"(xs[1][0]), "+v"(xs[1][1]),
                     "+v"(xs[1][2]), "+v"(xs[1][3]), "+v"(xs[2][0]), "+v"(xs[2][1]), "+v"(xs[2][2]), "+v"(xs[2][3]),
                     "+v"(xs[3][0]), "+v"(xs[3][1]), "+v"(xs[3][2]), "+v"(xs[3][3]),
                     "+v"(ys[0][0]), "+v"(ys[0][1]), "+v"(ys[0][2]), "+v"(ys[0][3]), "+v"(ys[1][0]), "+v"(ys[1][1]),
                     "+v"(ys[1][2]), "+v"(ys[1][3]), "+v"(ys[2][0]), "+v"(ys[2][1]), "+v"(ys[2][2]), "+v"(ys[2][3]),
                     "+v"(ys[3][0]), "+v"(ys[3][1]), "+v"(ys[3][2]), "+v"(ys[3][3]));
#undef STONK_A4_SWAP4
      const int mrow = wm0 + 16 * i + 8 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ro = (mrow + e) * ldc_b, ro4 = (mrow + 4 + e) * ldc_b;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(xs[jp][e] * p.alpha, rC, ro + coff[jp], 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(ys[jp][e] * p.alpha, rC, ro4 + coff[jp], 0, 0);
        }
      }
      // bias gradient of this block's 16 features: every column of the extra accumulator holds the same sums
      if (want_bias && wc == 0) {
        const int boff = r16 == 0 ? 0 : 0x40000000;   // (one lane per row adds)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(bacc[i][e] * p.alpha, rBias, (wm0 + 16 * i + 4 * qq + e) * 4 + boff, 0, 0);
      }
    }
    if (!more) break;
    cwi = nwi;
    cw = nw;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef STONK_TN_A4_DMA_OPERANDS
#undef STONK_A4_CLOBBERS_TN
}

}  // namespace

// weight-gradient form on the written-out loop; a.split_k already chosen, a.flags = grid cap (as gemm_tn_w4.hip)
int stonk_gemm_tn_a4_launch(const GemmArgs& a, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    n_cu = prop.multiProcessorCount;
  }
  static bool attr_done = false;   // (one process per GPU: the attribute is per function and device)
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_a4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.split_k;
  const int cap = (a.flags > 0 && a.flags < n_cu) ? a.flags : n_cu;
  const int grid = (int)(tiles < cap ? tiles : cap);
  hipLaunchKernelGGL(gemm_tn_a4_kernel, dim3(grid), dim3(256), LDS_BYTES, st, a);
  return stonk_launch_status();
}
