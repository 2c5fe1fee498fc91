// bf16 MFMA GEMM, 256 x (256 | 192) x 64 tiles, FOUR waves per workgroup (one per SIMD, 128 x 128 | 96 wave tiles),
// v_mfma_f32_16x16x32_bf16, operands by LDS-DMA (buffer_load ... lds), and a K loop whose instruction stream is
// WRITTEN OUT: tools/gen_gemm_a4.py generates gemm_a4_loop.inc, one inline-assembly block per output tile.
// Same contract as gemm_bf16.hip (C = epilogue(A[M,K] . B[N,K]^T)): bf16 output with the fused epilogues (alpha = 1, no K
// split), and the two forms the label-sparse decoders launch - fp16 output of the plain product (the logits) and
// C (fp32) += alpha * product over a SPLIT contraction through float atomics (d hidden = d logits . W, K = the vocabulary).
//
// Why a fourth NT kernel: gemm_w4.hip has this geometry in compiled C++ and keeps the matrix pipe
// 54-64 % busy - with one wave per SIMD every instruction that is not in an MFMA's shadow delays the next MFMA, and the
// compiler's scheduler, register staging (global -> VGPR -> ds_write) and 32x32x16 MFMAs (a lower clock under load than
// 16x16x32 at equal cycles: the guide's DVFS note) are what separated it from the vendor library's 93 %. Here:
//  * accumulators live in a[0:255] (the AGPR half of the unified file), fragments in v[128:255], the compiler keeps
//    v[0:127] for everything else - nothing is ever spilled inside the loop because the compiler does not own the loop;
//  * operands go global -> LDS directly (no staging registers, no ds_write), two K tiles ahead, and across output tiles:
//    the last two K tiles of a tile request the first two of the workgroup's NEXT tile, so the epilogue runs with the
//    next K loop's operands already on their way;
//  * every wait of the loop is "all but this K tile's own pieces" (s_waitcnt vmcnt(pieces)), so whatever else is older
//    in the queue - the previous tile's output stores, side-operand loads - can only strengthen a wait;
//  * the epilogue needs no LDS: a v_permlane16_swap per register pair turns the 16x16 accumulator layout (a lane = one
//    row, 4 consecutive columns per block) into 8 consecutive columns per lane, i.e. 16-byte stores and 16-byte side
//    operand loads (residual, saved GELU') at the same addresses, requested a row block ahead.
#include "gemm_common.h"
#ifndef STONK_A4_LOOP_INC
#define STONK_A4_LOOP_INC "gemm_a4_loop.inc"   // (tools/a4_sweep.py builds schedule variants against other files)
#endif
#include STONK_A4_LOOP_INC

using namespace stonk_gemm;

namespace {

constexpr int BM = 256, BK = 64;
constexpr int IMG_BYTES = 256 * BK * 2;       // one operand image of one K tile: 32 KiB
constexpr int LDS_BYTES = 4 * IMG_BYTES;      // two stages x (A image + B image) = 128 KiB

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Work {
  int m0, n0;
  int k0, nk;   // the item's K tiles: [k0, k0 + nk), nk even
};

// Exact (erf) GELU and its derivative on PAIRS of values: gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x), with
// erfc(|x| / sqrt 2) by Abramowitz & Stegun 7.1.26 as in common.h (|error| <= 1.5e-7) - the same formula, arranged so that
// everything but the reciprocal, the exponential and the sign select is packed fp32 arithmetic (v_pk_fma_f32 /
// v_pk_mul_f32: two values per issue slot): ~25 issue slots per pair where the scalar form takes ~23 per VALUE. The FFN-up
// epilogue evaluates this on every output element and was as long as its K loop.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
// v[8] -> gelu(v) in place; WITH_GRAD: u[8] = gelu'(v). The four pairs advance stage by stage (a dependent packed
// instruction right behind its producer costs a wait state; four independent ones between them cost none).
template <bool WITH_GRAD>
__device__ __forceinline__ void gelu8(float (&v)[8], float (&u)[8]) {
  const f32x2_ c1 = {1.061405429f, 1.061405429f}, c2 = {-1.453152027f, -1.453152027f}, c3 = {1.421413741f, 1.421413741f},
               c4 = {-0.284496736f, -0.284496736f}, c5 = {0.254829592f, 0.254829592f}, half = {0.5f, 0.5f};
  f32x2_ x[4], t[4], pl[4], a[4], e[4], h[4], cdf[4];
#define STONK_A4_STAGE(...)                                                                                    \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) { __VA_ARGS__; }                                                 \
  __builtin_amdgcn_sched_barrier(0);   /* keeps the stages apart: the scheduler would re-serialise the pairs */
  STONK_A4_STAGE(x[k] = ((f32x2_){v[2 * k], v[2 * k + 1]});
                 t[k] = ((f32x2_){fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x[k][0]), 1.0f),
                                  fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x[k][1]), 1.0f)});
                 a[k] = x[k] * (f32x2_){-0.72134752044448170368f, -0.72134752044448170368f})   // -x^2 / 2 in the exp2 domain ...
  STONK_A4_STAGE(t[k] = ((f32x2_){__builtin_amdgcn_rcpf(t[k][0]), __builtin_amdgcn_rcpf(t[k][1])}); a[k] = a[k] * x[k])
  STONK_A4_STAGE(e[k] = ((f32x2_){__builtin_amdgcn_exp2f(a[k][0]), __builtin_amdgcn_exp2f(a[k][1])});   // exp(-x^2 / 2)
                 pl[k] = __builtin_elementwise_fma(c1, t[k], c2))
  STONK_A4_STAGE(pl[k] = __builtin_elementwise_fma(pl[k], t[k], c3))
  STONK_A4_STAGE(pl[k] = __builtin_elementwise_fma(pl[k], t[k], c4))
  STONK_A4_STAGE(pl[k] = __builtin_elementwise_fma(pl[k], t[k], c5))
  STONK_A4_STAGE(pl[k] = pl[k] * t[k])
  STONK_A4_STAGE(pl[k] = pl[k] * e[k])                                       // erfc(|x| / sqrt 2) = 2 Phi(-|x|)
  STONK_A4_STAGE(h[k] = __builtin_elementwise_fma(pl[k], -half, half))       // 1/2 - Phi(-|x|) >= 0
  // Phi(x) = 1/2 + sign(x) (1/2 - Phi(-|x|)): a v_bfi per value - a compare + select would go through an SGPR pair, two
  // wait states each
  STONK_A4_STAGE(cdf[k] = ((f32x2_){__builtin_copysignf(h[k][0], x[k][0]), __builtin_copysignf(h[k][1], x[k][1])}) + half)
  if (WITH_GRAD) {
    STONK_A4_STAGE(a[k] = x[k] * (f32x2_){0.39894228040143267794f, 0.39894228040143267794f})
    STONK_A4_STAGE(const f32x2_ dg = __builtin_elementwise_fma(a[k], e[k], cdf[k]); u[2 * k] = dg[0]; u[2 * k + 1] = dg[1])
  }
  STONK_A4_STAGE(const f32x2_ g = x[k] * cdf[k]; v[2 * k] = g[0]; v[2 * k + 1] = g[1])
#undef STONK_A4_STAGE
}

// eight fp32 -> eight IEEE fp16, saturated at +-65504 as to_f16_sat (common.h) does
__device__ __forceinline__ u32x4 pack8h(const float* e) {
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  union {
    u32x4 v;
    f16x2_ h[4];
  } u;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    u.h[j] = __builtin_convertvector((f32x2_){fminf(fmaxf(e[2 * j], -65504.f), 65504.f), fminf(fmaxf(e[2 * j + 1], -65504.f), 65504.f)}, f16x2_);
  return u.v;
}

// eight fp32 -> eight bf16 by four v_cvt_pk_bf16_f32 (element-wise casts into a vector cost a conversion AND a v_perm each)
__device__ __forceinline__ bf16x8 pack8(const float* e) {
  typedef bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  union {
    bf16x8 v;
    bf16x2_ h[4];
  } u;
#pragma unroll
  for (int j = 0; j < 4; ++j) u.h[j] = __builtin_convertvector((f32x2_){e[2 * j], e[2 * j + 1]}, bf16x2_);
  return u.v;
}

template <int EPI, int BN_>
__global__ __launch_bounds__(256, 1) void gemm_a4_kernel(const GemmArgs p) {
  static_assert(BN_ == 256 || BN_ == 192, "tile widths: 256 or 192");
  constexpr int BN = BN_;
  constexpr int NBJ = BN_ / 32;       // 16-column blocks per wave
  constexpr int NPB = BN_ / 32;       // 8-row pieces of the B image per wave and K tile
  constexpr int BROWS_W = BN_ / 4;    // rows of the B image each wave stages
  constexpr int NACC = 2 * NBJ;       // accumulator registers in units of 16
  // The two decoder forms are compile-time variants: in the bf16 instances a work item is a whole tile and the output
  // descriptor is loop-invariant, exactly as before they existed - a handful of further scalars alive across the K loop
  // was enough to push every 256-wide instance into scratch (29-78 registers spilled, FFN-up 144 -> 175 us in the step).
  constexpr int OUT = EPI & STONK_EPI_OUT_MASK;
  constexpr bool SPLIT = OUT == STONK_EPI_OUT_F32_ATOMIC;   // work item = (tile, K share)
  constexpr bool REBASE = OUT != STONK_EPI_OUT_BF16;        // output addressed from the tile's first row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  int M = p.M;
  if (p.m_dev) {
    const int md = *p.m_dev;
    M = md < M ? md : M;
  }
  const int N = p.N;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  // K tiles per work item: all of them, or (split_k > 1: the atomic output) an even share - the last item of a tile takes
  // what is left (even as well: K / 64 is)
  // split_k is an upper bound: with a device-side row count the number of tiles is known here only, and more items than
  // workgroups would run as a second, mostly empty round (20 tiles x 13 shares on 256 CUs: 12 shares fill one round)
  const int nk_all = p.K / BK;        // even, >= 2 (launcher)
  const int tiles_mn = ntm * ntn;
  int sk = SPLIT ? p.split_k : 1;
  if (SPLIT && sk > 1 && tiles_mn > 0) {
    const int fit = gridDim.x / tiles_mn;
    sk = sk < fit ? sk : (fit > 1 ? fit : 1);
  }
  const int per = (SPLIT && sk > 1) ? ((nk_all + 2 * sk - 1) / (2 * sk)) * 2 : nk_all;
  const int nsp = SPLIT ? (nk_all + per - 1) / per : 1;
  const int total = tiles_mn * nsp;   // (the tiles of one K share are neighbours: they share operand panels in an XCD's L2)
  const int G = gridDim.x;

  // work item -> tile; the items of one round that the workgroups of one XCD take are neighbours (as gemm_w4.hip)
  auto get_work = [&](int w, Work& o) -> bool {
    if (w >= total) return false;
    int idx = w;
    if ((G & 7) == 0) {
      const int r = w / G, b = w - r * G;
      const int n_r = total - r * G < G ? total - r * G : G;
      const int x = b & 7, s = b >> 3, q = n_r >> 3, rem = n_r & 7;
      if (s >= q + (x < rem ? 1 : 0)) return false;
      idx = r * G + x * q + (x < rem ? x : rem) + s;
    }
    o.k0 = 0;
    o.nk = nk_all;
    if (SPLIT && nsp > 1) {
      const int sp = idx / tiles_mn;
      idx -= sp * tiles_mn;
      o.k0 = sp * per;
      o.nk = nk_all - o.k0 < per ? nk_all - o.k0 : per;
    }
    int rt, ct;
    if (ntm >= ntn) {
      rt = idx / ntn;
      ct = idx - rt * ntn;
    } else {
      ct = idx / ntm;
      rt = idx - ct * ntm;
    }
    o.m0 = rt * BM;
    o.n0 = ct * BN;
    return true;
  };

  // ---- LDS-DMA sources. Piece q of this wave = image rows 8 q .. 8 q + 7 of its 64 (A) / BROWS_W (B) rows; a lane moves
  // 16 bytes: image row l8 = lane >> 3, PHYSICAL chunk lane & 7, i.e. logical chunk (lane & 7) ^ ((row >> 1) & 7) - the
  // LDS image is lane-linear (the DMA's destination is M0 + 16 lane), the swizzle is on the source. One per-lane byte
  // offset per piece; the buffer's base is the tile's first row at the K tile's first column and its extent what is left
  // of the operand from there, so rows past the operand's end are out of range and arrive as zeros (the range check
  // covers the per-lane offset only - nothing rides in the scalar offset).
  const int l8 = lane >> 3;
  const int lda2 = (int)p.lda * 2, ldb2 = (int)p.ldb * 2;
  int voffA[8], voffB[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#ifdef STONK_A4_LINEAR_SRC   // timing experiment (tools/a4_sweep.py): unswizzled source, garbage results
    const int c = (lane & 7) << 4;
#else
    const int c = (((lane & 7) ^ (l8 >> 1) ^ (4 * (q & 1))) << 4);
#endif
    voffA[q] = (wave * 64 + 8 * q + l8) * lda2 + c;
    voffB[q] = (wave * BROWS_W + 8 * (q < NPB ? q : 0) + l8) * ldb2 + c;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int m0a = (int)lds0 + wave * 8192;
  const int m0b = (int)lds0 + 2 * IMG_BYTES + wave * (BROWS_W * 128);
  // ---- fragment reads: lane (r = lane & 15, qq = lane >> 4) takes row r of a 16-row block, k = 32 h + 8 qq .. +7
  const int r16 = lane & 15, qq = lane >> 4;
  int raA[2], raB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int o = r16 * 128 + ((((4 * h + qq) ^ (r16 >> 1)) & 7) << 4);
    raA[h] = (int)lds0 + wr * 16384 + o;
    raB[h] = (int)lds0 + 2 * IMG_BYTES + wc * (BN_ / 2) * 128 + o;
  }

  // (extents: the tile's own rows only - 256 x ld bytes always fits 32 bits, M x ld need not)
  auto cursor_of = [&](const Work& w, i32x4& a, i32x4& b) {
    const int kb = SPLIT ? w.k0 * (BK * 2) : 0;
    const unsigned long pa = (unsigned long)p.A + (unsigned long)((long)w.m0 * lda2) + (unsigned long)kb;
    const unsigned long pb = (unsigned long)p.B + (unsigned long)((long)w.n0 * ldb2) + (unsigned long)kb;
    if (REBASE) {
      const int ra = M - w.m0 < BM ? M - w.m0 : BM, rb = N - w.n0 < BN ? N - w.n0 : BN;
      a = (i32x4){(int)(unsigned)pa, (int)((pa >> 32) & 0xffff), ra * lda2 - kb, 0x00020000};
      b = (i32x4){(int)(unsigned)pb, (int)((pb >> 32) & 0xffff), rb * ldb2 - kb, 0x00020000};
    } else {   // (bf16 launches: M x ld fits 32 bits - the launcher's condition)
      a = (i32x4){(int)(unsigned)pa, (int)((pa >> 32) & 0xffff), (M - w.m0) * lda2, 0x00020000};
      b = (i32x4){(int)(unsigned)pb, (int)((pb >> 32) & 0xffff), (N - w.n0) * ldb2, 0x00020000};
    }
  };

  Work cw, nw;
  int cwi = blockIdx.x;
  if (!get_work(cwi, cw)) return;     // uniform: whole workgroup leaves together
  i32x4 curA, curB;                   // the operand cursors (s[36:39], s[40:43] inside the blocks)
  cursor_of(cw, curA, curB);

#define STONK_A4_VOFF_OPERANDS                                                                                          \
  [voffA0] "v"(voffA[0]), [voffA1] "v"(voffA[1]), [voffA2] "v"(voffA[2]), [voffA3] "v"(voffA[3]), [voffA4] "v"(voffA[4]), \
      [voffA5] "v"(voffA[5]), [voffA6] "v"(voffA[6]), [voffA7] "v"(voffA[7]), [voffB0] "v"(voffB[0]),                   \
      [voffB1] "v"(voffB[1]), [voffB2] "v"(voffB[2]), [voffB3] "v"(voffB[3]), [voffB4] "v"(voffB[4]),                   \
      [voffB5] "v"(voffB[5]), [voffB6] "v"(voffB[6]), [voffB7] "v"(voffB[7]), [m0a] "s"(m0a), [m0b] "s"(m0b)

  // K tiles 0 and 1 of the first tile
  if (BN_ == 256)
    asm volatile(STONK_A4_PROLOGUE_256 : "+{s[36:39]}"(curA), "+{s[40:43]}"(curB) : STONK_A4_VOFF_OPERANDS : "m0", "scc", "memory");
  else
    asm volatile(STONK_A4_PROLOGUE_192 : "+{s[36:39]}"(curA), "+{s[40:43]}"(curB) : STONK_A4_VOFF_OPERANDS : "m0", "scc", "memory");

  static_assert(OUT == STONK_EPI_OUT_BF16 || ((OUT == STONK_EPI_OUT_F16 || OUT == STONK_EPI_OUT_F32_ATOMIC) && (EPI & ~STONK_EPI_OUT_MASK) == 0),
                "fp16 / atomic fp32 output: the plain product");
  constexpr int CB = OUT == STONK_EPI_OUT_F32_ATOMIC ? 4 : 2;   // bytes per output element
  const int flags = EPI & ~STONK_EPI_OUT_MASK;
  const int ldc_b = (int)p.ldc * CB, ldr_b = (int)p.ldr * 2, ldx_b = (int)p.ldaux * 2;
  const __amdgpu_buffer_rsrc_t rC0 = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, REBASE ? 0 : M * ldc_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, M * ldr_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.aux, 0, M * ldx_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, N * 4, 0x00020000);
  constexpr bool SIDE_X = (EPI & STONK_EPI_GELU_BWD) != 0;
  constexpr bool SIDE_R = (EPI & STONK_EPI_RESID) != 0;
  static_assert(!(SIDE_X && SIDE_R), "one side operand per launch");
  constexpr bool SIDE = SIDE_X || SIDE_R;
  constexpr int NJP = NBJ / 2;        // pairs of column blocks = 16-byte pieces per lane and row block
  // a lane's 8 output columns of pair jp: the pair's first block for even qq, its second for odd qq; halves by qq >> 1
  const int ncol = wc * (BN_ / 2) + 16 * (qq & 1) + 8 * (qq >> 1);

  for (;;) {
    const bool more = get_work(cwi + G, nw);
    i32x4 nxA, nxB;
    cursor_of(more ? nw : cw, nxA, nxB);   // (no next tile: the cursor re-reads this tile's first K tiles, never consumed)
    f32x16 acc[16];
    int rem = (SPLIT ? cw.nk : nk_all) >> 1;
    // the bias of this lane's columns: requested before the K loop, used after it
    f32x4 bq[NJP][2];
    if (flags & STONK_EPI_BIAS) {
#pragma unroll
      for (int jp = 0; jp < NJP; ++jp)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          bq[jp][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBias, (cw.n0 + ncol + 32 * jp + 4 * h) * 4, 0, 0));
    }
#define STONK_A4_TILE_OPERANDS                                                                                        \
  STONK_A4_VOFF_OPERANDS, [raA0] "v"(raA[0]), [raA1] "v"(raA[1]), [raB0] "v"(raB[0]), [raB1] "v"(raB[1]),             \
      [nal] "s"(nxA[0]), [nah] "s"(nxA[1]), [nan] "s"(nxA[2]), [nbl] "s"(nxB[0]), [nbh] "s"(nxB[1]), [nbn] "s"(nxB[2])
#define STONK_A4_CLOBBERS                                                                                              \
  "m0", "scc", "memory", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", \
      "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153",  \
      "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167",  \
      "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181",  \
      "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195",  \
      "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",  \
      "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",  \
      "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237",  \
      "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251",  \
      "v252", "v253", "v254", "v255"
    if (BN_ == 256) {
      asm volatile(STONK_A4_TILE_256
                   : "={a[0:15]}"(acc[0]), "={a[16:31]}"(acc[1]), "={a[32:47]}"(acc[2]), "={a[48:63]}"(acc[3]),
                     "={a[64:79]}"(acc[4]), "={a[80:95]}"(acc[5]), "={a[96:111]}"(acc[6]), "={a[112:127]}"(acc[7]),
                     "={a[128:143]}"(acc[8]), "={a[144:159]}"(acc[9]), "={a[160:175]}"(acc[10]), "={a[176:191]}"(acc[11]),
                     "={a[192:207]}"(acc[12]), "={a[208:223]}"(acc[13]), "={a[224:239]}"(acc[14]), "={a[240:255]}"(acc[15]),
                     "+{s[36:39]}"(curA), "+{s[40:43]}"(curB), [rem] "+s"(rem)
                   : STONK_A4_TILE_OPERANDS
                   : STONK_A4_CLOBBERS);
    } else {
      asm volatile(STONK_A4_TILE_192
                   : "={a[0:15]}"(acc[0]), "={a[16:31]}"(acc[1]), "={a[32:47]}"(acc[2]), "={a[48:63]}"(acc[3]),
                     "={a[64:79]}"(acc[4]), "={a[80:95]}"(acc[5]), "={a[96:111]}"(acc[6]), "={a[112:127]}"(acc[7]),
                     "={a[128:143]}"(acc[8]), "={a[144:159]}"(acc[9]), "={a[160:175]}"(acc[10]), "={a[176:191]}"(acc[11]),
                     "+{s[36:39]}"(curA), "+{s[40:43]}"(curB), [rem] "+s"(rem)
                   : STONK_A4_TILE_OPERANDS
                   : STONK_A4_CLOBBERS);
    }

    // (the accumulators stay where they are - AGPR-class values - until the epilogue reads them one by one: copied out
    // wholesale into VGPRs after the block they would push everything that lives across the K loop into scratch)
#pragma unroll
    for (int g = 0; g < NACC; ++g) asm volatile("" : "+a"(acc[g]));
    // ---- epilogue: row block i (16 rows: a lane's row is r16), column-block pair jp -> 8 consecutive columns per lane
    const int wm0 = cw.m0 + wr * 128, wn0 = cw.n0 + ncol;
    // the output through a buffer that starts at the tile's first row and ends with its last one (M x ldc bytes need not
    // fit 32 bits: 16 384 rows of 175 104 logits)
    const int crows = M - cw.m0 < BM ? M - cw.m0 : BM;
    const __amdgpu_buffer_rsrc_t rC =
        REBASE ? __builtin_amdgcn_make_buffer_rsrc((char*)p.C + (long)cw.m0 * ldc_b, 0, crows * ldc_b, 0x00020000) : rC0;
    const int cm0 = REBASE ? cw.m0 : 0;   // row the output offsets count from
    bf16x8 sd[NJP];
    auto side_request = [&](const int i) {
      if (!SIDE) return;
      const int m = wm0 + 16 * i + r16;
#pragma unroll
      for (int jp = 0; jp < NJP; ++jp) {
        const int n = wn0 + 32 * jp;
        const int oob = n < N ? 0 : 0x40000000;
        sd[jp] = __builtin_bit_cast(bf16x8, SIDE_X ? __builtin_amdgcn_raw_buffer_load_b128(rX, m * ldx_b + n * 2 + oob, 0, 0)
                                                  : __builtin_amdgcn_raw_buffer_load_b128(rR, m * ldr_b + n * 2 + oob, 0, 0));
      }
    };
    side_request(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = wm0 + 16 * i + r16;
      bf16x8 cur[NJP];
#pragma unroll
      for (int jp = 0; jp < NJP; ++jp) cur[jp] = sd[jp];
      if (i + 1 < 8) side_request(i + 1);   // flies while this row block is processed
      // Accumulator blocks (i, 2 jp) = x and (i, 2 jp + 1) = y of every pair jp. v_permlane16_swap exchanges x's odd
      // rows of 16 lanes with y's even rows: afterwards (x, y) of a lane are columns 0-3 / 4-7 of its 8 consecutive output
      // columns. One block per row block (inline assembly: this toolchain's __builtin_amdgcn_permlane16_swap returns its
      // FIRST result twice; the no-ops stand in for the VALU <-> permlane wait states the compiler would have placed).
      float xs[NJP][4], ys[NJP][4];
#pragma unroll
      for (int jp = 0; jp < NJP; ++jp) {
        const int bx = NBJ * i + 2 * jp, by = bx + 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xs[jp][e] = acc[bx >> 2][4 * (bx & 3) + e];
          ys[jp][e] = acc[by >> 2][4 * (by & 3) + e];
        }
      }
#define STONK_A4_SWAP4(a, b) \
  "v_permlane16_swap_b32 %" #a ", %" #b "\n"
      if (NJP == 4) {
        asm volatile("s_nop 1\n" STONK_A4_SWAP4(0, 16) STONK_A4_SWAP4(1, 17) STONK_A4_SWAP4(2, 18) STONK_A4_SWAP4(3, 19)
                     STONK_A4_SWAP4(4, 20) STONK_A4_SWAP4(5, 21) STONK_A4_SWAP4(6, 22) STONK_A4_SWAP4(7, 23)
                     STONK_A4_SWAP4(8, 24) STONK_A4_SWAP4(9, 25) STONK_A4_SWAP4(10, 26) STONK_A4_SWAP4(11, 27)
                     STONK_A4_SWAP4(12, 28) STONK_A4_SWAP4(13, 29) STONK_A4_SWAP4(14, 30) STONK_A4_SWAP4(15, 31) "s_nop 1"
                     : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[0][2]), "+v"(xs[0][3]), "+v"(xs[1][0]), "+v"(xs[1][1]),
                       "+v"(xs[1][2]), "+v"(xs[1][3]), "+v"(xs[2][0]), "+v"(xs[2][1]), "+v"(xs[2][2]), "+v"(xs[2][3]),
                       "+v"(xs[NJP - 1][0]), "+v"(xs[NJP - 1][1]), "+v"(xs[NJP - 1][2]), "+v"(xs[NJP - 1][3]),
                       "+v"(ys[0][0]), "+v"(ys[0][1]), "+v"(ys[0][2]), "+v"(ys[0][3]), "+v"(ys[1][0]), "+v"(ys[1][1]),
                       "+v"(ys[1][2]), "+v"(ys[1][3]), "+v"(ys[2][0]), "+v"(ys[2][1]), "+v"(ys[2][2]), "+v"(ys[2][3]),
                       "+v"(ys[NJP - 1][0]), "+v"(ys[NJP - 1][1]), "+v"(ys[NJP - 1][2]), "+v"(ys[NJP - 1][3]));
      } else {
        asm volatile("s_nop 1\n" STONK_A4_SWAP4(0, 12) STONK_A4_SWAP4(1, 13) STONK_A4_SWAP4(2, 14) STONK_A4_SWAP4(3, 15)
                     STONK_A4_SWAP4(4, 16) STONK_A4_SWAP4(5, 17) STONK_A4_SWAP4(6, 18) STONK_A4_SWAP4(7, 19)
                     STONK_A4_SWAP4(8, 20) STONK_A4_SWAP4(9, 21) STONK_A4_SWAP4(10, 22) STONK_A4_SWAP4(11, 23) "s_nop 1"
                     : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[0][2]), "+v"(xs[0][3]), "+v"(xs[1][0]), "+v"(xs[1][1]),
                       "+v"(xs[1][2]), "+v"(xs[1][3]), "+v"(xs[2][0]), "+v"(xs[2][1]), "+v"(xs[2][2]), "+v"(xs[2][3]),
                       "+v"(ys[0][0]), "+v"(ys[0][1]), "+v"(ys[0][2]), "+v"(ys[0][3]), "+v"(ys[1][0]), "+v"(ys[1][1]),
                       "+v"(ys[1][2]), "+v"(ys[1][3]), "+v"(ys[2][0]), "+v"(ys[2][1]), "+v"(ys[2][2]), "+v"(ys[2][3]));
      }
#undef STONK_A4_SWAP4
#pragma unroll
      for (int jp = 0; jp < NJP; ++jp) {
        float v[8] = {xs[jp][0], xs[jp][1], xs[jp][2], xs[jp][3], ys[jp][0], ys[jp][1], ys[jp][2], ys[jp][3]};
        const int n = wn0 + 32 * jp;
        const int oob = n < N ? 0 : 0x40000000;   // columns past N: pushed out of the buffers' range
        if (flags & STONK_EPI_BIAS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += bq[jp][0][e];
            v[4 + e] += bq[jp][1][e];
          }
        }
        int rest = flags & ~STONK_EPI_BIAS;
        if (flags & STONK_EPI_GELU) {   // (with or without the saved pre-activation / saved GELU')
          constexpr bool GRAD = (EPI & STONK_EPI_SAVE_PREACT) && (EPI & STONK_EPI_AUX_GRAD);
          float u[8];
          if (!GRAD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = v[e];   // (the plain form saves the pre-activation itself)
          }
          gelu8<GRAD>(v, u);
          if (flags & STONK_EPI_SAVE_PREACT)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8(u)), rX, m * ldx_b + n * 2 + oob, 0, 0);
          rest &= ~(STONK_EPI_GELU | STONK_EPI_SAVE_PREACT);
        }
        SideOps so;
        so.aux = so.res = cur[jp];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        epilogue8_pre(v, p, rest & ~STONK_EPI_SAVE_PREACT, m, n, z4, z4, so);
        const int co = (m - cm0) * ldc_b + n * CB + oob;
        if (OUT == STONK_EPI_OUT_F32_ATOMIC) {
#pragma unroll
          for (int e = 0; e < 8; ++e) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v[e] * p.alpha, rC, co + 4 * e, 0, 0);
        } else if (OUT == STONK_EPI_OUT_F16) {
          __builtin_amdgcn_raw_buffer_store_b128(pack8h(v), rC, co, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8(v)), rC, co, 0, 0);
        }
      }
    }
    if (!more) break;
    cwi += G;
    cw = nw;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the cursor's last, unused pieces
#undef STONK_A4_TILE_OPERANDS
#undef STONK_A4_VOFF_OPERANDS
#undef STONK_A4_CLOBBERS
}

template <int EPI, int BN_>
int launch_a4(const GemmArgs& a, int grid, hipStream_t st) {
  static bool attr_done = false;   // (one process per GPU: the attribute is per function and device)
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_a4_kernel<EPI, BN_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_a4_kernel<EPI, BN_>), dim3(grid), dim3(256), LDS_BYTES, st, a);
  return stonk_launch_status();
}

}  // namespace

// Launcher used by stonk_gemm_nt_bf16 (gemm_bf16.hip). Requires K % 128 == 0, ld % 64 == 0, 256 rows of every operand within
// 2^30 bytes, 16-byte aligned side operands; bf16 output: alpha == 1 and split_k == 1; fp16 output: the same, no epilogue;
// atomic fp32 output: no epilogue, alpha as given, split_k an upper bound (the kernel takes as many shares as fill the grid
// once; a share is rounded up to an even number of K tiles).
// tile_n: 0 = choose, 256, 192. items_per_wg as gemm_w4.hip.
// Returns STONK_ESHAPE for an epilogue this kernel has no instance of (the caller then takes another kernel).
int stonk_gemm_a4_launch(const GemmArgs& a, int tile_n, int items_per_wg, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    n_cu = prop.multiProcessorCount;
  }
  constexpr int B = STONK_EPI_BIAS, G = STONK_EPI_GELU, SV = STONK_EPI_SAVE_PREACT, GB = STONK_EPI_GELU_BWD,
                R = STONK_EPI_RESID, D = STONK_EPI_DROPOUT, AG = STONK_EPI_AUX_GRAD;
  const int epi = a.flags & (B | G | SV | GB | R | D | AG);
  const int out = a.flags & STONK_EPI_OUT_MASK;
  if (out != STONK_EPI_OUT_BF16 && (epi != 0 || out == STONK_EPI_OUT_F32)) return STONK_ESHAPE;
  const long ntm = (a.M + BM - 1) / BM;
  const bool has192 = a.N % 192 == 0 && out != STONK_EPI_OUT_F16 &&
                      (epi == 0 || epi == B || epi == R || epi == (B | R) || epi == (B | R | D));
  if (tile_n == 192 && !has192) return STONK_ESHAPE;
  const int nk_all = a.K / BK;
  const int per = a.split_k > 1 ? ((nk_all + 2 * a.split_k - 1) / (2 * a.split_k)) * 2 : nk_all;
  const long nsp = (nk_all + per - 1) / per;   // (as the kernel counts them)
  if (tile_n == 0) {
    const long t256 = ntm * ((a.N + 255) / 256) * nsp, t192 = ntm * (a.N / 192) * nsp;
    const long c256 = ((t256 + n_cu - 1) / n_cu) * 256, c192 = ((t192 + n_cu - 1) / n_cu) * 192;
    tile_n = (has192 && c192 < c256) ? 192 : 256;
  }
  const long tiles = (tile_n == 192 ? ntm * (a.N / 192) : ntm * ((a.N + 255) / 256)) * nsp;
  const int grid = (int)(items_per_wg > 0 ? (tiles + items_per_wg - 1) / items_per_wg : (tiles < n_cu ? tiles : n_cu));
  if (out == STONK_EPI_OUT_F16) return launch_a4<STONK_EPI_OUT_F16, 256>(a, grid, st);
  // (the atomic form on 256 x 192 tiles only: its 256-wide instance does not fit the compiler's half of the register file)
  if (out == STONK_EPI_OUT_F32_ATOMIC) return tile_n == 192 ? launch_a4<STONK_EPI_OUT_F32_ATOMIC, 192>(a, grid, st) : STONK_ESHAPE;
  if (tile_n == 192) {
    switch (epi) {
      case 0: return launch_a4<0, 192>(a, grid, st);
      case B: return launch_a4<B, 192>(a, grid, st);
      case R: return launch_a4<R, 192>(a, grid, st);
      case B | R: return launch_a4<B | R, 192>(a, grid, st);
      default: return launch_a4<B | R | D, 192>(a, grid, st);
    }
  }
  switch (epi) {
    case 0: return launch_a4<0, 256>(a, grid, st);
    case B: return launch_a4<B, 256>(a, grid, st);
    case B | G: return launch_a4<B | G, 256>(a, grid, st);
    case B | G | SV: return launch_a4<B | G | SV, 256>(a, grid, st);
    case GB: return launch_a4<GB, 256>(a, grid, st);
    case B | G | SV | AG: return launch_a4<B | G | SV | AG, 256>(a, grid, st);
    case GB | AG: return launch_a4<GB | AG, 256>(a, grid, st);
    case R: return launch_a4<R, 256>(a, grid, st);
    case B | R: return launch_a4<B | R, 256>(a, grid, st);
    case B | R | D: return launch_a4<B | R | D, 256>(a, grid, st);
    default: return STONK_ESHAPE;
  }
}
