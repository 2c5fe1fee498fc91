// bf16 MFMA GEMM, 256x256x64 tiles, FOUR waves per workgroup, one wave per SIMD: each wave owns a 128x128 corner of
// the tile in 256 accumulator registers (16 blocks of v_mfma_f32_32x32x16_bf16). Same contract as gemm_bf16.hip
// (C = epilogue(alpha * A[M,K] . B[N,K]^T)).
//
// Why a third tiling: in the eight-wave kernel (gemm256.hip, wave tile 128x64) every K tile costs the CU 192 KiB of
// fragment reads + 64 KiB of LDS writes = 2048 LDS cycles at 128 B/clk - exactly the 2048 MFMA cycles of the K tile, so
// the matrix pipe can never be much more than half busy. A 128x128 wave tile reads 128 KiB per K tile (75 % of the LDS
// budget with the writes), which is what the vendor library's 256x256 kernels do as well.
//
// Operands travel global -> registers -> LDS -> fragment registers. With one wave per SIMD every instruction the wave
// issues competes with its own MFMAs, and an LDS-DMA piece (global_load_lds_dwordx4) costs the issuing wave 60+ cycles
// - measured on the first version of this kernel as 240 us of a 1017 us 8192^3 run (timing builds with the VAR template parameter) - where a
// plain 16-byte load and a ds_write_b128 cost a few cycles each.
//
// Schedule. The two K-tile stages (A image 256 rows x 128 B + B image, XOR-swizzled, 64 KiB each) alternate; a K tile
// is consumed in four k steps of 16 MFMAs (4 x 4 blocks, 16 different accumulators: no MFMA waits for its predecessor).
// k step (t, ks)
//   * multiplies with the fragments read during the previous k step (two fragment sets of 8 x ds_read_b128 alternate:
//     only 64 fragment registers are live, against 128 for a quadrant-by-quadrant order),
//   * reads the fragments of the next k step - for ks = 3 those of K tile t+1, from the other stage,
//   * writes four 1 KiB pieces (8 rows x 128 B) of K tile t+1 (ks = 3: t+2) from staging registers to the other stage,
//     behind ONE counted wait, vmcnt(12): the loads of the three younger k steps stay in flight (about 2000 cycles),
//   * loads the four pieces that the same k step of the NEXT K tile will write into those staging registers.
// One s_barrier per K tile, after k step 2: K tile t+1 is then complete in its stage (k step 3 may read it) and every
// wave is done reading K tile t-1's... i.e. the stage that k step 3 starts to refill.
// After a tile boundary the wait counts the epilogue's stores on top for four k steps (vmcnt retires in order), so the
// load stream never drains. The prefetch cursor runs across output tiles; when it runs out of work it re-reads its
// last position (never consumed), so the counts stay constant to the end.
#include <cstdlib>

#include "gemm_common.h"

using namespace stonk_gemm;

namespace {

constexpr int BM = 256, BK = 64;   // the tile is BM x BN_ with BN_ = 256 or 192 (template parameter)
constexpr int IMG_BYTES = 256 * BK * 2;             // one operand of one K tile: 32 KiB
constexpr int STAGE_BYTES = 2 * IMG_BYTES;          // 64 KiB
constexpr int LDS_RING = 2 * STAGE_BYTES;           // 128 KiB
constexpr int SLAB_BYTES = 32 * 64 * 4;             // per wave: 32 rows x 64 fp32 columns
constexpr int LDS_BYTES = LDS_RING + 4 * SLAB_BYTES;  // 160 KiB

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Work {
  int m0, n0;       // tile origin
  long k_begin;     // element offset of the first K tile
  int nk;           // K tiles in this work item
};

__device__ __forceinline__ void barrier() { __builtin_amdgcn_s_barrier(); }
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// VAR: timing experiments only (bit 0 no barrier, bit 2 no global loads / LDS writes in the loop, bit 3 no fragment
// reads in the loop, bit 4 loads but no LDS writes, bit 5 LDS writes but no loads) - anything but 0 computes garbage.
// BN_ = 192 (128x96 wave tiles, 4 x 3 MFMA blocks): N = 768 = 4 x 192 tiles the 32 768-row outputs of the step into 512
// work items = two full rounds of the 256 CUs, where 3 x 256 gives 384 = one and a half (the second round half empty: 75 %
// tile efficiency); 16 384 rows (frozen backbone) give 256 items = one full round instead of 192. The B image shrinks to
// 192 rows (six 1-KiB pieces per wave and K tile instead of eight), everything else keeps its place.
template <int OUT_MODE, int EPI, int BN_ = 256, int VAR = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmArgs p) {
  static_assert(BN_ == 256 || BN_ == 192, "tile widths: 256 or 192");
  constexpr int BN = BN_;
  constexpr int NB = BN_ / 64;        // 32-column blocks per wave
  constexpr int BROWS_W = BN_ / 4;    // rows of the B image each wave stages
  constexpr int NPB = BROWS_W / 8;    // ... in this many 8-row pieces per K tile
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
  int nk_total = p.K / BK;
  if (p.k_dev) {
    const int kd = (*p.k_dev + BK - 1) / BK;
    nk_total = kd < nk_total ? kd : nk_total;
  }
  const int nk_per = (nk_total + p.split_k - 1) / p.split_k;
  const int per_split = ntm * ntn;
  const int total = per_split * p.split_k;
  const int G = gridDim.x;

  // work item -> tile; items processed in the same round by the workgroups of one XCD are neighbours
  auto get_work = [&](int w, Work& o) -> bool {
    if (w >= total) return false;
    int idx = w;
    if ((G & 7) == 0) {
      // Round r hands items [r G, r G + n_r) to the workgroups so that every XCD (blockIdx & 7) takes ONE contiguous run
      // of them - also in a ragged round (n_r < G: the last one, or the only one when a device-side row count leaves fewer
      // items than the grid was sized for). The label-sparse decoder dgrad is such a launch: 30 live tiles x 8 K splits =
      // 240 items on a grid of 256; in natural order every XCD saw every K split and each L2 re-fetched the 269 MB weight
      // (2.7 GB of fabric reads per launch for 1.1 GB of operands); with the runs, XCD x works on K split x alone.
      const int r = w / G, b = w - r * G;
      const int n_r = total - r * G < G ? total - r * G : G;
      const int x = b & 7, s = b >> 3, q = n_r >> 3, rem = n_r & 7;
      if (s >= q + (x < rem ? 1 : 0)) return false;   // (only in a ragged round, which is the last)
      idx = r * G + x * q + (x < rem ? x : rem) + s;
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
    o.nk = nk < nk_per ? nk : nk_per;
    return true;
  };

  // ---- global -> staging registers. A K tile of one operand = 256 rows x 128 B = 32 pieces of 8 rows; wave w moves
  // pieces 8w..8w+7 of A and of B: 16 pieces per K tile, four per k step. Lane -> (row l8 = lane>>3, chunk lane&7 of the
  // LDS image); the image is XOR-swizzled (physical chunk = logical ^ ((row >> 1) & 7)) by permuting the SOURCE chunks,
  // so the LDS write is lane-linear. One wave per SIMD means address arithmetic competes with the MFMAs for the wave's
  // single issue stream (the first version spent 18 % of its cycles on SALU), so a piece costs none: buffer loads take
  // one of two loop-invariant per-lane byte offsets per operand ((64 wave + l8) * ld + swizzled chunk) plus a scalar
  // offset (tile origin * ld + K offset, bumped by 128 per K tile, + piece * 8 * ld: one SALU add), and rows past the
  // operand's end read as zeros by the buffer's range check - no clamping. The launcher guarantees rows * ld * 2 < 2^31 and ld % 64 == 0.
  const int l8 = lane >> 3;
  const int lda2 = (int)p.lda * 2, ldb2 = (int)p.ldb * 2;
  int voffA[2], voffB[2];   // even / odd pieces (the odd pieces' swizzled chunk is the even pieces' ^ 64)
  {
    const int c0 = ((lane & 7) ^ (l8 >> 1)) * 16;
    voffA[0] = (wave * 64 + l8) * lda2 + c0;
    voffA[1] = (wave * 64 + l8) * lda2 + (c0 ^ 64);
    voffB[0] = (wave * BROWS_W + l8) * ldb2 + c0;   // (BROWS_W % 16 == 0: the image rows' swizzle period)
    voffB[1] = (wave * BROWS_W + l8) * ldb2 + (c0 ^ 64);
  }
  const int pstepA = 8 * lda2, pstepB = 8 * ldb2;   // byte distance between consecutive pieces (uniform)
  // (An inline-asm variant of these loads, whose waits are all hand-counted, removes the compiler's conservative wait
  // after a tile boundary - pending loads are older than the epilogue's stores, and vmcnt retires in order - but the
  // compiler then spills in-flight staging registers around the epilogue: not shippable. See DESIGN.md section 4.2.)
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.M * lda2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, N * ldb2, 0x00020000);
  const int wofs = wave * 8192 + lane * 16;   // this lane's 16 bytes of piece 0 of this wave inside the A image
  const int wofsB = wave * (BROWS_W * 128) + lane * 16;   // ... inside the B image
  int soffA = 0, soffB = 0;   // scalar byte offsets of the K tile the prefetch cursor points at
  auto set_sources = [&](const Work& w) {
    soffA = w.m0 * lda2 + (int)w.k_begin * 2;
    soffB = w.n0 * ldb2 + (int)w.k_begin * 2;
  };
  // piece q of this wave: q < 8 -> A rows 64 wave + 8 q .., q >= 8 -> B rows 64 wave + 8 (q - 8) ..
  auto load_piece = [&](const int q, bf16x8& d) {
    if (q >= 8 + NPB) return;   // (BN_ = 192: the B image has six pieces per wave; q is a compile-time constant at every use)
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = (q < 8) ? __builtin_amdgcn_raw_buffer_load_b128(rsrcA, voffA[q & 1], soffA + (q & 7) * pstepA, 0)
                            : __builtin_amdgcn_raw_buffer_load_b128(rsrcB, voffB[q & 1], soffB + (q & 7) * pstepB, 0);
    d = __builtin_bit_cast(bf16x8, v);
  };
  auto write_piece = [&](const int stage, const int q, const bf16x8& d) {
    if (q >= 8 + NPB) return;
    *(bf16x8*)(smem + stage * IMG_BYTES + (q < 8 ? wofs : 2 * IMG_BYTES + wofsB) + (q & 7) * 1024) = d;
  };
  bf16x8 stg[4][4];   // [k step][piece]: loaded in k step ks of one K tile, written in k step ks of the next

  // ---- fragment reads: lane (r = lane & 31, hh = lane >> 5) takes row r of a 32-row block, k = 16 ks + 8 hh .. +7
  const int r = lane & 31, hh = lane >> 5;
  int lofsA[4], lofsB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int o = r * 128 + (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4);
    lofsA[ks] = o + wr * 16384;               // this wave's 128 rows of the A image
    lofsB[ks] = o + wc * (BN_ / 2) * 128 + 2 * IMG_BYTES;   // ... and of the B image (A images at 0 / 32 KiB, B at 64 / 96 KiB:
                                                  // either stage is within a 16-bit immediate of these two bases)
  }
  bf16x8 fr[2][8];   // [set][0-3: A row blocks, 4-7: B column blocks]
  auto read_frags = [&](const int stage, const int ks, bf16x8 (&f)[8]) {
    const char* s = smem + stage * IMG_BYTES;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f[b] = *(const bf16x8*)(s + b * 4096 + lofsA[ks]);
      if (b < NB) f[4 + b] = *(const bf16x8*)(s + b * 4096 + lofsB[ks]);
    }
  };

  f32x16 acc[4][NB];   // [32-row block][32-column block]
  f32x16 fzero;   // first k step of a work item: C = 0 as the MFMA's inline constant (no 256-register clear)
#pragma unroll
  for (int e = 0; e < 16; ++e) fzero[e] = 0.f;

  // ------------------------------------------------------------------ stream state
  Work cw, pw;
  int cwi = blockIdx.x;
  if (!get_work(cwi, cw)) return;                     // uniform: whole workgroup leaves together
  while (cw.nk <= 0) {                                // (k_dev may leave a K-split empty)
    cwi += G;
    if (!get_work(cwi, cw)) return;
  }
  int pwi = cwi;
  pw = cw;
  int pk = 0;
  bool p_valid = true;
  set_sources(pw);
  // advance the prefetch cursor by one K tile (after its last pieces have been requested)
  auto advance_prefetch = [&]() {
    if (!p_valid) return;   // out of work: the cursor keeps re-reading its last K tile (valid memory, never consumed)
    if (pk + 1 < pw.nk) {
      ++pk;
      soffA += BK * 2;
      soffB += BK * 2;
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

  // epilogue vm operations that can still be outstanding during the four k steps after a tile boundary
  constexpr int FL = EPI >= 0 ? EPI : 0;
  constexpr int S = (OUT_MODE == 0) ? ((FL & STONK_EPI_SAVE_PREACT) || EPI < 0 ? 64 : 32) : (OUT_MODE == 1 ? 64 : 0);
  // loads that stay in flight behind the wait of k step KS: the pieces requested by the three other k steps of a K tile
  // (16 pieces, four per k step, at BN_ = 256; 14 - k step 2 requests only two - at 192)
  constexpr int NLOADS = 8 + NPB;
#define STONK_W4_LOADS_OF(KS) ((4 * (((KS) + 1) & 3) + 4 <= NLOADS) ? 4 : (NLOADS - 4 * (((KS) + 1) & 3) > 0 ? NLOADS - 4 * (((KS) + 1) & 3) : 0))

  // One k step. ST = stage of the K tile being multiplied, KS = k step. Pieces written here: 4 ((KS + 1) & 3) .. +3 of
  // K tile t+1 (KS = 3: of t+2), into the stage not being multiplied... which for KS = 3 IS stage ST: safe, every wave
  // passed this K tile's barrier (after k step 2) and so finished reading ST's last fragments.
#define STONK_W4_KSTEP(ST, KS, FIRST)                                                                         \
  do {                                                                                                 \
    constexpr int WST = ((KS) == 3) ? (ST) : ((ST) ^ 1);                                               \
    constexpr int Q0 = 4 * (((KS) + 1) & 3);                                                           \
    constexpr int RST = ((KS) == 3) ? ((ST) ^ 1) : (ST);                                               \
    constexpr int RKS = ((KS) + 1) & 3;                                                                \
    constexpr int WAIT = NLOADS - STONK_W4_LOADS_OF(KS);                                               \
    constexpr int WAIT_POST = (WAIT + S) > 63 ? 63 : (WAIT + S);                                       \
    if (!(VAR & 4) && !(VAR & 16)) {                                                                   \
      if (post) wait_vm<WAIT_POST>();                                                                  \
      else wait_vm<WAIT>();                                                                            \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    /* four sub-blocks of 4 MFMAs (one accumulator row) with ONE memory instruction per MFMA gap: the next k step's \
       A and B fragment of block g, one staged piece to LDS, one load (a ds_write_b128 or a buffer load occupies the \
       wave's issue for 20-35 cycles - about what one 32-cycle MFMA hides, and no more) */                \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
      if (!(VAR & 8)) {                                                                                \
        fr[RKS & 1][g] = *(const bf16x8*)(smem + RST * IMG_BYTES + g * 4096 + lofsA[RKS]);             \
        if (g < NB) fr[RKS & 1][4 + g] = *(const bf16x8*)(smem + RST * IMG_BYTES + g * 4096 + lofsB[RKS]); \
      }                                                                                                \
      if (!(VAR & 4) && !(VAR & 16)) write_piece(WST, Q0 + g, stg[KS][g]);                             \
      if (!(VAR & 4) && !(VAR & 32)) load_piece(Q0 + g, stg[KS][g]);                                   \
      _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                 \
        /* swapped operands: D[n][m] - a lane holds one output row m and groups of 4 consecutive columns */ \
        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(KS) & 1][4 + j], fr[(KS) & 1][g],        \
                                                            (FIRST) ? fzero : acc[g][j], 0, 0, 0);       \
      }                                                                                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* MFMA */                                    \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* DS read */                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); /* DS write */                                \
      if (NB == 4) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* (three MFMAs per sub-block at BN_ = 192) */ \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* the buffer load (a VMEM read) */           \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
    if (VAR & 16) { _Pragma("unroll") for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(stg[KS][g])); } \
    if ((KS) == 2) {                                                                                   \
      advance_prefetch();                                                                              \
      wait_lgkm0();                                                                                    \
      if (!(VAR & 1)) barrier();                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
  } while (0)
#define STONK_W4_KTILE(ST, FIRST)    \
  do {                               \
    STONK_W4_KSTEP(ST, 0, FIRST);    \
    STONK_W4_KSTEP(ST, 1, false);    \
    STONK_W4_KSTEP(ST, 2, false);    \
    STONK_W4_KSTEP(ST, 3, false);    \
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
  read_frags(0, 0, fr[0]);
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: a wave drains its 128x128 corner in 8 rounds of 32 rows x 64 columns through a private 8 KiB slab
  // (XOR-swizzled), so that every global access - output, residual, saved pre-activation - is a 16-byte-per-lane row
  // segment (8 lanes = one 128-byte line of bf16). All of them are BUFFER accesses: a per-lane offset that does not
  // depend on the round (row-in-group * ld + column * size) + the round's uniform offset, one v_add - no 64-bit address
  // arithmetic, few registers (what the K loop keeps live across the epilogue is not spilled), and rows past M fall off
  // the end of the buffer instead of needing a predicate.
  // The uniform part is ADDED INTO the vector offset rather than passed as the instruction's scalar offset: a
  // buffer_store_dwordx4 with an SGPR offset reads its last data dwords late, and a following VALU write to those
  // registers (the next round's values) reached memory instead - seen here as the last 8 bytes of the round's last
  // store in the upper lanes carrying the NEXT round's data (the classic >64-bit-store data hazard; the toolchain does
  // not pad it on this target).
  typedef __attribute__((ext_vector_type(4))) unsigned eu32x4;
  f32x4 bq[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  auto store_tile = [&](const Work& w) {
    // the lane id is re-read from the hardware and the wave id made opaque: the epilogue's addresses are derived HERE,
    // not hoisted out of the K loop (where they would be spilled and reloaded behind a compiler-placed vmcnt(0))
    int lv = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wv = wave;
    asm volatile("" : "+v"(lv), "+s"(wv));
    const int r = lv & 31, hh = lv >> 5;
    char* ep = smem + LDS_RING + wv * SLAB_BYTES;
    const int flags = EPI >= 0 ? EPI : p.flags;
    const int rrow = lv >> 3, c8 = lv & 7;
    const int esz = OUT_MODE == 0 ? 2 : 4;
    const int ldc_b = (int)p.ldc * esz, ldr_b = (int)p.ldr * 2, ldx_b = (int)p.ldaux * 2;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, M * ldc_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, M * ldr_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.aux, 0, M * ldx_b, 0x00020000);
    const int wm0 = w.m0 + (wv >> 1) * 128, wn0 = w.n0 + (wv & 1) * (BN_ / 2);
    constexpr int WCOLS = BN_ / 2;   // columns of this wave's corner: the second 64-column round is half empty at BN_ = 192
    // Side operand (residual OR saved GELU input / GELU', never both here): the four 16-byte pieces a lane needs in round
    // k + 1 are requested as soon as round k has consumed its own - requested at their point of use, every round paid a full
    // memory latency (a 50 MB residual cost 35 us where its HBM time is 10). 16 registers, reused round after round
    // (a second set, one full round ahead, measured the same and spilled 70 more registers).
    constexpr bool SIDE_X = (EPI >= 0) && (EPI & STONK_EPI_GELU_BWD) != 0;
    constexpr bool SIDE_R = (EPI >= 0) && (EPI & STONK_EPI_RESID) != 0 && !SIDE_X;
    constexpr bool SIDE = SIDE_X || SIDE_R;
    bf16x8 sd[1][4];
    auto side_request = [&](const int round, bf16x8 (&dst)[4]) {
      if (!SIDE) return;
      const int bi2 = round >> 1, qb2 = round & 1;
      const int mrow0 = wm0 + bi2 * 32, n0 = wn0 + qb2 * 64;
      const int oob = (n0 + c8 * 8 < N && qb2 * 64 + c8 * 8 < WCOLS) ? 0 : 0x40000000;
      const int vo = rrow * (SIDE_X ? ldx_b : ldr_b) + c8 * 16 + oob;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int mrow = mrow0 + it * 8;
        dst[it] = __builtin_bit_cast(bf16x8, SIDE_X ? __builtin_amdgcn_raw_buffer_load_b128(rX, vo + mrow * ldx_b + n0 * 2, 0, 0)
                                                   : __builtin_amdgcn_raw_buffer_load_b128(rR, vo + mrow * ldr_b + n0 * 2, 0, 0));
      }
    };
    side_request(0, sd[0]);
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int round = bi * 2 + qb;
        const int mrow0 = wm0 + bi * 32;
        const int n0 = wn0 + qb * 64;
        const int n = n0 + c8 * 8;
        const bool n_ok = n < N && qb * 64 + c8 * 8 < WCOLS;
        const int oob = n_ok ? 0 : 0x40000000;   // columns past N (or past this wave's corner): pushed out of the buffer's range
        const f32x4 b0 = bq[qb][0], b1 = bq[qb][1];
        // write: lane holds row r, register group g -> columns 8g + 4hh .. +3 of each 32-column block
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (2 * qb + bj >= NB) continue;      // (BN_ = 192: no fourth column block; its slab columns are never stored)
            const int ch = bj * 8 + 2 * g + hh;   // 16-byte chunk of the 256-byte slab row
            const f32x16& c = acc[bi][2 * qb + bj];
            f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
            *(f32x4*)(ep + r * 256 + ((ch ^ (r & 15)) << 4)) = v * p.alpha;
          }
        __builtin_amdgcn_wave_barrier();
        if (OUT_MODE == 2) {
          // fp32 accumulate: one atomic wave-instruction = 64 consecutive floats of one row
#pragma unroll 4
          for (int rr = 0; rr < 32; ++rr) {
            const float x = *(const float*)(ep + rr * 256 + (((lv >> 2) ^ (rr & 15)) << 4) + (lv & 3) * 4);
            if (mrow0 + rr < M && n0 + lv < N && qb * 64 + lv < WCOLS) atomicAdd((float*)p.C + (long)(mrow0 + rr) * p.ldc + n0 + lv, x);
          }
        } else {
          const int voC = rrow * ldc_b + c8 * 8 * esz + oob;
          const int voR = rrow * ldr_b + c8 * 16 + oob, voX = rrow * ldx_b + c8 * 16 + oob;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + rrow;
            const int mm = mrow0 + row;
            const int mrow = mrow0 + it * 8;   // uniform part of the row
            const f32x4 q0 = *(const f32x4*)(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
            const f32x4 q1 = *(const f32x4*)(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
            float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            SideOps so;
            if (SIDE) {
              so.aux = so.res = sd[0][it];
            } else {   // run-time flags (EPI < 0) or both operands at once: requested here
              if (flags & STONK_EPI_GELU_BWD)
                so.aux = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rX, voX + mrow * ldx_b + n0 * 2, 0, 0));
              if (flags & STONK_EPI_RESID)
                so.res = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rR, voR + mrow * ldr_b + n0 * 2, 0, 0));
            }
            // (bias, GELU, GELU', dropout, residual on the 8 values; the saved pre-activation leaves from here as well)
            if (flags & STONK_EPI_BIAS) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] += b0[e];
                v[4 + e] += b1[e];
              }
            }
            if (flags & STONK_EPI_SAVE_PREACT) {
              bf16x8 u;
#pragma unroll
              for (int e = 0; e < 8; ++e) u[e] = (bf16)gelu_saved(v[e], (flags & STONK_EPI_AUX_GRAD) != 0);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(eu32x4, u), rX, voX + mrow * ldx_b + n0 * 2, 0, 0);
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            epilogue8_pre(v, p, flags & ~(STONK_EPI_BIAS | STONK_EPI_SAVE_PREACT), mm, n, z4, z4, so);
            if (OUT_MODE == 0) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(eu32x4, o), rC, voC + mrow * ldc_b + n0 * 2, 0, 0);
            } else {
              const eu32x4 lo = __builtin_bit_cast(eu32x4, (f32x4){v[0], v[1], v[2], v[3]});
              const eu32x4 hi = __builtin_bit_cast(eu32x4, (f32x4){v[4], v[5], v[6], v[7]});
              __builtin_amdgcn_raw_buffer_store_b128(lo, rC, voC + mrow * ldc_b + n0 * 4, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(hi, rC, voC + mrow * ldc_b + n0 * 4 + 16, 0, 0);
            }
          }
          if (round + 1 < 8) side_request(round + 1, sd[0]);   // flies while the next round is written to the slab
        }
        __builtin_amdgcn_wave_barrier();
      }
  };

  // ------------------------------------------------------------------ stream of K tiles
  int mode = 0;   // 1: the next K tile follows a full-tile boundary (its waits count the boundary's stores)
  for (;;) {   // one output tile (work item) per iteration; the load stream runs across iterations
    // The bias of this lane's 2 x 8 output columns is requested HERE, a whole K loop before the epilogue uses it: a
    // load issued in the epilogue would be waited for on the spot (draining the in-flight operand loads with it), and a
    // load issued after the tile's first store could not complete, as far as vmcnt can tell, before that store retires.
    // Buffer loads: columns past N read as zero, no branch.
    if ((EPI >= 0 ? EPI : p.flags) & STONK_EPI_BIAS) {
      const int lv = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, N * 4, 0x00020000);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int n = cw.n0 + wc * (BN_ / 2) + qb * 64 + (lv & 7) * 8 + h * 4;
          bq[qb][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBias, n * 4, 0, 0));
        }
    }
    {   // the first pair of K tiles: accumulators start from zero; after a full-tile boundary the first K tile's waits
        // count the boundary's stores
      const bool post = mode != 0;
      STONK_W4_KTILE(0, true);
    }
    {
      const bool post = false;
      STONK_W4_KTILE(1, false);
    }
    for (int ck = 2; ck < cw.nk; ck += 2) {   // (the launcher guarantees an even number of K tiles per work item)
      const bool post = false;
      STONK_W4_KTILE(0, false);
      STONK_W4_KTILE(1, false);
    }
    store_tile(cw);
    // the store count the post-boundary waits assume is only exact for a tile without masked rows / columns
    mode = (cw.m0 + BM <= M && cw.n0 + BN <= N) ? 1 : 0;
    Work nw;
    bool more;
    do {
      cwi += G;
      more = get_work(cwi, nw);
    } while (more && nw.nk <= 0);
    if (!more) break;
    cw = nw;
  }
  wait_vm<0>();   // (the cursor's last, unused loads)
#undef STONK_W4_KTILE
#undef STONK_W4_KSTEP
#undef STONK_W4_LOADS_OF
}

template <int OUT_MODE, int EPI, int BN_ = 256>
int launch_w4(const GemmArgs& a, int grid, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<OUT_MODE, EPI, BN_>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_w4_kernel<OUT_MODE, EPI, BN_>), dim3(grid), dim3(256), LDS_BYTES, st, a);
  return stonk_launch_status();
}

}  // namespace

// Launcher used by stonk_gemm_nt_bf16 (gemm_bf16.hip). Requires K % 64 == 0 and an EVEN number of K tiles per work item
// (the K loop is unrolled by two; every GEMM of the STonKGs step has K = 768, 2304 or 3072). tile_n: 0 = choose, 256, 192.
// items_per_wg > 0: grid = work items / items_per_wg instead of one persistent workgroup per CU (STONK_GEMM_DISPATCHED: 1,
// _DISPATCHED2: 2) - the hardware dispatcher is then the work queue and a workgroup that gets its CU late delays nobody
// else; the workgroups still take their items in XCD-contiguous runs, and with two items the second one's operands are
// prefetched across the tile boundary as in the persistent form.
int stonk_gemm_w4_launch(const GemmArgs& a, int out_mode, int tile_n, int items_per_wg, hipStream_t st) {
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
  const long ntm = (a.M + BM - 1) / BM;
  // 192-wide tiles where they quantise better: time ~ rounds of the CUs x tile width (the launches they are built for:
  // N = 768 at 32 768 rows, two full rounds instead of one and a half; at 16 384 rows one full round instead of 3/4)
  const bool has192 = out_mode == 0 && a.N % 192 == 0 && (epi == 0 || epi == B || epi == R || epi == (B | R) || epi == (B | R | D));
  if (tile_n == 192 && !has192) return STONK_ESHAPE;
  if (tile_n == 0) {
    const long t256 = ntm * ((a.N + 255) / 256) * a.split_k, t192 = ntm * (a.N / 192) * a.split_k;
    const long c256 = ((t256 + n_cu - 1) / n_cu) * 256, c192 = ((t192 + n_cu - 1) / n_cu) * 192;
    tile_n = (has192 && c192 < c256) ? 192 : 256;
  }
  if (tile_n == 192) {
    const long tiles = ntm * (a.N / 192) * a.split_k;
    const int grid = (int)(items_per_wg > 0 ? (tiles + items_per_wg - 1) / items_per_wg : (tiles < n_cu ? tiles : n_cu));
    switch (epi) {
      case 0: return launch_w4<0, 0, 192>(a, grid, st);
      case B: return launch_w4<0, B, 192>(a, grid, st);
      case R: return launch_w4<0, R, 192>(a, grid, st);
      case B | R: return launch_w4<0, B | R, 192>(a, grid, st);
      default: return launch_w4<0, B | R | D, 192>(a, grid, st);
    }
  }
  const long tiles = ntm * ((a.N + 255) / 256) * a.split_k;
  const int grid = (int)(items_per_wg > 0 ? (tiles + items_per_wg - 1) / items_per_wg : (tiles < n_cu ? tiles : n_cu));
  if (out_mode == 1) return epi == 0 ? launch_w4<1, 0>(a, grid, st) : launch_w4<1, -1>(a, grid, st);
  if (out_mode == 2) return launch_w4<2, 0>(a, grid, st);
  switch (epi) {   // the combinations the STonKGs step uses are compiled with constant flags
    case 0: return launch_w4<0, 0>(a, grid, st);
    case B: return launch_w4<0, B>(a, grid, st);
    case B | G: return launch_w4<0, B | G>(a, grid, st);
    case B | G | SV: return launch_w4<0, B | G | SV>(a, grid, st);
    case GB: return launch_w4<0, GB>(a, grid, st);
    case B | G | SV | AG: return launch_w4<0, B | G | SV | AG>(a, grid, st);
    case GB | AG: return launch_w4<0, GB | AG>(a, grid, st);
    case R: return launch_w4<0, R>(a, grid, st);
    case B | R: return launch_w4<0, B | R>(a, grid, st);
    case B | R | D: return launch_w4<0, B | R | D>(a, grid, st);
    default: return launch_w4<0, -1>(a, grid, st);
  }
}
