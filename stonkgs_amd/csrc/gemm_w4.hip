// bf16 MFMA GEMM, 256x256x64 tiles, FOUR waves per workgroup, one wave per SIMD: each wave owns a 128x128 corner of
// the tile in 256 accumulator registers (16 blocks of v_mfma_f32_32x32x16_bf16). Same contract as gemm_bf16.hip
// (C = epilogue(alpha * A[M,K] . B[N,K]^T)).
//
// Why a third tiling: in the eight-wave kernel (gemm256.hip, wave tile 128x64) every K tile costs the CU 192 KiB of
// fragment reads + 64 KiB of LDS-DMA writes = 2048 LDS cycles at 128 B/clk - exactly the 2048 MFMA cycles of the K
// tile, so the matrix pipe can never be more than about half busy. A 128x128 wave tile reads 128 KiB per K tile
// (75 % of the LDS budget with the DMA writes), which is what the vendor library's 256x256 kernels do as well.
//
// Schedule. A K tile is staged as four 16 KiB half-tiles cut by use (A-first = rows 0-63 of each wave-row's 128,
// B-first = columns 0-63 of each wave-column's 128, B-second, A-second), in a ring of EIGHT slots (two K tiles). A K
// tile is four phases of 16 MFMAs, one 64x64 quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0). With half-tiles numbered
// h = 4t + {0 A-first, 1 B-first, 2 B-second, 3 A-second} and phases phi = 4t + p, phase phi
//   * multiplies the quadrant whose fragments are already in registers,
//   * reads half-tile phi+2 from LDS into the fragment registers the NEXT phase needs (8 ds_read_b128),
//   * DMAs half-tile phi+9 (global_load_lds_dwordx4, 4 per wave) into the slot that phase phi-1 finished reading,
//   * ends with  s_waitcnt lgkmcnt(0) ; s_waitcnt vmcnt(24) ; s_barrier : half-tile phi+3 has landed (six younger
//     half-tiles = 24 DMA instructions stay in flight, about 1.5 us of prefetch distance) and every wave is done reading
//     the slot the next phase refills.
// B-first fragments are kept from phase 0 to phase 3, so the two B register sets swap roles every K tile (the K loop is
// unrolled by two; slot numbers are compile-time constants of (phase, parity)).
// The prefetch cursor runs across output tiles, and because it is seven half-tiles ahead, the waits of the six phases
// after a tile boundary simply count the epilogue's stores on top (vmcnt retires in order): no drain, no special
// boundary protocol. When the cursor runs out of work it re-reads its last position (never consumed), so the wait
// counts stay constant to the end.
#include <cstdlib>

#include "gemm_common.h"

using namespace stonk_gemm;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;            // 16 KiB
constexpr int LDS_RING = 8 * HALF_BYTES;            // 128 KiB
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

// VAR: timing experiments only (bit 0 no barrier, bit 1 all fragment reads in the first two sub-steps, bit 2 no DMA in the
// loop, bit 3 no fragment reads in the loop) - anything but 0 / 2 computes garbage.
template <int OUT_MODE, int EPI, int VAR = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmArgs p) {
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
      const int r = w / G, b = w - r * G;
      const int cand = r * G + (b & 7) * (G >> 3) + (b >> 3);
      if ((r + 1) * G <= total) idx = cand;  // full rounds only; the ragged last round keeps natural order
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

  // ---- DMA: half-tile image = 128 rows x 128 B; wave w moves image rows [32w, 32w+32) in four 1 KiB pieces (8 rows
  // each). LDS-DMA writes lane-linearly, so lane -> (row l8 = lane>>3, physical chunk lane&7) and the XOR swizzle
  // (physical = logical ^ ((row >> 1) & 7)) is applied to the SOURCE chunk. One wave per SIMD means every bookkeeping
  // instruction competes with the MFMAs for the wave's single issue stream, so a piece costs NO vector arithmetic: its
  // address is a uniform 64-bit base (SALU: cursor base + first row of the piece * ld) plus one of four loop-invariant
  // per-lane offsets (l8 * ld + swizzled chunk; the odd pieces' chunk is the even pieces' ^ 64). Pieces that would
  // start past the last 8 rows of the operand are pulled back to it as a whole (valid memory; such rows are never
  // stored). The launcher guarantees rows * ld * 2 < 2^31, ld % 64 == 0 and at least 8 rows.
  const int l8 = lane >> 3;
  const int lda2 = (int)p.lda * 2, ldb2 = (int)p.ldb * 2;
  uint32_t voffA[2], voffB[2];
  {
    const uint32_t c0 = (uint32_t)(((lane & 7) ^ (l8 >> 1)) * 16);
    voffA[0] = (uint32_t)(l8 * lda2) + c0;
    voffA[1] = (uint32_t)(l8 * lda2) + (c0 ^ 64u);
    voffB[0] = (uint32_t)(l8 * ldb2) + c0;
    voffB[1] = (uint32_t)(l8 * ldb2) + (c0 ^ 64u);
  }
  const int wave_lds = wave * 4096;
  const int wave_origin = (wave >> 1) * 128 + (wave & 1) * 32;
  const char* curA = (const char*)p.A;   // operand bases advanced to the K tile the prefetch cursor points at
  const char* curB = (const char*)p.B;
  int p_m0 = 0, p_n0 = 0;
  auto set_sources = [&](const Work& w) {
    p_m0 = w.m0;
    p_n0 = w.n0;
    curA = (const char*)p.A + w.k_begin * 2;
    curB = (const char*)p.B + w.k_begin * 2;
  };
  // kind: 0 A-first, 1 B-first, 2 B-second, 3 A-second
  auto issue_piece = [&](const int kind, const int slot, const int i) {
    int wl = wave_lds, wo = wave_origin;
    asm volatile("" : "+s"(wl), "+s"(wo));   // opaque: destinations are re-derived here, not kept in 32 SGPRs
    char* dst = smem + wl + slot * HALF_BYTES + i * 1024;
    const bool isA = (kind == 0 || kind == 3);
    const bool second = kind >= 2;
    int row0 = (isA ? p_m0 : p_n0) + wo + (second ? 64 : 0) + i * 8;
    const int last = (isA ? p.M : N) - 8;
    row0 = row0 < last ? row0 : last;
    const char* base = (isA ? curA : curB) + (long)(row0 * (isA ? lda2 : ldb2));
    const uint32_t voff = isA ? voffA[i & 1] : voffB[i & 1];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue = [&](const int kind, const int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece(kind, slot, i);
  };

  // ---- fragment reads: lane (r = lane & 31, hh = lane >> 5) takes row r of a 32-row block, k = 16 ks + 8 hh .. +7
  const int r = lane & 31, hh = lane >> 5;
  int lofs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) lofs[ks] = r * 128 + (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4);
  const int a_base = wr * 8192, b_base = wc * 8192;   // this wave's 64 image rows of an A / B half
  bf16x8 fa0[2][4], fa1[2][4], fb[2][2][4];            // [row block][k step]
  auto read_half = [&](const int slot, const int wbase, bf16x8 (&f)[2][4]) {
    const char* s = smem + slot * HALF_BYTES + wbase;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) f[rb][ks] = *(const bf16x8*)(s + rb * 4096 + lofs[ks]);
  };

  f32x16 acc[2][2][2][2];   // [A half][B half][32-row block][32-col block]
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][i][j][e] = 0.f;
  };
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
  // advance the prefetch cursor by one K tile (after its A-second half-tile has been issued)
  auto advance_prefetch = [&]() {
    if (!p_valid) return;   // out of work: the cursor keeps re-reading its last K tile (valid memory, never consumed)
    if (pk + 1 < pw.nk) {
      ++pk;
      curA += BK * 2;
      curB += BK * 2;
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

  // epilogue vm operations that can still be outstanding during the six phases after a tile boundary
  constexpr int FL = EPI >= 0 ? EPI : 0;
  constexpr int S = (OUT_MODE == 0) ? ((FL & STONK_EPI_SAVE_PREACT) || EPI < 0 ? 64 : 32) : (OUT_MODE == 1 ? 64 : 0);
  constexpr int WAIT_POST = (24 + S) > 63 ? 63 : (24 + S);
  auto phase_end = [&](const bool post) {
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkm0();
    if (post) wait_vm<WAIT_POST>();
    else wait_vm<24>();
    if (!(VAR & 1)) barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // One phase = four sub-steps, one 16-deep k step each: 4 MFMAs on four different accumulator blocks (so no MFMA
  // waits for its predecessor's result), two of the next phase's fragment reads and one DMA piece, placed in the MFMA
  // gaps (one wave per SIMD: the same wave has to feed the matrix pipe AND issue the loads). The sub-steps are fenced
  // so the scheduler cannot chain the four k steps of one accumulator back to back.
  auto phase = [&](f32x16 (&c)[2][2], const bf16x8 (&a)[2][4], const bf16x8 (&b)[2][4], const int rslot, const int rbase,
                   bf16x8 (&f)[2][4], const int kind, const int islot) {
    const char* rs = smem + rslot * HALF_BYTES + rbase;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (!(VAR & 8)) {
        if (VAR & 2) {
          if (ks < 2) {
            f[0][2 * ks] = *(const bf16x8*)(rs + lofs[2 * ks]);
            f[1][2 * ks] = *(const bf16x8*)(rs + 4096 + lofs[2 * ks]);
            f[0][2 * ks + 1] = *(const bf16x8*)(rs + lofs[2 * ks + 1]);
            f[1][2 * ks + 1] = *(const bf16x8*)(rs + 4096 + lofs[2 * ks + 1]);
          }
        } else {
          f[0][ks] = *(const bf16x8*)(rs + lofs[ks]);
          f[1][ks] = *(const bf16x8*)(rs + 4096 + lofs[ks]);
        }
      }
      if (!(VAR & 4)) issue_piece(kind, islot, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // swapped operands: D[n][m] - a lane holds one output row m and groups of 4 consecutive columns
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][ks], a[i][ks], c[i][j], 0, 0, 0);
        }
      if (VAR & 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (ks < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (ks < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // VALU: the DMA piece's address
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // the LDS-DMA (a VMEM read)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: half-tiles 0..6, then the two read-only phases phi = -2, -1
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  issue(3, 3);
  advance_prefetch();
  issue(0, 4);
  issue(1, 5);
  issue(2, 6);
  wait_vm<24>();
  barrier();
  read_half(0, a_base, fa0);
  issue(3, 7);
  advance_prefetch();
  phase_end(false);
  read_half(1, b_base, fb[0]);
  issue(0, 0);
  phase_end(false);

  // One K tile; PAR = parity of the K tile in the stream (selects ring stage and the B register roles).
  // post01 / post23: phases 0-1 / 2-3 fall within the six phases after a tile boundary (waits count its stores too)
#define STONK_W4_KTILE(PAR)                                                                                      \
  do {                                                                                                           \
    /* phase 0: (A0, B-first) ; read B-second ; DMA B-first of tile t+2 */                                       \
    phase(acc[0][0], fa0, fb[PAR], (4 * (PAR) + 2) & 7, b_base, fb[(PAR) ^ 1], 1, (4 * (PAR) + 1) & 7);          \
    phase_end(post01);                                                                                           \
    /* phase 1: (A0, B-second) ; read A-second ; DMA B-second of t+2 */                                          \
    phase(acc[0][1], fa0, fb[(PAR) ^ 1], (4 * (PAR) + 3) & 7, a_base, fa1, 2, (4 * (PAR) + 2) & 7);              \
    phase_end(post01);                                                                                           \
    /* phase 2: (A1, B-second) ; read A-first of t+1 ; DMA A-second of t+2, then the cursor moves on */          \
    phase(acc[1][1], fa1, fb[(PAR) ^ 1], (4 * (PAR) + 4) & 7, a_base, fa0, 3, (4 * (PAR) + 3) & 7);              \
    phase_end(post23);                                                                                           \
    advance_prefetch();                                                                                          \
    /* phase 3: (A1, B-first) ; read B-first of t+1 into the B-second registers ; DMA A-first of t+3 */          \
    phase(acc[1][0], fa1, fb[PAR], (4 * (PAR) + 5) & 7, b_base, fb[(PAR) ^ 1], 0, (4 * (PAR) + 4) & 7);          \
    phase_end(post23);                                                                                           \
  } while (0)

  // ---- epilogue: a wave drains its 128x128 corner in 8 rounds of 32 rows x 64 columns through a private 8 KiB slab
  // (XOR-swizzled), so that every global access - output, residual, saved pre-activation - is a 16-byte-per-lane row
  // segment (8 lanes = one 128-byte line of bf16).
  auto store_tile = [&](const Work& w) {
    char* ep = smem + LDS_RING + wave * SLAB_BYTES;
    const int flags = EPI >= 0 ? EPI : p.flags;
    const int rrow = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const int mrow0 = w.m0 + wr * 128 + qa * 64 + bi * 32;
          const int n0 = w.n0 + wc * 128 + qb * 64;
          const int n = n0 + c8 * 8;
          const bool n_ok = n < N;
          f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
          if ((flags & STONK_EPI_BIAS) && n_ok) {
            b0 = *(const f32x4*)(p.bias + n);
            b1 = *(const f32x4*)(p.bias + n + 4);
          }
          // write: lane holds row r, register group g -> columns 8g + 4hh .. +3 of each 32-column block
#pragma unroll
          for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ch = bj * 8 + 2 * g + hh;   // 16-byte chunk of the 256-byte slab row
              const f32x16& c = acc[qa][qb][bi][bj];
              f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
              *(f32x4*)(ep + r * 256 + ((ch ^ (r & 15)) << 4)) = v * p.alpha;
            }
          __builtin_amdgcn_wave_barrier();
          if (OUT_MODE == 2) {
            // fp32 accumulate: one atomic wave-instruction = 64 consecutive floats of one row
#pragma unroll 4
            for (int rr = 0; rr < 32; ++rr) {
              const float x = *(const float*)(ep + rr * 256 + (((lane >> 2) ^ (rr & 15)) << 4) + (lane & 3) * 4);
              if (mrow0 + rr < M && n0 + lane < N) atomicAdd((float*)p.C + (long)(mrow0 + rr) * p.ldc + n0 + lane, x);
            }
          } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int row = it * 8 + rrow;
              const int mm = mrow0 + row;
              const f32x4 q0 = *(const f32x4*)(ep + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
              const f32x4 q1 = *(const f32x4*)(ep + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
              if (mm < M && n_ok) {
                float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                SideOps so;
                side_prefetch(so, p, flags, mm, n, true);
                epilogue8_pre(v, p, flags, mm, n, b0, b1, so);
                if (OUT_MODE == 0) {
                  bf16x8 o;
#pragma unroll
                  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                  *(bf16x8*)((bf16*)p.C + (long)mm * p.ldc + n) = o;
                } else {
                  float* dst = (float*)p.C + (long)mm * p.ldc + n;
                  *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
                  *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
  };

  // ------------------------------------------------------------------ stream of K tiles
  int mode = 0;   // 1: the next two K tiles follow a full-tile boundary (their first six phases count its stores)
  for (;;) {   // one output tile (work item) per iteration; the DMA stream runs across iterations
    zero_acc();
    for (int ck = 0; ck < cw.nk; ck += 2) {   // (the launcher guarantees an even number of K tiles per work item)
      {
        const bool post01 = mode != 0, post23 = mode != 0;
        STONK_W4_KTILE(0);
      }
      {
        const bool post01 = mode != 0, post23 = false;
        STONK_W4_KTILE(1);
      }
      mode = 0;
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
  // the ring may still be receiving the cursor's last (unused) half-tiles: let them land before the LDS is released
  wait_vm<0>();
}

template <int OUT_MODE, int EPI, int VAR = 0>
int launch_w4(const GemmArgs& a, int grid, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<OUT_MODE, EPI, VAR>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_w4_kernel<OUT_MODE, EPI, VAR>), dim3(grid), dim3(256), LDS_BYTES, st, a);
  return stonk_launch_status();
}

}  // namespace

// Launcher used by stonk_gemm_nt_bf16 (gemm_bf16.hip). Requires K % 64 == 0 and an EVEN number of K tiles per work item
// (the K loop is unrolled by two; every GEMM of the STonKGs step has K = 768, 2304 or 3072).
int stonk_gemm_w4_launch(const GemmArgs& a, int out_mode, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    n_cu = prop.multiProcessorCount;
  }
  const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.split_k;
  const int grid = (int)(tiles < n_cu ? tiles : n_cu);
  constexpr int B = STONK_EPI_BIAS, G = STONK_EPI_GELU, SV = STONK_EPI_SAVE_PREACT, GB = STONK_EPI_GELU_BWD,
                R = STONK_EPI_RESID, D = STONK_EPI_DROPOUT;
  const int epi = a.flags & (B | G | SV | GB | R | D);
  if (out_mode == 1) return epi == 0 ? launch_w4<1, 0>(a, grid, st) : launch_w4<1, -1>(a, grid, st);
  if (out_mode == 2) return launch_w4<2, 0>(a, grid, st);
  switch (epi) {   // the combinations the STonKGs step uses are compiled with constant flags
    case 0: {
      static const int var = getenv("STONK_W4_VAR") ? atoi(getenv("STONK_W4_VAR")) : 0;   // timing experiments
      switch (var) {
        case 1: return launch_w4<0, 0, 1>(a, grid, st);
        case 2: return launch_w4<0, 0, 2>(a, grid, st);
        case 3: return launch_w4<0, 0, 3>(a, grid, st);
        case 4: return launch_w4<0, 0, 4>(a, grid, st);
        case 8: return launch_w4<0, 0, 8>(a, grid, st);
        case 12: return launch_w4<0, 0, 12>(a, grid, st);
        case 13: return launch_w4<0, 0, 13>(a, grid, st);
        default: return launch_w4<0, 0>(a, grid, st);
      }
    }
    case B: return launch_w4<0, B>(a, grid, st);
    case B | G: return launch_w4<0, B | G>(a, grid, st);
    case B | G | SV: return launch_w4<0, B | G | SV>(a, grid, st);
    case GB: return launch_w4<0, GB>(a, grid, st);
    case R: return launch_w4<0, R>(a, grid, st);
    case B | R: return launch_w4<0, B | R>(a, grid, st);
    case B | R | D: return launch_w4<0, B | R | D>(a, grid, st);
    default: return launch_w4<0, -1>(a, grid, st);
  }
}
