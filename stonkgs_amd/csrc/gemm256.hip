// bf16 MFMA GEMM, 256x256x64 tiles, persistent, LDS-DMA pipelined across barriers AND across output tiles.
// Same contract as gemm_bf16.hip (C = epilogue(alpha * A[M,K] . B[N,K]^T)); used for the large launches of the
// STonKGs step, where the 128x128 two-barrier kernel tops out at ~0.65-0.95 PFLOP/s.
//
// Structure (CDNA4 playbook, "256^2 8-phase" family; the schedule below is derived in DESIGN.md section 4.1):
//  * 512 threads = 8 waves as 2(M) x 4(N); a wave owns 128x64 of the tile = 2x2 quadrants of 64x32, i.e. 128
//    accumulator VGPRs of v_mfma_f32_16x16x32_bf16; ONE workgroup per CU (128 KiB LDS), two waves per SIMD.
//  * A K tile (64 deep) is staged as FOUR 16 KiB half-tiles cut by USE, not by position:
//      A-first  = the first 64 rows of each wave-row's 128  (tile rows 0-63, 128-191)
//      B-first  = the first 32 cols of each wave-col's 64   (tile cols 0-31, 64-95, 128-159, 192-223)
//      B-second, A-second = the rest.
//    A K tile is consumed in 4 phases, one accumulator quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0). Phase 0
//    needs A-first+B-first, phase 1 B-second, phase 2 A-second, phase 3 nothing new - so half-tiles are DMA'd
//    (global_load_lds_dwordx4, 2 per wave per half-tile) in exactly that order, one per phase, into the OTHER
//    K-tile buffer, 3-4 phases before their first read. Waits are counted (s_waitcnt vmcnt(4): two half-tiles
//    stay in flight across every barrier), barriers are raw s_barrier, never __syncthreads (which would drain).
//  * every phase is  [ds_read operand sub-block | issue DMA | wait] s_barrier [16 MFMA] s_barrier ; the second
//    wave-row runs one barrier behind the first, so on each SIMD one wave is in its MFMA section while its
//    partner is in its LDS/DMA section.
//  * RAW: a half-tile is read one phase after the wait that retires it (all waves wait, then a barrier).
//    WAR: a slot is re-staged >= 4 phases after its last read. Both hold for the lagging wave-row too.
//  * persistent: a workgroup walks its work items (tile x K-split) as ONE stream of K tiles, so the DMA for the next
//    output tile's first K tiles is in flight while the current tile's epilogue stores run.
#include "gemm_common.h"

using namespace stonk_gemm;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;       // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // 64 KiB per K tile
constexpr int LDS_STAGES = 2 * STAGE_BYTES;    // 128 KiB of K-tile stages
constexpr int LDS_BYTES = LDS_STAGES + 8 * 4096;  // + a private 4 KiB epilogue slab per wave = 160 KiB
// slot order inside a stage = DMA / first-use order
constexpr int SLOT_A0 = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_A1 = 3;

__device__ __forceinline__ void wait_vm4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier() { __builtin_amdgcn_s_barrier(); }

struct Work {
  int m0, n0;       // tile origin
  long k_begin;     // element offset of the first K tile
  int nk;           // K tiles in this work item
};

// EPI >= 0: the epilogue flags are a compile-time constant (no dead side-operand loads for the compiler to guard with
// vmcnt(0) - such a wait inside the epilogue rounds also waits for the previous round's STORES and costs ~2 us per
// round); EPI < 0: flags read from the arguments at run time (rare combinations).
//
// TN = true: the weight-gradient form. A = dY [K tokens][M features], B = X [K tokens][N features] (both row-major as
// backward holds them), C[M][N] += sum_t A[t][m] B[t][n], bias[m] += sum_t A[t][m]. Same schedule, same half-tile
// cut; only the DMA source addressing (a half-tile image is [64 tokens][256 B]) and the fragment reads
// (ds_read_b64_tr_b16 pairs, image swizzle c ^ (((r&3)<<2)|((r>>2)&3))) differ. 256x256 tiles halve the operand bytes
// per flop against the 128x128 TN kernel, which runs at the L2's pace on the small weights of this model.
template <int OUT_MODE, int EPI, bool TN>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  int M = p.M;
  if (p.m_dev) {
    const int md = *p.m_dev;
    M = md < M ? md : M;
  }
  const int N = p.N;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  int nk_total = TN ? (p.K + BK - 1) / BK : p.K / BK;
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

  // ---- DMA sources of the prefetch cursor. Nothing per-lane is kept across the loop (a spilled address would be
  // reloaded behind a compiler-inserted vmcnt(0) and serialise the DMA pipeline): each issue recomputes its two
  // row addresses from wave-uniform scalars + (lane >> 3), ~10 VALU beside 16 MFMAs.
  const int l8 = lane >> 3;                                   // row within an 8-row DMA piece
  const int cofs = ((lane & 7) ^ (l8 & 7)) * 16;              // swizzled source chunk (bytes) for LDS slot (lane & 7)
  long pk_bytes = 0;  // byte offset (along K) of the K tile the prefetch cursor points at
  int p_m0 = 0, p_n0 = 0;
  auto set_sources = [&](const Work& w) {
    p_m0 = w.m0;
    p_n0 = w.n0;
    pk_bytes = TN ? w.k_begin : w.k_begin * 2;
  };
  // issue half-tile `slot` of the K tile at the prefetch cursor into stage `buf`
  auto issue = [&](int slot, int buf) {
    char* dst = smem + buf * STAGE_BYTES + slot * HALF_BYTES + wave * 2048;
    const bool isA = (slot == SLOT_A0 || slot == SLOT_A1);
    if (TN) {
      // image [64 token rows][256 B]; a wave-instruction = 4 rows; lane -> row (lane>>4), physical chunk (lane&15)
      const char* base = (const char*)(isA ? p.A : p.B);
      const long ld2 = (isA ? p.lda : p.ldb) * 2;
      int lv = lane;
      asm volatile("" : "+v"(lv));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 4 + (lv >> 4);
        const int c = (lv & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));   // logical chunk for physical slot (lane & 15)
        int feat;
        if (isA) feat = p_m0 + (c >> 3) * 128 + (c & 7) * 8 + (slot == SLOT_A1 ? 64 : 0);
        else feat = p_n0 + (c >> 2) * 64 + (c & 3) * 8 + (slot == SLOT_B1 ? 32 : 0);
        const int lim = (isA ? M : N) - 8;
        feat = feat < lim ? feat : lim;   // columns past the matrix: valid memory, results never stored
        const long tok = pk_bytes + r;    // (token index of the K tile's first row is kept in pk_bytes for TN)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + tok * ld2 + feat * 2),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
      return;
    }
    const char* base = (const char*)(isA ? p.A : p.B) + pk_bytes;
    const long ld2 = (isA ? p.lda : p.ldb) * 2;
    const int lim = (isA ? M : N) - 1;
    int l8v = l8, cv = cofs;
    asm volatile("" : "+v"(l8v), "+v"(cv));  // opaque: keeps the address arithmetic here instead of hoisted + spilled
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // image row r = (wave*2+i)*8 + l8 of the 128-row half-tile:
      //  A halves: rows 0-63 -> wave-row 0, 64-127 -> wave-row 1 (tile row (r>>6)*128 + (r&63), +64 for A-second)
      //  B halves: rows 32j..32j+31 -> wave-col j            (tile col (r>>5)*64 + (r&31), +32 for B-second)
      int row;
      if (isA) row = p_m0 + (wave >> 2) * 128 + ((wave & 3) * 2 + i) * 8 + (slot == SLOT_A1 ? 64 : 0) + l8v;
      else row = p_n0 + (wave >> 1) * 64 + ((wave & 1) * 2 + i) * 8 + (slot == SLOT_B1 ? 32 : 0) + l8v;
      row = row < lim ? row : lim;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + row * ld2 + cv),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes) inside a half-tile image
  const int frag_off = (lane & 15) * 128;
  const int kc = lane >> 4, swz = lane & 7;
  const int a_row0 = wr * 64 * 128;   // this wave-row's 64 rows of an A half
  const int b_row0 = wc * 32 * 128;   // this wave-col's 32 rows of a B half

  f32x4 acc[2][2][4][2];
  // The bias enters through the accumulators' initial value (C = bias + A.B, alpha == 1 enforced by the launcher): a
  // lane's 4 consecutive columns of fragment tile (b, j) are the same for every row block, so 4 quads per lane per
  // tile. They are loaded for the NEXT tile at the top of the tile boundary, BEFORE its DMA and stores are issued, so
  // the wait the compiler puts in front of init_acc is a counted one that the boundary's own wait already satisfied.
  f32x4 bq[2][2];
  auto load_bias_quads = [&](const Work& w) {
    const int flags = EPI >= 0 ? EPI : p.flags;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = w.n0 + wc * 64 + b * 32 + j * 16 + (lane >> 4) * 4;
        bq[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((flags & STONK_EPI_BIAS) && n < N) bq[b][j] = *(const f32x4*)(p.bias + n);
      }
  };
  auto init_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][b][i][j] = bq[b][j];
  };
  bf16x8 fa[2][4], fb0[2][2], fb1[2][2];  // [k-step][tile]
  // TN: transposed-read offsets. Lane (g = lane>>4, q = (lane&15)>>2, pq = lane&3) addresses token row 8g + 4hi + q
  // (+32 per k-step: swizzle unchanged), 4 features at 4*pq of a 16-feature tile; it receives feature (lane&15).
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  int toffa[2][4], toffb[2][2];
  if (TN) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      const int row = 8 * g + 4 * hi + q;
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) toffa[hi][i] = row * 256 + (((wr * 8 + 2 * i + (pq >> 1)) ^ sw) << 4) + ((pq & 1) << 3);
#pragma unroll
      for (int j = 0; j < 2; ++j) toffb[hi][j] = row * 256 + (((wc * 4 + 2 * j + (pq >> 1)) ^ sw) << 4) + ((pq & 1) << 3);
    }
  }
  auto tr_pair = [&](const char* s, int off_lo, int off_hi) -> bf16x8 {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s + off_lo));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s + off_hi));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto read_a = [&](int buf, int slot) {
    if (TN) {
      const char* s = smem + buf * STAGE_BYTES + slot * HALF_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[ks][i] = tr_pair(s + ks * 8192, toffa[0][i], toffa[1][i]);
      return;
    }
    const char* s = smem + buf * STAGE_BYTES + slot * HALF_BYTES + a_row0 + frag_off;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = *(const bf16x8*)(s + i * 2048 + (((ks * 4 + kc) ^ swz) << 4));
  };
  auto read_b = [&](int buf, int slot, bf16x8 (&fb)[2][2]) {
    if (TN) {
      const char* s = smem + buf * STAGE_BYTES + slot * HALF_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[ks][j] = tr_pair(s + ks * 8192, toffb[0][j], toffb[1][j]);
      return;
    }
    const char* s = smem + buf * STAGE_BYTES + slot * HALF_BYTES + b_row0 + frag_off;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[ks][j] = *(const bf16x8*)(s + j * 2048 + (((ks * 4 + kc) ^ swz) << 4));
  };
  // TN bias gradient: column sums of the A operand on the matrix pipe. One accumulator per A sub-block: fragment
  // tile i is multiplied by a selector whose rows 4i..4i+3 are ones, so its sums land in the lanes with lane>>4 == i.
  f32x4 accb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto bias_mma = [&](int a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool on = ((lane & 15) >> 2) == i;
      bf16x8 sel;
#pragma unroll
      for (int e = 0; e < 8; ++e) sel[e] = on ? (bf16)1.0f : (bf16)0.0f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, fa[ks][i], accb[a], 0, 0, 0);
    }
  };
  auto mma = [&](f32x4 (&c)[4][2], const bf16x8 (&fb)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // swapped operands: D[n][m], so a lane ends up with 4 consecutive output columns of one row
          c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], c[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- epilogue: the tile leaves through a PRIVATE 4 KiB LDS slab per wave (16 rows x 64 cols fp32 per round, XOR
  // swizzled), so that every global access of the epilogue - output, residual, saved pre-activation - is a full
  // 16-byte-per-lane row segment (4 lanes = one 128-byte line) instead of 8-byte pieces scattered over 16 rows
  // (those cost more than the whole K loop at K = 768). Wave-private: no workgroup barrier, the DMA stream of the
  // next tile keeps running underneath.
  // Side operand of the epilogue (residual OR saved pre-activation, never both on this kernel): the 16 row segments of
  // this lane are requested in two batches of 8 - the first at the very top of the tile boundary, before the
  // boundary's DMA, so that the boundary's own wait (E2) covers it; the second right after that wait, so it flies while
  // rounds 0-3 are processed. Latency is paid about once per tile instead of once per 16-row round. The vectors live
  // in registers the operand fragments have just vacated (8 x 4 VGPRs per batch).
  constexpr int FL = EPI >= 0 ? EPI : 0;
  constexpr bool HAS_SIDE = EPI >= 0 && (FL & (STONK_EPI_RESID | STONK_EPI_GELU_BWD)) != 0 && OUT_MODE != 2;
  bf16x8 side_a[4][2], side_b[4][2];
  auto prefetch_side = [&](const Work& w, int batch, bf16x8 (&dst)[4][2]) {
    if (!HAS_SIDE) return;
    const int n = w.n0 + wc * 64 + (lane & 7) * 8;
    const int mbase = w.m0 + wr * 128 + batch * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mm = mbase + i * 16 + h * 8 + (lane >> 3);
        dst[i][h] = side_load1(p, FL, mm, n, mm < M && n < N);
      }
  };
  auto store_tile = [&](const Work& w) {
    char* ep = smem + LDS_STAGES + wave * 4096;
    const int flags = (EPI >= 0 ? EPI : p.flags) & ~STONK_EPI_BIAS;
    const int wm = lane & 15, wq = lane >> 4;   // write side: row, 4-column group inside a 16x16 fragment tile
    const int rrow = lane >> 3, c8 = lane & 7;  // read side: 8 lanes x 8 columns = one 128-byte (bf16) line per row
    const int n0 = w.n0 + wc * 64;
    const int n = n0 + c8 * 8;
    const bool n_ok = n < N;
    const int mbase = w.m0 + wr * 128;
    // (the bias is already inside the accumulators: they were INITIALISED with it, see init_acc)
    const f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    SideOps side[2][2];   // run-time-flag instance only: fetched one round ahead
    if (OUT_MODE != 2 && !HAS_SIDE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mm = mbase + h * 8 + rrow;
        side_prefetch(side[0][h], p, flags, mm, n, mm < M && n_ok);
      }
    }
#pragma unroll
    for (int rd = 0; rd < 8; ++rd) {
      const int a = rd >> 2, i = rd & 3;
      const int m = mbase + a * 64 + i * 16;
      if (OUT_MODE != 2 && !HAS_SIDE && rd + 1 < 8) {
        const int mnext = mbase + ((rd + 1) >> 2) * 64 + ((rd + 1) & 3) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mm = mnext + h * 8 + rrow;
          side_prefetch(side[(rd + 1) & 1][h], p, flags, mm, n, mm < M && n_ok);
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ch = b * 8 + j * 4 + wq;  // 16-byte chunk of the 256-byte row
          *(f32x4*)(ep + wm * 256 + ((ch ^ wm) << 4)) = acc[a][b][i][j] * p.alpha;
        }
      __builtin_amdgcn_wave_barrier();
      if (OUT_MODE == 2) {
        // fp32 accumulate: one atomic wave-instruction = 64 consecutive floats of one row (256 contiguous bytes)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = *(const float*)(ep + r * 256 + (((lane >> 2) ^ r) << 4) + (lane & 3) * 4);
          if (m + r < M && n0 + lane < N) atomicAdd((float*)p.C + (long)(m + r) * p.ldc + n0 + lane, x);
        }
      } else {
        f32x4 q[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = h * 8 + rrow;
          q[h][0] = *(const f32x4*)(ep + row * 256 + (((2 * c8) ^ row) << 4));
          q[h][1] = *(const f32x4*)(ep + row * 256 + (((2 * c8 + 1) ^ row) << 4));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mm = m + h * 8 + rrow;
          if (mm < M && n_ok) {
            float v[8] = {q[h][0][0], q[h][0][1], q[h][0][2], q[h][0][3], q[h][1][0], q[h][1][1], q[h][1][2], q[h][1][3]};
            if (HAS_SIDE) {
              SideOps so;
              so.aux = so.res = (rd < 4) ? side_a[rd & 3][h] : side_b[rd & 3][h];
              epilogue8_pre(v, p, flags, mm, n, b0, b1, so);
            } else {
              epilogue8_pre(v, p, flags, mm, n, b0, b1, side[rd & 1][h]);
            }
            if (OUT_MODE == 0) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
              *(bf16x8*)((bf16*)p.C + (long)mm * p.ldc + n) = o;
            } else if (OUT_MODE == 3) {
              f16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = to_f16_sat(v[e]);
              *(f16x8*)((_Float16*)p.C + (long)mm * p.ldc + n) = o;
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
  Work cw, pw;            // compute / prefetch work items
  int cwi = blockIdx.x;   // their indices
  if (!get_work(cwi, cw) ) return;                    // uniform: whole workgroup leaves together
  while (cw.nk <= 0) {                                // (k_dev may leave a K-split empty)
    cwi += G;
    if (!get_work(cwi, cw)) return;
  }
  int pwi = cwi;
  pw = cw;
  int pk = 0;             // K tile (within pw) the prefetch cursor points at
  bool p_valid = true;
  set_sources(pw);
  // advance the prefetch cursor by one K tile (after its 4 half-tiles have been issued)
  auto advance_prefetch = [&]() {
    ++pk;
    pk_bytes += TN ? BK : BK * 2;
    if (pk >= pw.nk) {
      do {
        pwi += G;
        p_valid = get_work(pwi, pw);
      } while (p_valid && pw.nk <= 0);
      pk = 0;
      if (p_valid) set_sources(pw);
    }
  };

  load_bias_quads(cw);
  init_acc();
  // prologue: the whole first K tile
  issue(SLOT_A0, 0);
  issue(SLOT_B0, 0);
  issue(SLOT_B1, 0);
  issue(SLOT_A1, 0);
  advance_prefetch();
  wait_vm4();   // A-first, B-first landed (this wave's part)
  barrier();
  if (wr == 1) barrier();   // second wave-row runs one barrier behind

  // TN: the first column tile's wave-column 0 also produces the bias gradient of its 128 features
  bool want_bias = TN && p.bias != nullptr && cw.n0 == 0 && wc == 0;
  int buf = 0;    // stage holding the compute K tile
  // ---- tile boundary protocol. vmcnt retires in issue order, so a DMA wait placed AFTER the epilogue's stores in
  // program order would also wait for those stores (128 KiB per workgroup draining at the store path's pace). The
  // boundary therefore (E1) issues the WHOLE next-but-one K tile into the stage that has just been consumed, (E2)
  // waits for the next K tile to land, and only then (E3) issues the stores. The two K tiles after a boundary use
  // waits whose counts skip over the S stores (`mode` 1: no issue, one wait; `mode` 2: normal issue, waits +S); the
  // first wait that has to see the stores retired comes 7 phases after they were issued.
  const int S = (OUT_MODE == 0) ? (((EPI >= 0 ? EPI : p.flags) & STONK_EPI_SAVE_PREACT) ? 32 : 16)
                                : (OUT_MODE == 1 ? 32 : (OUT_MODE == 3 ? 16 : 0));
  auto wait_keep = [&](bool plus4) {   // at most S (+4) youngest operations may stay outstanding
    if (S == 16) { if (plus4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
    else if (S == 32) { if (plus4) asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); }
    else { if (plus4) wait_vm4(); else wait_vm0(); }
  };
  int mode = 0;           // 0 steady state, 1 / 2 = first / second K tile after a tile boundary
  bool ahead = false;     // mode 1: the K tile after this one was issued at the boundary
  for (;;) {      // one output tile (work item) per iteration; the DMA stream runs across iterations
    for (int ck = 0; ck < cw.nk; ++ck) {
      const int nb = buf ^ 1;
      const bool do_issue = p_valid && mode != 1;
      // ---------------- phase 0: quadrant (A0, B0)
      read_a(buf, SLOT_A0);
      read_b(buf, SLOT_B0, fb0);
      if (do_issue) issue(SLOT_A0, nb);
      if (mode == 0) { if (do_issue) wait_vm4(); else wait_vm0(); }            // retires B-second of this K tile
      else if (mode == 2) wait_keep(do_issue);
      barrier();
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      mma(acc[0][0], fb0);
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 1: quadrant (A0, B1)
      read_b(buf, SLOT_B1, fb1);
      if (do_issue) issue(SLOT_B0, nb);
      if (mode == 0) { if (do_issue) wait_vm4(); else wait_vm0(); }            // retires A-second of this K tile
      else if (mode == 2) wait_keep(do_issue);
      barrier();
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      mma(acc[0][1], fb1);
      if (TN && want_bias) bias_mma(0);
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 2: quadrant (A1, B1)
      read_a(buf, SLOT_A1);   // B1 fragments are still in registers
      if (do_issue) issue(SLOT_B1, nb);
      barrier();
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      mma(acc[1][1], fb1);
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- phase 3: quadrant (A1, B0): B0 fragments kept since phase 0 - no LDS read here, so a stage is
      // read for the last time in phase 2 and may be re-staged right at the tile boundary
      if (TN) read_b(buf, SLOT_B0, fb0);   // TN keeps no B0 fragments across phases 1-2 (register budget); it also
                                           // skips the boundary's early re-staging, so this late read is safe
      if (do_issue) issue(SLOT_A1, nb);
      if (mode == 1) { if (ahead) wait_keep(true); }                             // retires A-first, B-first of the next K tile
      else if (do_issue) wait_vm4();
      else wait_vm0();
      barrier();
      if (TN) wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      mma(acc[1][0], fb0);
      if (TN && want_bias) bias_mma(1);
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (do_issue) advance_prefetch();
      buf = nb;
      mode = (mode == 1) ? 2 : 0;
    }
    // ---- tile boundary: stage buf^1 (the K tile just finished) is free, stage buf holds the next K tile
    Work nw;
    bool more;
    do {
      cwi += G;
      more = get_work(cwi, nw);
    } while (more && nw.nk <= 0);
    if (more) load_bias_quads(nw);   // ahead of everything the boundary issues
    prefetch_side(cw, 0, side_a);
    ahead = p_valid && !TN;
    if (ahead) {   // E1
      issue(SLOT_A0, buf ^ 1);
      issue(SLOT_B0, buf ^ 1);
      issue(SLOT_B1, buf ^ 1);
      issue(SLOT_A1, buf ^ 1);
      advance_prefetch();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // E2: everything older than those 8 DMA has landed
    } else {
      wait_vm0();
    }
    prefetch_side(cw, 1, side_b);
    store_tile(cw);   // E3
    if (TN && want_bias) {   // lanes with lane>>4 == i hold the column sums of feature tile i (all four registers equal)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int m = cw.m0 + wr * 128 + a * 64 + (lane >> 4) * 16 + (lane & 15);
        if (m < M) atomicAdd((float*)p.bias + m, accb[a][0] * p.alpha);
        accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    mode = TN ? 0 : 1;
    if (!more) break;
    cw = nw;
    want_bias = TN && p.bias != nullptr && cw.n0 == 0 && wc == 0;
    init_acc();
  }
  if (wr == 0) barrier();   // balance the stagger barrier
}

}  // namespace

namespace {
template <int OUT_MODE, int EPI, bool TN = false>
int launch256(const GemmArgs& a, int grid, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<OUT_MODE, EPI, TN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<OUT_MODE, EPI, TN>), dim3(grid), dim3(512), LDS_BYTES, st, a);
  return stonk_launch_status();
}
}  // namespace

static int cu_count() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu = prop.multiProcessorCount;
  }
  return n_cu;
}

// weight-gradient form (see gemm_tn.hip for the contract): C fp32 [M,N] += alpha * A[K,M]^T . B[K,N], bias[M] += colsum
int stonk_gemm256_tn_launch(const GemmArgs& a, hipStream_t st) {
  const int n_cu = cu_count();
  if (n_cu <= 0) return (int)hipGetLastError();
  const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.split_k;
  const int grid = (int)(tiles < n_cu ? tiles : n_cu);
  return launch256<2, 0, true>(a, grid, st);
}

int stonk_gemm256_launch(const GemmArgs& a, int out_mode, hipStream_t st) {
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
                R = STONK_EPI_RESID, D = STONK_EPI_DROPOUT, AG = STONK_EPI_AUX_GRAD;
  const int epi = a.flags & (B | G | SV | GB | R | D | AG);
  if (out_mode == 1) return epi == 0 ? launch256<1, 0>(a, grid, st) : launch256<1, -1>(a, grid, st);
  if (out_mode == 2) return launch256<2, 0>(a, grid, st);
  if (out_mode == 3) return launch256<3, 0>(a, grid, st);
  switch (epi) {   // the combinations the STonKGs step uses are compiled with constant flags
    case 0: return launch256<0, 0>(a, grid, st);
    case B: return launch256<0, B>(a, grid, st);
    case B | G: return launch256<0, B | G>(a, grid, st);
    case B | G | SV: return launch256<0, B | G | SV>(a, grid, st);
    case GB: return launch256<0, GB>(a, grid, st);
    case B | G | SV | AG: return launch256<0, B | G | SV | AG>(a, grid, st);
    case GB | AG: return launch256<0, GB | AG>(a, grid, st);
    case R: return launch256<0, R>(a, grid, st);
    case B | R: return launch256<0, B | R>(a, grid, st);
    default: return launch256<0, -1>(a, grid, st);
  }
}
