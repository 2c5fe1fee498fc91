// Fused scaled-dot-product attention for the STonKGs encoder (head_dim 64, bf16 MFMA, fp32 softmax):
// forward (flash-style online softmax, never materialises the S x S scores) and backward (dQ kernel +
// dK/dV kernel, both recomputing P from the forward's log-sum-exp; no atomics, bitwise reproducible).
//
// Replaces hf:models/bert/modeling_bert.py BertSelfAttention :164-203 (eager path :111-136):
//   softmax(Q K^T / sqrt(d) + (1 - attention_mask) * -inf) -> dropout -> . V
// The key-padding mask is read straight from the int64 attention_mask [B,S] (0 = masked); the frozen LM
// backbone passes no mask (quirk Q5: ref:src/stonkgs/models/stonkgs_model.py:178).
//
// Layout: q/k/v are column slices of one [T, 3H] projection output (row stride `ld`), head h at columns
// h*64..h*64+63 - no head-major permute is ever written. MFMA orientation follows the CDNA4 playbook:
//  * forward and dQ compute S^T = K.Q^T (key index in the accumulator registers, query on the lane), so
//    softmax statistics are per-lane scalars and P^T feeds the next MFMA as B operand with no LDS trip;
//  * dK/dV compute S = Q.K^T (key on the lane) so P and dS feed dV^T / dK^T the same way.
// K^T / V^T / Q^T / dO^T operands come from row-major LDS tiles through ds_read_b64_tr_b16.
#include "common.h"
#include "stonk_flags.h"

namespace {

constexpr int HD = 64;        // head dim
constexpr int TK = 64;        // rows per LDS tile
constexpr int ROWB = 128;     // bytes per tile row
constexpr int TILEB = TK * ROWB;
constexpr int STAGEB = 2 * TILEB + 2 * TK * 4;   // two [64][64] bf16 tiles + two rows of 64 floats
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
// Additive key-padding bias in RAW score units, entered as the MFMA accumulator's initial value. A power of two: the
// products are absorbed (every masked score is exactly NEG_MASK, as the reference's finfo.min absorbs them) and
// NEG_MASK * scale is exact, so exp2(fma(s, scale, -max)) is exactly 1 for a row whose keys are all masked.
constexpr float NEG_MASK = -0x1p100f;
constexpr float NEG_INIT = -0x1p120f;

// The block multipliers of one row (four consecutive entries of STONK_C2_BLK) or of one column (every fourth), selected
// per lane at kernel entry from literals (three compares and twelve selects, no memory access).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ uint32_t pick4(int i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d));
}
__device__ __forceinline__ u32x4 c2_of_row(int row) {
  const int i = row & 3;
  u32x4 o;
#pragma unroll
  for (int u = 0; u < 4; ++u) o[u] = pick4(i, STONK_C2_BLK[u], STONK_C2_BLK[4 + u], STONK_C2_BLK[8 + u], STONK_C2_BLK[12 + u]);
  return o;
}
__device__ __forceinline__ u32x4 c2_of_col(int col) {
  const int i = col & 3;
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pick4(i, STONK_C2_BLK[4 * j], STONK_C2_BLK[4 * j + 1], STONK_C2_BLK[4 * j + 2], STONK_C2_BLK[4 * j + 3]);
  return o;
}

struct AttnArgs {
  const bf16* q;
  const bf16* k;
  const bf16* v;
  long ld;            // row stride (elements) of q/k/v
  const long* mask;   // [B,S] or null; with `cu`: one word per ROW of the packed layout (required)
  const int* cu;      // null, or [B+1] row offsets: sequence b = rows cu[b] .. cu[b+1]-1 of q/k/v/out/dout/dq/dk/dv (<= S rows)
  const int* qoff;    // null, or [B+1]: only the first qoff[b+1] - qoff[b] rows of sequence b are QUERIES (keys: all of them)
  int dq_writes_delta;   // the dQ kernel stores delta (default); 0 when attn_delta_kernel has, and dK/dV may be reading it
  bf16* out;          // fwd: context [T, ldo]; bwd: the forward's context (read)
  long ldo;
  float* lse;         // [B, NH, S] natural-log LSE of the scaled+masked scores
  // backward only
  const bf16* dout;   // [T, lddo]
  long lddo;
  float* delta;       // [B, NH, S] rowsum(dO * O): written by the dQ kernel, read by the dK/dV kernel
  bf16* dq;
  bf16* dk;
  bf16* dv;
  long ldd;           // row stride of dq/dk/dv
  int B, NH, S;
  float scale;
  uint32_t drop_thr32;
  float drop_scale;
  uint32_t seed;      // mixed (stonk_seed_mix)
};

// 16-byte chunk swizzle of a 128-byte tile row. Rows 2t and 2t+1 sit in opposite bank halves and share a swizzle; the
// swizzle is a bit rotation of t so that (a) 16 consecutive rows reading one logical chunk (ds_read_b128 fragments) hit
// 16 different bank groups and (b) rows r and r+2 of a transposed 4-row x 64-byte read (ds_read_b64_tr_b16) fall into
// different chunk halves - with the plain (row >> 1) & 7 they collided two ways (SQ_LDS_BANK_CONFLICT 20 % of LDS cycles).
__device__ __forceinline__ int swz(int row) {
  const int t = (row >> 1) & 7;
  return ((t & 1) << 2) | (t >> 1);
}
// byte offset of element (row, col) inside a [64][64] bf16 tile image
__device__ __forceinline__ int tile_off(int row, int col) {
  return row * ROWB + ((((col >> 3) ^ swz(row)) << 4) | ((col & 7) << 1));
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// Maximum of a 32x32 MFMA accumulator tile's 16 registers (and a running value) in eight v_max3_f32. fmaxf on MFMA
// results would make the compiler canonicalise every input first (one more VALU op per element), hence the asm - but the
// hazard recogniser does not see asm operands, so the statement carries its own wait for the MFMA that produced its
// inputs (an XDL write needs up to 19 wait states before a VALU read of the same registers).
__device__ __forceinline__ float max16_mfma(const f32x16& a) {
  float d;
  asm("s_nop 15\n\ts_nop 3\n\t"
      "v_max3_f32 %0, %1, %2, %3\n\t"
      "v_max3_f32 %0, %0, %4, %5\n\t"
      "v_max3_f32 %0, %0, %6, %7\n\t"
      "v_max3_f32 %0, %0, %8, %9\n\t"
      "v_max3_f32 %0, %0, %10, %11\n\t"
      "v_max3_f32 %0, %0, %12, %13\n\t"
      "v_max3_f32 %0, %0, %14, %15\n\t"
      "v_max3_f32 %0, %0, %16, %16"
      : "=&v"(d)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
        "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
  return d;
}
// same, folded into a running maximum (the compiler orders the two tiles' MFMAs freely, so this one waits as well)
__device__ __forceinline__ float max16_chain(float t, const f32x16& a) {
  float d;
  asm("s_nop 15\n\ts_nop 3\n\t"
      "v_max3_f32 %0, %17, %1, %2\n\t"
      "v_max3_f32 %0, %0, %3, %4\n\t"
      "v_max3_f32 %0, %0, %5, %6\n\t"
      "v_max3_f32 %0, %0, %7, %8\n\t"
      "v_max3_f32 %0, %0, %9, %10\n\t"
      "v_max3_f32 %0, %0, %11, %12\n\t"
      "v_max3_f32 %0, %0, %13, %14\n\t"
      "v_max3_f32 %0, %0, %15, %16"
      : "=&v"(d)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
        "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(t));
  return d;
}

// Packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two floats per issue slot) and the packed
// conversion (one v_cvt_pk_bf16_f32 per pair; element-wise casts into a bf16 vector cost a conversion AND a v_perm each).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ bf16x8 pack8(const float* e) {
  union {
    bf16x8 v;
    bf16x2 h[4];
  } u;
#pragma unroll
  for (int j = 0; j < 4; ++j) u.h[j] = __builtin_convertvector((f32x2){e[2 * j], e[2 * j + 1]}, bf16x2);
  return u.v;
}

// A/B fragment of a 32x32x16 MFMA read by rows: lane (r, hh) gets tile[row0 + r][16*st + 8*hh .. +7]
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row0, int st, int r, int hh) {
  return *(const bf16x8*)(tile + tile_off(row0 + r, 16 * st + 8 * hh));
}

// Transposed A fragment: lane (r = column c0 + (lane & 31), hh) gets tile[row0 + 8*(j>>2) + 4*hh + (j&3)][col]
// for j = 0..7: exactly the k order of an accumulator tile reused as B operand (guide section 3).
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
  const int row = row0 + 4 * (g >> 1) + (i >> 2);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + tile_off(row, col)));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + tile_off(row + 8, col)));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// stage a [64][64] bf16 tile: 512 16-byte chunks over 256 threads
struct Stage2 {
  bf16x8 c[2];
};
// rows row0 .. row0+63 of the sequence that starts at `base`, through a buffer that begins at the tile's first row and
// ends with the sequence's last one: rows past it arrive as ZEROS - finite values the callers mask out (a packed
// sequence's length need not be a multiple of the tile). The descriptor is scalar arithmetic per tile and the per-thread
// offsets (TileOff) are two loop-invariant registers per row stride: with per-thread 64-bit addresses the compiler
// rebuilt row * ld for every load of every tile - 13 v_lshl_add_u64, 8 v_mul_lo_u32 and 4 v_mad_u64_u32 per tile, a fifth
// of the forward kernel's vector instructions (tools/isa_mix.py).
struct TileOff {
  int o[2];
};
__device__ __forceinline__ TileOff tile_voff(long ld, int tid) {
  TileOff t;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    t.o[i] = (id >> 3) * (int)(ld * 2) + (id & 7) * 16;
  }
  return t;
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ void stage_load(Stage2& s, const bf16* base, long ld, int row0, int n, const TileOff& vo) {
#ifdef STONK_ATTN_ABLATE_LOADS   // timing experiment (tools/attn_probe.py): every tile re-reads the sequence's first - cache hits
  row0 = 0;
#endif
  row0 = __builtin_amdgcn_readfirstlane(row0);
  const int rows = n - row0 < TK ? n - row0 : TK;   // >= 1: the caller only asks for tiles that start inside the sequence
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)row0 * ld), 0, (rows - 1) * (int)(ld * 2) + ROWB, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) s.c[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, vo.o[i], 0, 0));
}
__device__ __forceinline__ void stage_store(const Stage2& s, char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    *(bf16x8*)(tile + tile_off(id >> 3, (id & 7) * 8)) = s.c[i];
  }
}

#ifdef STONK_ATTN_ABLATE_BARRIER   // timing experiment: the waves of a workgroup run free (races: garbage results)
#define TILE_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define TILE_BARRIER() __syncthreads()
#endif

// Every kernel below runs the same pipeline: tile t+1 is fetched into registers while tile t is computed from LDS
// stage t&1, written to the other stage when the compute is done, ONE barrier per tile.
//
// Dropout: element (query row, key), row = the flat (b,h,q) index, is kept iff
// stonk_blk_keep(stonk_blk_round1(rowkey(row >> 2), colkey(key >> 2)), C2_BLK[4 (row & 3) + (key & 3)]) - one first hash
// round per 4 x 4 BLOCK of scores (common.h). Both keys are linear, so the kernels with the query on the lane add a
// compile-time constant to the tile's key-quad key per four elements and the kernel with the key on the lane does the same
// with the row-quad key; either keeps its four multipliers in registers. The 1/(1-p) factor never touches an
// element: it is folded into the output normalisation (forward, dV) or into delta and the final scale (dQ, dK).

// Workgroup -> (128-row block, head, sequence). Workgroups are dispatched round-robin over the 8 XCDs in linear order, so
// with the plain grid mapping the S/128 blocks of one head (which all read that head's K and V, or Q and dO) land on
// different XCDs and each L2 fetches the head again: 455 MB of fabric reads per forward launch against 151 MB of Q/K/V
// (rocprofv3 FETCH_SIZE). Here every XCD takes one contiguous run of the linear (sequence, head, block) order, so a
// head's blocks share an L2 and run at the same time.
//
// Which (sequence, head) pairs an XCD gets matters as well once fully masked key tiles are skipped: sequences differ in
// length, and a contiguous run of sequences per XCD leaves the XCD with the longest ones working alone at the end (a batch
// sorted by length: no gain from the skipping at all). When the pairs divide by 8, XCD x takes pairs x, x + 8, ... - one
// or two heads of EVERY sequence.
struct AttnBlock { int xb, h, b; };
__device__ __forceinline__ AttnBlock attn_block() {
  const int nx = gridDim.x, nh = gridDim.y;
  const int n = nx * nh * gridDim.z;
  const int lin = blockIdx.x + nx * (blockIdx.y + nh * blockIdx.z);
  const int q = n >> 3, rm = n & 7, x = lin & 7;
  int l = ((x < rm) ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + (lin >> 3);
  if (((nh * gridDim.z) & 7) == 0) {
    const int j = lin >> 3;
    l = ((j / nx) * 8 + x) * nx + j % nx;
  }
  AttnBlock o;
  o.xb = l % nx;
  const int t = l / nx;
  o.h = t % nh;
  o.b = t / nh;
  return o;
}

// Rows of the block's sequence: [tok0, tok0 + n). Padded layout: n = S. Packed layout (p.cu): the sequence's own extent -
// a 128-row block that starts past it has nothing to do and leaves (the grid is sized for S rows per sequence).
#define SEQ_EXTENT()                                                     \
  long tok0 = (long)b * S;                                               \
  int n = S;                                                             \
  if (p.cu) {                                                            \
    tok0 = p.cu[b];                                                      \
    n = p.cu[b + 1] - p.cu[b];                                           \
    n = n < S ? n : S;                                                   \
  }                                                                      \
  if (blk.xb * 128 >= n) return

// Query rows of the block's sequence: all n, or - the last encoder layer, whose output is read at a sequence's first rows
// only (stonk_unpad_plan puts the read rows there) - its first p.qoff[b+1] - p.qoff[b].
#define QUERY_EXTENT()                          \
  int nq = n;                                   \
  if (p.qoff) {                                 \
    const int lim = p.qoff[b + 1] - p.qoff[b];  \
    nq = lim < n ? lim : n;                     \
  }

// Key tiles (64 keys) of a sequence that hold at least one unmasked key, one bit per tile: a fully masked tile adds
// exactly nothing to any query (its scores are -2^100 in raw units, their exp2 is 0), so the kernels with the keys in the
// tile loop walk the set bits only - in the STonKGs layout the padding of the text half, 112 of 512 positions on average
// in the benchmark's batches. In two halves, so that the mask words travel while the kernel's other first loads do (asked
// for first, they are also the first to arrive): live_issue() requests them, live_finish() returns the bits - 0 when NO
// key of the sequence is unmasked (the reference then attends uniformly: the callers walk every tile), all ones without a
// mask or beyond 1024 keys. live_finish() contains two barriers.
struct LiveTiles {
  long mv[4];
};
template <bool HAS_MASK>
__device__ __forceinline__ void live_issue(LiveTiles& t, const long* mask, long tok0, int S, int tid) {
  if (!HAS_MASK || S > 1024) return;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int key = c * 256 + tid;   // the 64 lanes of a wave look at one tile
    t.mv[c] = key < S ? mask[tok0 + key] : 0;   // (S: the rows of THIS sequence)
  }
}
// With `kbias` (forward, dQ): the additive bias of EVERY key of the sequence (0, or NEG_MASK for a masked key and for the
// keys past the sequence's end) goes to LDS here, once - the tile loop reads it from there instead of loading and
// converting a tile's 64 mask words with every tile (a load per tile whose 64-bit word also cost the forward kernel
// the two registers that pushed it into scratch).
template <bool HAS_MASK>
__device__ __forceinline__ uint64_t live_finish(const LiveTiles& t, const long* mask, long tok0, int n, int S, int tid,
                                                float* kbias) {
  const int nt = (n + TK - 1) / TK;
  const uint64_t all = nt >= 64 ? ~0ull : ((1ull << nt) - 1);
  if (!HAS_MASK) return all;
  if (S > 1024) {   // (no tile skipping beyond 1024 keys: the words are only turned into the bias)
    if (kbias) {
      for (int key = tid; key < S; key += 256) kbias[key] = (key < n && mask[tok0 + key] != 0) ? 0.f : NEG_MASK;
      __syncthreads();
    }
    return all;
  }
  __shared__ int tile_live[16];
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint64_t b = __ballot(t.mv[c] != 0);
    if (lane == 0) tile_live[c * 4 + wave] = b != 0;
    if (kbias && c * 256 + tid < S) kbias[c * 256 + tid] = t.mv[c] != 0 ? 0.f : NEG_MASK;
  }
  __syncthreads();
  const uint64_t m = __ballot(lane < nt && tile_live[lane & 15] != 0);
  __syncthreads();
  return m;
}

// ------------------------------------------------------------------ forward
template <bool HAS_MASK, bool DROPOUT>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGEB];
  extern __shared__ __attribute__((aligned(16))) float kbias[];   // [S] with a mask (dynamic: the launcher sizes it)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const AttnBlock blk = attn_block();
  const int b = blk.b, h = blk.h;
  const int S = p.S;
  const int q0 = blk.xb * 128 + wave * 32;
  SEQ_EXTENT();
  QUERY_EXTENT();
  if (blk.xb * 128 >= nq) return;
  const float sc2 = p.scale * LOG2E;
  LiveTiles lt;
  live_issue<HAS_MASK>(lt, p.mask, tok0, n, tid);

  const int qr = q0 + r < n ? q0 + r : n - 1;   // query rows past the sequence re-read its last row; nothing is stored for them
  bf16x8 qf[4];
  {
    const bf16* qrow = p.q + (tok0 + qr) * p.ld + h * HD;
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8*)(qrow + 16 * st + 8 * hh);
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;
  float m = NEG_INIT, l = 0.f;   // m in scaled log2 units

  const bf16* kbase = p.k + tok0 * p.ld + h * HD;
  const bf16* vbase = p.v + tok0 * p.ld + h * HD;
  const int ntiles = (n + TK - 1) / TK;
  const uint32_t rk = stonk_rowkey((uint32_t)((b * p.NH + h) * S + q0 + r) >> 2, p.seed);   // row QUAD of this lane's query
  const uint32_t ck_lane = stonk_colkey((uint32_t)hh);   // key QUADS: this lane's keys start at 4 hh = quad hh
  u32x4 c2u = {0u, 0u, 0u, 0u};   // the block multipliers of this lane's row (q0 and the flat (b,h) offset are multiples of 4)
  if (DROPOUT) c2u = c2_of_row(r);

  Stage2 sk, sv;
  const TileOff vo = tile_voff(p.ld, tid);
  auto load_tile = [&](int kt) {
    stage_load(sk, kbase, p.ld, kt * TK, n, vo);
    stage_load(sv, vbase, p.ld, kt * TK, n, vo);
  };
  auto store_tile = [&](int buf) {
    char* base = lds + buf * STAGEB;
    stage_store(sk, base, tid);
    stage_store(sv, base + TILEB, tid);
  };
  load_tile(0);   // (before the live bits are known: tile 0 almost always is - [CLS])
  uint64_t rem = live_finish<HAS_MASK>(lt, p.mask, tok0, n, S, tid, kbias);
  if (rem == 0) rem = ntiles >= 64 ? ~0ull : ((1ull << ntiles) - 1);
  int kt = __builtin_ctzll(rem);
  rem &= rem - 1;
  if (kt != 0) load_tile(kt);
  store_tile(0);
  __syncthreads();

  for (int it = 0;; ++it) {   // the live key tiles, in order
    const char* Ks = lds + (it & 1) * STAGEB;
    const char* Vs = Ks + TILEB;
    const float* Mb = kbias + kt * TK;   // (the sequence's key bias, written once by live_finish)
    const int nx = rem ? __builtin_ctzll(rem) : -1;
    rem &= rem - 1;
    if (nx >= 0) load_tile(nx);
    // S^T = K . Q^T (+ key bias through the accumulator) for the two 32-key halves of the tile, raw units
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 mb = {0.f, 0.f, 0.f, 0.f};
        if (HAS_MASK) mb = *(const f32x4*)(Mb + sub * 32 + 8 * g + 4 * hh);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[sub][4 * g + j] = mb[j];
      }
#pragma unroll
      for (int st = 0; st < 4; ++st) s[sub] = mfma32(row_frag(Ks, sub * 32, st, r, hh), qf[st], s[sub]);
    }
    float tmax = max16_chain(max16_mfma(s[1]), s[0]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax * sc2);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
    const float negm = -m_new;
    f32x2 rs2 = {0.f, 0.f};
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const f32x2 a = pk_fma((f32x2){s[sub][i], s[sub][i + 1]}, (f32x2){sc2, sc2}, (f32x2){negm, negm});
        const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        s[sub][i] = e[0];
        s[sub][i + 1] = e[1];
        rs2 += e;
      }
    float rs = rs2[0] + rs2[1];
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    if (!__all(alpha == 1.f)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] *= alpha;
        o[1][i] *= alpha;
      }
    }
    // O^T += V^T . P^T
    const uint32_t ckt = ck_lane + (uint32_t)(kt * TK / 4) * STONK_G_COL;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; j += 4) {   // elements j .. j + 3 are the keys 4m .. 4m + 3 of one quad
          uint32_t y8 = 0;
          if (DROPOUT) {
            const uint32_t cj = (uint32_t)((sub * 32 + 16 * ks + 8 * (j >> 2)) >> 2) * STONK_G_COL;
            y8 = stonk_blk_round1(rk, ckt + cj);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            e[j + u] = s[sub][8 * ks + j + u];
            if (DROPOUT) e[j + u] = stonk_blk_keep(y8, c2u[u], p.drop_thr32) ? e[j + u] : 0.f;
          }
        }
        const bf16x8 pf = pack8(e);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(tr_frag(Vs, sub * 32 + 16 * ks, dt * 32, lane), pf, o[dt]);
      }
    if (nx >= 0) store_tile((it + 1) & 1);
    TILE_BARRIER();
    if (nx < 0) break;
    kt = nx;
  }
  const float inv = (DROPOUT ? p.drop_scale : 1.f) / l;
  if (q0 + r >= nq) return;
  bf16* orow = p.out + (tok0 + q0 + r) * p.ldo + h * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v = {(bf16)(o[dt][4 * g] * inv), (bf16)(o[dt][4 * g + 1] * inv), (bf16)(o[dt][4 * g + 2] * inv),
                  (bf16)(o[dt][4 * g + 3] * inv)};
      *(bf16x4*)(orow + dt * 32 + 8 * g + 4 * hh) = v;
    }
  if (hh == 0 && p.lse) p.lse[(long)(b * p.NH + h) * S + q0 + r] = (m + __builtin_amdgcn_logf(l)) * LN2;
}

// ------------------------------------------------------------------ backward: delta = rowsum(dO * O) on its own
// (what lets the dQ and the dK/dV kernels run side by side on two streams: the dQ kernel otherwise produces it)
__global__ __launch_bounds__(128) void attn_delta_kernel(const AttnArgs p) {
  const AttnBlock blk = attn_block();
  const int b = blk.b, h = blk.h;
  const int S = p.S;
  SEQ_EXTENT();
  QUERY_EXTENT();
  const int q = blk.xb * 128 + threadIdx.x;
  if (q >= nq) return;
  const bf16* drow = p.dout + (tok0 + q) * p.lddo + h * HD;
  const bf16* orow = p.out + (tok0 + q) * p.ldo + h * HD;
  float dlt = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bf16x8 d = *(const bf16x8*)(drow + 8 * c), o = *(const bf16x8*)(orow + 8 * c);
#pragma unroll
    for (int j = 0; j < 8; ++j) dlt += (float)o[j] * (float)d[j];
  }
  p.delta[(long)(b * p.NH + h) * S + q] = dlt;
}

// ------------------------------------------------------------------ backward: dQ (query on the lane)
// Also produces delta = rowsum(dO * O) for its query rows (the dK/dV kernel, launched after it, reads it).
template <bool HAS_MASK, bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGEB];
  extern __shared__ __attribute__((aligned(16))) float kbias[];   // [S] with a mask
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const AttnBlock blk = attn_block();
  const int b = blk.b, h = blk.h;
  const int S = p.S;
  const int q0 = blk.xb * 128 + wave * 32;
  SEQ_EXTENT();
  QUERY_EXTENT();
  if (blk.xb * 128 >= nq) return;
  LiveTiles lt;
  live_issue<HAS_MASK>(lt, p.mask, tok0, n, tid);

  const bool qlive = q0 + r < nq;
  const int qr = qlive ? q0 + r : n - 1;   // rows past the sequence re-read its last row; nothing is stored for them
  bf16x8 qf[4], dof[4];
  float dlt = 0.f;
  {
    const bf16* qrow = p.q + (tok0 + qr) * p.ld + h * HD;
    const bf16* drow = p.dout + (tok0 + qr) * p.lddo + h * HD;
    const bf16* orow = p.out + (tok0 + qr) * p.ldo + h * HD;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      qf[st] = *(const bf16x8*)(qrow + 16 * st + 8 * hh);
      dof[st] = *(const bf16x8*)(drow + 16 * st + 8 * hh);
      const bf16x8 of = *(const bf16x8*)(orow + 16 * st + 8 * hh);
#pragma unroll
      for (int j = 0; j < 8; ++j) dlt += (float)of[j] * (float)dof[st][j];
    }
  }
  dlt += __shfl_xor(dlt, 32, 64);
  const long stat = (long)(b * p.NH + h) * S + q0 + r;
  if (hh == 0 && qlive && p.dq_writes_delta) p.delta[stat] = dlt;
  const float lse_q = p.lse[stat - (q0 + r) + qr];
  // dS = P * (drop(dP) - delta) with drop(dP) = keep ? dP / (1-p) : 0  ==  (1/(1-p)) * P * ((keep ? dP : 0) - (1-p) delta)
  const float dlt_s = DROPOUT ? dlt / p.drop_scale : dlt;
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) dq[0][i] = dq[1][i] = 0.f;

  const bf16* kbase = p.k + tok0 * p.ld + h * HD;
  const bf16* vbase = p.v + tok0 * p.ld + h * HD;
  const int ntiles = (n + TK - 1) / TK;
  const uint32_t rk = stonk_rowkey((uint32_t)stat >> 2, p.seed);   // row quad, key quads and multipliers as in the forward
  const uint32_t ck_lane = stonk_colkey((uint32_t)hh);
  u32x4 c2u = {0u, 0u, 0u, 0u};
  if (DROPOUT) c2u = c2_of_row(r);

  Stage2 sk, sv;
  const TileOff vo = tile_voff(p.ld, tid);
  auto load_tile = [&](int kt) {
    stage_load(sk, kbase, p.ld, kt * TK, n, vo);
    stage_load(sv, vbase, p.ld, kt * TK, n, vo);
  };
  auto store_tile = [&](int buf) {
    char* base = lds + buf * STAGEB;
    stage_store(sk, base, tid);
    stage_store(sv, base + TILEB, tid);
  };
  load_tile(0);   // (before the live bits are known)
  uint64_t rem = live_finish<HAS_MASK>(lt, p.mask, tok0, n, S, tid, kbias);
  // No unmasked key in the whole sequence: the reference's finfo.min absorbs every score and it attends uniformly, P = 1/S.
  // The forward gets there by the same absorption; its log-sum-exp (-2^100-sized) cannot carry log S, so the backward
  // kernels rebuild P from a zero score scale and lse = log S. (Never the case in a STonKGs batch - [CLS] is always live.)
  const bool uniform = rem == 0;
  if (uniform) rem = ntiles >= 64 ? ~0ull : ((1ull << ntiles) - 1);
  const float sc2 = uniform ? 0.f : p.scale * LOG2E;
  const float nlse2 = uniform ? -__builtin_amdgcn_logf((float)n) : -lse_q * LOG2E;   // (v_log_f32 is log2)
  int kt = __builtin_ctzll(rem);
  rem &= rem - 1;
  if (kt != 0) load_tile(kt);
  store_tile(0);
  __syncthreads();

  for (int it = 0;; ++it) {   // the live key tiles, in order (a fully masked tile gives dS = 0)
    const char* Ks = lds + (it & 1) * STAGEB;
    const char* Vs = Ks + TILEB;
    const float* Mb = kbias + kt * TK;   // (the sequence's key bias, written once by live_finish)
    const int nx = rem ? __builtin_ctzll(rem) : -1;
    rem &= rem - 1;
    if (nx >= 0) load_tile(nx);
    const uint32_t ckt = ck_lane + (uint32_t)(kt * TK / 4) * STONK_G_COL;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 mb = {0.f, 0.f, 0.f, 0.f};
        if (HAS_MASK) mb = *(const f32x4*)(Mb + sub * 32 + 8 * g + 4 * hh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[4 * g + j] = mb[j];
          dp[4 * g + j] = 0.f;
        }
      }
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        s = mfma32(row_frag(Ks, sub * 32, st, r, hh), qf[st], s);      // S^T[k][q] (+ key bias)
        dp = mfma32(row_frag(Vs, sub * 32, st, r, hh), dof[st], dp);   // dP^T[k][q] = sum_d V[k][d] dO[q][d]
      }
      uint32_t y8q = 0;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const f32x2 a = pk_fma((f32x2){s[i], s[i + 1]}, (f32x2){sc2, sc2}, (f32x2){nlse2, nlse2});
        const f32x2 pr = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        f32x2 dpv = {dp[i], dp[i + 1]};
        if (DROPOUT) {
          if ((i & 2) == 0) y8q = stonk_blk_round1(rk, ckt + (uint32_t)((sub * 32 + 8 * (i >> 2)) >> 2) * STONK_G_COL);
          dpv[0] = stonk_blk_keep(y8q, c2u[i & 2], p.drop_thr32) ? dpv[0] : 0.f;
          dpv[1] = stonk_blk_keep(y8q, c2u[(i & 2) + 1], p.drop_thr32) ? dpv[1] : 0.f;
        }
        const f32x2 ds = pr * (dpv - (f32x2){dlt_s, dlt_s});  // dS^T (up to the folded 1/(1-p))
        s[i] = ds[0];
        s[i + 1] = ds[1];
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = s[8 * ks + j];
        const bf16x8 dsf = pack8(e);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(tr_frag(Ks, sub * 32 + 16 * ks, dt * 32, lane), dsf, dq[dt]);
      }
    }
    if (nx >= 0) store_tile((it + 1) & 1);
    TILE_BARRIER();
    if (nx < 0) break;
    kt = nx;
  }
  const float fs = DROPOUT ? p.scale * p.drop_scale : p.scale;
  if (!qlive) return;
  bf16* orow = p.dq + (tok0 + q0 + r) * p.ldd + h * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v = {(bf16)(dq[dt][4 * g] * fs), (bf16)(dq[dt][4 * g + 1] * fs), (bf16)(dq[dt][4 * g + 2] * fs),
                  (bf16)(dq[dt][4 * g + 3] * fs)};
      *(bf16x4*)(orow + dt * 32 + 8 * g + 4 * hh) = v;
    }
}

// ------------------------------------------------------------------ backward: dK, dV (key on the lane)
template <bool HAS_MASK, bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGEB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const AttnBlock blk = attn_block();
  const int b = blk.b, h = blk.h;
  const int S = p.S;
  const int k0 = blk.xb * 128 + wave * 32;
  SEQ_EXTENT();
  QUERY_EXTENT();
  LiveTiles lt;
  live_issue<HAS_MASK>(lt, p.mask, tok0, n, tid);

  const bool klive = k0 + r < n;
  const int kr = klive ? k0 + r : n - 1;   // keys past the sequence re-read its last row, count as masked, store nothing
  bf16x8 kf[4], vf[4];
  {
    const bf16* krow = p.k + (tok0 + kr) * p.ld + h * HD;
    const bf16* vrow = p.v + (tok0 + kr) * p.ld + h * HD;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      kf[st] = *(const bf16x8*)(krow + 16 * st + 8 * hh);
      vf[st] = *(const bf16x8*)(vrow + 16 * st + 8 * hh);
    }
  }
  float mb = 0.f;
  if (HAS_MASK) mb = (klive && p.mask[tok0 + kr] != 0) ? 0.f : NEG_MASK;
  // Masked keys get exactly zero gradient (P = 0 for every query), unless NO key of the sequence is unmasked (the reference
  // then attends uniformly): a wave whose 32 keys are all masked skips its arithmetic - its accumulators stay zero and it
  // only helps to stage the tiles - and a workgroup whose 128 keys are all masked writes its zeros and leaves.
  bool uniform = false;   // no unmasked key at all: P = 1/S, see the dQ kernel (set below, once the first loads are under way)
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f;

  const bf16* qbase = p.q + tok0 * p.ld + h * HD;
  const bf16* dbase = p.dout + tok0 * p.lddo + h * HD;
  const long statbase = (long)(b * p.NH + h) * S;
  const int ntiles = (nq + TK - 1) / TK;   // QUERY tiles
  // row QUADS: an accumulator group g of this lane holds the query rows 8 g + 4 hh .. + 3 of a 32-row block = one quad
  const uint32_t rk_lane = stonk_rowkey((uint32_t)(statbase >> 2) + (uint32_t)hh, p.seed);
  const uint32_t ck = stonk_colkey((uint32_t)((k0 + r) >> 2));   // this lane's key: its quad ...
  u32x4 c2j = {0u, 0u, 0u, 0u};                                  // ... and the block multipliers of its column
  if (DROPOUT) c2j = c2_of_col(r);
  const float inv_ds = DROPOUT ? 1.f / p.drop_scale : 1.f;

  Stage2 sq, sd;
  // row statistics of a tile: threads 0..63 carry -lse (log2 units), 64..127 -(1-p) * delta. The loaded word is only
  // touched when the tile is stored (arithmetic at the load would wait for it, and for the tile loads issued before it,
  // at the top of every iteration).
  float sreg = 0.f;
  bool slive = true;   // the statistics word in flight belongs to a query of this sequence
  const float* sptr = tid < TK ? p.lse + statbase + tid : p.delta + statbase + tid - TK;
  const float sfac = tid < TK ? -LOG2E : -inv_ds;
  const float nlog2s = -__builtin_amdgcn_logf((float)n);
  const TileOff voq = tile_voff(p.ld, tid), vod = tile_voff(p.lddo, tid);
  auto load_tile = [&](int qt) {
    stage_load(sq, qbase, p.ld, qt * TK, n, voq);
    stage_load(sd, dbase, p.lddo, qt * TK, n, vod);
    if (tid < 2 * TK) {
      slive = qt * TK + (tid & (TK - 1)) < nq;
      sreg = sptr[qt * TK];      // (inside the [B,NH,S] statistics arrays also past the sequence: S % 128 == 0)
    }
  };
  // a query row past the sequence's end (its Q / dO arrive as zeros) gets -lse = NEG_MASK, which makes its
  // P exactly zero for every key, and delta = 0: it adds nothing to dK and dV
  auto store_tile = [&](int buf) {
    char* base = lds + buf * STAGEB;
    stage_store(sq, base, tid);
    stage_store(sd, base + TILEB, tid);
    if (tid < 2 * TK)
      ((float*)(base + 2 * TILEB))[tid] = !slive ? (tid < TK ? NEG_MASK : 0.f) : (uniform && tid < TK) ? nlog2s : sreg * sfac;
  };
  load_tile(0);   // (before the live bits are known; a workgroup that then leaves has asked for one tile in vain)
  const uint64_t seq = live_finish<HAS_MASK>(lt, p.mask, tok0, n, S, tid, nullptr);
  uniform = seq == 0;
  const float sc2 = uniform ? 0.f : p.scale * LOG2E;
  bool wave_live = true, wg_live = true;
  if (HAS_MASK) {
    wave_live = uniform || __ballot(mb == 0.f) != 0;
    wg_live = uniform || ((seq >> (2 * blk.xb)) & 3) != 0;
  }
  if (wg_live) store_tile(0);
  __syncthreads();

  for (int qt = 0; qt < (wg_live ? ntiles : 0); ++qt) {
    const char* Qs = lds + (qt & 1) * STAGEB;
    const char* Ds = Qs + TILEB;
    const float* Ls = (const float*)(Qs + 2 * TILEB);
    const float* Dl = Ls + TK;
    if (qt + 1 < ntiles) load_tile(qt + 1);
    const uint32_t rkt = rk_lane + (uint32_t)(qt * TK / 4) * STONK_G_ROW;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (!wave_live) break;
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i] = mb;
        dp[i] = 0.f;
      }
      // LDS operands are requested a phase ahead of the MFMAs that use them, and the scheduler is told to leave them
      // there: left alone it sinks every read next to its MFMA (fewer live registers) and each of the tile's 32 MFMAs
      // then waits out an LDS round trip - with two waves per SIMD nobody hides that (the loop was load / wait / MFMA).
      bf16x8 qa[4], da[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        qa[st] = row_frag(Qs, sub * 32, st, r, hh);
        da[st] = row_frag(Ds, sub * 32, st, r, hh);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        s = mfma32(qa[st], kf[st], s);    // S[q][k] (+ key bias)
        dp = mfma32(da[st], vf[st], dp);  // dP[q][k] = sum_d dO[q][d] V[k][d]
      }
      // the transposed operands of dV^T / dK^T: requested now, used after the softmax arithmetic below
      bf16x8 td[2][2], tq[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          td[ks][dt] = tr_frag(Ds, sub * 32 + 16 * ks, dt * 32, lane);
          tq[ks][dt] = tr_frag(Qs, sub * 32 + 16 * ks, dt * 32, lane);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 nls = *(const f32x4*)(Ls + sub * 32 + 8 * g + 4 * hh);
        const f32x4 dl = *(const f32x4*)(Dl + sub * 32 + 8 * g + 4 * hh);   // -(1-p) * delta
        uint32_t y8 = 0;
        if (DROPOUT) y8 = stonk_blk_round1(rkt + (uint32_t)(sub * 8 + 2 * g) * STONK_G_ROW, ck);
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const int i = 4 * g + j;
          const f32x2 a = pk_fma((f32x2){s[i], s[i + 1]}, (f32x2){sc2, sc2}, (f32x2){nls[j], nls[j + 1]});
          const f32x2 pr = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          f32x2 pd = pr;
          if (DROPOUT) {
#pragma unroll
            for (int u = 0; u < 2; ++u) pd[u] = stonk_blk_keep(y8, c2j[j + u], p.drop_thr32) ? pr[u] : 0.f;
          }
          // dS = P (drop(dP) - delta) = dropped P . dP + P . (-delta): one select per score instead of two
          const f32x2 ds = pk_fma(pd, (f32x2){dp[i], dp[i + 1]}, pr * (f32x2){dl[j], dl[j + 1]});
          s[i] = pd[0];                    // dropped P (feeds dV), up to the folded 1/(1-p)
          s[i + 1] = pd[1];
          dp[i] = ds[0];                   // dS (feeds dK), up to the folded 1/(1-p)
          dp[i + 1] = ds[1];
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float e[8], f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          e[j] = s[8 * ks + j];
          f[j] = dp[8 * ks + j];
        }
        const bf16x8 pf = pack8(e), dsf = pack8(f);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = mfma32(td[ks][dt], pf, dv[dt]);   // dV^T += dO^T . P
          dk[dt] = mfma32(tq[ks][dt], dsf, dk[dt]);  // dK^T += Q^T . dS
        }
      }
    }
    if (qt + 1 < ntiles) store_tile((qt + 1) & 1);
    TILE_BARRIER();
  }
  const float fk = DROPOUT ? p.scale * p.drop_scale : p.scale;
  const float fv = DROPOUT ? p.drop_scale : 1.f;
  if (!klive) return;
  bf16* krow = p.dk + (tok0 + k0 + r) * p.ldd + h * HD;
  bf16* vrow = p.dv + (tok0 + k0 + r) * p.ldd + h * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 a = {(bf16)(dk[dt][4 * g] * fk), (bf16)(dk[dt][4 * g + 1] * fk), (bf16)(dk[dt][4 * g + 2] * fk),
                  (bf16)(dk[dt][4 * g + 3] * fk)};
      bf16x4 c = {(bf16)(dv[dt][4 * g] * fv), (bf16)(dv[dt][4 * g + 1] * fv), (bf16)(dv[dt][4 * g + 2] * fv),
                  (bf16)(dv[dt][4 * g + 3] * fv)};
      *(bf16x4*)(krow + dt * 32 + 8 * g + 4 * hh) = a;
      *(bf16x4*)(vrow + dt * 32 + 8 * g + 4 * hh) = c;
    }
}

int check_common(const void* q, const void* k, const void* v, int64_t ld, int B, int NH, int S, int D) {
  STONK_CHECK_ARG(q && k && v, STONK_EINVAL);
  STONK_CHECK_ARG(D == HD, STONK_ESHAPE);
  // (the kernels keep one bit per 64-key tile of a sequence in a 64-bit word: 4096 keys)
  STONK_CHECK_ARG(B >= 0 && NH > 0 && S > 0 && S % 128 == 0 && S <= 4096, STONK_ESHAPE);
  STONK_CHECK_ARG(ld % 8 == 0, STONK_EALIGN);
  STONK_CHECK_ARG((uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0, STONK_EALIGN);
  STONK_CHECK_ARG((long)B * NH * S < (1L << 32), STONK_ESHAPE);  // 32-bit dropout row counter
  return STONK_OK;
}

}  // namespace

extern "C" int stonk_attention_fwd(const void* q, const void* k, const void* v, int64_t ld,
                                   const int64_t* attention_mask, const int* seq_offsets, const int* q_offsets, void* out,
                                   int64_t ldo, float* lse, int B, int NH, int S, int D, float scale, float drop_p,
                                   uint32_t seed, void* stream) {
  int rc = check_common(q, k, v, ld, B, NH, S, D);
  if (rc) return rc;
  STONK_CHECK_ARG(out && ldo % 4 == 0, STONK_EINVAL);
  STONK_CHECK_ARG(!seq_offsets || attention_mask, STONK_EINVAL);   // packed rows carry their key mask
  STONK_CHECK_ARG(!q_offsets || seq_offsets, STONK_EINVAL);
  STONK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, STONK_EINVAL);
  if (B == 0) return STONK_OK;
  AttnArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ld = ld;
  a.mask = (const long*)attention_mask; a.cu = seq_offsets; a.qoff = q_offsets; a.out = (bf16*)out; a.ldo = ldo; a.lse = lse;
  a.B = B; a.NH = NH; a.S = S; a.scale = scale;
  a.drop_thr32 = stonk_drop_thr32(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.seed = stonk_seed_mix(seed);
  const dim3 grid(S / 128, NH, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool hm = attention_mask != nullptr, dr = drop_p > 0.f;
  const size_t kb = (size_t)S * sizeof(float);   // the sequence's key bias
  if (hm && dr) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, block, kb, st, a);
  else if (hm) hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, block, kb, st, a);
  else if (dr) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, block, 0, st, a);
  return stonk_launch_status();
}

static int attention_bwd_launch(int phases, const void* q, const void* k, const void* v, int64_t ld,
                                const int64_t* attention_mask, const int* seq_offsets, const int* q_offsets, const void* out,
                                int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta_ws, void* dq,
                                void* dk, int64_t ldd, void* dv, int B, int NH, int S, int D, float scale, float drop_p,
                                uint32_t seed, void* stream) {
  int rc = check_common(q, k, v, ld, B, NH, S, D);
  if (rc) return rc;
  STONK_CHECK_ARG(out && dout && lse && delta_ws && dq && dk && dv, STONK_EINVAL);
  STONK_CHECK_ARG(!seq_offsets || attention_mask, STONK_EINVAL);
  STONK_CHECK_ARG(!q_offsets || seq_offsets, STONK_EINVAL);
  STONK_CHECK_ARG(ldo % 8 == 0 && lddo % 8 == 0 && ldd % 4 == 0, STONK_EALIGN);
  STONK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, STONK_EINVAL);
  if (B == 0) return STONK_OK;
  AttnArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ld = ld;
  a.mask = (const long*)attention_mask; a.cu = seq_offsets; a.qoff = q_offsets; a.lse = (float*)lse; a.out = (bf16*)out; a.ldo = ldo;
  a.dout = (const bf16*)dout; a.lddo = lddo; a.delta = delta_ws;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.ldd = ldd;
  a.B = B; a.NH = NH; a.S = S; a.scale = scale;
  a.drop_thr32 = stonk_drop_thr32(drop_p); a.drop_scale = 1.f / (1.f - drop_p); a.seed = stonk_seed_mix(seed);
  a.dq_writes_delta = phases == STONK_ATTN_BWD_ALL;   // (split calls: the DELTA phase produces it, before DQ and DKV)
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(S / 128, NH, B), block(256);
  const bool hm = attention_mask != nullptr, dr = drop_p > 0.f;
  if ((phases & STONK_ATTN_BWD_DELTA) && phases != STONK_ATTN_BWD_ALL)
    hipLaunchKernelGGL(attn_delta_kernel, grid, dim3(128), 0, st, a);
#define LAUNCH_BWD(HM, DR)                                                                                         \
  do {                                                                                                             \
    if (phases & STONK_ATTN_BWD_DQ)                                                                                \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<HM, DR>), grid, block, HM ? (size_t)S * sizeof(float) : 0, st, a);    \
    if (phases & STONK_ATTN_BWD_DKV) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HM, DR>), grid, block, 0, st, a);     \
  } while (0)
  if (hm && dr) LAUNCH_BWD(true, true);
  else if (hm) LAUNCH_BWD(true, false);
  else if (dr) LAUNCH_BWD(false, true);
  else LAUNCH_BWD(false, false);
#undef LAUNCH_BWD
  return stonk_launch_status();
}

extern "C" int stonk_attention_bwd(const void* q, const void* k, const void* v, int64_t ld,
                                   const int64_t* attention_mask, const int* seq_offsets, const int* q_offsets,
                                   const void* out, int64_t ldo, const void* dout,
                                   int64_t lddo, const float* lse, float* delta_ws, void* dq, void* dk, int64_t ldd,
                                   void* dv, int B, int NH, int S, int D, float scale, float drop_p, uint32_t seed,
                                   void* stream) {
  return attention_bwd_launch(STONK_ATTN_BWD_ALL, q, k, v, ld, attention_mask, seq_offsets, q_offsets, out, ldo, dout, lddo,
                              lse, delta_ws, dq, dk, ldd, dv, B, NH, S, D, scale, drop_p, seed, stream);
}

extern "C" int stonk_attention_bwd_phases(int phases, const void* q, const void* k, const void* v, int64_t ld,
                                          const int64_t* attention_mask, const int* seq_offsets, const int* q_offsets,
                                          const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                                          float* delta_ws, void* dq, void* dk, int64_t ldd, void* dv, int B, int NH, int S,
                                          int D, float scale, float drop_p, uint32_t seed, void* stream) {
  STONK_CHECK_ARG(phases > 0 && phases <= STONK_ATTN_BWD_ALL, STONK_EINVAL);
  return attention_bwd_launch(phases, q, k, v, ld, attention_mask, seq_offsets, q_offsets, out, ldo, dout, lddo, lse,
                              delta_ws, dq, dk, ldd, dv, B, NH, S, D, scale, drop_p, seed, stream);
}
