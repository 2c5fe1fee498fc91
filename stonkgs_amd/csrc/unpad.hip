// Row plan of the UNPADDED trainable encoder.
//
// The reference runs its encoder over all 512 positions of every sequence (ref:src/stonkgs/models/stonkgs_model.py:204-210
// -> hf:models/bert/modeling_bert.py:164-203 with an additive key mask): a padded text position is a QUERY whose output
// nobody reads unless the position carries a label (the reference labels 15 % of the PADDED half,
// ref:src/stonkgs/data/indra_for_pretraining.py:33-77), and it is never a KEY. Dropping the rows that are neither live
// keys, nor labelled, nor position 0 (the pooler's input) therefore changes no loss term and no gradient - every per-row
// operation (linear layers, LayerNorm, GELU, residuals) is independent per row and attention only ever reads live keys.
// In the benchmark's batches (text length uniform in [32, 256]) that is 95 of 512 positions.
//
// This kernel pair turns (attention_mask, labels) into the packed layout the engine then runs on:
// Order of a sequence's packed rows: its READ rows first (labelled positions and position 0, in position order), then its
// other kept rows (position order). Attention does not care about the order of a sequence's rows (positions live in the
// embeddings), every other operation is row-wise - and the rows whose last-layer output is read sit at the head of each
// sequence, where the last layer's attention can stop (stonk_attention_* `q_limit`).
//   row_of_pos [B*S]  packed row of padded position b*S+s, or -1 for a dropped position
//   pos_of_row [B*S]  padded position of packed row i (i < total), -1 beyond
//   seq_offsets[B+1]  first packed row of every sequence; [B] = total (what stonk_attention_* take as `seq_offsets`)
//   row_mask  [B*S]   attention_mask gathered to packed rows (int64, 0 beyond total)
// and, optionally, the READ rows - packed rows whose LAST-layer output something reads: labelled positions (the decoders)
// and position 0 (the pooler). The last encoder layer's feed-forward block, the pooler and the head transform are
// row-wise, so they only need to run on those (about 77 of a sequence's ~410 rows):
//   read_rows  [B*S]  packed row of the j-th read row (position order), -1 beyond their count
//   read_of_pos[B*S]  index into read_rows of padded position b*S+s, or -1
//   read_offsets[B+1] first read row of every sequence ([B] = their count); read_rows[read_offsets[b]] is position 0 of b
// A sequence WITHOUT any live key keeps all of its positions (the reference then attends uniformly over all S keys).
// Integer work, bit-exact by construction; tests compare with a numpy restatement (oracle/masking_oracle.py).
#include "common.h"
#include "stonk_flags.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ bool read_position(const long* text_labels, const long* ent_labels, int b, int s, int half) {
  if (s == 0) return true;
  if (s < half) return text_labels && text_labels[(long)b * half + s] != -100;
  return ent_labels && s - half < half && ent_labels[(long)b * half + (s - half)] != -100;
}

__device__ __forceinline__ bool keep_position(const long* mask, const long* text_labels, const long* ent_labels, int b,
                                              int s, int S, int half, bool any_live) {
  if (!any_live || s == 0) return true;
  if (mask[(long)b * S + s] != 0) return true;
  if (s < half) return text_labels && text_labels[(long)b * half + s] != -100;
  return ent_labels && s - half < half && ent_labels[(long)b * half + (s - half)] != -100;
}

// exclusive block scan of one int per thread (256 threads = 4 waves)
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
  __shared__ int wsum[TPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < TPB / 64; ++j) {
    if (j < w) off += wsum[j];
    tot += wsum[j];
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

// pass 1, one workgroup per sequence: how many positions it keeps
__global__ __launch_bounds__(TPB) void unpad_count_kernel(const long* __restrict__ mask,
                                                          const long* __restrict__ text_labels,
                                                          const long* __restrict__ ent_labels, int B, int S, int half,
                                                          int* __restrict__ counts, int* __restrict__ live_flags) {
  const int b = blockIdx.x;
  __shared__ int any_live_s;
  if (threadIdx.x == 0) any_live_s = 0;
  __syncthreads();
  int live = 0;
  for (int s = threadIdx.x; s < S; s += TPB) live |= mask[(long)b * S + s] != 0;
  if (__ballot(live) != 0 && (threadIdx.x & 63) == 0) atomicOr(&any_live_s, 1);
  __syncthreads();
  const bool any_live = any_live_s != 0;
  int cnt = 0, rd = 0;
  for (int s = threadIdx.x; s < S; s += TPB) {
    cnt += keep_position(mask, text_labels, ent_labels, b, s, S, half, any_live);
    rd += read_position(text_labels, ent_labels, b, s, half);
  }
  int total, total_rd;
  block_excl_scan(cnt, &total);
  block_excl_scan(rd, &total_rd);
  if (threadIdx.x == 0) {
    counts[b] = total;
    live_flags[b] = any_live;
    counts[2 * B + b] = total_rd;   // (workspace: [0,B) kept, [B,2B) live flags, [2B,3B) read rows)
  }
}

// pass 2, one workgroup per sequence: its offset (sum of the counts before it), then the maps. Positions are dealt to
// threads in CONTIGUOUS chunks so that the packed order is the position order.
__global__ __launch_bounds__(TPB) void unpad_fill_kernel(const long* __restrict__ mask, const long* __restrict__ text_labels,
                                                         const long* __restrict__ ent_labels, int B, int S, int half,
                                                         const int* __restrict__ counts, const int* __restrict__ live_flags,
                                                         int* __restrict__ row_of_pos, int* __restrict__ pos_of_row,
                                                         int* __restrict__ seq_offsets, long* __restrict__ row_mask,
                                                         int* __restrict__ read_rows, int* __restrict__ read_of_pos,
                                                         int* __restrict__ read_offsets) {
  const int b = blockIdx.x;
  int before = 0, all = 0, rbefore = 0, rall = 0;
  for (int j = threadIdx.x; j < B; j += TPB) {
    const int c = counts[j], rc = counts[2 * B + j];
    all += c;
    rall += rc;
    if (j < b) {
      before += c;
      rbefore += rc;
    }
  }
  __shared__ int red[4][TPB / 64];
  {
    int x = before, y = all, z = rbefore, w = rall;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      x += __shfl_xor(x, o, 64);
      y += __shfl_xor(y, o, 64);
      z += __shfl_xor(z, o, 64);
      w += __shfl_xor(w, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = x;
      red[1][threadIdx.x >> 6] = y;
      red[2][threadIdx.x >> 6] = z;
      red[3][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    before = all = rbefore = rall = 0;
#pragma unroll
    for (int j = 0; j < TPB / 64; ++j) {
      before += red[0][j];
      all += red[1][j];
      rbefore += red[2][j];
      rall += red[3][j];
    }
    __syncthreads();
  }
  const bool any_live = live_flags[b] != 0;
  const int per = (S + TPB - 1) / TPB;
  const int s0 = threadIdx.x * per, s1 = s0 + per < S ? s0 + per : S;
  int cnt_r = 0, cnt_o = 0;   // read rows / other kept rows of this thread's chunk
  for (int s = s0; s < s1; ++s) {
    const bool rdp = read_position(text_labels, ent_labels, b, s, half);
    cnt_r += rdp;
    cnt_o += !rdp && keep_position(mask, text_labels, ent_labels, b, s, S, half, any_live);
  }
  int total;
  int row_r = before + block_excl_scan(cnt_r, &total);
  const int n_read_b = total;                         // (= counts[2B + b])
  int row_o = before + n_read_b + block_excl_scan(cnt_o, &total);
  for (int s = s0; s < s1; ++s) {
    const long p = (long)b * S + s;
    const bool rdp = read_position(text_labels, ent_labels, b, s, half);
    const bool kept = rdp || keep_position(mask, text_labels, ent_labels, b, s, S, half, any_live);
    const int row = rdp ? row_r++ : (kept ? row_o++ : -1);
    row_of_pos[p] = row;
    if (kept) {
      pos_of_row[row] = (int)p;
      row_mask[row] = mask[p];
    }
    if (read_rows) {
      const int ri = rdp ? rbefore + (row - before) : -1;   // read rows are the first n_read_b rows of the sequence
      read_of_pos[p] = ri;
      if (rdp) read_rows[ri] = row;
    }
  }
  if (threadIdx.x == 0) {
    seq_offsets[b] = before;
    if (b == B - 1) seq_offsets[B] = all;
    if (read_rows) {
      read_offsets[b] = rbefore;
      if (b == B - 1) read_offsets[B] = rall;
    }
  }
  // rows past the total belong to no position
  const long cap = (long)B * S;
  for (long i = all + (long)b * TPB + threadIdx.x; i < cap; i += (long)B * TPB) {
    pos_of_row[i] = -1;
    row_mask[i] = 0;
  }
  if (read_rows)
    for (long i = rall + (long)b * TPB + threadIdx.x; i < cap; i += (long)B * TPB) read_rows[i] = -1;
}

}  // namespace

extern "C" int64_t stonk_unpad_workspace_ints(int B) { return B > 0 ? 3L * B : 0; }

extern "C" int stonk_unpad_plan(const int64_t* attention_mask, const int64_t* text_labels, const int64_t* ent_labels, int B,
                                int S, int half, int* row_of_pos, int* pos_of_row, int* seq_offsets, int64_t* row_mask,
                                int* read_rows, int* read_of_pos, int* read_offsets, int* workspace, int64_t ws_ints,
                                void* stream) {
  STONK_CHECK_ARG(attention_mask && row_of_pos && pos_of_row && seq_offsets && row_mask && workspace, STONK_EINVAL);
  STONK_CHECK_ARG((read_rows && read_of_pos && read_offsets) || (!read_rows && !read_of_pos && !read_offsets), STONK_EINVAL);
  STONK_CHECK_ARG(B >= 0 && S > 0 && half >= 0 && half <= S && (long)B * S < (1L << 31), STONK_ESHAPE);
  STONK_CHECK_ARG(ws_ints >= 3L * B, STONK_EINVAL);
  if (B == 0) return STONK_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(unpad_count_kernel, dim3(B), dim3(TPB), 0, st, (const long*)attention_mask, (const long*)text_labels,
                     (const long*)ent_labels, B, S, half, workspace, workspace + B);
  hipLaunchKernelGGL(unpad_fill_kernel, dim3(B), dim3(TPB), 0, st, (const long*)attention_mask, (const long*)text_labels,
                     (const long*)ent_labels, B, S, half, workspace, workspace + B, row_of_pos, pos_of_row, seq_offsets,
                     (long*)row_mask, read_rows, read_of_pos, read_offsets);
  return stonk_launch_status();
}
