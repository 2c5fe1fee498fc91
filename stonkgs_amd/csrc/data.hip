// On-device batch assembly and dynamic masking: what the reference does once, offline, in
// Python lists (ref:src/stonkgs/data/indra_for_pretraining.py:33-77 masking, :80-126 NSP negatives, :190-239 row
// assembly) done per step on the GPU from the tokenised text ids and the random-walk table, so that every step sees
// fresh masks and fresh negative pairs and no pre-masked 13.8 M-row frame has to exist.
//
// The reference's semantics are kept: exactly int(len * 0.15) positions of each PADDED 256-token half are chosen
// uniformly without replacement; each chosen position becomes [MASK] with probability 0.8, stays with 0.1, becomes a
// uniform id in [0, vocab_len-1] with 0.1; labels hold the original id there and -100 elsewhere; the entity half of a
// negative row comes from another row of the batch and gets NSP label 1. The random stream cannot be Python's Mersenne
// Twister: it is counter-based (lowbias32 of (seed, row, half, position)), restated bit for bit in numpy by
// oracle/masking_oracle.py - integer work, so the parity tests demand equality. The host path that IS bit-exact with the
// reference's own draws stays in stonkgs_amd/data.py (replace_mlm_tokens).
#include "common.h"

namespace {

constexpr uint32_t K_SEED = 0x9E3779B9u, K_SEED_ADD = 0x7F4A7C15u, K_ROW = 0x85EBCA77u, K_POS = 0x9E3779B1u;
constexpr uint32_t K_D1 = 0x68E31DA4u, K_D2 = 0xB5297A4Du, K_D3 = 0x1B56C4E9u, K_NEG = 0x2545F491u, K_PART = 0x632BE5ABu;

__device__ __forceinline__ uint32_t base_key(uint32_t seed) { return stonk_hash32(seed * K_SEED + K_SEED_ADD); }

// one wave per (row, half): keys of the half's positions in LDS, rank by counting, decisions by three more hashes
__global__ __launch_bounds__(64) void mlm_mask_kernel(const long* __restrict__ ids_in, long* __restrict__ ids_out,
                                                      long* __restrict__ text_labels, long* __restrict__ ent_labels,
                                                      int S, int half, uint32_t vocab_text, uint32_t vocab_ent,
                                                      long mask_id, int k_text, int k_ent, uint32_t seed) {
  extern __shared__ uint32_t keys[];
  const int b = blockIdx.x >> 1, h = blockIdx.x & 1, lane = threadIdx.x;
  const uint32_t rowkey = stonk_hash32(base_key(seed) ^ ((uint32_t)(b * 2 + h) * K_ROW));
  const long* src = ids_in + (long)b * S + h * half;
  long* dst = ids_out + (long)b * S + h * half;
  long* lab = (h == 0 ? text_labels : ent_labels) + (long)b * half;
  const uint32_t vocab = h == 0 ? vocab_text : vocab_ent;
  const int k = h == 0 ? k_text : k_ent;
  for (int pos = lane; pos < half; pos += 64) keys[pos] = stonk_hash32(rowkey + (uint32_t)pos * K_POS);
  __syncthreads();
  for (int pos = lane; pos < half; pos += 64) {
    const uint32_t key = keys[pos];
    int rank = 0;
    for (int j = 0; j < half; ++j) {
      const uint32_t kj = keys[j];
      rank += (kj < key) || (kj == key && j < pos);
    }
    const long tok = src[pos];
    long out = tok, label = -100;
    if (rank < k) {
      label = tok;
      if ((stonk_hash32(key ^ K_D1) >> 8) < 13421773u) out = mask_id;                    // < 0.8
      else if ((stonk_hash32(key ^ K_D2) >> 8) < 8388608u) out = tok;                    // < 0.5: keep
      else out = (long)(((uint64_t)stonk_hash32(key ^ K_D3) * (uint64_t)vocab) >> 32);   // uniform id
    }
    dst[pos] = out;
    lab[pos] = label;
  }
}

// one thread per output element: text half copied, entity half = walk(source) [SEP] walk(target) [SEP] of this row or,
// for a negative, of its partner row
__global__ void assemble_kernel(const long* __restrict__ text_ids, const long* __restrict__ text_attention,
                                const long* __restrict__ source, const long* __restrict__ target,
                                const long* __restrict__ walks, long n_nodes, int walk_len, long* __restrict__ ids_out,
                                long* __restrict__ attention_out, long* __restrict__ type_out, long* __restrict__ nsp_out,
                                int B, int S, int half, long sep_id, uint32_t neg_thr32, uint32_t seed,
                                int* __restrict__ err) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * S) return;
  const int b = (int)(gid / S), p = (int)(gid - (long)b * S);
  const uint32_t base = base_key(seed);
  const bool neg = B > 1 && stonk_hash32(base ^ ((uint32_t)b * K_NEG)) < neg_thr32;
  int row = b;
  if (neg) {
    row = (int)(((uint64_t)stonk_hash32(base ^ ((uint32_t)b * K_PART)) * (uint64_t)B) >> 32);
    if (row == b) row = (b + 1) % B;
  }
  if (p == 0) nsp_out[b] = neg ? 1 : 0;
  if (p < half) {
    ids_out[gid] = text_ids[(long)b * half + p];
    attention_out[gid] = text_attention[(long)b * half + p];
    type_out[gid] = 0;
    return;
  }
  const int e = p - half;
  long v = sep_id;
  if (e != walk_len && e != 2 * walk_len + 1) {
    const bool second = e > walk_len;
    const long node = second ? target[row] : source[row];
    if (node < 0 || node >= n_nodes) {
      atomicOr(err, 1);   // a pair names a node without a walk (the reference raises KeyError on the dict lookup)
      v = 0;
    } else {
      v = walks[node * walk_len + (second ? e - walk_len - 1 : e)];
    }
  }
  ids_out[gid] = v;
  attention_out[gid] = 1;
  type_out[gid] = 1;
}

}  // namespace

extern "C" int stonk_mlm_mask(const int64_t* ids_in, int64_t* ids_out, int64_t* text_labels, int64_t* ent_labels, int B,
                              int S, int half, int64_t vocab_text, int64_t vocab_ent, int64_t mask_id, int k_text,
                              int k_ent, uint32_t seed, void* stream) {
  STONK_CHECK_ARG(ids_in && ids_out && text_labels && ent_labels, STONK_EINVAL);
  STONK_CHECK_ARG(B >= 0 && half > 0 && S == 2 * half && half <= 8192, STONK_ESHAPE);
  STONK_CHECK_ARG(vocab_text > 0 && vocab_ent > 0 && vocab_text < (1L << 32) && vocab_ent < (1L << 32), STONK_EINVAL);
  STONK_CHECK_ARG(k_text >= 0 && k_text <= half && k_ent >= 0 && k_ent <= half, STONK_EINVAL);
  if (B == 0) return STONK_OK;
  hipLaunchKernelGGL(mlm_mask_kernel, dim3(2 * B), dim3(64), half * sizeof(uint32_t), (hipStream_t)stream,
                     (const long*)ids_in, (long*)ids_out, (long*)text_labels, (long*)ent_labels, S, half,
                     (uint32_t)vocab_text, (uint32_t)vocab_ent, (long)mask_id, k_text, k_ent, seed);
  return stonk_launch_status();
}

extern "C" int stonk_assemble_rows(const int64_t* text_ids, const int64_t* text_attention, const int64_t* source,
                                   const int64_t* target, const int64_t* walks, int64_t n_nodes, int walk_len,
                                   int64_t* ids_out, int64_t* attention_out, int64_t* type_out, int64_t* nsp_out, int B,
                                   int S, int half, int64_t sep_id, float negative_rate, uint32_t seed, int* err_flag,
                                   void* stream) {
  STONK_CHECK_ARG(text_ids && text_attention && source && target && walks && ids_out && attention_out && type_out &&
                      nsp_out && err_flag, STONK_EINVAL);
  STONK_CHECK_ARG(B >= 0 && half > 0 && S == 2 * half && 2 * walk_len + 2 == half && n_nodes > 0, STONK_ESHAPE);
  STONK_CHECK_ARG(negative_rate >= 0.f && negative_rate < 1.f, STONK_EINVAL);
  if (B == 0) return STONK_OK;
  const long n = (long)B * S;
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long*)text_ids, (const long*)text_attention, (const long*)source, (const long*)target,
                     (const long*)walks, (long)n_nodes, walk_len, (long*)ids_out, (long*)attention_out, (long*)type_out,
                     (long*)nsp_out, B, S, half, (long)sep_id, stonk_drop_thr32(negative_rate), seed, err_flag);
  return stonk_launch_status();
}
