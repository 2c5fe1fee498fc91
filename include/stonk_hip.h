/* stonk_hip.h - C ABI of libstonk_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the STonKGs
 * pre-training hot path (STonKGsForPreTraining forward / backward / optimizer step).
 *
 * The reference (stonkgs v0.1.6-dev) has no FFI of its own: its hot path is a Python nn.Module whose
 * arithmetic is issued by HuggingFace `modeling_bert` into torch ATen. This header is therefore the
 * boundary a maintainer would bind (ctypes) to replace those ATen calls; every entry
 * cites the reference / HF site it replaces ("ref:" = /root/reference, "hf:" = transformers).
 *
 * Conventions:
 *  - plain pointers and sizes only; all pointers are DEVICE addresses unless stated; bf16 = 16-bit brain
 *    float stored as uint16_t; "ld*" are row strides in ELEMENTS;
 *  - every launcher is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing, keeps no
 *    global state (a launcher that needs scratch memory takes a caller workspace and has a `*_workspace_floats` query),
 *    never throws; it returns 0 on success, <0 for a rejected argument (STONK_E*), >0 for a
 *    hipError_t raised by the launch;
 *  - dropout masks are regenerated from (seed, element index) by a counter-based hash, never stored.
 */
#ifndef STONK_HIP_H
#define STONK_HIP_H
#include <stdint.h>
#include "../stonkgs_amd/csrc/stonk_flags.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STONK_OK 0
#define STONK_EINVAL (-1) /* null pointer / inconsistent argument */
#define STONK_ESHAPE (-2) /* unsupported shape */
#define STONK_EALIGN (-3) /* pointer or stride not aligned as required */

int stonk_abi_version(void);

/* C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T), A/B bf16, fp32 accumulate on MFMA. N % 128 == 0, K % 64 == 0.
 * `flags`: STONK_EPI_* (output type, bias, GELU, residual, saved pre-activation, GELU', dropout). STONK_EPI_AUX_GRAD
 * modifies the two users of `aux`: SAVE_PREACT (next to GELU) then stores gelu'(pre-activation) and GELU_BWD multiplies
 * by `aux` as is - the training step uses the pair, so the erf/exp of GELU' are evaluated once, in the forward epilogue.
 * m_dev / k_dev (nullable): effective M / K read from device memory at run time (label-sparse decoders).
 * `kernel`: STONK_GEMM_AUTO (the launcher picks one of its kernels from shape and epilogue), an explicit
 * STONK_GEMM_TILE128 / _WAVE8 / _WAVE4 / _WAVE4_192 / _ASM4 / _ASM4_192 (an explicit kernel that cannot take the arguments
 * is refused with STONK_ESHAPE; stonk_flags.h says what each takes), or STONK_GEMM_DISPATCHED / _DISPATCHED2: AUTO's choice without a persistent grid (one / two work items per
 * workgroup), for launches that run while another stream - a collective - holds CUs.
 * Replaces torch addmm/mm of hf:models/bert/modeling_bert.py:154-156 (Q,K,V), :289-293 (attn out), :334-337
 * (FFN up + GELU), :347-351 (FFN down), :476-480 (head transform); ref:src/stonkgs/models/stonkgs_model.py:70-71
 * (text / entity decoders) and their autograd backward. */
int stonk_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                       int flags, const float* bias, const void* resid, int64_t ldr, void* aux, int64_t ldaux,
                       float alpha, int split_k, const int* m_dev, const int* k_dev, float drop_p, uint32_t seed,
                       int kernel, void* stream);

/* Weight / bias gradient straight from row-major activations: dW[M',N'] += alpha * dY[T,M']^T . X[T,N'],
 * db[M'] += alpha * colsum(dY) (nullable). M', N' % 128 == 0; rows of dY / X in [k, roundup64(k)) must read as zero when
 * the token count k (<= K) comes from k_dev (the 256x256 kernels range-check tokens themselves). split_k >= 1: 128x128
 * tiles with that many K splits; split_k == 0: the four-wave 256x256 kernel (M', N' >= 256, lda / ldb % 64 == 0) with an
 * automatic split over all CUs; split_k <= -16: the same kernel held to -split_k CUs' worth of workgroups (launches on
 * a second stream beside other work); split_k == -1: the older eight-wave 256x256 form.
 * Autograd's weight/bias gradients of every nn.Linear on the path. */
int stonk_gemm_tn_bf16(const void* dY, int64_t lda, const void* X, int64_t ldb, float* dW, int64_t ldc, float* dbias,
                       int M, int N, int K, float alpha, int split_k, const int* k_dev, void* stream);

/* y = dropout(LayerNorm(x)); x,y bf16 [rows,H]; gamma/beta fp32; mean/rstd fp32 [rows] saved for backward.
 * Replaces nn.LayerNorm(eps=1e-12) + nn.Dropout at hf:modeling_bert.py:106-107, :291-292, :349-350, :479. */
int stonk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        int64_t rows, int H, float eps, int flags, float drop_p, uint32_t seed, void* stream);

/* dx = LayerNorm'(dy) (dy first masked by the forward's output dropout when STONK_LN_DROPOUT);
 * dx_drop (nullable) = dropout-masked copy of dx for the branch that went through nn.Dropout before the residual
 * add; dgamma/dbeta (fp32, nullable) are ACCUMULATED - through per-workgroup partials when a workspace of
 * >= 1024 * 2 * H floats is passed (nullable: falls back to atomics). Autograd backward of the sites above.
 * stonk_layernorm_bwd_workspace_floats(rows, H) is the size to pass (the only launcher that takes a workspace). */
int64_t stonk_layernorm_bwd_workspace_floats(int64_t rows, int H);
int stonk_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        void* dx, void* dx_drop, float* dgamma, float* dbeta, int64_t rows, int H, int flags,
                        float drop_p_in, uint32_t seed_in, float drop_p_out, uint32_t seed_out, float* partial_ws,
                        int64_t ws_floats, void* stream);
/* With STONK_LN_DEFER_REDUCE in `flags` (workspace required) stonk_layernorm_bwd leaves the per-workgroup dgamma / dbeta
 * partial sums in `partial_ws`; this adds them into dgamma / dbeta - same `rows` and `H` as that call - on any stream
 * ordered after it. The training step runs it on its weight-gradient stream: only the optimizer waits for these two
 * vectors (autograd accumulates them at the same point: hf:modeling_bert.py:107 / :292 / :350 / :479 backward). */
int stonk_layernorm_bwd_reduce(const float* partial_ws, int64_t rows, int H, float* dgamma, float* dbeta, void* stream);

/* inputs_embeds + position + token-type embeddings -> LayerNorm -> dropout, in one pass:
 *   row (b,s<half)  = text_hidden[b*half+s]          (frozen LM backbone output, bf16)
 *   row (b,s>=half) = kg_table[input_ids[b,s]]       (fp32 node2vec table; ids 100/102/103 = LM special vectors)
 * Replaces the Python gather loop, torch.stack/cat and the CPU fp32 round trip of
 * ref:src/stonkgs/models/stonkgs_model.py:182-200 plus BertEmbeddings hf:modeling_bert.py:98-108.
 * An id outside [0, kg_rows) sets bit 0 of *err_flag (the reference raises KeyError at :185).
 * pos_of_row (nullable) / n_rows: the PACKED layout of stonk_unpad_plan - output row i (i < n_rows) is padded position
 * pos_of_row[i]; rows whose entry is -1 (the tail up to n_rows) are written as zeros with mean 0, rstd 1. */
int stonk_joint_embed_ln_fwd(const int64_t* input_ids, const int64_t* token_type_ids, const void* text_hidden,
                             const float* kg_table, const float* pos_emb, const float* type_emb, const float* gamma,
                             const float* beta, void* sum_out, void* y, float* mean, float* rstd, int B, int S,
                             int half, int H, int64_t kg_rows, int type_rows, float eps, int flags, float drop_p,
                             uint32_t seed, int* err_flag, const int* pos_of_row, int64_t n_rows, void* stream);

/* Frozen LM backbone embeddings: word_emb[input_ids[:, :S]] + pos + type[0] -> LayerNorm -> dropout.
 * Replaces BertEmbeddings of `self.lm_backbone(input_ids[:, :half])`, ref:stonkgs_model.py:178. */
int stonk_text_embed_ln_fwd(const int64_t* input_ids, int64_t ld_ids, const float* word_emb, const float* pos_emb,
                            const float* type_emb, const float* gamma, const float* beta, void* y, int B, int S, int H,
                            int64_t vocab, float eps, int flags, float drop_p, uint32_t seed, int* err_flag,
                            void* stream);

/* d(position_embeddings) and d(token_type_embeddings) from d(embedding sum) (bf16 [B*S,H]); accumulates.
 * row_of_pos (nullable): packed layout - dx row of padded position p is row_of_pos[p], -1 = dropped (no gradient). */
int stonk_embed_grad(const void* dx, const int64_t* token_type_ids, float* dpos, float* dtype, int B, int S, int H,
                     int type_rows, const int* row_of_pos, void* stream);

/* Row plan of the unpadded trainable encoder. A padded text position is never a key, and its output is read only if it
 * carries a label (the reference labels 15 % of the PADDED half, ref:src/stonkgs/data/indra_for_pretraining.py:33-77) or
 * is position 0 (the pooler's input): every other padded row can be dropped without changing a loss term or a gradient
 * of ref:src/stonkgs/models/stonkgs_model.py:204-245. Kept: attention_mask != 0, s == 0, text_labels[b,s] != -100
 * (s < half), ent_labels[b,s-half] != -100 (labels nullable); a sequence WITHOUT any unmasked key keeps all S positions
 * (the reference then attends uniformly over them). Outputs (int32 unless stated, device): row_of_pos [B*S] (-1 =
 * dropped), pos_of_row [B*S] (-1 past the total), seq_offsets [B+1] ([B] = total; the `seq_offsets` of
 * stonk_attention_*), row_mask int64 [B*S] (attention_mask per packed row, 0 past the total).
 * Optionally (all three or none) the READ rows - the packed rows whose last-layer output something reads (labelled
 * positions: the decoders; position 0: the pooler), on which alone the last layer's row-wise feed-forward block, the pooler
 * and the head transform need to run: read_rows [B*S] (packed row of the j-th read row, -1 past their count),
 * read_of_pos [B*S] (index into read_rows of a padded position, or -1), read_offsets [B+1] ([B] = their count;
 * read_rows[read_offsets[b]] is position 0 of sequence b). workspace: device ints, stonk_unpad_workspace_ints(B) of them. */
int64_t stonk_unpad_workspace_ints(int B);
int stonk_unpad_plan(const int64_t* attention_mask, const int64_t* text_labels, const int64_t* ent_labels, int B, int S,
                     int half, int* row_of_pos, int* pos_of_row, int* seq_offsets, int64_t* row_mask, int* read_rows,
                     int* read_of_pos, int* read_offsets, int* workspace, int64_t ws_ints, void* stream);

/* Fused attention, head_dim 64, S % 128 == 0, S <= 4096: out = dropout(softmax(q k^T * scale + mask)) v.
 * q/k/v: column slices of the [T, 3H] projection (row stride ld), head h at columns h*64..; attention_mask int64
 * [B,S] (0 = masked key) or NULL; lse fp32 [B,NH,S]. Replaces hf:modeling_bert.py:188-203 (eager :111-136).
 * PACKED layout (seq_offsets != NULL, int32 [B+1], device): sequence b is rows seq_offsets[b] .. seq_offsets[b+1]-1 of
 * q/k/v/out (any length <= S, no alignment), attention_mask is then REQUIRED and holds one word per packed ROW; lse (and
 * delta_ws) keep the [B,NH,S] layout. What the trainable encoder runs on once the rows that nothing reads - padding that
 * is neither a live key nor a labelled position - are dropped (stonk_unpad_plan); rows outside every sequence are never
 * read or written. q_offsets (nullable, int32 [B+1], packed layout only): sequence b's QUERIES are its first
 * q_offsets[b+1] - q_offsets[b] rows (its keys: all of them) - out / lse / dq are written for those rows only, dk / dv
 * receive their contributions only. The last encoder layer's form: stonk_unpad_plan puts a sequence's read rows first. */
int stonk_attention_fwd(const void* q, const void* k, const void* v, int64_t ld, const int64_t* attention_mask,
                        const int* seq_offsets, const int* q_offsets, void* out, int64_t ldo, float* lse, int B, int NH,
                        int S, int D, float scale, float drop_p, uint32_t seed, void* stream);
/* Backward of the above (recomputes P from lse; no atomics). delta_ws: fp32 [B,NH,S] scratch. With q_offsets, dq rows
 * past a sequence's queries are NOT written (zero them if something reads them) and dout rows past them must be finite. */
int stonk_attention_bwd(const void* q, const void* k, const void* v, int64_t ld, const int64_t* attention_mask,
                        const int* seq_offsets, const int* q_offsets, const void* out, int64_t ldo, const void* dout,
                        int64_t lddo,
                        const float* lse, float* delta_ws, void* dq, void* dk, int64_t ldd, void* dv, int B, int NH,
                        int S, int D, float scale, float drop_p, uint32_t seed, void* stream);

/* The same in separately launched parts (`phases`: STONK_ATTN_BWD_DELTA | _DQ | _DKV), so that the dQ and the dK / dV
 * kernels - independent once delta = rowsum(dO * O) exists - can be put on two streams: run DELTA first, order the DKV
 * call's stream after it, then DQ and DKV in any order or side by side. phases = STONK_ATTN_BWD_ALL is stonk_attention_bwd. */
int stonk_attention_bwd_phases(int phases, const void* q, const void* k, const void* v, int64_t ld,
                               const int64_t* attention_mask, const int* seq_offsets, const int* q_offsets, const void* out,
                               int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta_ws, void* dq,
                               void* dk, int64_t ldd, void* dv, int B, int NH, int S, int D, float scale, float drop_p,
                               uint32_t seed, void* stream);

/* out[c][r] = in[r][c] (bf16). Rows >= *rows_dev (nullable) read as zero; colsum (nullable, fp32) += column sums
 * of `in` (bias gradients). Feeds wgrad operands to stonk_gemm_nt_bf16. */
int stonk_transpose_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int cols,
                         float* colsum, const int* rows_dev, void* stream);
int stonk_transpose_f32_to_bf16(const float* in, void* out, int64_t rows, int cols, int64_t ld_out, void* stream);
/* n bf16 transposes in ONE launch (the W^T copies of every weight after an optimizer step). desc_dev: device array of n
 * 56-byte entries {const void* in; void* out; int64 ld_in, ld_out, rows; int32 cols, first_tile, col_tiles, pad}, sorted
 * by first_tile; an entry covers ceil(rows/64) * col_tiles tiles of 64x64 (col_tiles = ceil(cols/64)), total_tiles = the
 * sum; cols % 8 == 0, ld_out >= roundup64(rows), rows past `rows` are written as zeros up to the tile edge. */
int stonk_transpose_bf16_batched(const void* desc_dev, int n, int total_tiles, void* stream);
int stonk_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* On-device dynamic masking: ids_in [B,S] -> ids_out [B,S] + text / entity labels [B,half]. Per half,
 * exactly k_* positions (the reference uses int(half * 0.15)) chosen uniformly without replacement - padding included,
 * as the reference masks the padded sequence; 80 % -> mask_id, 10 % kept, 10 % uniform id in [0, vocab_*-1]; labels =
 * original id there, -100 elsewhere. Counter-based random stream (seed, row, half, position), restated bit for bit by
 * oracle/masking_oracle.py. Replaces ref:src/stonkgs/data/indra_for_pretraining.py:33-77 (replace_mlm_tokens), whose
 * own Mersenne-Twister draws stay reproducible on the host path (stonkgs_amd/data.py). */
int stonk_mlm_mask(const int64_t* ids_in, int64_t* ids_out, int64_t* text_labels, int64_t* ent_labels, int B, int S,
                   int half, int64_t vocab_text, int64_t vocab_ent, int64_t mask_id, int k_text, int k_ent, uint32_t seed,
                   void* stream);

/* On-device row assembly: text_ids / text_attention [B,half] (already padded), source / target [B] node indices into
 * walks [n_nodes, walk_len] (2 * walk_len + 2 == half) -> ids_out / attention_out / type_out [B,S], nsp_out [B]. Entity
 * half = walks[source] [SEP] walks[target] [SEP]; with probability negative_rate a row takes the entity half of another
 * row of the batch and NSP label 1 (the reference appends 25 % such rows offline: the same 1 in 5). A node index outside
 * the table sets bit 0 of *err_flag (the reference raises KeyError). Replaces
 * ref:src/stonkgs/data/indra_for_pretraining.py:190-239 (row assembly) and :80-126 (negative NSP samples). */
int stonk_assemble_rows(const int64_t* text_ids, const int64_t* text_attention, const int64_t* source,
                        const int64_t* target, const int64_t* walks, int64_t n_nodes, int walk_len, int64_t* ids_out,
                        int64_t* attention_out, int64_t* type_out, int64_t* nsp_out, int B, int S, int half,
                        int64_t sep_id, float negative_rate, uint32_t seed, int* err_flag, void* stream);

/* Labelled-row compaction for the MLM / ELM heads (labels != -100), count kept on the device.
 * rows_out[i] = b*S + offset + pos of the i-th labelled position - or, with row_of_pos (nullable; packed layout of
 * stonk_unpad_plan), that position's packed row. Semantics of nn.CrossEntropyLoss(ignore_index=-100) at
 * ref:stonkgs_model.py:229-240. */
int stonk_label_compact(const int64_t* labels, int64_t n, int half, int S, int offset, int* rows_out, int* targets_out,
                        int* count_out, const int* row_of_pos, void* stream);
int stonk_gather_rows_bf16(const void* src, int64_t ld_src, const int* rows, const int* count_dev, void* dst,
                           int64_t ld_dst, int cols, int64_t cap, void* stream);
int stonk_scatter_rows_bf16(const void* src, int64_t ld_src, const int* rows, const int* count_dev, void* dst,
                            int64_t ld_dst, int cols, void* stream);
int stonk_scatter_rows_f32_to_bf16(const float* src, int64_t ld_src, const int* rows, const int* count_dev, void* dst,
                                   int64_t ld_dst, int cols, void* stream);

/* Per labelled row: loss_sum += logsumexp(logits[row,:ncols]) - logits[row,target];
 * dlogits (bf16, nullable, cap_rows rows allocated) = (softmax - onehot) * grad_scale / count, rows
 * [count, roundup64(count)) zeroed. Bit 3 of *err_flag: target out of range. */
int stonk_softmax_xent_fwd_bwd(const float* logits, int64_t ld, int ncols, int npad, const int* targets,
                               const int* count_dev, float* loss_sum, void* dlogits, int64_t ld_d, float grad_scale,
                               int cap_rows, int* err_flag, void* stream);

/* The same on fp16 logits (stonk_gemm_nt_bf16 with STONK_EPI_OUT_F16): 6 bytes of HBM traffic per logit instead of 10 -
 * the training step's label-sparse decoders use this pair; all arithmetic stays fp32. ld % 8 == 0. */
int stonk_softmax_xent_f16_fwd_bwd(const void* logits_f16, int64_t ld, int ncols, int npad, const int* targets,
                                   const int* count_dev, float* loss_sum, void* dlogits, int64_t ld_d, float grad_scale,
                                   int cap_rows, int* err_flag, void* stream);
/* NSP loss (ref:stonkgs_model.py:241-243): loss_sum_cnt[0] += sum, [1] += number of labels. */
int stonk_nsp_xent_fwd_bwd(const float* logits, const int64_t* labels, int B, int C, float* loss_sum_cnt, float* dlogits,
                           float grad_scale, int* err_flag, void* stream);
/* loss_out[0..3] = total, text MLM, entity MLM, NSP (ref:stonkgs_model.py:245). */
int stonk_loss_finalize(const float* text_sum, const int* text_cnt, const float* ent_sum, const int* ent_cnt,
                        const float* nsp_sum_cnt, float* loss_out, void* stream);

/* BertPooler / NSP classifier on fp32 master weights (hf:modeling_bert.py:457-463, :523-527). */
int stonk_small_linear_fwd(const void* x, int64_t ldx, const float* W, const float* bias, float* y, int M, int N, int K,
                           int act, void* stream);
int stonk_small_linear_bwd(const float* dy, const float* y, const void* x, int64_t ldx, const float* W, float* dW,
                           float* db, float* dx_f32, void* dx_bf16_accum, int64_t ld_dxb, int M, int N, int K, int act,
                           void* stream);

/* Classification head plumbing (ref:src/stonkgs/models/stonkgs_finetuning.py:310-330): dropout on the pooled fp32
 * vector (replayable from the seed), and *out = *num / *den for the mean of a device-side loss sum. */
int stonk_dropout_f32(const float* x, float* y, int64_t n, float p, uint32_t seed, void* stream);
int stonk_ratio_f32(const float* num, const float* den, float* out, void* stream);
/* The classification head's other two losses (ref:stonkgs_finetuning.py:328-338): `mode` = STONK_LOSS_MSE (regression,
 * nn.MSELoss), STONK_LOSS_MSE_BROADCAST (num_labels = 1 with 1-D labels: torch broadcasts [B,1] against [B] to [B,B], and
 * so does this), STONK_LOSS_BCE (multi-label, nn.BCEWithLogitsLoss). logits / targets fp32 [B,C] ([B] targets for the
 * broadcast mode); *loss_out = the mean; dlogits (nullable) = d(loss * grad_scale)/d(logits). */
int stonk_elementwise_loss_fwd_bwd(const float* logits, const float* targets, int B, int C, int mode, float* loss_out,
                                   float* dlogits, float grad_scale, void* stream);

/* du = dg * gelu'(u), bf16 elementwise (backward of hf:modeling_bert.py:478, the head transform's activation). */
int stonk_gelu_bwd_bf16(const void* dg, const void* u, void* du, int64_t n, void* stream);

/* Optimizer step pieces (hf:trainer.py:1780-1796 as driven by ref:src/stonkgs/models/stonkgs_pretraining.py:171-223):
 * *out_accum += sum(x^2), summed in a fixed order (bitwise repeatable: data-parallel replicas rely on it) through a caller
 * workspace of stonk_sumsq_workspace_floats() floats - zero it once after allocating it; the kernel leaves it ready for
 * the next launch; concurrent launches (different streams) need different workspaces;
 * fused clip_grad_norm_(max_grad_norm) + AdamW + bf16 weight refresh + grad zeroing. */
int64_t stonk_sumsq_workspace_floats(void);
int stonk_sumsq_f32(const float* x, int64_t n, float* out_accum, float* workspace, int64_t ws_floats, void* stream);
/* stonk_adamw_step works on any piece of the flat buffers (pointers into p / g / p_bf16 at the piece, m / v wherever the
 * caller keeps that piece's state: an optimizer sharded over data-parallel ranks updates its 1/world of every gradient
 * bucket). decay_spans (nullable, device): n_spans sorted [lo, hi) element ranges of the WHOLE flat buffer that receive
 * the decoupled weight decay (HF Trainer decays weights, not biases / LayerNorm); span_base = the piece's offset in the
 * flat buffer (% 4 == 0). Without a table, weight_decay applies to every element. */
int stonk_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float bias_corr1, float bias_corr2, const float* gnorm_sq_dev,
                     float max_grad_norm, float grad_scale, const int64_t* decay_spans, int n_spans, int64_t span_base,
                     void* stream);
int stonk_scale_f32(float* x, int64_t n, float s, void* stream);

/* ---- Data-parallel gradient exchange (csrc/comm.hip): RCCL collectives on a stream the LIBRARY owns, handed over by
 * events. Replaces torch DistributedDataParallel's bucketed all-reduce, which the reference gets from HF Trainer when it
 * is launched distributed (ref:src/stonkgs/models/stonkgs_pretraining.py:215-223), and - reduce-scatter / all-gather -
 * DeepSpeed ZeRO-2's exchange when `deepspeed=True` (:174-175).
 *
 *   stonk_comm_unique_id   rank 0 fills 128 bytes (an ncclUniqueId); the CALLER carries them to the other ranks (a file,
 *                          a socket, torch's store - the library opens no connection of its own for that);
 *   stonk_comm_init        one communicator per process (one process per GPU): RCCL over xGMI inside a node; creates the
 *                          communicator's stream and two events. `*comm_out` is an opaque handle;
 *   stonk_comm_*_async     `buf` / `send` / `recv` are device pointers, `n` counts ELEMENTS, dtype 0 = fp32, 1 = bf16, the
 *                          reduction is a sum (the mean is the optimizer's grad_scale). The collective is ordered behind
 *                          everything `after_stream` has enqueued at the call (event hand-off: the producer stream does
 *                          not wait, the host does not block) and runs on the communicator's stream. allreduce: in place.
 *                          reduce_scatter: send holds world * recv_n elements, recv (which may be the rank's own slice of
 *                          send) receives the sum of slice `rank`; allgather: recv holds world * send_n elements (send
 *                          may be the rank's own slice of recv);
 *   stonk_comm_wait        `stream` waits (on the device) for every collective issued so far; no host synchronisation;
 *   stonk_comm_stream      the communicator's hipStream_t (for a profiler range, or to order caller work behind it);
 *   stonk_comm_destroy     drains the stream, destroys communicator, stream and events.
 * Return codes as everywhere (0, STONK_E*, a hipError_t); an RCCL failure r is reported as 10000 + r. RCCL itself is
 * resolved at run time: without it these entry points return STONK_EINVAL and the rest of the library is unaffected. */
int stonk_comm_unique_id(void* id_out_128_bytes);
int stonk_comm_init(void** comm_out, int world, int rank, const void* unique_id_128_bytes, int device);
int stonk_comm_allreduce_async(void* comm, void* buf, int64_t n, int dtype, void* after_stream);
int stonk_comm_reduce_scatter_async(void* comm, const void* send, void* recv, int64_t recv_n, int dtype, void* after_stream);
int stonk_comm_allgather_async(void* comm, const void* send, void* recv, int64_t send_n, int dtype, void* after_stream);
int stonk_comm_wait(void* comm, void* stream);
void* stonk_comm_stream(void* comm);
int stonk_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* STONK_HIP_H */
