// Epilogue / mode flags shared by the HIP sources and include/stonk_hip.h (which includes this file).
#pragma once
// --- stonk_gemm_nt_bf16 `flags` ---
#define STONK_EPI_OUT_BF16 0        /* C is bf16 */
#define STONK_EPI_OUT_F32 1         /* C is fp32 */
#define STONK_EPI_OUT_F32_ATOMIC 2  /* C (fp32) += result, via global_atomic_add_f32; required for split_k > 1 */
#define STONK_EPI_OUT_F16 3         /* C is IEEE fp16, clamped to +-65504 (label-sparse decoder logits: fp32's range is not
                                       needed, and 11 significant bits against bf16's 8 keep softmax terms to 0.05 %);
                                       no other epilogue flag */
#define STONK_EPI_OUT_MASK 3
#define STONK_EPI_BIAS (1 << 2)         /* + bias[n] (fp32) */
#define STONK_EPI_GELU (1 << 3)         /* exact erf GELU */
#define STONK_EPI_RESID (1 << 4)        /* + resid[m][n] (bf16), after activation/dropout */
#define STONK_EPI_SAVE_PREACT (1 << 5)  /* aux[m][n] (bf16) = value before the activation */
#define STONK_EPI_GELU_BWD (1 << 6)     /* result *= gelu'(aux[m][n]) */
#define STONK_EPI_DROPOUT (1 << 7)      /* inverted dropout (drop_p, seed) before the residual add */
#define STONK_EPI_AUX_GRAD (1 << 8)     /* aux holds gelu'(pre-activation), not the pre-activation: SAVE_PREACT (with GELU) \
                                           stores it, GELU_BWD multiplies by it - the erf/exp of the backward epilogue \
                                           are paid once, in the forward one, which evaluates them anyway */
// --- stonk_gemm_nt_bf16 `kernel`: which of the three NT kernels runs the launch ---
#define STONK_GEMM_AUTO 0     /* the launcher's choice, from the shape and the epilogue (measured on MI355X, gemm_bf16.hip) */
#define STONK_GEMM_TILE128 1  /* 128x128x64 tiles, two workgroups per CU, a grid the hardware schedules dynamically: the \
                                 form to ask for when another stream (a collective) holds CUs during the launch */
#define STONK_GEMM_WAVE8 2    /* persistent 256x256x64, eight waves, one workgroup per CU (gemm256.hip) */
#define STONK_GEMM_WAVE4 3    /* persistent 256x256x64, four waves with 128x128 wave tiles (gemm_w4.hip) */
#define STONK_GEMM_WAVE4_192 4 /* the same kernel on 256x192 tiles (128x96 wave tiles): N % 192 == 0, bf16 output, the \
                                 epilogues of the N = 768 launches (none, bias, residual, bias + residual [+ dropout]) */
#define STONK_GEMM_DISPATCHED 5 /* the launcher's choice as with AUTO, launched so that the hardware dispatcher hands out the \
                                 work: the four-wave kernel with ONE work item per workgroup (grid = tiles; 2-8 % slower alone - no \
                                 prefetch across tile boundaries), 128x128 tiles where AUTO would take the eight-wave kernel. For a \
                                 launch beside which another stream (a collective) holds CUs: a persistent grid with a static tile \
                                 split waits for its last workgroup to get a CU */
#define STONK_GEMM_DISPATCHED2 6 /* the same with TWO work items per workgroup (grid = tiles / 2): every second tile boundary                                  keeps its prefetch, the dispatcher still hands out the work in pieces far smaller than a                                  CU's share; where DISPATCHED runs 128x128 tiles this is DISPATCHED */
#define STONK_GEMM_ASM4 7     /* persistent 256x256x64, four waves, 16x16x32 MFMAs, LDS-DMA operands, the K loop a written-out \
                                 instruction stream (gemm_a4.hip): bf16 output (split_k == 1, the step's epilogues), or the \
                                 plain product as fp16 (STONK_EPI_OUT_F16: the label-sparse logits; any M x ldc) */
#define STONK_GEMM_ASM4_192 8 /* the same on 256x192 tiles: N % 192 == 0, the epilogues of the N = 768 launches; also the plain \
                                 product added into fp32 by atomics (STONK_EPI_OUT_F32_ATOMIC, alpha as given) over a K split - \
                                 there split_k is an UPPER bound: the kernel takes as many shares as fill its grid once */
// --- stonk_layernorm_* `flags` ---
#define STONK_LN_DROPOUT (1 << 0)
#define STONK_LN_DEFER_REDUCE (1 << 1) /* stonk_layernorm_bwd: leave the dgamma / dbeta partial sums in the workspace; \
                                        stonk_layernorm_bwd_reduce adds them (on any stream, after this launch) */
// --- stonk_small_linear_* `act` ---
#define STONK_SMALL_TANH 1
#define STONK_SMALL_X_F32 16 /* x is fp32 (default bf16) */
// --- stonk_elementwise_loss_fwd_bwd `mode` (ref:src/stonkgs/models/stonkgs_finetuning.py:328-338) ---
#define STONK_LOSS_MSE 0            /* regression: MSELoss over [B,C] */
#define STONK_LOSS_MSE_BROADCAST 1  /* regression with num_labels = 1 and 1-D labels: torch's [B,1] x [B] -> [B,B] broadcast */
#define STONK_LOSS_BCE 2            /* multi-label: BCEWithLogitsLoss over [B,C] */
// --- stonk_attention_bwd_phases `phases`: which kernels of the attention backward a call launches ---
#define STONK_ATTN_BWD_DELTA 1 /* delta = rowsum(dO * O) by a kernel of its own: must have completed before DKV starts */
#define STONK_ATTN_BWD_DQ 2    /* dQ (stores delta itself only in the ALL form) */
#define STONK_ATTN_BWD_DKV 4   /* dK, dV (reads delta) */
#define STONK_ATTN_BWD_ALL 7   /* = stonk_attention_bwd: dQ (producing delta), then dK / dV, on one stream */
