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
#define STONK_EPI_DEBUG_REGSTAGE (1 << 16) /* A/B test: register staging instead of LDS-DMA (128x128 kernel) */
#define STONK_EPI_DEBUG_V1 (1 << 17)       /* force the 128x128 two-barrier kernel */
#define STONK_EPI_DEBUG_V2 (1 << 18)       /* force the persistent 256x256 kernel */
#define STONK_EPI_DEBUG_SIDE_V1 (1 << 19)  /* A/B test: keep side-operand epilogues on the 128x128 kernel */
#define STONK_EPI_DEBUG_W4 (1 << 20)       /* force the four-wave 256x256 kernel (gemm_w4.hip) */
// --- stonk_layernorm_* `flags` ---
#define STONK_LN_DROPOUT (1 << 0)
// --- stonk_small_linear_* `act` ---
#define STONK_SMALL_TANH 1
#define STONK_SMALL_X_F32 16 /* x is fp32 (default bf16) */
