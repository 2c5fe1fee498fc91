"""ctypes binding of the C-ABI kernel library ``csrc/libstonk_hip.so`` (declared in ``include/stonk_hip.h``).

The product path has no CPU fallback: if the shared library is missing or a launcher returns a
non-zero status, this module raises. PyTorch is used by the callers only for device memory and
streams; every pointer crossing this boundary is a raw device address.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (STONK_HIP_LIB: a differently built copy of the SAME library - the host-side AddressSanitizer build of `make asan`)
LIB_PATH = os.environ.get("STONK_HIP_LIB") or os.path.join(_HERE, "csrc", "libstonk_hip.so")

# ---- flags (mirror of csrc/stonk_flags.h) ----
EPI_OUT_BF16 = 0
EPI_OUT_F32 = 1
EPI_OUT_F32_ATOMIC = 2
EPI_OUT_F16 = 3
EPI_BIAS = 1 << 2
EPI_GELU = 1 << 3
EPI_RESID = 1 << 4
EPI_SAVE_PREACT = 1 << 5
EPI_GELU_BWD = 1 << 6
EPI_DROPOUT = 1 << 7
EPI_AUX_GRAD = 1 << 8   # aux = gelu'(pre-activation): stored by SAVE_PREACT (with GELU), multiplied in by GELU_BWD
GEMM_AUTO, GEMM_TILE128, GEMM_WAVE8, GEMM_WAVE4, GEMM_WAVE4_192, GEMM_DISPATCHED, GEMM_DISPATCHED2 = 0, 1, 2, 3, 4, 5, 6   # stonk_gemm_nt_bf16 `kernel`
GEMM_ASM4, GEMM_ASM4_192 = 7, 8   # the written-out four-wave kernel (gemm_a4.hip), 256x256 / 256x192 tiles
LN_DROPOUT = 1 << 0
LN_DEFER_REDUCE = 1 << 1   # stonk_layernorm_bwd leaves its partial sums for stonk_layernorm_bwd_reduce
SMALL_TANH = 1
SMALL_X_F32 = 16
LOSS_MSE, LOSS_MSE_BROADCAST, LOSS_BCE = 0, 1, 2
ATTN_BWD_DELTA, ATTN_BWD_DQ, ATTN_BWD_DKV, ATTN_BWD_ALL = 1, 2, 4, 7

_vp, _i32, _i64, _f32, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32

# name -> argtypes; every launcher returns int (0 ok, <0 bad argument, >0 hipError_t)
_SIGNATURES = {
    "stonk_gemm_nt_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _i64,
                           _f32, _i32, _vp, _vp, _f32, _u32, _i32, _vp],
    "stonk_gemm_tn_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp],
    "stonk_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _f32, _u32, _vp],
    "stonk_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _u32, _f32, _u32,
                            _vp, _i64, _vp],
    "stonk_layernorm_bwd_reduce": [_vp, _i64, _i32, _vp, _vp, _vp],
    "stonk_joint_embed_ln_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                 _i64, _i32, _f32, _i32, _f32, _u32, _vp, _vp, _i64, _vp],
    "stonk_unpad_plan": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "stonk_text_embed_ln_fwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _f32, _i32, _f32,
                                _u32, _vp, _vp],
    "stonk_embed_grad": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "stonk_attention_fwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp],
    "stonk_attention_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i32,
                            _i32, _i32, _i32, _f32, _f32, _u32, _vp],
    "stonk_attention_bwd_phases": [_i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp,
                                   _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp],
    "stonk_transpose_bf16": [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp],
    "stonk_transpose_f32_to_bf16": [_vp, _vp, _i64, _i32, _i64, _vp],
    "stonk_transpose_bf16_batched": [_vp, _i32, _i32, _vp],
    "stonk_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "stonk_mlm_mask": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _u32, _vp],
    "stonk_assemble_rows": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _f32, _u32, _vp,
                            _vp],
    "stonk_label_compact": [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "stonk_gather_rows_bf16": [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp],
    "stonk_scatter_rows_bf16": [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp],
    "stonk_scatter_rows_f32_to_bf16": [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp],
    "stonk_softmax_xent_fwd_bwd": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp, _vp],
    "stonk_softmax_xent_f16_fwd_bwd": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp, _vp],
    "stonk_nsp_xent_fwd_bwd": [_vp, _vp, _i32, _i32, _vp, _vp, _f32, _vp, _vp],
    "stonk_loss_finalize": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "stonk_small_linear_fwd": [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "stonk_small_linear_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "stonk_dropout_f32": [_vp, _vp, _i64, _f32, _u32, _vp],
    "stonk_ratio_f32": [_vp, _vp, _vp, _vp],
    "stonk_elementwise_loss_fwd_bwd": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _f32, _vp],
    "stonk_gelu_bwd_bf16": [_vp, _vp, _vp, _i64, _vp],
    "stonk_sumsq_f32": [_vp, _i64, _vp, _vp, _i64, _vp],
    "stonk_adamw_step": [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _f32, _f32,
                         _vp, _i32, _i64, _vp],
    "stonk_scale_f32": [_vp, _i64, _f32, _vp],
    # data-parallel gradient exchange: RCCL on a library-owned stream (csrc/comm.hip)
    "stonk_comm_unique_id": [_vp],
    "stonk_comm_init": [C.POINTER(C.c_void_p), _i32, _i32, _vp, _i32],
    "stonk_comm_allreduce_async": [_vp, _vp, _i64, _i32, _vp],
    "stonk_comm_reduce_scatter_async": [_vp, _vp, _vp, _i64, _i32, _vp],
    "stonk_comm_allgather_async": [_vp, _vp, _vp, _i64, _i32, _vp],
    "stonk_comm_wait": [_vp, _vp],
    "stonk_comm_destroy": [_vp],
}


class StonkHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the C-ABI library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StonkHipError(
                f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
                "There is no CPU fallback for the STonKGs hot path."
            )
        # torch bundles its own libamdhip64 (same soname as /opt/rocm's): import it FIRST so that this library
        # binds to the HIP runtime torch already initialised - one runtime per process, shared streams
        import torch  # noqa: F401

        handle = C.CDLL(LIB_PATH)
        missing = [name for name in _SIGNATURES if not hasattr(handle, name)]
        if missing and os.environ.get("STONK_DEV_PARTIAL") == "1":  # kernel bring-up only
            for name in missing:
                _SIGNATURES.pop(name)
        elif missing:  # header/library mismatch: refuse to run on a stale build
            raise StonkHipError(f"{LIB_PATH} does not export {missing}; rebuild with `make`")
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        handle.stonk_abi_version.argtypes = []
        handle.stonk_abi_version.restype = C.c_int
        handle.stonk_layernorm_bwd_workspace_floats.argtypes = [_i64, _i32]   # a size query, not a launcher: returns the size
        handle.stonk_layernorm_bwd_workspace_floats.restype = C.c_int64
        handle.stonk_sumsq_workspace_floats.argtypes = []
        handle.stonk_sumsq_workspace_floats.restype = C.c_int64
        handle.stonk_unpad_workspace_ints.argtypes = [_i32]
        handle.stonk_unpad_workspace_ints.restype = C.c_int64
        handle.stonk_comm_stream.argtypes = [_vp]
        handle.stonk_comm_stream.restype = C.c_void_p
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(list(_SIGNATURES) + ["stonk_abi_version", "stonk_layernorm_bwd_workspace_floats",
                                     "stonk_sumsq_workspace_floats", "stonk_unpad_workspace_ints", "stonk_comm_stream"])


def check(status: int, name: str) -> None:
    if status != 0:
        kind = "bad argument" if status < 0 else "hipError_t"
        raise StonkHipError(f"{name} failed: status {status} ({kind})")


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)


def ptr(t) -> int:
    """Raw device address of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
