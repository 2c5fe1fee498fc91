"""GPU parity: bf16 MFMA GEMM (C-ABI stonk_gemm_nt_bf16) against torch fp32 matmul on the same bf16 inputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(hip, A, B, out_dtype=torch.bfloat16, M=None, flags=0, bias=None, resid=None, aux=None, alpha=1.0,
          split_k=1, C=None, m_dev=None, k_dev=None, drop_p=0.0, seed=0, kernel=0):
    M = A.shape[0] if M is None else M
    N, K = B.shape
    if C is None:
        C = torch.empty(M, N, device=A.device, dtype=out_dtype)
    hip.call("stonk_gemm_nt_bf16", hip.ptr(A), A.stride(0), hip.ptr(B), B.stride(0), hip.ptr(C), C.stride(0), M, N, K,
             flags, hip.ptr(bias), hip.ptr(resid), 0 if resid is None else resid.stride(0), hip.ptr(aux),
             0 if aux is None else aux.stride(0), alpha, split_k, hip.ptr(m_dev), hip.ptr(k_dev), drop_p, seed,
             kernel, hip.stream_ptr())
    return C


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


T128, WAVE8, WAVE4 = 1, 2, 3   # stonk_gemm_nt_bf16 `kernel` (STONK_GEMM_TILE128 / _WAVE8 / _WAVE4; 0 = AUTO)
KERNELS = {"v1": T128, "v2_256": WAVE8, "w4_256": WAVE4, "auto": 0}


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (1000, 768, 3072), (77, 2304, 768),
                                   (2000, 1152, 1024), (4096, 29056, 128)])
def test_gemm_plain_bf16_and_f32(hip, M, N, K, kernel):
    A, B = _rand((M, K), seed=1), _rand((N, K), seed=2)
    ref = A.float() @ B.float().t()
    dbg = KERNELS[kernel]
    if kernel == "w4_256" and (K // 64) % 2:   # the four-wave kernel walks K tiles in pairs: an explicit request is refused
        with pytest.raises(hip.StonkHipError, match="-2"):
            _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32, kernel=dbg)
        return
    out32 = _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32, kernel=dbg)
    torch.cuda.synchronize()
    err = (out32 - ref).abs().max().item()
    assert err <= 1e-3 * math.sqrt(K), f"fp32-out max err {err}"
    out16 = _gemm(hip, A, B, torch.bfloat16, flags=hip.EPI_OUT_BF16, kernel=dbg)
    torch.testing.assert_close(out16.float(), ref, rtol=1.6e-2, atol=1e-2 * math.sqrt(K) / 8)
    if kernel in ("v1", "v2_256", "auto"):   # fp16 output (label-sparse logits): the fp32 result rounded once, to 11 bits
        outh = _gemm(hip, A, B, torch.float16, flags=hip.EPI_OUT_F16, kernel=dbg)
        torch.testing.assert_close(outh.float(), out32.half().float(), rtol=1e-3, atol=1e-3)


def test_gemm_f16_output_saturates_and_takes_no_epilogue(hip):
    A = torch.full((128, 64), 200.0, device="cuda", dtype=torch.bfloat16)
    B = torch.full((128, 64), 200.0, device="cuda", dtype=torch.bfloat16)
    B[1] = -200.0
    out = _gemm(hip, A, B, torch.float16, flags=hip.EPI_OUT_F16)   # 64 * 200 * 200 = 2.56e6 > 65504
    assert torch.isfinite(out).all() and float(out[0, 0]) == 65504.0 and float(out[0, 1]) == -65504.0
    C = torch.empty(128, 128, device="cuda", dtype=torch.float16)
    bias = torch.zeros(128, device="cuda")
    rc = hip.lib().stonk_gemm_nt_bf16(hip.ptr(A), 64, hip.ptr(B), 64, hip.ptr(C), 128, 128, 128, 64,
                                      hip.EPI_OUT_F16 | hip.EPI_BIAS, hip.ptr(bias), 0, 0, 0, 0, 1.0, 1, 0, 0, 0.0, 0,
                                      0, hip.stream_ptr())
    assert rc == -1


def test_gemm_asymmetric_identity(hip):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    K = 128
    A = torch.eye(K, device="cuda", dtype=torch.bfloat16)
    B = (torch.arange(256 * K, device="cuda").reshape(256, K) % 251).to(torch.bfloat16)
    out = _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32)
    torch.testing.assert_close(out, B.float().t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("dbg", [T128, WAVE8, WAVE4])
def test_gemm_epilogues(hip, dbg):
    M, N, K = (304, 256, 256) if dbg == WAVE4 else (300, 256, 192)   # four-wave kernel: K tiles in pairs, M % 8
    A, B = _rand((M, K), 0.5, 3), _rand((N, K), 0.5, 4)
    bias = torch.randn(N, device="cuda")
    resid = _rand((M, N), 1.0, 5)
    pre = A.float() @ B.float().t() + bias
    # bias + gelu + saved pre-activation
    aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT, kernel=dbg, bias=bias, aux=aux)
    torch.testing.assert_close(aux.float(), pre, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(pre), rtol=1e-2, atol=2e-2)
    # bias + residual
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_RESID, kernel=dbg, bias=bias, resid=resid)
    torch.testing.assert_close(out.float(), pre + resid.float(), rtol=1e-2, atol=3e-2)
    # gelu backward: result * gelu'(aux)
    u = _rand((M, N), 1.0, 6)
    out = _gemm(hip, A, B, flags=hip.EPI_GELU_BWD, kernel=dbg, aux=u)
    uf = u.float().requires_grad_(True)
    (gp,) = torch.autograd.grad(torch.nn.functional.gelu(uf).sum(), uf)
    torch.testing.assert_close(out.float(), (A.float() @ B.float().t()) * gp, rtol=1e-2, atol=3e-2)
    # the same pair with STONK_EPI_AUX_GRAD: forward leaves gelu'(pre-activation) in aux, backward multiplies by aux as is
    aux2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT | hip.EPI_AUX_GRAD, kernel=dbg, bias=bias,
                aux=aux2)
    pf = pre.clone().requires_grad_(True)
    (gpre,) = torch.autograd.grad(torch.nn.functional.gelu(pf).sum(), pf)
    torch.testing.assert_close(aux2.float(), gpre, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(pre), rtol=1e-2, atol=2e-2)
    out = _gemm(hip, A, B, flags=hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD, kernel=dbg, aux=u)
    torch.testing.assert_close(out.float(), (A.float() @ B.float().t()) * u.float(), rtol=1e-2, atol=3e-2)
    # the modifier alone is refused
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    rc = hip.lib().stonk_gemm_nt_bf16(hip.ptr(A), K, hip.ptr(B), K, hip.ptr(C), N, M, N, K, hip.EPI_AUX_GRAD, 0, 0, 0, 0,
                                      0, 1.0, 1, 0, 0, 0.0, 0, 0, hip.stream_ptr())
    assert rc == -1


@pytest.mark.parametrize("dbg,M,N,K,sk", [(T128, 256, 128, 2048, 8), (WAVE8, 768, 768, 8192, 7),
                                          (WAVE4, 768, 768, 8192, 8), (WAVE4, 304, 512, 4096, 16),
                                          (WAVE8, 300, 256, 4096, 32), (0, 3072, 768, 32768, 8)])
def test_gemm_splitk_atomic_accumulates(hip, dbg, M, N, K, sk):
    A, B = _rand((M, K), 0.3, 7), _rand((N, K), 0.3, 8)
    C = torch.ones(M, N, device="cuda")
    _gemm(hip, A, B, flags=hip.EPI_OUT_F32_ATOMIC, kernel=dbg, split_k=sk, C=C)
    torch.testing.assert_close(C, 1.0 + A.float() @ B.float().t(), rtol=1e-4, atol=2e-3 * (K / 2048) ** 0.5)


def test_gemm256_device_counts_and_races(hip):
    """Persistent 256x256 kernel: device-side M and K, many work items per workgroup, repeated launches must be
    bit-identical (an LDS-DMA race would show up as rare differing tiles)."""
    M, N, K = 16384, 768, 1024
    A, B = _rand((M, K), 0.5, 21), _rand((N, K), 0.5, 22)
    m_dev = torch.tensor([2432 - 5], device="cuda", dtype=torch.int32)
    C = torch.full((M, N), -7.0, device="cuda")
    _gemm(hip, A, B, flags=hip.EPI_OUT_F32, C=C, m_dev=m_dev, kernel=WAVE8)
    ref = A[:2427].float() @ B.float().t()
    torch.testing.assert_close(C[:2427], ref, rtol=1e-4, atol=2e-3)
    assert (C[2427:] == -7.0).all()
    # device-side contraction length (label-sparse wgrad): K tiles past ceil(k/64) are skipped
    k_dev = torch.tensor([300], device="cuda", dtype=torch.int32)
    A2, B2 = _rand((1024, 4096), 0.5, 23), _rand((768, 4096), 0.5, 24)
    A2[:, 300:] = 0  # what the zero-filling transpose guarantees up to the 64 round-up
    C2 = torch.zeros(1024, 768, device="cuda")
    _gemm(hip, A2, B2, flags=hip.EPI_OUT_F32_ATOMIC, split_k=4, C=C2, k_dev=k_dev, kernel=WAVE8)
    torch.testing.assert_close(C2, A2[:, :320].float() @ B2[:, :320].float().t(), rtol=1e-4, atol=2e-3)
    # race screen
    A3, B3 = _rand((8192, 768), 1.0, 25), _rand((3072, 768), 0.05, 26)
    first = _gemm(hip, A3, B3, kernel=WAVE8)
    torch.testing.assert_close(first.float(), A3.float() @ B3.float().t(), rtol=2e-2, atol=2e-2)
    for _ in range(20):
        assert torch.equal(_gemm(hip, A3, B3, kernel=WAVE8), first)


def test_gemm_w4_device_rows_races_and_fused_epilogue(hip):
    """Four-wave 256x256 kernel: device-side M with a ragged last tile, many work items per workgroup, bit-identical
    repeats (a ring-slot race would show as rare differing tiles), and the bias + dropout + residual epilogue."""
    M, N, K = 16384, 768, 1024
    A, B = _rand((M, K), 0.5, 31), _rand((N, K), 0.5, 32)
    m_dev = torch.tensor([2432 - 5], device="cuda", dtype=torch.int32)
    C = torch.full((M, N), -7.0, device="cuda")
    _gemm(hip, A, B, flags=hip.EPI_OUT_F32, C=C, m_dev=m_dev, kernel=WAVE4)
    torch.testing.assert_close(C[:2427], A[:2427].float() @ B.float().t(), rtol=1e-4, atol=2e-3)
    assert (C[2427:] == -7.0).all()
    A3, B3 = _rand((8192, 768), 1.0, 35), _rand((3072, 768), 0.05, 36)
    first = _gemm(hip, A3, B3, kernel=WAVE4)
    torch.testing.assert_close(first.float(), A3.float() @ B3.float().t(), rtol=2e-2, atol=2e-2)
    for _ in range(20):
        assert torch.equal(_gemm(hip, A3, B3, kernel=WAVE4), first)
    # N = 768 output with K = 3072 (FFN down): bias + dropout + residual, compared through the kept elements
    A4, B4 = _rand((4096, 3072), 0.5, 37), _rand((768, 3072), 0.05, 38)
    bias = torch.randn(768, device="cuda")
    resid = _rand((4096, 768), 1.0, 39)
    fl = hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT
    out = _gemm(hip, A4, B4, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=77, kernel=WAVE4).float()
    pre = A4.float() @ B4.float().t() + bias
    delta = out - resid.float()                      # = keep ? pre / 0.9 : 0
    kept = (delta.abs() > 1e-3) | (pre.abs() < 1e-2)
    assert abs(1.0 - kept.float().mean().item() - 0.1) < 0.01
    torch.testing.assert_close(torch.where(kept, delta, torch.zeros_like(delta)),
                               torch.where(kept, pre / 0.9, torch.zeros_like(pre)), rtol=3e-2, atol=6e-2)
    # same mask from the 128x128 kernel (one dropout stream per (seed, row, column), whatever the tiling)
    out_v1 = _gemm(hip, A4, B4, flags=hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT, kernel=T128, bias=bias,
                   resid=resid, drop_p=0.1, seed=77).float()
    assert ((out_v1 - resid.float()).abs() > 1e-3).eq((delta.abs() > 1e-3)).float().mean().item() > 0.999


def test_gemm_device_row_count_and_dropout(hip):
    M, N, K = 1024, 256, 128
    A, B = _rand((M, K), 0.5, 9), _rand((N, K), 0.5, 10)
    m_dev = torch.tensor([333], device="cuda", dtype=torch.int32)
    C = torch.full((M, N), -7.0, device="cuda")
    _gemm(hip, A, B, flags=hip.EPI_OUT_F32, C=C, m_dev=m_dev)
    ref = A.float() @ B.float().t()
    torch.testing.assert_close(C[:333], ref[:333], rtol=1e-4, atol=1e-3)
    assert (C[333:] == -7.0).all()  # rows past the device-side count are never written
    # dropout: keep-rate and scaling
    out = _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32 | hip.EPI_DROPOUT, drop_p=0.1, seed=123)
    kept = out != 0
    rate = 1.0 - kept.float().mean().item()
    assert abs(rate - 0.1) < 0.01, rate
    torch.testing.assert_close(out[kept], (ref / 0.9)[kept], rtol=1e-4, atol=1e-3)
    out2 = _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32 | hip.EPI_DROPOUT, drop_p=0.1, seed=123)
    assert torch.equal(out, out2)  # same (seed, index) -> same mask, bit for bit


@pytest.mark.parametrize("M,N,K,fl", [(4096, 768, 768, 0), (2048, 768, 3072, "resid"), (2048, 3072, 768, "gelu_bwd"), (1024, 2304, 768, "bias"),
                                      (1024, 1536, 768, 0), (300, 768, 768, 0), (1280, 768, 768, 0)])
def test_gemm_dispatched_is_auto_without_a_persistent_grid(hip, M, N, K, fl):
    """STONK_GEMM_DISPATCHED: the launcher's own choice, one work item per workgroup (the form for launches beside a
    collective). Same kernel, same tiles, same arithmetic: bit-identical to AUTO - also where AUTO takes the eight-wave
    kernel (wide plain launches; DISPATCHED then uses the 128x128 one: equal within rounding) or 128x128 tiles anyway."""
    A, B = _rand((M, K), 1.0, 61), _rand((N, K), 0.05, 62)
    kw = {}
    flags = 0
    if fl == "resid":
        flags, kw = hip.EPI_RESID, {"resid": _rand((M, N), 1.0, 63)}
    elif fl == "gelu_bwd":
        flags, kw = hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD, {"aux": _rand((M, N), 1.0, 64)}
    elif fl == "bias":
        flags, kw = hip.EPI_BIAS, {"bias": torch.randn(N, device="cuda")}
    auto = _gemm(hip, A, B, flags=flags, kernel=hip.GEMM_AUTO, **kw)
    for kernel in (hip.GEMM_DISPATCHED, hip.GEMM_DISPATCHED2):   # (one / two work items per workgroup; odd tile counts included)
        disp = _gemm(hip, A, B, flags=flags, kernel=kernel, **kw)
        if N == 1536:
            torch.testing.assert_close(disp.float(), auto.float(), rtol=2e-2, atol=2e-2)
        else:
            assert torch.equal(auto, disp)


def test_gemm_bad_arguments(hip):
    A, B = _rand((128, 64)), _rand((100, 64))
    with pytest.raises(hip.StonkHipError):
        _gemm(hip, A, B)  # N not a multiple of 128


@pytest.mark.parametrize("T,Mo,No,sk", [(64, 128, 128, 1), (4096, 768, 768, 12), (8192, 2304, 768, 4), (2048, 768, 3072, 3),
                                        (640, 29056, 128, 1)])
def test_gemm_tn_weight_and_bias_gradient(hip, T, Mo, No, sk):
    """dW += dY^T X and db += colsum(dY) straight from row-major [token][feature] operands (transposed LDS reads)."""
    dY, X = _rand((T, Mo), 0.5, 31), _rand((T, No), 0.5, 32)
    dW = torch.full((Mo, No), 0.25, device="cuda")
    db = torch.full((Mo,), -1.0, device="cuda")
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, hip.ptr(db), Mo, No, T, 0.5, sk, 0,
             hip.stream_ptr())
    ref = 0.25 + 0.5 * (dY.float().t() @ X.float())
    torch.testing.assert_close(dW, ref, rtol=1e-4, atol=2e-3 * (T / 2048) ** 0.5)
    torch.testing.assert_close(db, -1.0 + 0.5 * dY.float().sum(0), rtol=1e-4, atol=2e-3 * (T / 2048) ** 0.5)


def test_gemm_tn_asymmetric_and_device_token_count(hip):
    # exact integer data catches any fragment / swizzle mix-up: dY = one-hot rows, X = row index pattern
    T, Mo, No = 256, 128, 256
    dY = torch.zeros(T, Mo, device="cuda", dtype=torch.bfloat16)
    dY[torch.arange(T), (torch.arange(T) * 7) % Mo] = 1.0
    X = ((torch.arange(T * No, device="cuda").reshape(T, No) % 61) - 30).to(torch.bfloat16)
    dW = torch.zeros(Mo, No, device="cuda")
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, 0, Mo, No, T, 1.0, 2, 0,
             hip.stream_ptr())
    assert torch.equal(dW, dY.float().t() @ X.float())
    # token count from device memory: rows in [k, roundup64(k)) are zero by contract, later K steps are skipped
    k = 100
    dY2, X2 = _rand((T, Mo), 1.0, 33), _rand((T, No), 1.0, 34)
    dY2[k:128] = 0
    X2[k:128] = 0
    k_dev = torch.tensor([k], device="cuda", dtype=torch.int32)
    dW2 = torch.zeros(Mo, No, device="cuda")
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY2), Mo, hip.ptr(X2), No, hip.ptr(dW2), No, 0, Mo, No, T, 1.0, 1,
             hip.ptr(k_dev), hip.stream_ptr())
    torch.testing.assert_close(dW2, dY2[:k].float().t() @ X2[:k].float(), rtol=1e-4, atol=1e-3)


def test_gemm_tn_256_exact_and_matches_small_kernel(hip):
    """The four-wave 256x256 weight-gradient kernel (split_k = 0): exact integer data (any fragment / sub-tile mix-up
    shows), bias sums, agreement with the 128x128 kernel and the eight-wave form (split_k = -1)."""
    T, Mo, No = 2048, 768, 1024
    dY = torch.zeros(T, Mo, device="cuda", dtype=torch.bfloat16)
    dY[torch.arange(T), (torch.arange(T) * 7) % Mo] = 1.0
    dY[torch.arange(T), (torch.arange(T) * 13 + 5) % Mo] += 2.0
    X = ((torch.arange(T * No, device="cuda").reshape(T, No) % 61) - 30).to(torch.bfloat16)
    dW = torch.zeros(Mo, No, device="cuda")
    db = torch.zeros(Mo, device="cuda")
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, hip.ptr(db), Mo, No, T, 1.0, 0, 0,
             hip.stream_ptr())
    assert torch.equal(dW, dY.float().t() @ X.float())
    assert torch.equal(db, dY.float().sum(0))
    # random data: 256x256 path vs forced 128x128 path vs torch
    T, Mo, No = 16384, 3072, 768
    dY, X = _rand((T, Mo), 0.5, 41), _rand((T, No), 0.5, 42)
    ref = dY.float().t() @ X.float()
    outs = []
    for sk in (0, -1, 4):
        dW = torch.zeros(Mo, No, device="cuda")
        db = torch.zeros(Mo, device="cuda")
        hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, hip.ptr(db), Mo, No, T, 1.0, sk,
                 0, hip.stream_ptr())
        torch.testing.assert_close(dW, ref, rtol=1e-4, atol=8e-3)
        torch.testing.assert_close(db, dY.float().sum(0), rtol=1e-4, atol=8e-3)
        outs.append(dW)
    # odd feature counts (decoder: 29056 = 113.5 x 256) and a device-side token count
    T, Mo, No = 4096, 29056, 512
    dY, X = _rand((T, Mo), 0.5, 43), _rand((T, No), 0.5, 44)
    k = 2432
    dY[k:2496] = 0
    X[k:2496] = 0
    k_dev = torch.tensor([k], device="cuda", dtype=torch.int32)
    dW = torch.zeros(Mo, No, device="cuda")
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, 0, Mo, No, T, 1.0, 0,
             hip.ptr(k_dev), hip.stream_ptr())
    torch.testing.assert_close(dW, dY[:k].float().t() @ X[:k].float(), rtol=1e-4, atol=4e-3)
    # the four-wave kernel range-checks tokens itself: rows past the device-side count may hold anything
    dY[k:] = 1000.0
    X[k:] = -1000.0
    dW2 = torch.zeros(Mo, No, device="cuda")
    hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW2), No, 0, Mo, No, T, 1.0, 0,
             hip.ptr(k_dev), hip.stream_ptr())
    torch.testing.assert_close(dW2, dW, rtol=1e-5, atol=1e-4)


def test_gemm_tn_256_operand_beyond_4gb(hip):
    """The four-wave weight-gradient kernel re-bases its buffer resources per K tile: tokens that start more than 4 GB into
    an operand (the entity decoder's dlogits are [16 384 x 175 104] bf16 = 5.7 GB) must contribute, with and without a
    device-side token count. Exact data: one non-zero per token row of dY, small integers in X."""
    T, Mo, No = 8448, 262144, 256          # T x Mo x 2 B = 4.43 GB; tokens >= 8192 lie beyond 2^32 bytes
    free, _ = torch.cuda.mem_get_info()
    if free < 12 * (1 << 30):
        pytest.skip("needs 12 GB of free HBM")
    dY = torch.zeros(T, Mo, device="cuda", dtype=torch.bfloat16)
    cols = (torch.arange(T, device="cuda") * 977) % Mo
    dY[torch.arange(T, device="cuda"), cols] = 1.0
    X = ((torch.arange(T * No, device="cuda").reshape(T, No) % 13) - 6).to(torch.bfloat16)
    for k in (T, 8300):                    # all tokens; a device-side count that ends beyond the 4 GB mark
        k_dev = torch.tensor([k], device="cuda", dtype=torch.int32)
        dW = torch.zeros(Mo, No, device="cuda")
        hip.call("stonk_gemm_tn_bf16", hip.ptr(dY), Mo, hip.ptr(X), No, hip.ptr(dW), No, 0, Mo, No, T, 1.0, 0,
                 hip.ptr(k_dev) if k < T else 0, hip.stream_ptr())
        ref = torch.zeros(Mo, No, device="cuda")
        ref.index_add_(0, cols[:k], X[:k].float())
        assert torch.equal(dW, ref), k
        assert float(dW[cols[8200]].abs().sum()) > 0      # a token beyond the 4 GB mark did land
    del dY, X, dW, ref
    torch.cuda.empty_cache()


WAVE4_192 = 4


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (2427, 768, 3072), (8192, 2304, 768), (300, 384, 256), (16384, 768, 2304)])
def test_four_wave_kernel_on_192_wide_tiles(hip, M, N, K):
    """STONK_GEMM_WAVE4_192: 256x192 tiles (128x96 wave tiles, 4 x 3 MFMA blocks) - what AUTO takes for the N = 768
    launches of the step. Plain product (ragged last row tile), the side-operand epilogues it is compiled for, the same
    dropout mask as the 128x128 kernel, a device-side row count, bit-identical repeats; shapes it cannot take are refused."""
    A, B = _rand((M, K), 0.5, 51), _rand((N, K), 0.05, 52)
    ref = A.float() @ B.float().t()
    out = _gemm(hip, A, B, kernel=WAVE4_192)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    for _ in range(5):
        assert torch.equal(_gemm(hip, A, B, kernel=WAVE4_192), out)
    bias = torch.randn(N, device="cuda")
    resid = _rand((M, N), 1.0, 53)
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS, bias=bias, kernel=WAVE4_192)
    torch.testing.assert_close(out.float(), ref + bias, rtol=2e-2, atol=2e-2)
    out = _gemm(hip, A, B, flags=hip.EPI_RESID, resid=resid, kernel=WAVE4_192)
    torch.testing.assert_close(out.float(), ref + resid.float(), rtol=2e-2, atol=3e-2)
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_RESID, bias=bias, resid=resid, kernel=WAVE4_192)
    torch.testing.assert_close(out.float(), ref + bias + resid.float(), rtol=2e-2, atol=3e-2)
    fl = hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT
    d192 = _gemm(hip, A, B, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=9, kernel=WAVE4_192).float() - resid.float()
    d128 = _gemm(hip, A, B, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=9, kernel=T128).float() - resid.float()
    pre = ref + bias
    kept = (d192.abs() > 1e-3) | (pre.abs() < 1e-2)
    assert abs(1.0 - kept.float().mean().item() - 0.1) < 0.02
    assert (d192.abs() > 1e-3).eq(d128.abs() > 1e-3).float().mean().item() > 0.999      # one mask per (seed, row, column)
    torch.testing.assert_close(torch.where(kept, d192, torch.zeros_like(d192)),
                               torch.where(kept, pre / 0.9, torch.zeros_like(pre)), rtol=3e-2, atol=6e-2)
    m_dev = torch.tensor([max(1, M - 37)], device="cuda", dtype=torch.int32)
    C = torch.full((M, N), -7.0, device="cuda", dtype=torch.bfloat16)
    _gemm(hip, A, B, C=C, m_dev=m_dev, kernel=WAVE4_192)
    torch.testing.assert_close(C[:M - 37].float(), ref[:M - 37], rtol=2e-2, atol=2e-2)
    assert (C[M - 37:] == -7.0).all()
    # AUTO picks a tile width itself and agrees
    auto = _gemm(hip, A, B, flags=hip.EPI_RESID, resid=resid)
    torch.testing.assert_close(auto.float(), ref + resid.float(), rtol=2e-2, atol=3e-2)
    with pytest.raises(hip.StonkHipError, match="-2"):   # no 192-wide instance for a GELU epilogue / fp32 output
        _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_GELU, bias=bias, kernel=WAVE4_192)
    with pytest.raises(hip.StonkHipError, match="-2"):
        _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32, kernel=WAVE4_192)


def test_192_wide_tiles_need_n_divisible_by_192(hip):
    A, B = _rand((512, 256), 0.5, 61), _rand((256, 256), 0.05, 62)
    with pytest.raises(hip.StonkHipError, match="-2"):
        _gemm(hip, A, B, kernel=WAVE4_192)


ASM4, ASM4_192 = 7, 8   # STONK_GEMM_ASM4 / _ASM4_192: the four-wave kernel with the written-out K loop (gemm_a4.hip)


@pytest.mark.parametrize("kernel", [ASM4, ASM4_192])
@pytest.mark.parametrize("M,N,K", [(256, 768, 128), (1000, 768, 768), (2050, 1536, 3072), (26408, 768, 768), (4096, 2304, 768),
                                   (16384, 768, 2304)])
def test_written_out_four_wave_kernel(hip, kernel, M, N, K):
    """STONK_GEMM_ASM4 / _ASM4_192 (what AUTO takes for the bf16 launches of the step since round 4): plain product with a
    ragged last row tile and several tiles per workgroup, bit-identical repeats (a misplaced wait of the hand-placed LDS-DMA
    pipeline would show as rare differing tiles), every epilogue instance against torch fp32 on the same bf16 inputs, the
    same dropout mask as the 128x128 kernel, a device-side row count; what it has no instance of is refused."""
    if kernel == ASM4_192 and N % 192:
        A, B = _rand((M, K), 0.5, 71), _rand((N, K), 0.05, 72)
        with pytest.raises(hip.StonkHipError, match="-2"):
            _gemm(hip, A, B, kernel=kernel)
        return
    A, B = _rand((M, K), 0.5, 71), _rand((N, K), 0.05, 72)
    ref = A.float() @ B.float().t()
    out = _gemm(hip, A, B, kernel=kernel)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    for _ in range(10):
        assert torch.equal(_gemm(hip, A, B, kernel=kernel), out)
    bias = torch.randn(N, device="cuda")
    resid = _rand((M, N), 1.0, 73)
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS, bias=bias, kernel=kernel)
    torch.testing.assert_close(out.float(), ref + bias, rtol=2e-2, atol=2e-2)
    out = _gemm(hip, A, B, flags=hip.EPI_RESID, resid=resid, kernel=kernel)
    torch.testing.assert_close(out.float(), ref + resid.float(), rtol=2e-2, atol=3e-2)
    out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_RESID, bias=bias, resid=resid, kernel=kernel)
    torch.testing.assert_close(out.float(), ref + bias + resid.float(), rtol=2e-2, atol=3e-2)
    fl = hip.EPI_BIAS | hip.EPI_RESID | hip.EPI_DROPOUT
    da = _gemm(hip, A, B, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=9, kernel=kernel).float() - resid.float()
    d128 = _gemm(hip, A, B, flags=fl, bias=bias, resid=resid, drop_p=0.1, seed=9, kernel=T128).float() - resid.float()
    pre = ref + bias
    kept = (da.abs() > 1e-3) | (pre.abs() < 1e-2)
    assert abs(1.0 - kept.float().mean().item() - 0.1) < 0.02
    assert (da.abs() > 1e-3).eq(d128.abs() > 1e-3).float().mean().item() > 0.999      # one mask per (seed, row, column)
    torch.testing.assert_close(torch.where(kept, da, torch.zeros_like(da)),
                               torch.where(kept, pre / 0.9, torch.zeros_like(pre)), rtol=3e-2, atol=6e-2)
    cut = max(1, M - 37)
    m_dev = torch.tensor([cut], device="cuda", dtype=torch.int32)
    C = torch.full((M, N), -7.0, device="cuda", dtype=torch.bfloat16)
    _gemm(hip, A, B, C=C, m_dev=m_dev, kernel=kernel)
    torch.testing.assert_close(C[:cut].float(), ref[:cut], rtol=2e-2, atol=2e-2)
    assert (C[cut:] == -7.0).all()
    if kernel == ASM4:   # the GELU family (256-wide tiles only)
        gelu = torch.nn.functional.gelu
        out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_GELU, bias=bias, kernel=kernel)
        torch.testing.assert_close(out.float(), gelu(pre), rtol=1e-2, atol=2e-2)
        for ag in (0, hip.EPI_AUX_GRAD):
            aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            out = _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT | ag, bias=bias, aux=aux, kernel=kernel)
            torch.testing.assert_close(out.float(), gelu(pre), rtol=1e-2, atol=2e-2)
            if ag:
                pf = pre.clone().requires_grad_(True)
                (gpre,) = torch.autograd.grad(gelu(pf).sum(), pf)
                torch.testing.assert_close(aux.float(), gpre, rtol=1e-2, atol=1e-2)
            else:
                torch.testing.assert_close(aux.float(), pre, rtol=1e-2, atol=2e-2)
        u = _rand((M, N), 1.0, 74)
        uf = u.float().requires_grad_(True)
        (gp,) = torch.autograd.grad(gelu(uf).sum(), uf)
        out = _gemm(hip, A, B, flags=hip.EPI_GELU_BWD, aux=u, kernel=kernel)
        torch.testing.assert_close(out.float(), ref * gp, rtol=1e-2, atol=3e-2)
        out = _gemm(hip, A, B, flags=hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD, aux=u, kernel=kernel)
        torch.testing.assert_close(out.float(), ref * u.float(), rtol=1e-2, atol=3e-2)
    else:
        with pytest.raises(hip.StonkHipError, match="-2"):
            _gemm(hip, A, B, flags=hip.EPI_BIAS | hip.EPI_GELU, bias=bias, kernel=kernel)
    with pytest.raises(hip.StonkHipError, match="-2"):   # fp32 output, a scaled product, an odd number of K tiles
        _gemm(hip, A, B, torch.float32, flags=hip.EPI_OUT_F32, kernel=kernel)
    with pytest.raises(hip.StonkHipError, match="-2"):
        _gemm(hip, A, B, alpha=0.5, kernel=kernel)
    if K >= 192:
        with pytest.raises(hip.StonkHipError, match="-2"):
            _gemm(hip, A[:, :K - 64], B[:, :K - 64], kernel=kernel)


@pytest.mark.parametrize("M,N,K,cut", [(1000, 768, 128, None), (4096, 29056, 768, 2427), (16384, 70016, 128, None)])
def test_written_out_kernel_fp16_logits(hip, M, N, K, cut):
    """The label-sparse decoders' forward on the written-out kernel (explicitly, and as AUTO takes it): the fp32 product
    rounded once to fp16, saturated; a device-side row count leaves the rows past it alone; an output of more than 2^31
    bytes (the entity decoder's 16 384 x 175 104 capacity) is addressed from each tile's first row."""
    A, B = _rand((M, K), 0.5, 81), _rand((N, K), 0.5, 82)
    m_dev = None if cut is None else torch.tensor([cut], device="cuda", dtype=torch.int32)
    rows = M if cut is None else cut
    for kernel in (ASM4, 0):
        C = torch.full((M, N), -7.0, device="cuda", dtype=torch.float16)
        _gemm(hip, A, B, flags=hip.EPI_OUT_F16, C=C, m_dev=m_dev, kernel=kernel)
        for r0 in range(0, rows, 4096):   # (in slabs: the fp32 reference of the largest case is 4.6 GB)
            r1 = min(rows, r0 + 4096)
            ref = (A[r0:r1].float() @ B.float().t()).half()
            torch.testing.assert_close(C[r0:r1].float(), ref.float(), rtol=1e-3, atol=1e-3)
        assert (C[rows:] == -7.0).all()
    big = torch.full((256, 128), 200.0, device="cuda", dtype=torch.bfloat16)
    Bb = big[:128].clone()
    Bb[1] = -200.0
    out = _gemm(hip, big, Bb, torch.float16, flags=hip.EPI_OUT_F16, kernel=ASM4)   # 128 * 200 * 200 > 65504
    assert torch.isfinite(out).all() and float(out[0, 0]) == 65504.0 and float(out[0, 1]) == -65504.0
    with pytest.raises(hip.StonkHipError, match="-2"):
        _gemm(hip, A, B, torch.float16, flags=hip.EPI_OUT_F16, kernel=ASM4_192)


@pytest.mark.parametrize("kernel", [ASM4, ASM4_192])
@pytest.mark.parametrize("M,N,K,sk,cut", [(768, 768, 8192, 8, None), (304, 1536, 4096, 16, None), (300, 768, 4224, 5, None),
                                          (2048, 768, 175104, 13, 1229), (2048, 768, 29056, 8, 2040)])
def test_written_out_kernel_split_k_atomic(hip, kernel, M, N, K, sk, cut):
    """The decoders' dgrad form: C (fp32) += alpha * A . B^T over a split contraction (a split's share is rounded up to an even
    number of K tiles: 66 K tiles in 5 splits = 14 + 14 + 14 + 14 + 10), device-side row count, 175 104-long contraction."""
    A, B = _rand((M, K), 0.3, 83), _rand((N, K), 0.3, 84)
    m_dev = None if cut is None else torch.tensor([cut], device="cuda", dtype=torch.int32)
    rows = M if cut is None else cut
    C = torch.ones(M, N, device="cuda")
    if kernel == ASM4:   # (256 x 192 tiles only: the 256-wide instance would need scratch)
        with pytest.raises(hip.StonkHipError, match="-2"):
            _gemm(hip, A, B, flags=hip.EPI_OUT_F32_ATOMIC, kernel=kernel, split_k=sk, C=C, m_dev=m_dev)
        return
    _gemm(hip, A, B, flags=hip.EPI_OUT_F32_ATOMIC, kernel=kernel, split_k=sk, C=C, m_dev=m_dev, alpha=0.5)
    ref = 1.0 + 0.5 * (A[:rows].float() @ B.float().t())
    torch.testing.assert_close(C[:rows], ref, rtol=1e-4, atol=2e-3 * (K / 2048) ** 0.5)
    assert (C[rows:] == 1.0).all()
    C1 = torch.zeros(M, N, device="cuda")
    _gemm(hip, A, B, flags=hip.EPI_OUT_F32_ATOMIC, kernel=kernel, split_k=1, C=C1, m_dev=m_dev)
    torch.testing.assert_close(C1[:rows], A[:rows].float() @ B.float().t(), rtol=1e-4, atol=2e-3 * (K / 2048) ** 0.5)
