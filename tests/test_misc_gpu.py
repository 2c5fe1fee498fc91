"""GPU parity for the HBM-bound helper kernels: transposes, label compaction, gather/scatter, fused softmax
cross-entropy, NSP loss, pooler/NSP heads, grad-norm + AdamW."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


@pytest.mark.parametrize("rows,cols", [(64, 64), (1000, 768), (130, 2304)])
def test_transpose_bf16_colsum(hip, rows, cols):
    x = _rand((rows, cols), 1.0, 1)
    rpad = (rows + 63) // 64 * 64
    out = torch.full((cols, rpad), 7.0, device="cuda", dtype=torch.bfloat16)
    cs = torch.zeros(cols, device="cuda")
    hip.call("stonk_transpose_bf16", hip.ptr(x), cols, hip.ptr(out), rpad, rows, cols, hip.ptr(cs), 0, hip.stream_ptr())
    assert torch.equal(out[:, :rows], x.t())
    assert (out[:, rows:] == 0).all()
    torch.testing.assert_close(cs, x.float().sum(0), rtol=1e-4, atol=1e-3)


def test_transpose_device_row_count(hip):
    rows, cols, live = 1024, 128, 300
    x = _rand((rows, cols), 1.0, 2)
    out = torch.full((cols, rows), 7.0, device="cuda", dtype=torch.bfloat16)
    cnt = torch.tensor([live], device="cuda", dtype=torch.int32)
    hip.call("stonk_transpose_bf16", hip.ptr(x), cols, hip.ptr(out), rows, rows, cols, 0, hip.ptr(cnt), hip.stream_ptr())
    assert torch.equal(out[:, :live], x[:live].t())
    assert (out[:, live:320] == 0).all()  # zero up to the 64-row round-up: what a K tile of the wgrad GEMM reads


def test_transpose_f32_and_cast(hip):
    w = _rand((1000, 256), 0.02, 3, torch.float32)
    out = torch.full((256, 1024), 7.0, device="cuda", dtype=torch.bfloat16)
    hip.call("stonk_transpose_f32_to_bf16", hip.ptr(w), hip.ptr(out), 1000, 256, 1024, hip.stream_ptr())
    assert torch.equal(out[:, :1000], w.to(torch.bfloat16).t())
    assert (out[:, 1000:] == 0).all()
    y = torch.empty(1000 * 256 - 3, device="cuda", dtype=torch.bfloat16)
    hip.call("stonk_cast_f32_to_bf16", hip.ptr(w), hip.ptr(y), y.numel(), hip.stream_ptr())
    assert torch.equal(y, w.flatten()[: y.numel()].to(torch.bfloat16))


def test_batched_transpose_refreshes_many_tensors_in_one_launch(hip):
    """stonk_transpose_bf16_batched: a device-side table of (in, out, ld_in, ld_out, rows, cols, first_tile, col_tiles)
    entries, one 64x64 tile per workgroup - the W^T copies of every weight after an optimizer step. Ragged row counts
    (the text decoder's 28 996 rows of a 29 056-row slab) are zero-filled up to the tile edge."""
    import struct

    shapes = [(768, 768), (1000, 256), (3072, 768), (128, 64), (28996 % 4096 + 4096, 128)]
    srcs, outs, entries, first = [], [], [], 0
    for i, (rows, cols) in enumerate(shapes):
        x = _rand((rows, cols), 1.0, 40 + i)
        rpad = (rows + 63) // 64 * 64
        out = torch.full((cols, rpad), 7.0, device="cuda", dtype=torch.bfloat16)
        col_tiles = (cols + 63) // 64
        entries.append(struct.pack("<QQqqqiiii", hip.ptr(x), hip.ptr(out), cols, rpad, rows, cols, first, col_tiles, 0))
        first += ((rows + 63) // 64) * col_tiles
        srcs.append(x)
        outs.append(out)
    table = torch.frombuffer(bytearray(b"".join(entries)), dtype=torch.uint8).cuda()
    hip.call("stonk_transpose_bf16_batched", hip.ptr(table), len(shapes), first, hip.stream_ptr())
    torch.cuda.synchronize()
    for x, out, (rows, cols) in zip(srcs, outs, shapes):
        assert torch.equal(out[:, :rows], x.t())
        assert (out[:, rows:] == 0).all()


@pytest.mark.parametrize("B,half,frac", [(1, 7, 0.5), (3, 333, 0.15), (64, 256, 0.15), (150, 256, 0.15), (64, 256, 0.0), (17, 256, 1.0)])
def test_label_compact_any_size_with_a_row_map(hip, B, half, frac):
    """stonk_label_compact in chunks of 16 384 labels: fewer than one round of 1024, exactly one chunk, several chunks with a
    ragged last one, nothing labelled, everything labelled - order, targets, count; with the packed layout's row map."""
    S = 2 * half
    g = torch.Generator().manual_seed(B * 1000 + half)
    labels = torch.full((B, half), -100, dtype=torch.long)
    pick = torch.rand(B, half, generator=g) < frac
    labels[pick] = torch.randint(0, 5000, (int(pick.sum()),), generator=g)
    labels = labels.cuda()
    n = B * half
    row_of_pos = torch.randperm(B * S, generator=g).to(torch.int32).cuda()
    for rmap in (None, row_of_pos):
        rows = torch.full((n,), -1, device="cuda", dtype=torch.int32)
        tg = torch.full((n,), -1, device="cuda", dtype=torch.int32)
        cnt = torch.full((1,), -5, device="cuda", dtype=torch.int32)
        hip.call("stonk_label_compact", hip.ptr(labels), n, half, S, half, hip.ptr(rows), hip.ptr(tg), hip.ptr(cnt),
                 hip.ptr(rmap), hip.stream_ptr())
        idx = (labels.view(-1) != -100).nonzero().squeeze(1)
        c = cnt.item()
        assert c == idx.numel()
        pos = (idx // half) * S + half + idx % half
        exp_rows = pos if rmap is None else rmap[pos].long()
        assert torch.equal(rows[:c].long(), exp_rows) and torch.equal(tg[:c].long(), labels.view(-1)[idx])
        assert (rows[c:] == -1).all() and (tg[c:] == -1).all()


def test_label_compact_gather_scatter(hip):
    B, half, S, H = 5, 256, 512, 64
    g = torch.Generator().manual_seed(0)
    labels = torch.full((B, half), -100, dtype=torch.long)
    pick = torch.rand(B, half, generator=g) < 0.15
    labels[pick] = torch.randint(0, 1000, (int(pick.sum()),), generator=g)
    labels = labels.cuda()
    n = B * half
    rows = torch.full((n,), -1, device="cuda", dtype=torch.int32)
    tg = torch.full((n,), -1, device="cuda", dtype=torch.int32)
    cnt = torch.zeros(1, device="cuda", dtype=torch.int32)
    hip.call("stonk_label_compact", hip.ptr(labels), n, half, S, half, hip.ptr(rows), hip.ptr(tg), hip.ptr(cnt), 0,
             hip.stream_ptr())
    idx = (labels.view(-1) != -100).nonzero().squeeze(1)
    c = cnt.item()
    assert c == idx.numel()
    exp_rows = (idx // half) * S + half + idx % half
    assert torch.equal(rows[:c].long(), exp_rows)
    assert torch.equal(tg[:c].long(), labels.view(-1)[idx])
    # gather / scatter
    src = _rand((B * S, H), 1.0, 4)
    cap = n
    dst = torch.full((cap, H), 9.0, device="cuda", dtype=torch.bfloat16)
    hip.call("stonk_gather_rows_bf16", hip.ptr(src), H, hip.ptr(rows), hip.ptr(cnt), hip.ptr(dst), H, H, cap,
             hip.stream_ptr())
    assert torch.equal(dst[:c], src[exp_rows])
    lim = (c + 127) // 128 * 128
    assert (dst[c:lim] == 0).all()
    back = torch.zeros_like(src)
    hip.call("stonk_scatter_rows_bf16", hip.ptr(dst), H, hip.ptr(rows), hip.ptr(cnt), hip.ptr(back), H, H,
             hip.stream_ptr())
    ref = torch.zeros_like(src)
    ref[exp_rows] = src[exp_rows]
    assert torch.equal(back, ref)


@pytest.mark.parametrize("N", [1000, 28996])
def test_softmax_xent(hip, N):
    R, cap = 37, 128
    npad = (N + 127) // 128 * 128
    logits = torch.zeros(cap, npad, device="cuda")
    logits[:, :N] = _rand((cap, N), 3.0, 5, torch.float32)
    tg = torch.randint(0, N, (cap,), device="cuda", dtype=torch.int32)
    cnt = torch.tensor([R], device="cuda", dtype=torch.int32)
    loss = torch.zeros(1, device="cuda")
    dl = torch.full((cap, npad), 5.0, device="cuda", dtype=torch.bfloat16)
    err = torch.zeros(1, device="cuda", dtype=torch.int32)
    hip.call("stonk_softmax_xent_fwd_bwd", hip.ptr(logits), npad, N, npad, hip.ptr(tg), hip.ptr(cnt), hip.ptr(loss),
             hip.ptr(dl), npad, 1.0, cap, hip.ptr(err), hip.stream_ptr())
    assert (dl[R:64] == 0).all() and (dl[64:] == 5.0).all()  # pad rows up to the 64-row K step are zeroed, no more
    x = logits[:R, :N].clone().requires_grad_(True)
    ref = F.cross_entropy(x, tg[:R].long())
    ref.backward()
    torch.testing.assert_close(loss[0] / R, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dl[:R, :N].float(), x.grad, rtol=1e-2, atol=1e-5)
    assert (dl[:R, N:] == 0).all()
    assert err.item() == 0
    # the fp16-logit entry point: same numbers as the fp32 one fed the rounded logits
    lh = logits.half()
    loss_h = torch.zeros(1, device="cuda")
    dl_h = torch.full((cap, npad), 5.0, device="cuda", dtype=torch.bfloat16)
    hip.call("stonk_softmax_xent_f16_fwd_bwd", hip.ptr(lh), npad, N, npad, hip.ptr(tg), hip.ptr(cnt), hip.ptr(loss_h),
             hip.ptr(dl_h), npad, 1.0, cap, hip.ptr(err), hip.stream_ptr())
    xh = lh[:R, :N].float().requires_grad_(True)
    ref_h = F.cross_entropy(xh, tg[:R].long())
    ref_h.backward()
    torch.testing.assert_close(loss_h[0] / R, ref_h, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dl_h[:R, :N].float(), xh.grad, rtol=1e-2, atol=1e-5)
    assert (dl_h[R:64] == 0).all() and (dl_h[64:] == 5.0).all() and (dl_h[:R, N:] == 0).all()
    assert abs(float(loss_h[0] - loss[0])) / R < 2e-3   # and close to the fp32-logit loss


def test_nsp_and_finalize(hip):
    B = 64
    logits = _rand((B, 2), 1.0, 6, torch.float32)
    labels = torch.randint(0, 2, (B,), device="cuda")
    acc = torch.zeros(2, device="cuda")
    dl = torch.empty(B, 2, device="cuda")
    err = torch.zeros(1, device="cuda", dtype=torch.int32)
    hip.call("stonk_nsp_xent_fwd_bwd", hip.ptr(logits), hip.ptr(labels), B, 2, hip.ptr(acc), hip.ptr(dl), 1.0,
             hip.ptr(err), hip.stream_ptr())
    x = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(x, labels)
    ref.backward()
    torch.testing.assert_close(acc[0] / acc[1], ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dl, x.grad, rtol=1e-4, atol=1e-6)
    ts = torch.tensor([10.0], device="cuda")
    tc = torch.tensor([4], device="cuda", dtype=torch.int32)
    es = torch.tensor([9.0], device="cuda")
    ec = torch.tensor([3], device="cuda", dtype=torch.int32)
    out = torch.zeros(4, device="cuda")
    hip.call("stonk_loss_finalize", hip.ptr(ts), hip.ptr(tc), hip.ptr(es), hip.ptr(ec), hip.ptr(acc), hip.ptr(out),
             hip.stream_ptr())
    torch.testing.assert_close(out, torch.stack([2.5 + 3.0 + ref.detach(), torch.tensor(2.5, device="cuda"),
                                                 torch.tensor(3.0, device="cuda"), ref.detach()]))


def test_small_linear_fwd_bwd(hip):
    B, S, H = 6, 4, 128
    seq = _rand((B * S, H), 1.0, 7)
    W = _rand((H, H), 0.05, 8, torch.float32)
    b = _rand((H,), 0.05, 9, torch.float32)
    y = torch.empty(B, H, device="cuda")
    hip.call("stonk_small_linear_fwd", hip.ptr(seq), S * H, hip.ptr(W), hip.ptr(b), hip.ptr(y), B, H, H, hip.SMALL_TANH,
             hip.stream_ptr())
    x0 = seq.view(B, S, H)[:, 0].float().requires_grad_(True)
    Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.tanh(x0 @ Wr.t() + br)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)
    dy = _rand((B, H), 1.0, 10, torch.float32)
    ref.backward(dy)
    dW = torch.zeros_like(W)
    db = torch.zeros_like(b)
    dseq = _rand((B * S, H), 1.0, 11)
    dseq0 = dseq.clone()
    hip.call("stonk_small_linear_bwd", hip.ptr(dy), hip.ptr(y), hip.ptr(seq), S * H, hip.ptr(W), hip.ptr(dW), hip.ptr(db),
             0, hip.ptr(dseq), S * H, B, H, H, hip.SMALL_TANH, hip.stream_ptr())
    torch.testing.assert_close(dW, Wr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db, br.grad, rtol=1e-4, atol=1e-5)
    got = dseq.view(B, S, H)[:, 0].float() - dseq0.view(B, S, H)[:, 0].float()
    torch.testing.assert_close(got, x0.grad, rtol=5e-2, atol=3e-2)  # bf16 read-modify-write
    assert torch.equal(dseq.view(B, S, H)[:, 1:], dseq0.view(B, S, H)[:, 1:])
    # fp32-input classifier without activation
    W2 = _rand((2, H), 0.05, 12, torch.float32)
    b2 = _rand((2,), 0.05, 13, torch.float32)
    y2 = torch.empty(B, 2, device="cuda")
    hip.call("stonk_small_linear_fwd", hip.ptr(y), H, hip.ptr(W2), hip.ptr(b2), hip.ptr(y2), B, 2, H, hip.SMALL_X_F32,
             hip.stream_ptr())
    torch.testing.assert_close(y2, y @ W2.t() + b2, rtol=1e-4, atol=1e-5)
    dx = torch.empty(B, H, device="cuda")
    dW2 = torch.zeros_like(W2)
    db2 = torch.zeros_like(b2)
    dy2 = _rand((B, 2), 1.0, 14, torch.float32)
    hip.call("stonk_small_linear_bwd", hip.ptr(dy2), 0, hip.ptr(y), H, hip.ptr(W2), hip.ptr(dW2), hip.ptr(db2),
             hip.ptr(dx), 0, 0, B, 2, H, hip.SMALL_X_F32, hip.stream_ptr())
    torch.testing.assert_close(dx, dy2 @ W2, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dW2, dy2.t() @ y, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db2, dy2.sum(0), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("clip", [False, True])
def test_adamw_matches_torch(hip, clip):
    n = 4096 * 3 + 8
    p0 = _rand((n,), 0.05, 15, torch.float32)
    scale = 30.0 if clip else 0.01
    grads = [_rand((n,), scale, 16 + i, torch.float32) for i in range(3)]
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for step, g in enumerate(grads, 1):
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        gg = g.clone()
        nrm = torch.zeros(1, device="cuda")
        ws = torch.zeros(int(hip.lib().stonk_sumsq_workspace_floats()), device="cuda")
        hip.call("stonk_sumsq_f32", hip.ptr(gg), n, hip.ptr(nrm), hip.ptr(ws), ws.numel(), hip.stream_ptr())
        torch.testing.assert_close(nrm[0], (g.double() ** 2).sum().float(), rtol=1e-5, atol=0)
        hip.call("stonk_adamw_step", hip.ptr(p), hip.ptr(gg), hip.ptr(m), hip.ptr(v), hip.ptr(pb), n, 1e-3, 0.9, 0.999,
                 1e-8, 0.01, 1 - 0.9 ** step, 1 - 0.999 ** step, hip.ptr(nrm), 1.0, 1.0, 0, 0, 0, hip.stream_ptr())
        assert (gg == 0).all()  # zero_grad fused
        torch.testing.assert_close(p, ref_p.data, rtol=2e-5, atol=2e-7)
        assert torch.equal(pb, p.to(torch.bfloat16))


def test_sumsq_is_bitwise_repeatable(hip):
    """The global grad-norm feeds the clip coefficient of every parameter update: data-parallel ranks must get the same
    bits from the same summed gradients, so the reduction order is fixed (no per-block atomics). 20 launches over a
    buffer large enough for the full 1024-block grid, plus a ragged tail."""
    n = 50_000_003
    g = _rand((n,), 1.0, 77, torch.float32)
    out = torch.zeros(20, device="cuda")
    ws = torch.zeros(int(hip.lib().stonk_sumsq_workspace_floats()), device="cuda")
    for i in range(20):
        hip.call("stonk_sumsq_f32", hip.ptr(g), n, out[i:].data_ptr(), hip.ptr(ws), ws.numel(), hip.stream_ptr())
    torch.cuda.synchronize()
    assert (out == out[0]).all(), out.tolist()
    torch.testing.assert_close(out[0], (g.double() ** 2).sum().float(), rtol=1e-5, atol=0)
    assert int(ws[-1].view(torch.int32).item()) == 0          # the ticket word is left ready for the next launch
    # two launches in flight on two streams, each with its own workspace (the state used to be library globals, where
    # they would have shared slots and ticket): both get the bits of the serial launches
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ws2 = torch.zeros_like(ws)
    g2 = g * 0.5
    both = torch.zeros(2, 8, device="cuda")
    torch.cuda.synchronize()
    for i in range(8):
        with torch.cuda.stream(s1):
            hip.call("stonk_sumsq_f32", hip.ptr(g), n, both[0, i:].data_ptr(), hip.ptr(ws), ws.numel(), hip.stream_ptr())
        with torch.cuda.stream(s2):
            hip.call("stonk_sumsq_f32", hip.ptr(g2), n, both[1, i:].data_ptr(), hip.ptr(ws2), ws2.numel(), hip.stream_ptr())
    torch.cuda.synchronize()
    assert (both[0] == out[0]).all() and (both[1] == both[1, 0]).all()
    torch.testing.assert_close(both[1, 0], (g2.double() ** 2).sum().float(), rtol=1e-5, atol=0)
    assert hip.lib().stonk_sumsq_f32(hip.ptr(g), n, out.data_ptr(), hip.ptr(ws), 8, hip.stream_ptr()) == -1   # short workspace
