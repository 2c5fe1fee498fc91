"""GPU parity: LayerNorm forward/backward and the embedding front-ends against torch fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class _LazyWs:
    """Partial-sum workspace for stonk_layernorm_bwd (1024 workgroups x 2H floats), created on first use."""

    def __init__(self):
        self.t = None

    def get(self):
        if self.t is None:
            self.t = torch.empty(1024 * 2 * 1024, device="cuda")
        return self.t

    def data_ptr(self):
        return self.get().data_ptr()

    def numel(self):
        return self.get().numel()


LN_WS = _LazyWs()


def _rand(shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


@pytest.mark.parametrize("rows,H", [(5, 64), (1000, 768), (333, 1024), (4099, 512), (70, 256), (9000, 768)])
def test_layernorm_fwd_bwd(hip, rows, H):
    x = _rand((rows, H), 2.0, 1)
    dy = _rand((rows, H), 1.0, 2)
    gamma = torch.randn(H, device="cuda") * 0.5 + 1.0
    beta = torch.randn(H, device="cuda") * 0.1
    y = torch.empty_like(x)
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    hip.call("stonk_layernorm_fwd", hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd),
             rows, H, 1e-12, 0, 0.0, 0, hip.stream_ptr())
    xf = x.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (H,), gf, bf, 1e-12)
    torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean, x.float().mean(-1), rtol=1e-4, atol=1e-4)
    ref.backward(dy.float())
    dx = torch.empty_like(x)
    dgamma = torch.zeros(H, device="cuda")
    dbeta = torch.zeros(H, device="cuda")
    hip.call("stonk_layernorm_bwd", hip.ptr(dy), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dx),
             0, hip.ptr(dgamma), hip.ptr(dbeta), rows, H, 0, 0.0, 0, 0.0, 0, hip.ptr(LN_WS), LN_WS.numel(), hip.stream_ptr())
    torch.testing.assert_close(dx.float(), xf.grad, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dgamma, gf.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(dbeta, bf.grad, rtol=1e-3, atol=1e-2)
    # STONK_LN_DEFER_REDUCE: the partial sums stay in the workspace until stonk_layernorm_bwd_reduce adds them - on another
    # stream, ordered by an event, as the training step does; same dx, the same sums (the partials are added in the same order)
    dx2 = torch.empty_like(x)
    dg2 = torch.full((H,), 3.0, device="cuda")
    db2 = torch.full((H,), -2.0, device="cuda")
    ws2 = torch.empty(int(hip.lib().stonk_layernorm_bwd_workspace_floats(rows, H)), device="cuda")
    hip.call("stonk_layernorm_bwd", hip.ptr(dy), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dx2),
             0, hip.ptr(dg2), hip.ptr(db2), rows, H, hip.LN_DEFER_REDUCE, 0.0, 0, 0.0, 0, hip.ptr(ws2), ws2.numel(),
             hip.stream_ptr())
    torch.cuda.synchronize()
    assert (dg2 == 3.0).all() and (db2 == -2.0).all() and torch.equal(dx2, dx)     # nothing added yet
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        hip.call("stonk_layernorm_bwd_reduce", hip.ptr(ws2), rows, H, hip.ptr(dg2), hip.ptr(db2), hip.stream_ptr())
    torch.cuda.synchronize()
    torch.testing.assert_close(dg2 - 3.0, dgamma, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(db2 + 2.0, dbeta, rtol=1e-5, atol=1e-4)
    with pytest.raises(hip.StonkHipError, match="-1"):   # deferring needs the workspace
        hip.call("stonk_layernorm_bwd", hip.ptr(dy), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dx2),
                 0, hip.ptr(dg2), hip.ptr(db2), rows, H, hip.LN_DEFER_REDUCE, 0.0, 0, 0.0, 0, 0, 0, hip.stream_ptr())


@pytest.mark.parametrize("rows,H", [(512, 768), (300, 1024), (129, 512), (200, 640)])
def test_layernorm_dropout_consistency(hip, rows, H):
    """H = 512 / 768 / 1024 take the lane-owned-column backward kernel, 640 the generic one: same masks, same maths."""
    x = _rand((rows, H), 1.0, 3)
    gamma = torch.ones(H, device="cuda")
    beta = torch.zeros(H, device="cuda")
    y0 = torch.empty_like(x)
    y1 = torch.empty_like(x)
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    hip.call("stonk_layernorm_fwd", hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y0), hip.ptr(mean), hip.ptr(rstd),
             rows, H, 1e-12, 0, 0.0, 0, hip.stream_ptr())
    hip.call("stonk_layernorm_fwd", hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y1), hip.ptr(mean), hip.ptr(rstd),
             rows, H, 1e-12, hip.LN_DROPOUT, 0.1, 77, hip.stream_ptr())
    kept = y1 != 0
    assert abs((1 - kept.float().mean().item()) - 0.1) < 0.01
    torch.testing.assert_close(y1[kept].float(), (y0.float() / 0.9)[kept], rtol=1e-2, atol=1e-2)
    # backward with the same (seed) masks dy identically, and dx_drop masks dx with the out-seed
    dy = _rand((rows, H), 1.0, 4)
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x)
    dg = torch.zeros(H, device="cuda")
    db = torch.zeros(H, device="cuda")
    hip.call("stonk_layernorm_bwd", hip.ptr(dy), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dx),
             hip.ptr(dxd), hip.ptr(dg), hip.ptr(db), rows, H, hip.LN_DROPOUT, 0.1, 77, 0.1, 78, hip.ptr(LN_WS), LN_WS.numel(), hip.stream_ptr())
    dy_m = torch.where(kept, dy.float() / 0.9, torch.zeros((), device="cuda"))
    xf = x.float().requires_grad_(True)
    F.layer_norm(xf, (H,), gamma, beta, 1e-12).backward(dy_m)
    torch.testing.assert_close(dx.float(), xf.grad, rtol=2e-2, atol=2e-2)
    kept2 = dxd != 0
    assert abs((1 - kept2.float().mean().item()) - 0.1) < 0.01
    torch.testing.assert_close(dxd[kept2].float(), (dx.float() / 0.9)[kept2], rtol=1e-2, atol=1e-2)


def test_joint_embed_ln(hip):
    B, S, half, H, KG = 3, 16, 8, 64, 50
    ids = torch.randint(0, KG + 3, (B, S), device="cuda")
    tt = torch.cat([torch.zeros(B, half), torch.ones(B, S - half)], 1).long().cuda()
    text_h = _rand((B * half, H), 1.0, 5)
    table = _rand((KG + 3, H), 0.3, 6, torch.float32)
    pos = _rand((S, H), 0.02, 7, torch.float32)
    typ = _rand((2, H), 0.02, 8, torch.float32)
    gamma = torch.randn(H, device="cuda") * 0.1 + 1
    beta = torch.randn(H, device="cuda") * 0.1
    ssum = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(ssum)
    mean = torch.empty(B * S, device="cuda")
    rstd = torch.empty(B * S, device="cuda")
    err = torch.zeros(1, device="cuda", dtype=torch.int32)
    hip.call("stonk_joint_embed_ln_fwd", hip.ptr(ids), hip.ptr(tt), hip.ptr(text_h), hip.ptr(table), hip.ptr(pos),
             hip.ptr(typ), hip.ptr(gamma), hip.ptr(beta), hip.ptr(ssum), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd),
             B, S, half, H, KG + 3, 2, 1e-12, 0, 0.0, 0, hip.ptr(err), 0, 0, hip.stream_ptr())
    emb = torch.cat([text_h.float().view(B, half, H), table[ids[:, half:]]], 1) + pos[None] + typ[tt]
    torch.testing.assert_close(ssum.float().view(B, S, H), emb, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(y.float().view(B, S, H), F.layer_norm(emb, (H,), gamma, beta, 1e-12), rtol=1e-2, atol=1e-2)
    assert err.item() == 0
    ids[1, half + 2] = KG + 3  # out-of-table entity id: the reference raises KeyError; the kernel raises the flag
    hip.call("stonk_joint_embed_ln_fwd", hip.ptr(ids), hip.ptr(tt), hip.ptr(text_h), hip.ptr(table), hip.ptr(pos),
             hip.ptr(typ), hip.ptr(gamma), hip.ptr(beta), hip.ptr(ssum), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd),
             B, S, half, H, KG + 3, 2, 1e-12, 0, 0.0, 0, hip.ptr(err), 0, 0, hip.stream_ptr())
    assert err.item() & 1


def test_text_embed_ln_and_embed_grad(hip):
    B, S, H, V = 4, 8, 64, 300
    ids = torch.randint(0, V, (B, 2 * S), device="cuda")
    word = _rand((V, H), 0.02, 9, torch.float32)
    pos = _rand((2 * S, H), 0.02, 10, torch.float32)
    typ = _rand((2, H), 0.02, 11, torch.float32)
    gamma = torch.ones(H, device="cuda")
    beta = torch.zeros(H, device="cuda")
    y = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
    err = torch.zeros(1, device="cuda", dtype=torch.int32)
    hip.call("stonk_text_embed_ln_fwd", hip.ptr(ids), ids.stride(0), hip.ptr(word), hip.ptr(pos), hip.ptr(typ),
             hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), B, S, H, V, 1e-12, 0, 0.0, 0, hip.ptr(err), hip.stream_ptr())
    emb = word[ids[:, :S]] + pos[None, :S] + typ[0]
    torch.testing.assert_close(y.float().view(B, S, H), F.layer_norm(emb, (H,), gamma, beta, 1e-12), rtol=1e-2, atol=1e-2)
    # embedding grads
    dx = _rand((B * S, H), 1.0, 12)
    tt = torch.randint(0, 2, (B, S), device="cuda")
    dpos = torch.zeros(S, H, device="cuda")
    dtyp = torch.zeros(2, H, device="cuda")
    hip.call("stonk_embed_grad", hip.ptr(dx), hip.ptr(tt), hip.ptr(dpos), hip.ptr(dtyp), B, S, H, 2, 0, hip.stream_ptr())
    d = dx.float().view(B, S, H)
    torch.testing.assert_close(dpos, d.sum(0), rtol=1e-4, atol=1e-4)
    for t in (0, 1):
        torch.testing.assert_close(dtyp[t], (d * (tt == t)[..., None]).sum((0, 1)), rtol=1e-4, atol=1e-4)
