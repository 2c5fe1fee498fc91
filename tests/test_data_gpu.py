"""GPU parity of the on-device batch assembly + dynamic masking: equality with the numpy
restatement (integer work: bit-exact), and the reference's distributional semantics (ref:indra_for_pretraining.py:33-77:
exactly int(256 * 0.15) labels per padded half, 80 / 10 / 10 split, uniform positions; :80-126 negatives)."""
import numpy as np
import pytest
import torch

from oracle import masking_oracle as mo

pytestmark = pytest.mark.gpu


def _inputs(B, half, V, K, n_nodes, seed):
    rng = np.random.RandomState(seed)
    n = rng.randint(8, half + 1, B)
    text = np.zeros((B, half), dtype=np.int64)
    att = np.zeros((B, half), dtype=np.int64)
    for b in range(B):
        text[b, : n[b]] = rng.randint(min(1000, V // 2), V, n[b])
        text[b, 0], text[b, n[b] - 1] = 101, 102
        att[b, : n[b]] = 1
    walks = rng.randint(0, K, (n_nodes, half // 2 - 1)).astype(np.int64)
    src, tgt = rng.randint(0, n_nodes, B).astype(np.int64), rng.randint(0, n_nodes, B).astype(np.int64)
    return text, att, src, tgt, walks


@pytest.mark.parametrize("B,half,V,K", [(1, 256, 28996, 175094), (5, 256, 28996, 175094), (64, 256, 28996, 175094),
                                        (3, 128, 300, 50)])
def test_device_batcher_equals_numpy_restatement(hip, B, half, V, K):
    from stonkgs_amd.data import DeviceBatcher

    text, att, src, tgt, walks = _inputs(B, half, V, K, 500, seed=B)
    bat = DeviceBatcher(torch.from_numpy(walks), V, K, half=half, seed=11)
    for step in (0, 7):
        got = {k: v.cpu().numpy() for k, v in bat(text, att, src, tgt, step).items()}
        seed = bat.step_seed(step)
        raw, a, t, nsp = mo.assemble_rows(text, att, src, tgt, walks, negative_rate=0.2, seed=seed)
        ids, tl, el = mo.mlm_mask(raw, half, V, K, seed=seed ^ 0x5BD1E995)
        assert np.array_equal(got["input_ids"], ids)
        assert np.array_equal(got["masked_lm_labels"], tl) and np.array_equal(got["ent_masked_lm_labels"], el)
        assert np.array_equal(got["attention_mask"], a) and np.array_equal(got["token_type_ids"], t)
        assert np.array_equal(got["next_sentence_labels"], nsp)
    bat.check_errors()
    # another step = other masks; same step = same batch
    a0, a1, a0b = bat(text, att, src, tgt, 0), bat(text, att, src, tgt, 1), bat(text, att, src, tgt, 0)
    assert all(torch.equal(a0[k], a0b[k]) for k in a0)
    assert not torch.equal(a0["masked_lm_labels"], a1["masked_lm_labels"])


def test_device_masking_keeps_the_reference_semantics(hip):
    from stonkgs_amd.data import MASK_ID, DeviceBatcher

    B, half, V, K = 512, 256, 28996, 175094
    text, att, src, tgt, walks = _inputs(B, half, V, K, 2000, seed=99)
    bat = DeviceBatcher(torch.from_numpy(walks), V, K, seed=3)
    out = {k: v.cpu() for k, v in bat(text, att, src, tgt, 5).items()}
    raw, _, _, nsp = mo.assemble_rows(text, att, src, tgt, walks, seed=bat.step_seed(5))
    raw = torch.from_numpy(raw)
    for name, off, vocab in (("masked_lm_labels", 0, V), ("ent_masked_lm_labels", half, K)):
        lab = out[name]
        sel = lab != -100
        assert (sel.sum(1) == int(half * 0.15)).all()                    # exactly 38 per half, padding included
        orig = raw[:, off:off + half]
        new = out["input_ids"][:, off:off + half]
        assert torch.equal(lab[sel], orig[sel]) and torch.equal(new[~sel], orig[~sel])
        n = int(sel.sum())
        masked = (new[sel] == MASK_ID).float().mean().item()
        kept = ((new[sel] == orig[sel]) & (orig[sel] != MASK_ID)).float().mean().item()
        assert abs(masked - 0.8) < 4 * (0.16 / n) ** 0.5 + 1e-3          # 80 % -> [MASK] (+ the odd random hit of 103)
        assert abs(kept - 0.1) < 4 * (0.09 / n) ** 0.5 + 2e-3            # 10 % unchanged
        rnd = new[sel][(new[sel] != MASK_ID) & (new[sel] != orig[sel])]
        assert rnd.min() >= 0 and rnd.max() < vocab and rnd.float().mean().item() == pytest.approx(vocab / 2, rel=0.08)
        # positions uniform over the padded half: each is chosen with probability 38 / 256
        per_pos = sel.float().mean(0)
        assert (per_pos - 38 / 256).abs().max() < 5 * (38 / 256 * (1 - 38 / 256) / B) ** 0.5
    # negatives: about 1 row in 5, entity half = the partner's walks, NSP label 1
    assert abs(out["next_sentence_labels"].float().mean().item() - 0.2) < 0.08
    assert torch.equal(out["next_sentence_labels"], torch.from_numpy(nsp))
    assert (out["attention_mask"][:, half:] == 1).all() and (out["token_type_ids"][:, half:] == 1).all()
    assert torch.equal(out["attention_mask"][:, :half], torch.from_numpy(att))
    bat.check_errors()


def test_device_batcher_feeds_a_training_step_and_reports_bad_nodes(hip):
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import DeviceBatcher
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg = STonKGsConfig(vocab_size=2000, kg_vocab_size=600, num_hidden_layers=2, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    text, att, src, tgt, walks = _inputs(4, 256, cfg.vocab_size, cfg.kg_vocab_size, 50, seed=5)
    bat = DeviceBatcher(torch.from_numpy(walks), cfg.vocab_size, cfg.kg_vocab_size, seed=1)
    model = STonKGsForPreTraining(cfg, kg_embeddings=torch.randn(cfg.kg_vocab_size, cfg.hidden_size, dtype=torch.float64) * 0.3)
    tr = Trainer(model, TrainingArguments(max_steps=10, per_device_train_batch_size=4))
    losses = [float(tr.training_step(model, bat(text, att, src, tgt, s))) for s in range(2)]
    assert all(np.isfinite(losses)) and losses[0] != losses[1]
    model.engine.check_errors()
    bad = src.copy()
    bad[2] = 50                                                          # no such node
    bat(text, att, bad, tgt, 0)
    with pytest.raises(KeyError):
        bat.check_errors()
    with pytest.raises(ValueError):
        bat(text[:, :100], att[:, :100], src, tgt, 0)


@pytest.mark.parametrize("B,S", [(64, 512), (5, 256), (1, 512), (300, 128)])
def test_unpad_plan_equals_numpy_restatement(hip, B, S):
    """Row plan of the unpadded encoder (csrc/unpad.hip) against oracle/masking_oracle.unpad_plan - integer work, equality:
    prefix masks of every length, holes inside the text half, labelled padding positions, a sequence without any live key
    (keeps everything), one that is all live, masked positions inside the ENTITY half with and without labels."""
    half = S // 2
    rng = np.random.RandomState(B + S)
    am = np.ones((B, S), dtype=np.int64)
    n_text = rng.randint(1, half + 1, B)
    for b in range(B):
        am[b, n_text[b]:half] = 0
    if B > 2:
        am[1, :] = 0                       # no live key at all
        am[2, 3:9] = 0                     # a hole in the text half
        am[2, half + 5:half + 40] = 0      # masked entity positions
    tl = np.full((B, half), -100, dtype=np.int64)
    el = np.full((B, half), -100, dtype=np.int64)
    for lab in (tl, el):
        pick = rng.rand(B, half) < 0.15
        lab[pick] = rng.randint(0, 1000, int(pick.sum()))
    dev = lambda a: torch.from_numpy(a).cuda()
    d_am, d_tl, d_el = dev(am), dev(tl), dev(el)
    for use_labels in (True, False):
        rop = torch.full((B * S,), -7, dtype=torch.int32, device="cuda")
        por = torch.full((B * S,), -7, dtype=torch.int32, device="cuda")
        cu = torch.full((B + 1,), -7, dtype=torch.int32, device="cuda")
        rm = torch.full((B * S,), -7, dtype=torch.int64, device="cuda")
        ws = torch.zeros(int(hip.lib().stonk_unpad_workspace_ints(B)), dtype=torch.int32, device="cuda")
        rr = torch.full((B * S,), -7, dtype=torch.int32, device="cuda")
        rofp = torch.full((B * S,), -7, dtype=torch.int32, device="cuda")
        ro = torch.full((B + 1,), -7, dtype=torch.int32, device="cuda")
        hip.call("stonk_unpad_plan", hip.ptr(d_am), hip.ptr(d_tl) if use_labels else 0, hip.ptr(d_el) if use_labels else 0,
                 B, S, half, hip.ptr(rop), hip.ptr(por), hip.ptr(cu), hip.ptr(rm), hip.ptr(rr), hip.ptr(rofp), hip.ptr(ro),
                 hip.ptr(ws), ws.numel(), hip.stream_ptr())
        e_rop, e_por, e_cu, e_rm, e_rr, e_rofp, e_ro = mo.unpad_plan(am, tl if use_labels else None,
                                                                     el if use_labels else None, read=True)
        assert np.array_equal(cu.cpu().numpy(), e_cu)
        assert np.array_equal(rop.cpu().numpy(), e_rop) and np.array_equal(por.cpu().numpy(), e_por)
        assert np.array_equal(rm.cpu().numpy(), e_rm)
        assert np.array_equal(ro.cpu().numpy(), e_ro) and np.array_equal(rr.cpu().numpy(), e_rr)
        assert np.array_equal(rofp.cpu().numpy(), e_rofp)
        n_rd = int(e_ro[-1])
        assert (e_por[e_rr[e_ro[:-1]]] % S == 0).all() and n_rd >= B      # every sequence's first read row is its position 0
        # the three read outputs are optional as a group
        rop2 = torch.empty_like(rop)
        hip.call("stonk_unpad_plan", hip.ptr(d_am), hip.ptr(d_tl) if use_labels else 0, hip.ptr(d_el) if use_labels else 0,
                 B, S, half, hip.ptr(rop2), hip.ptr(por), hip.ptr(cu), hip.ptr(rm), 0, 0, 0, hip.ptr(ws), ws.numel(),
                 hip.stream_ptr())
        assert torch.equal(rop2, rop)
        assert hip.lib().stonk_unpad_plan(hip.ptr(d_am), 0, 0, B, S, half, hip.ptr(rop2), hip.ptr(por), hip.ptr(cu),
                                          hip.ptr(rm), hip.ptr(rr), 0, 0, hip.ptr(ws), ws.numel(), hip.stream_ptr()) == -1
        total = int(e_cu[-1])
        assert 0 < total <= B * S and (use_labels is False or total >= int((am != 0).sum()))
