"""The unpadded trainable encoder (Engine.unpad; csrc/unpad.hip + the packed layout of the attention / embedding
kernels) against the padded one and against the oracle: dropping the rows that are neither live keys, nor labelled, nor
position 0 changes no loss term and no gradient of ref:src/stonkgs/models/stonkgs_model.py:204-245 - the reference computes
those rows (hf:models/bert/modeling_bert.py:164-203 over all 512 positions) and never reads them."""
import numpy as np
import pytest
import torch

from oracle import stonkgs_oracle as orc
from tests.golden_util import load_case

pytestmark = pytest.mark.gpu


def _model(cfg, sd, tsv_rows, cls=None, **kw):
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    c = STonKGsConfig(**{k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                      "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **kw)
    model = (cls or STonKGsForPreTraining)(c, kg_embeddings=tsv_rows)
    model.load_state_dict(sd, strict=False)
    return model


def _rel(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    if float(b.norm()) < 1e-6:
        return float((a - b).abs().max())
    return float((a - b).norm() / b.norm())


def _batches(cfg, B):
    """Text lengths from 2 tokens to the whole half; one batch with a hole in a text mask, masked entity positions and a
    sequence WITHOUT any live key (keeps all of its rows; the reference then attends uniformly)."""
    from stonkgs_amd.data import synthetic_batch

    b0 = synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=31, min_text=2)
    b1 = {k: v.clone() for k, v in synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings,
                                                   seed=32, min_text=16).items()}
    half = cfg.max_position_embeddings // 2
    b1["attention_mask"][0, 3:9] = 0
    b1["attention_mask"][0, half + 10:half + 30] = 0
    b1["attention_mask"][1, :] = 0
    b1["attention_mask"][2, :] = 1
    return [b0, b1]


def test_unpadded_step_equals_padded_step(hip):
    """Same weights, same batch, dropout off: loss terms and EVERY gradient tensor of the packed run against the padded
    run of the same kernels, and the packed run did drop rows. The two are not the same arithmetic: packing moves the
    64-key tile boundaries of the attention kernels, so the online softmax rounds its bf16 probabilities against
    different running maxima - differences of bf16 rounding size (measured: at most 3e-3 relative on a gradient tensor,
    on the position embeddings, whose rows sum over only B terms). So besides the direct comparison (1e-2), both runs are
    held against the ORACLE's fp32 gradients: the packed run is as close to the truth as the padded one, tensor by tensor."""
    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    B = 6
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    for batch in _batches(cfg, B):
        truth = orc.train_step({k: v.clone() for k, v in sd.items()}, cfg, table, batch, orc.AdamState(), max_grad_norm=0.0)["grads"]
        res = []
        for unpad, prune, pattn in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
            m = _model(cfg, sd, tsv_rows)
            m.train()
            m.engine.unpad, m.engine.prune_last_ffn, m.engine.prune_last_attn = unpad, prune, pattn
            loss = float(m.forward_backward(batch))
            m.engine.join_wgrad()
            m.engine.check_errors()
            torch.cuda.synchronize()
            res.append((loss, [float(t) for t in m.last_loss_terms],
                        {k: v.detach().clone() for k, v in m.named_grad_views().items()}, list(m.engine.rows_executed)))
        (l0, t0, g0, r0), (l1, t1, g1, r1), (l2, t2, g2, r2), (l3, t3, g3, r3) = res
        # ... and the last layer's attention block behind the QKV projection on the read rows too (query limits)
        assert r3[:6] == r2[:6] and r3[6] < r2[6] // 2 and abs(l0 - l3) < 1e-4 * abs(l0) and np.allclose(t0, t3, rtol=2e-4, atol=1e-5)
        errs3 = sorted(((_rel(g3[k], g0[k]), k) for k in g0), reverse=True)
        print("pruned last layer (attention too) vs padded, worst gradient tensors:", [(round(e, 5), k) for e, k in errs3[:3]])
        assert errs3[0][0] < 1e-2 and errs3[len(errs3) // 2][0] < 2e-3, errs3[:3]
        assert r0[0] == r0[1] == B * cfg.max_position_embeddings and r1[0] < r1[1]      # rows were dropped
        # last-layer feed-forward block / pooler / head transform on the READ rows only (labelled + position 0): the same
        # loss and gradients again, on a fraction of the rows (B * (1 + 2 * int(half * 0.15)) of them, rounded up to 64)
        n_read = B + int((batch["masked_lm_labels"][:, 1:] != -100).sum()) + int((batch["ent_masked_lm_labels"] != -100).sum())
        assert r1[5] == r1[0] and r2[0] == r1[0] and r2[5] == (n_read + 63) // 64 * 64 and r2[5] < r2[0] // 2
        assert abs(l0 - l2) < 1e-4 * abs(l0) and np.allclose(t0, t2, rtol=2e-4, atol=1e-5)
        errs2 = sorted(((_rel(g2[k], g0[k]), k) for k in g0), reverse=True)
        print("pruned last layer vs padded, worst gradient tensors:", [(round(e, 5), k) for e, k in errs2[:3]])
        assert errs2[0][0] < 1e-2 and errs2[len(errs2) // 2][0] < 2e-3, errs2[:3]
        kept = int(((batch["attention_mask"] != 0).any(1, keepdim=True) == 0).sum()) * cfg.max_position_embeddings
        assert r1[0] >= int((batch["attention_mask"] != 0).sum()) + kept - 64
        assert abs(l0 - l1) < 1e-4 * abs(l0), (l0, l1)
        assert np.allclose(t0, t1, rtol=2e-4, atol=1e-5)
        errs = sorted(((_rel(g1[k], g0[k]), k) for k in g0), reverse=True)
        print("packed vs padded, worst gradient tensors:", [(round(e, 5), k) for e, k in errs[:4]],
              "median", round(errs[len(errs) // 2][0], 6))
        assert errs[0][0] < 1e-2 and errs[len(errs) // 2][0] < 2e-3, errs[:3]
        for k, ref in truth.items():                       # against fp32 truth: packing loses nothing
            if k not in g0:
                continue
            e0, e1, e2, e3 = (_rel(g[k].cpu(), ref) for g in (g0, g1, g2, g3))
            for e in (e1, e2, e3):
                assert e < max(1.3 * e0 + 2e-3, 8e-3) if float(ref.norm()) > 1e-6 else e < 1e-3, (k, e0, e1, e2, e3)
        total0 = torch.sqrt(sum((g.double() ** 2).sum() for g in g0.values()))
        total1 = torch.sqrt(sum((g.double() ** 2).sum() for g in g1.values()))
        assert abs(float(total0) - float(total1)) < 1e-4 * float(total0)


def test_unpadded_training_matches_the_oracle(hip):
    """The packed path against the CPU oracle (which, like the reference, runs all 512 positions): three optimizer steps."""
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    B = 4
    batches = _batches(cfg, B)
    model = _model(cfg, sd, tsv_rows)
    assert model.engine.unpad
    tr = Trainer(model, TrainingArguments(max_steps=10, learning_rate=1e-3, per_device_train_batch_size=B))
    got = [float(tr.training_step(model, batches[i % 2])) for i in range(3)]
    model.engine.check_errors()
    assert model.engine.rows_executed[0] < model.engine.rows_executed[1]
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    osd = {k: v.clone() for k, v in sd.items()}
    state = orc.AdamState()
    ref = [float(orc.train_step(osd, cfg, table, batches[i % 2], state, base_lr=1e-3, max_steps=10)["loss"]) for i in range(3)]
    assert np.abs(np.array(got) - np.array(ref)).max() < 1e-2, (got, ref)
    params = dict(model.named_parameters())
    for k in ("bert.encoder.layer.0.attention.self.query.weight", "bert.embeddings.position_embeddings.weight",
              "cls.predictions.entity_decoder.weight", "bert.pooler.dense.weight"):
        d_got, d_ref = params[k].detach().cpu() - sd[k], osd[k] - sd[k]
        cos = torch.nn.functional.cosine_similarity(d_got.flatten(), d_ref.flatten(), dim=0).item()
        assert cos > 0.97, (k, cos)


def test_autograd_bridge_and_dataclass_forward_choose_the_layout(hip):
    """`model(**batch)` in training mode returns a tuple without hidden states: packed; `return_dict=True` hands out
    hidden_states for every position: padded. Both give the same loss, and loss.backward() drives the packed backward."""
    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    batch = _batches(cfg, 4)[0]
    m = _model(cfg, sd, tsv_rows)
    m.train()
    before = list(m.engine.rows_executed)
    out = m(**batch)
    after = list(m.engine.rows_executed)
    assert after[0] - before[0] < after[1] - before[1]
    out[0].backward()
    g_packed = {k: v.detach().clone() for k, v in m.named_grad_views().items()}
    m.zero_grad()
    d = m(**batch, return_dict=True)
    assert d.hidden_states.shape == (4, cfg.max_position_embeddings, cfg.hidden_size)
    assert m.engine.rows_executed[0] - after[0] == m.engine.rows_executed[1] - after[1]
    assert abs(float(d.loss) - float(out[0])) < 1e-4 * abs(float(d.loss))
    d.loss.backward()
    g_padded = m.named_grad_views()
    worst = max((_rel(g_packed[k], g_padded[k]), k) for k in g_packed)
    assert worst[0] < 1e-2, worst


def test_unpadded_classification_step_equals_padded(hip):
    """Config 5's head (ref:src/stonkgs/models/stonkgs_finetuning.py:277-338) reads position 0 only: padding rows are
    dropped in training; loss and gradients equal the padded run."""
    from stonkgs_amd.stonkgs_model import STonKGsForSequenceClassification

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    batch = _batches(cfg, 5)[0]
    labels = torch.tensor([0, 2, 1, 1, 0])
    res = []
    for unpad in (False, True):
        m = _model(cfg, sd, tsv_rows, cls=STonKGsForSequenceClassification, num_labels=3)
        m.train()
        m.engine.unpad = unpad
        inputs = {k: batch[k] for k in ("input_ids", "attention_mask", "token_type_ids")}
        loss = float(m.forward_backward(dict(inputs, labels=labels)))
        m.engine.join_wgrad()
        torch.cuda.synchronize()
        res.append((loss, {k: v.detach().clone() for k, v in m.named_grad_views().items()}, list(m.engine.rows_executed)))
    (l0, g0, r0), (l1, g1, r1) = res
    assert r1[0] < r0[0] and abs(l0 - l1) < 1e-4 * abs(l0)
    worst = max((_rel(g1[k], g0[k]), k) for k in g0)
    assert worst[0] < 1e-2, worst


def test_backbone_prefetch_changes_nothing(hip):
    """TrainingArguments.prefetch_backbone: the NEXT batch's frozen-backbone forward runs on its own stream beside the
    current step (it depends on token ids and frozen weights only). Same losses with the hint, without it, and with a WRONG
    hint (the prefetched result is dropped and the forward runs inline); host batches are converted once."""
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    batches = _batches(cfg, 4) + _batches(cfg, 4)[::-1]
    runs = {}
    for mode in ("off", "hint", "wrong", "host"):
        model = _model(cfg, sd, tsv_rows)
        tr = Trainer(model, TrainingArguments(max_steps=10, learning_rate=1e-3, per_device_train_batch_size=4,
                                              prefetch_backbone=mode != "off"))
        bs = batches if mode == "host" else [{k: v.cuda() for k, v in b.items()} for b in batches]
        losses = []
        for i, b in enumerate(bs):
            nxt = None if mode == "off" or i + 1 == len(bs) else (bs[i + 1] if mode != "wrong" else bs[0])
            losses.append(float(tr.training_step(model, b, next_inputs=nxt)))
            if mode in ("hint", "host") and nxt is not None:
                assert model.engine._prefetch is not None and model.engine.next_input_ids is None
        model.engine.check_errors()
        torch.cuda.synchronize()
        runs[mode] = losses
    for mode in ("hint", "wrong", "host"):
        # the same kernels on the same data (the loss sums are float atomics: equal up to their arrival order)
        assert abs(runs[mode][0] - runs["off"][0]) < 1e-5 * abs(runs["off"][0]), mode
        assert np.allclose(runs[mode], runs["off"], rtol=0, atol=2e-3), (mode, runs[mode], runs["off"])


def test_forward_only_call_between_forward_and_backward_leaves_the_plan_alone(hip):
    """`encode()` (embedding extraction: forward only) may run between a training forward and its backward: it works in
    scratch buffers of its own throughout - embeddings, layers, pooler, row plan - so the pending backward still finds the
    activations and the maps of ITS batch."""
    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    b0, b1 = _batches(cfg, 4)
    grads = []
    for interleave in (False, True):
        m = _model(cfg, sd, tsv_rows)
        m.train()
        out = m(**b0)
        if interleave:
            m.eval()
            _, pooled = m.encode(b1["input_ids"][:3], b1["attention_mask"][:3], b1["token_type_ids"][:3], pooled_only=True)
            assert bool(torch.isfinite(pooled).all())
            m.train()
        out[0].backward()
        m.engine.join_wgrad()
        torch.cuda.synchronize()
        grads.append({k: v.detach().clone() for k, v in m.named_grad_views().items()})
    worst = max((_rel(grads[1][k], grads[0][k]), k) for k in grads[0])
    assert worst[0] < 1e-4, worst
