"""CPU: the oracle (oracle/stonkgs_oracle.py) against golden vectors produced by the REFERENCE itself
(oracle/make_golden.py ran the reference's forward + HF BERT + torch AdamW in the authoring container)."""
import random

import numpy as np
import pytest
import torch

from oracle import stonkgs_oracle as orc
from tests.golden_util import GOLDEN, load_case


@pytest.fixture(scope="module", params=["g1_tiny", "g2_hipsmall"])
def case(request):
    return load_case(request.param)


def _table(cfg, sd, tsv_rows):
    return orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))


def test_forward_matches_reference(case):
    cfg, sd, tsv_rows, batch, gold, meta = case
    with torch.no_grad():
        table = _table(cfg, sd, tsv_rows)
        for sid in (100, 102, 103):  # quirk Q2
            np.testing.assert_allclose(table[sid].numpy(), gold[f"special_{sid}"], rtol=1e-5, atol=2e-6)
        out = orc.forward(sd, cfg, table, **batch)
    for k in ("loss", "masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"):
        assert abs(float(out[k]) - float(gold[k])) <= 1e-5, k
    np.testing.assert_allclose(out["nsp_logits"].numpy(), gold["nsp_logits"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["pooler_output"].numpy(), gold["pooler_output"], rtol=1e-5, atol=1e-5)
    if "text_logits" in gold:
        np.testing.assert_allclose(out["hidden_states"].numpy(), gold["hidden_states"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["text_logits"].numpy(), gold["text_logits"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["ent_logits"].numpy(), gold["ent_logits"], rtol=1e-4, atol=1e-5)
    else:
        np.testing.assert_allclose(out["hidden_states"][:, ::7, ::3].numpy(), gold["hidden_states_s"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["text_logits"][:, ::5, ::3].numpy(), gold["text_logits_s"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["ent_logits"][batch["ent_masked_lm_labels"] != -100].numpy(),
                                   gold["ent_logits_lab"], rtol=1e-4, atol=1e-5)


def test_gradients_and_two_optimizer_steps_match_reference(case):
    cfg, sd, tsv_rows, batch, gold, meta = case
    with torch.no_grad():
        table = _table(cfg, sd, tsv_rows)
    sd = {k: v.clone() for k, v in sd.items()}
    state = orc.AdamState()
    r1 = orc.train_step(sd, cfg, table, batch, state, base_lr=1e-4, max_steps=200)
    assert abs(float(r1["grad_norm"]) - float(gold["grad_norm"])) <= 1e-4 * float(gold["grad_norm"])
    for k in meta["grad_keys"]:
        np.testing.assert_allclose(r1["grads"][k].numpy(), gold["grad::" + k], rtol=2e-4, atol=2e-6, err_msg=k)
    # dead parameters (quirk Q4) never receive a gradient in the reference either
    assert set(meta["dead_parameters"]).isdisjoint(orc.trainable_names(sd))
    r2 = orc.train_step(sd, cfg, table, batch, state, base_lr=1e-4, max_steps=200)
    np.testing.assert_allclose([float(r1["loss"]), float(r2["loss"])], gold["step_losses"], rtol=0, atol=2e-5)
    for k in meta["grad_keys"]:
        np.testing.assert_allclose(sd[k].numpy(), gold["after2::" + k], rtol=0, atol=2e-6, err_msg=k)


def test_masking_is_bit_exact():
    gold = dict(np.load(GOLDEN + "/masking.npz"))
    for seed in (0, 1, 1234):
        random.seed(seed)
        t_in, t_lab = orc.replace_mlm_tokens(list(range(1000, 1256)), 28996)
        e_in, e_lab = orc.replace_mlm_tokens([(7 * i) % 175094 for i in range(256)], 175094)
        assert t_in == gold[f"text_in_{seed}"].tolist() and t_lab == gold[f"text_lab_{seed}"].tolist()
        assert e_in == gold[f"ent_in_{seed}"].tolist() and e_lab == gold[f"ent_lab_{seed}"].tolist()
        assert sum(l != -100 for l in t_lab) == 38  # int(256 * 0.15)
        random.seed(seed)
        pairs = orc.negative_nsp_index_pairs(8)
        rows = [list(range(i * 10, i * 10 + 8)) for i in range(8)]
        neg_ids = [rows[i][:4] + rows[j][4:] for i, j in pairs]
        assert neg_ids == gold[f"neg_input_ids_{seed}"].tolist()
        assert [[j] * 4 for _, j in pairs] == gold[f"neg_ent_labels_{seed}"].tolist()
        assert gold[f"neg_nsp_{seed}"].tolist() == [1] * len(pairs)


def test_entity_index_space_quirk_q1():
    K = 175094
    exp = {0: 0, 99: 99, 100: None, 101: 100, 102: None, 103: None, 104: 101, K - 1: K - 4, K + 2: K - 1}
    for e, r in exp.items():
        assert orc.kg_row_of_entity_id(e) == r
    sv = {s: torch.full((2,), -float(s)) for s in (100, 102, 103)}
    big = torch.arange(200, dtype=torch.float64)[:, None].repeat(1, 2)
    t = orc.build_kg_table(big, sv)
    assert t.shape[0] == 203 and t[99, 0] == 99 and t[100, 0] == -100 and t[101, 0] == 100 and t[104, 0] == 101
    assert t[202, 0] == 199


def test_out_of_table_entity_raises_keyerror(case):
    cfg, sd, tsv_rows, batch, gold, meta = case
    with torch.no_grad():
        table = _table(cfg, sd, tsv_rows)
        bad = {k: v.clone() for k, v in batch.items()}
        bad["input_ids"][0, -1] = cfg.kg_vocab_size + 3
        with pytest.raises(KeyError):
            orc.forward(sd, cfg, table, **bad)


def _cls_case(name):
    import json
    import os

    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    gold = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    cfg = orc.OracleConfig(**meta["config"])
    sd = orc.init_state_dict(cfg, seed=meta["weight_seed"])
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - meta["weights_checksum"]) < 1e-6
    gw = torch.Generator().manual_seed(meta["classifier_seed"])
    nl, H = meta["num_labels"], cfg.hidden_size
    sd["classifier.weight"] = (torch.randn(nl, H, generator=gw) * 0.02).to(torch.bfloat16).float()
    sd["classifier.bias"] = (torch.randn(nl, generator=gw) * 0.02).to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(meta["table_seed"])
    rows = torch.randn(cfg.kg_vocab_size, H, generator=g, dtype=torch.float64) * meta["table_std"]
    return cfg, sd, rows, gold, meta


@pytest.mark.parametrize("name", ["g6_classification", "g13_cls_regression_1d", "g14_cls_regression", "g15_cls_multilabel"])
def test_classification_head_matches_reference(name):
    """The reference's STonKGsForSequenceClassification.forward (ref:stonkgs_finetuning.py:259-346), its three loss
    branches: G6 single-label CE (ragged batch of 5, 3 classes); G13 regression with num_labels = 1 and 1-D float labels
    (MSELoss broadcasts [B,1] x [B] to [B,B] - the reference's own behaviour, restated as is); G14 regression over
    [B,3]; G15 multi-label BCEWithLogitsLoss."""
    cfg, sd, rows, gold, meta = _cls_case(name)
    with torch.no_grad():
        table = orc.build_kg_table(rows, orc.special_vectors(sd, cfg))
    names = [k for k in sd if k.startswith("bert.") and "word_embeddings" not in k] + ["classifier.weight", "classifier.bias"]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(params)
    out = orc.forward_classification(work, cfg, table, torch.from_numpy(gold["input_ids"]),
                                     torch.from_numpy(gold["attention_mask"]), torch.from_numpy(gold["token_type_ids"]),
                                     torch.from_numpy(gold["labels"]),
                                     problem_type=meta.get("problem_type", "single_label_classification"))
    assert abs(float(out["loss"]) - float(gold["loss"])) < 1e-5
    np.testing.assert_allclose(out["logits"].detach().numpy(), gold["logits"], rtol=1e-4, atol=1e-5)
    out["loss"].backward()
    for k in meta["grad_keys"]:
        np.testing.assert_allclose(params[k].grad.numpy(), gold["grad::" + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_masking_oracle_properties():
    """The numpy restatement of the device masking stream (oracle/masking_oracle.py) keeps the reference's invariants
    (ref:indra_for_pretraining.py:33-77): exactly int(len * 0.15) labels per half, labels = original ids, untouched
    positions unchanged, deterministic in the seed."""
    import numpy as np

    from oracle import masking_oracle as mo

    rng = np.random.RandomState(0)
    ids = rng.randint(0, 300, (4, 64)).astype(np.int64)
    out, tl, el = mo.mlm_mask(ids, 32, 300, 50, seed=9)
    out2, tl2, el2 = mo.mlm_mask(ids, 32, 300, 50, seed=9)
    assert np.array_equal(out, out2) and np.array_equal(tl, tl2) and np.array_equal(el, el2)
    for lab, off in ((tl, 0), (el, 32)):
        sel = lab != -100
        assert (sel.sum(1) == int(32 * 0.15)).all()
        assert np.array_equal(lab[sel], ids[:, off:off + 32][sel])
        assert np.array_equal(out[:, off:off + 32][~sel], ids[:, off:off + 32][~sel])
    assert not np.array_equal(mo.mlm_mask(ids, 32, 300, 50, seed=10)[1], tl)
    walks = rng.randint(0, 50, (20, 15)).astype(np.int64)
    text = rng.randint(1, 300, (6, 32)).astype(np.int64)
    a, att, typ, nsp = mo.assemble_rows(text, np.ones_like(text), np.arange(6), np.arange(6) + 3, walks, seed=4)
    assert a.shape == (6, 64) and (a[:, 32 + 15] == 102).all() and (a[:, 63] == 102).all()
    assert np.array_equal(a[:, :32], text) and set(np.unique(nsp)) <= {0, 1} and (typ[:, 32:] == 1).all()
    for b in range(6):
        if nsp[b] == 0:
            assert np.array_equal(a[b, 32:47], walks[b]) and np.array_equal(a[b, 48:63], walks[b + 3])


def _slice_of(spec):
    return tuple(slice(a, b, c) for a, b, c in spec)


def test_shape_true_case_matches_reference():
    """SURVEY section 8c's G2 at the real depth / width / head count (12L, 768h, 12 heads, S = 512, V = 28 996; K = 4 096):
    the oracle against the reference's own forward and backward - loss terms, sampled outputs, the global gradient norm,
    the norm of EVERY gradient tensor and sampled slices of seven of them."""
    cfg, sd, tsv_rows, batch, gold, meta = load_case("g3_shapetrue")
    torch.set_num_threads(8)
    with torch.no_grad():
        table = _table(cfg, sd, tsv_rows)
        for sid in (100, 102, 103):
            np.testing.assert_allclose(table[sid].numpy(), gold[f"special_{sid}"], rtol=1e-4, atol=1e-5)
    res = orc.train_step({k: v.clone() for k, v in sd.items()}, cfg, table, batch, orc.AdamState(), max_grad_norm=0.0,
                         return_outputs=True)
    for k in ("loss", "masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"):
        assert abs(float(res[k]) - float(gold[k])) <= 2e-5, k
    out = res["outputs"]
    np.testing.assert_allclose(out["pooler_output"].numpy(), gold["pooler_output"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["nsp_logits"].numpy(), gold["nsp_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["hidden_states"][:, ::37, ::11].numpy(), gold["hidden_states_s"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(out["text_logits"][batch["masked_lm_labels"] != -100][:, ::97].numpy(),
                               gold["text_logits_lab_s"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(out["ent_logits"][batch["ent_masked_lm_labels"] != -100][:, ::29].numpy(),
                               gold["ent_logits_lab_s"], rtol=1e-3, atol=2e-5)
    assert abs(float(res["grad_norm"]) - float(gold["grad_norm"])) <= 1e-4 * float(gold["grad_norm"])
    assert set(meta["grad_names"]) == set(orc.trainable_names(sd))
    for name, ref_norm in zip(meta["grad_names"], gold["grad_norms"]):
        got = float(res["grads"][name].double().norm())
        assert abs(got - ref_norm) <= 2e-4 * ref_norm + 1e-7, name
    for k in meta["grad_keys"]:
        np.testing.assert_allclose(res["grads"][k][_slice_of(meta["grad_slices"][k])].numpy(), gold["grad_s::" + k],
                                   rtol=1e-3, atol=2e-6, err_msg=k)


def test_reference_written_checkpoint_and_tsv_table():
    """Row f3 on the oracle side: the state dict the reference-side model wrote (tests/golden/g9_ref_checkpoint, HF layout)
    and the node2vec TSV as the reference's prepare_df read it (g8_table.*) reproduce the reference's outputs."""
    import json
    import os

    gold = dict(np.load(os.path.join(GOLDEN, "g9_ref_checkpoint.npz")))
    meta = json.load(open(os.path.join(GOLDEN, "g9_ref_checkpoint.json")))
    tab = dict(np.load(os.path.join(GOLDEN, "g8_table.npz")))
    cfg = orc.OracleConfig(**meta["config"])
    sd = torch.load(os.path.join(GOLDEN, "g9_ref_checkpoint", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    assert sorted(sd.keys()) == meta["state_dict_keys"]
    seeded = orc.init_state_dict(cfg, seed=meta["weight_seed"])
    for k, v in seeded.items():
        assert torch.equal(sd[k], v), k    # the reference-side writer stored exactly the seeded weights, under HF's names
    tsv_rows = torch.from_numpy(tab["values"])
    with torch.no_grad():
        table = _table(cfg, sd, tsv_rows)
        batch = {k: torch.from_numpy(gold[k]) for k in ("input_ids", "attention_mask", "token_type_ids", "masked_lm_labels",
                                                        "ent_masked_lm_labels", "next_sentence_labels")}
        out = orc.forward(sd, cfg, table, **batch)
    assert abs(float(out["loss"]) - float(gold["loss"])) <= 1e-5
    np.testing.assert_allclose(out["pooler_output"].numpy(), gold["pooler_output"], rtol=1e-5, atol=1e-5)
    for e in (7, 101, 104):        # quirk Q1 through the TSV: id 101 -> row 100, id 104 -> row 101
        np.testing.assert_array_equal(table[e].numpy(), gold[f"table_row_{e}"].astype(np.float32))
    np.testing.assert_allclose(table[102].numpy(), gold["special_102"], rtol=1e-5, atol=2e-6)


def test_weight_decay_grouping_matches_torch_adamw_with_hf_groups():
    """The oracle's decoupled weight decay (unused by the reference, which trains with 0) against torch.optim.AdamW with the
    two parameter groups HF Trainer builds (hf:trainer.py get_decay_parameter_names: no decay on biases / LayerNorm)."""
    cfg, sd, tsv_rows, batch, gold, meta = load_case("g1_tiny")
    with torch.no_grad():
        table = _table(cfg, sd, tsv_rows)
    names = orc.trainable_names(sd)
    params = {k: torch.nn.Parameter(sd[k].clone()) for k in names}
    decay = [params[k] for k in names if not (k.endswith(".bias") or "LayerNorm" in k)]
    rest = [params[k] for k in names if (k.endswith(".bias") or "LayerNorm" in k)]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": rest, "weight_decay": 0.0}], lr=1e-2,
                            betas=(0.9, 0.999), eps=1e-8)
    osd = {k: v.clone() for k, v in sd.items()}
    state = orc.AdamState()
    for step in range(2):
        work = dict(sd)
        work.update(params)
        loss = orc.forward(work, cfg, table, **batch)["loss"]
        opt.zero_grad()
        loss.backward()
        for p in params.values():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        for g in opt.param_groups:
            g["lr"] = orc.linear_schedule_lr(1e-2, step, 200)
        opt.step()
        orc.train_step(osd, cfg, table, batch, state, base_lr=1e-2, max_steps=200, weight_decay=0.1)
    for k in names:
        np.testing.assert_allclose(osd[k].numpy(), params[k].detach().numpy(), rtol=0, atol=3e-6, err_msg=k)


def test_oracle_tracks_the_reference_fp32_loss_curve():
    """G12 / G11: the reference's own fp32 training run (200 steps at the small shape; the first two of the 60 at
    12L / 768h) against the oracle's Trainer step on the same weights and batches: the CPU restatement IS the reference's
    curve (measured max |d| 2.4e-6 over 200 steps), which is what lets the GPU tests use either."""
    import numpy as np

    from tests.golden_util import load_curve_case

    for name, n, tol in (("g12_curve_small", 200, 2e-4), ("g11_curve_shapetrue", 2, 2e-4)):
        cfg, sd, rows, batches, ref32, ref16, meta = load_curve_case(name)
        with torch.no_grad():
            table = orc.build_kg_table(rows, orc.special_vectors(sd, cfg))
        osd = {k: v.clone() for k, v in sd.items()}
        state = orc.AdamState()
        got = [float(orc.train_step(osd, cfg, table, batches[i % len(batches)], state, base_lr=meta["learning_rate"],
                                    max_steps=meta["steps"])["loss"]) for i in range(n)]
        d = np.abs(np.array(got) - ref32[:n])
        assert d.max() < tol, (name, d.max())
        # the reference's reduced-precision run is NOT within north_star's 1e-3 of its own fp32 run
        assert np.abs(ref16 - ref32).max() > 1e-2
