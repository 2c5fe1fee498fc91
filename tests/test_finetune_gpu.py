"""GPU parity of BASELINE config 5 (STonKGsForSequenceClassification) against the reference-made golden case G6 and
the oracle on ragged batch sizes; and of the wide configuration (hidden 1024 / 16 heads / FFN 4096, the shape family
of BASELINE config 4) against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import stonkgs_oracle as orc
from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    if b.norm() < 1e-5:
        return (a - b).norm().item() / 1e-2
    return ((a - b).norm() / b.norm()).item()


def _g6(name="g6_classification"):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    gold = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    cfg = orc.OracleConfig(**meta["config"])
    sd = orc.init_state_dict(cfg, seed=meta["weight_seed"])
    gw = torch.Generator().manual_seed(meta["classifier_seed"])
    nl, H = meta["num_labels"], cfg.hidden_size
    sd["classifier.weight"] = (torch.randn(nl, H, generator=gw) * 0.02).to(torch.bfloat16).float()
    sd["classifier.bias"] = (torch.randn(nl, generator=gw) * 0.02).to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(meta["table_seed"])
    rows = torch.randn(cfg.kg_vocab_size, H, generator=g, dtype=torch.float64) * meta["table_std"]
    return cfg, sd, rows, gold, meta


def _build_cls(cfg, sd, rows, num_labels, dropout=0.0):
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.stonkgs_model import STonKGsForSequenceClassification

    c = STonKGsConfig(**{k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                      "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
                      hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, num_labels=num_labels)
    model = STonKGsForSequenceClassification(c, kg_embeddings=rows)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("decoder" in k for k in missing), (missing, unexpected)
    return model


def test_classification_matches_reference_golden(hip):
    cfg, sd, rows, gold, meta = _g6()
    model = _build_cls(cfg, sd, rows, meta["num_labels"])
    batch = {k: torch.from_numpy(gold[k]) for k in ("input_ids", "attention_mask", "token_type_ids", "labels")}
    model.eval()
    with torch.no_grad():
        out = model(**batch, return_dict=True)
    assert abs(float(out.loss) - float(gold["loss"])) < 5e-3
    assert _rel(out.logits, gold["logits"]) < 3e-2
    # training mode, p = 0: loss.backward() through the autograd bridge
    model.train()
    model._store.grad.zero_()
    loss, logits = model(**batch)
    assert logits.shape == (5, 3)
    loss.backward()
    params = dict(model.named_parameters())
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.requires_grad))
    assert abs(float(total) - float(gold["grad_norm"])) < 4e-2 * float(gold["grad_norm"])
    for k in meta["grad_keys"]:
        assert _rel(params[k].grad, gold["grad::" + k]) < 8e-2, k
    # the inherited pre-training heads are present in the checkpoint but frozen and outside the gradient buffer
    assert "cls.predictions.entity_decoder.weight" in model.state_dict()
    assert not params["cls.predictions.entity_decoder.weight"].requires_grad
    model.engine.check_errors()


@pytest.mark.parametrize("name", ["g13_cls_regression_1d", "g14_cls_regression", "g15_cls_multilabel"])
def test_regression_and_multilabel_heads_match_reference_golden(hip, name):
    """The other two loss branches of ref:src/stonkgs/models/stonkgs_finetuning.py:328-338 against reference-made goldens:
    MSELoss (num_labels = 1 with 1-D labels - torch broadcasts [B,1] x [B] to [B,B] and the reference inherits it - and
    [B,3] with problem_type set) and BCEWithLogitsLoss (inferred from float labels, as the reference infers it)."""
    cfg, sd, rows, gold, meta = _g6(name)
    model = _build_cls(cfg, sd, rows, meta["num_labels"])
    if meta["problem"] == "regression":
        model.config.problem_type = "regression"       # (the golden set it too: float labels with num_labels > 1 infer multi-label)
    batch = {k: torch.from_numpy(gold[k]) for k in ("input_ids", "attention_mask", "token_type_ids", "labels")}
    model.eval()
    with torch.no_grad():
        out = model(**batch, return_dict=True)
    assert model.config.problem_type == meta["problem_type"]
    assert abs(float(out.loss) - float(gold["loss"])) < 5e-3, (float(out.loss), float(gold["loss"]))
    assert _rel(out.logits, gold["logits"]) < 3e-2
    model.train()
    model._store.grad.zero_()
    loss, logits = model(**batch)
    loss.backward()
    params = dict(model.named_parameters())
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.requires_grad))
    assert abs(float(total) - float(gold["grad_norm"])) < 4e-2 * float(gold["grad_norm"])
    for k in meta["grad_keys"]:
        assert _rel(params[k].grad, gold["grad::" + k]) < 8e-2, k
    # the fused step takes the same branch
    model._store.grad.zero_()
    l2 = float(model.forward_backward(dict(batch)))
    assert abs(l2 - float(loss)) < 1e-5 + 1e-5 * abs(l2)
    model.engine.check_errors()


@pytest.mark.parametrize("B", [8, 16, 5])
def test_classification_mixed_batch_sizes_against_oracle(hip, B):
    """Config 5: per-device batch 8 / 16 and a ragged last batch of 5 through the fused Trainer step."""
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg, sd, rows, gold, meta = _g6()
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["classifier.weight"], sd2["classifier.bias"] = sd["classifier.weight"][:2].clone(), sd["classifier.bias"][:2].clone()
    model = _build_cls(cfg, sd2, rows, 2)
    b = synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=50 + B, min_text=16)
    labels = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(B))
    inputs = {"input_ids": b["input_ids"], "attention_mask": b["attention_mask"], "token_type_ids": b["token_type_ids"],
              "labels": labels}
    with torch.no_grad():
        table = orc.build_kg_table(rows, orc.special_vectors(sd2, cfg))
        ref = orc.forward_classification(sd2, cfg, table, **inputs)
    tr = Trainer(model, TrainingArguments(learning_rate=5e-5, max_steps=100, per_device_train_batch_size=B))
    loss = float(tr.training_step(model, inputs))
    assert abs(loss - float(ref["loss"])) < 5e-3
    before = sd2["classifier.weight"]
    after = model.classifier.weight.detach().cpu()
    assert (after - before).abs().max() > 0 and (after - before).abs().max() <= 5e-5 * 1.01  # one Adam step of lr
    model.engine.check_errors()


def test_wide_configuration_hidden_1024(hip):
    """Shape family of BASELINE config 4 (hidden 1024, 16 heads, FFN 4096; 2 layers to keep the CPU oracle quick)."""
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    dims = dict(vocab_size=640, kg_vocab_size=400, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16,
                intermediate_size=4096, max_position_embeddings=256)
    ocfg = orc.OracleConfig(**dims)
    sd = orc.init_state_dict(ocfg, seed=41)
    rows = torch.randn(400, 1024, generator=torch.Generator().manual_seed(42), dtype=torch.float64) * 0.3
    model = STonKGsForPreTraining(STonKGsConfig(**dims, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
                                  kg_embeddings=rows)
    model.load_state_dict(sd, strict=False)
    b = synthetic_batch(2, 640, 400, 256, seed=43, min_text=16)
    with torch.no_grad():
        table = orc.build_kg_table(rows, orc.special_vectors(sd, ocfg))
    res = orc.train_step({k: v.clone() for k, v in sd.items()}, ocfg, table, b, orc.AdamState(), max_grad_norm=0.0)
    model.train()
    loss = float(model.forward_backward(b))
    assert abs(loss - float(res["loss"])) < 3e-2
    gv = model.named_grad_views()
    for k in ("bert.encoder.layer.0.attention.self.query.weight", "cls.predictions.entity_decoder.weight",
              "bert.embeddings.position_embeddings.weight", "bert.encoder.layer.1.output.dense.bias"):
        assert _rel(gv[k], res["grads"][k]) < 8e-2, k
    model.engine.check_errors()


def test_cross_validated_fine_tuning_learns_a_separable_task(hip):
    """f4: the cross-validation driver end to end on the HIP path - fresh model per fold, Trainer steps with a ragged last
    batch, batched prediction, weighted F1. Labels are a function of the input (entity half shifted per class), so two
    a dozen short epochs must beat chance clearly on held-out folds."""
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_finetuning import run_sequence_classification_cv

    cfg, sd, rows, gold, meta = _g6()
    n = 30
    b = synthetic_batch(n, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=77, min_text=16)
    labels = np.arange(n) % 2
    ids = b["input_ids"].clone()
    half = cfg.max_position_embeddings // 2
    ids[:, half:] = torch.where(torch.from_numpy(labels)[:, None] == 1, ids[:, half:] % 40, 150 + ids[:, half:] % 40)
    data = {"input_ids": ids.tolist(), "attention_mask": b["attention_mask"].tolist(),
            "token_type_ids": b["token_type_ids"].tolist(), "labels": labels}
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["classifier.weight"], sd2["classifier.bias"] = sd["classifier.weight"][:2].clone(), sd["classifier.bias"][:2].clone()
    made = []

    def factory(num_labels):
        assert num_labels == 2
        made.append(_build_cls(cfg, sd2, rows, num_labels))
        return made[-1]

    f1, frame = run_sequence_classification_cv(data, model_factory=factory, epochs=12, lr=1e-3, batch_size=8, n_splits=3)
    assert len(f1) == 3 and len(made) == 3 and len(frame) == n and sorted(frame["index"].tolist()) == list(range(n))
    assert set(frame.columns) == {"split", "index", "predicted_label", "true_label"}
    assert np.mean(f1) > 0.75, f1
