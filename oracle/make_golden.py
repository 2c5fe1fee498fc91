"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF  (authoring container only).

TEST INFRASTRUCTURE. Reads /root/reference at run time (never copies it): imports the reference's
``stonkgs.models.stonkgs_model`` and ``stonkgs.data.indra_for_pretraining`` unmodified, with the package
``__init__`` files and the network-touching ``stonkgs.constants`` replaced by stubs (recipe: SURVEY.md
Appendix A), builds the reference model from a LOCAL BertConfig with seeded random weights, runs the
reference's own ``forward`` / HF BERT / torch AdamW on CPU, and stores inputs + expected outputs as arrays.

Nothing here runs on the GPU box; the fixtures it writes are data only (ints / floats / config numbers).

    python oracle/make_golden.py          # rewrites tests/golden/*.npz, *.json
    python oracle/make_golden.py shape    # only g3_shapetrue; `f3` = only the TSV / checkpoint cases; `splits` = only g7;
                                          # `curve` / `curve_small` = only the loss-curve cases g11 / g12;
                                          # `bf16_grads` = only g16 / g17 (reference gradients under bf16 autocast); `cls` = only g6
"""
from __future__ import annotations

import importlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import stonkgs_oracle as orc  # noqa: E402

REF = "/root/reference/src/stonkgs"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name, path in (("stonkgs", REF), ("stonkgs.models", REF + "/models"), ("stonkgs.data", REF + "/data")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]  # namespace stub: skips every __init__.py (pybel / pystow / network)
        sys.modules[name] = pkg
    const = types.ModuleType("stonkgs.constants")
    for k in ("EMBEDDINGS_PATH", "NLP_MODEL_TYPE", "PRETRAINING_DIR", "PRETRAINING_PATH", "RANDOM_WALKS_PATH",
              "VOCAB_FILE", "CELL_LINE_DIR", "CELL_TYPE_DIR", "CORRECT_DIR", "DEEPSPEED_CONFIG_PATH", "DISEASE_DIR",
              "LOCATION_DIR", "MLFLOW_FINETUNING_TRACKING_URI", "ORGAN_DIR", "PRETRAINED_STONKGS_PATH",
              "RELATION_TYPE_DIR", "SPECIES_DIR", "STONKGS_OUTPUT_DIR"):
        setattr(const, k, "/nonexistent/" + k)
    sys.modules["stonkgs.constants"] = const
    kgb = types.ModuleType("stonkgs.models.kg_baseline_model")
    kgb.prepare_df = lambda path, sep="\t": {}  # never called: the hub-fetching __init__ is bypassed
    sys.modules["stonkgs.models.kg_baseline_model"] = kgb
    sm = importlib.import_module("stonkgs.models.stonkgs_model")
    pre = importlib.import_module("stonkgs.data.indra_for_pretraining")
    return sm, pre


def import_reference_finetuning():
    """ref:src/stonkgs/models/stonkgs_finetuning.py imports mlflow (experiment logging, absent here) at module level
    for its cross-validation driver only; an empty module object lets the CLASS (forward :259-346) import unmodified."""
    if "mlflow" not in sys.modules:
        sys.modules["mlflow"] = types.ModuleType("mlflow")
    return importlib.import_module("stonkgs.models.stonkgs_finetuning")


def build_reference_model(sm, cfg: orc.OracleConfig, sd, tsv_rows):
    from transformers import BertConfig, BertForPreTraining, BertModel

    hf_cfg = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings,
                        type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")
    hf_cfg.update({"kg_vocab_size": cfg.kg_vocab_size})

    class RefModel(sm.STonKGsForPreTraining):  # only the hub-fetching __init__ is replaced; forward is the reference's
        def __init__(self, c):
            BertForPreTraining.__init__(self, c)
            self.cls.predictions = sm.STonKGsELMPredictionHead(c)
            self.lm_backbone = BertModel(c)
            for p in self.lm_backbone.parameters():
                p.requires_grad = False
            self.lm_sep_id, self.lm_mask_id, self.lm_unk_id = 102, 103, 100

    model = RefModel(hf_cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if "position_ids" not in k and "decoder" not in k]
    assert not bad and not unexpected, (bad, unexpected)
    model.eval()
    # kg_backbone exactly as ref:stonkgs_model.py:123-141 builds it (TSV order -> numeric_indices, then specials)
    K = cfg.kg_vocab_size
    numeric_indices = list(range(K + 3))
    for sid in (102, 103, 100):
        numeric_indices.remove(sid)
    model.kg_backbone = {i: torch.tensor(tsv_rows[r].numpy()) for r, i in enumerate(numeric_indices)}
    with torch.no_grad():
        for sid in (102, 103, 100):
            model.kg_backbone[sid] = model.lm_backbone(torch.tensor([[sid]]))[0][0][0]
    return model


def make_batch(cfg, B, seed, pre, pad_rows=True):
    """Synthetic batch in the reference's schema; masking by the REFERENCE's replace_mlm_tokens."""
    half = cfg.half_length
    rng = np.random.RandomState(seed)
    random.seed(seed)
    ids, am, tt, ml, el = [], [], [], [], []
    for b in range(B):
        n_real = half if not pad_rows or b == 0 else int(rng.randint(half // 4, half))
        text = [101] + list(rng.randint(104, cfg.vocab_size, n_real - 2)) + [102]
        walk = list(rng.randint(0, cfg.kg_vocab_size, half // 2 - 1)) + [102]
        ent = walk + list(rng.randint(0, cfg.kg_vocab_size, half // 2 - 1)) + [102]
        t_in, t_lab = pre.replace_mlm_tokens(tokens=[int(x) for x in text], vocab_len=cfg.vocab_size)
        e_in, e_lab = pre.replace_mlm_tokens(tokens=[int(x) for x in ent], vocab_len=cfg.kg_vocab_size)
        pad = half - n_real
        ids.append(t_in + [0] * pad + e_in)
        am.append([1] * n_real + [0] * pad + [1] * half)
        tt.append([0] * half + [1] * half)
        ml.append(t_lab + [-100] * pad)
        el.append(e_lab)
    nsp = [int(x) for x in rng.randint(0, 2, B)]
    return {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(am), "token_type_ids": torch.tensor(tt),
            "masked_lm_labels": torch.tensor(ml), "ent_masked_lm_labels": torch.tensor(el),
            "next_sentence_labels": torch.tensor(nsp)}


GRAD_KEYS = ["bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.0.attention.self.key.bias",
             "bert.embeddings.position_embeddings.weight", "bert.embeddings.token_type_embeddings.weight",
             "bert.embeddings.LayerNorm.weight", "bert.pooler.dense.weight", "cls.seq_relationship.weight",
             "cls.predictions.transform.dense.weight", "cls.predictions.transform.LayerNorm.bias",
             "cls.predictions.text_decoder.weight", "cls.predictions.entity_decoder.weight"]


def model_case(name, sm, pre, cfg: orc.OracleConfig, B, seed, full_logits):
    sd = orc.init_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tsv_rows = (torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3)
    model = build_reference_model(sm, cfg, sd, tsv_rows)
    last = f"bert.encoder.layer.{cfg.num_hidden_layers - 1}.output.dense.bias"
    grad_keys = GRAD_KEYS + [last]
    batch = make_batch(cfg, B, seed + 2, pre)
    out = model(**batch, return_dict=True)
    tup = model(**batch)  # return_dict falsy -> tuple (quirk Q8)
    assert torch.equal(tup[0], out.loss) and torch.equal(tup[1][1], out.prediction_logits[1])
    model.zero_grad()
    out.loss.backward()
    params = dict(model.named_parameters())
    dead = [k for k, p in params.items() if p.requires_grad and p.grad is None]
    arrays = {k: v.numpy() for k, v in batch.items()}
    with torch.no_grad():
        half = cfg.half_length
        lt = torch.nn.functional.cross_entropy(out.prediction_logits[0].reshape(-1, cfg.vocab_size),
                                               batch["masked_lm_labels"].reshape(-1))
        le = torch.nn.functional.cross_entropy(out.prediction_logits[1].reshape(-1, cfg.kg_vocab_size),
                                               batch["ent_masked_lm_labels"].reshape(-1))
        ln = torch.nn.functional.cross_entropy(out.seq_relationship_logits, batch["next_sentence_labels"])
        arrays.update(loss=out.loss.numpy(), masked_lm_loss=lt.numpy(), ent_masked_lm_loss=le.numpy(),
                      next_sentence_loss=ln.numpy(), nsp_logits=out.seq_relationship_logits.numpy(),
                      pooler_output=out.pooler_output.numpy())
        hs = out.hidden_states
        tl, el = out.prediction_logits
        if full_logits:
            arrays.update(hidden_states=hs.numpy(), text_logits=tl.numpy(), ent_logits=el.numpy())
        else:  # keep the fixture small: labelled rows + a strided sample
            arrays.update(hidden_states_s=hs[:, ::7, ::3].numpy(),
                          text_logits_lab=tl[batch["masked_lm_labels"] != -100].numpy(),
                          ent_logits_lab=el[batch["ent_masked_lm_labels"] != -100].numpy(),
                          text_logits_s=tl[:, ::5, ::3].numpy(), ent_logits_s=el[:, ::5, ::3].numpy())
        # the special-token vectors of quirk Q2 and two gathered rows of the entity half
        for sid in (100, 102, 103):
            arrays[f"special_{sid}"] = model.kg_backbone[sid].numpy()
    for k in grad_keys:
        arrays["grad::" + k] = params[k].grad.numpy()
    total_norm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None))
    arrays["grad_norm"] = np.float32(total_norm.item())

    # G5: two optimizer steps exactly as the reference's Trainer configures them (hf:trainer.py:1780-1796)
    from transformers import get_linear_schedule_with_warmup

    train_params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(train_params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    sched = get_linear_schedule_with_warmup(opt, 0, 200)
    step_losses = []
    for _ in range(2):
        model.zero_grad()
        loss = model(**batch)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(train_params, 1.0)
        opt.step()
        sched.step()
        step_losses.append(loss.item())
    arrays["step_losses"] = np.array(step_losses, dtype=np.float32)
    for k in grad_keys:
        arrays["after2::" + k] = params[k].detach().numpy()
    meta = {"config": {k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                    "num_attention_heads", "intermediate_size",
                                                    "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
            "B": B, "weight_seed": seed, "table_seed": seed + 1, "batch_seed": seed + 2, "table_std": 0.3,
            "weights_checksum": float(sum(v.double().abs().sum() for v in sd.values())),
            "table_checksum": float(tsv_rows.abs().sum()), "dead_parameters": sorted(dead), "grad_keys": grad_keys,
            "torch": torch.__version__}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "loss", float(out.loss), "grad_norm", float(total_norm), "dead", len(dead))


def classification_case(name, sm, ft, pre, cfg: orc.OracleConfig, B, seed, num_labels, problem=None):
    """G6: STonKGsForSequenceClassification (config 5) - the reference's fine-tuning forward, loss and gradients.
    `problem`: None = integer class labels (the reference infers single_label_classification, :318-326);
    "regression_1d" = num_labels 1 with float labels of shape [B] (MSELoss broadcasts [B,1] x [B] to [B,B]: the
    reference's own behaviour), "regression" = float labels [B, num_labels], "multi_label" = 0/1 float labels [B, num_labels]."""
    from transformers import BertConfig, BertForPreTraining, BertModel

    sd = orc.init_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    gw = torch.Generator().manual_seed(seed + 3)
    sd["classifier.weight"] = (torch.randn(num_labels, cfg.hidden_size, generator=gw) * 0.02).to(torch.bfloat16).float()
    sd["classifier.bias"] = (torch.randn(num_labels, generator=gw) * 0.02).to(torch.bfloat16).float()
    hf_cfg = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings,
                        type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager",
                        num_labels=num_labels)
    hf_cfg.update({"kg_vocab_size": cfg.kg_vocab_size})
    if problem == "regression":   # (num_labels > 1 with float labels is inferred as multi-label: regression must be set)
        hf_cfg.problem_type = "regression"

    class RefCls(ft.STonKGsForSequenceClassification):  # forward is the reference's; only the hub-fetching init is not
        def __init__(self, c):
            BertForPreTraining.__init__(self, c)
            self.cls.predictions = sm.STonKGsELMPredictionHead(c)
            self.lm_backbone = BertModel(c)
            for p in self.lm_backbone.parameters():
                p.requires_grad = False
            self.lm_sep_id, self.lm_mask_id, self.lm_unk_id = 102, 103, 100
            self.num_labels = c.num_labels
            self.config = c
            self.bert = BertModel(c)
            self.dropout = torch.nn.Dropout(c.hidden_dropout_prob)
            self.classifier = torch.nn.Linear(c.hidden_size, c.num_labels)

    model = RefCls(hf_cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(("position_ids" in k or "decoder" in k) for k in missing), (missing, unexpected)
    model.eval()
    K = cfg.kg_vocab_size
    numeric_indices = [i for i in range(K + 3) if i not in (102, 103, 100)]
    model.kg_backbone = {i: torch.tensor(tsv_rows[r].numpy()) for r, i in enumerate(numeric_indices)}
    with torch.no_grad():
        for sid in (102, 103, 100):
            model.kg_backbone[sid] = model.lm_backbone(torch.tensor([[sid]]))[0][0][0]
    batch = make_batch(cfg, B, seed + 2, pre)
    rng = np.random.RandomState(seed + 4)
    if problem is None:
        labels = torch.tensor(rng.randint(0, num_labels, B))
    elif problem == "regression_1d":
        labels = torch.tensor(rng.randn(B).astype(np.float32))
    elif problem == "regression":
        labels = torch.tensor(rng.randn(B, num_labels).astype(np.float32))
    else:
        labels = torch.tensor(rng.randint(0, 2, (B, num_labels)).astype(np.float32))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                    token_type_ids=batch["token_type_ids"], labels=labels, return_dict=True)
    assert problem is None or model.config.problem_type == ("multi_label_classification" if problem == "multi_label"
                                                            else "regression")
    model.zero_grad()
    out.loss.backward()
    params = dict(model.named_parameters())
    keys = ["classifier.weight", "classifier.bias", "bert.pooler.dense.weight",
            "bert.encoder.layer.0.attention.self.query.weight", "bert.embeddings.position_embeddings.weight",
            f"bert.encoder.layer.{cfg.num_hidden_layers - 1}.output.dense.bias"]
    arrays = {k: batch[k].numpy() for k in ("input_ids", "attention_mask", "token_type_ids")}
    arrays.update(labels=labels.numpy(), loss=out.loss.detach().numpy(), logits=out.logits.detach().numpy())
    for k in keys:
        arrays["grad::" + k] = params[k].grad.numpy()
    total_norm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None))
    arrays["grad_norm"] = np.float32(total_norm.item())
    meta = {"config": {k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                    "num_attention_heads", "intermediate_size",
                                                    "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
            "B": B, "weight_seed": seed, "table_seed": seed + 1, "batch_seed": seed + 2, "classifier_seed": seed + 3,
            "num_labels": num_labels, "table_std": 0.3, "problem": problem, "problem_type": model.config.problem_type,
            "weights_checksum": float(sum(v.double().abs().sum() for k, v in sd.items() if not k.startswith("classifier"))),
            "table_checksum": float(tsv_rows.abs().sum()), "grad_keys": keys, "torch": torch.__version__}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "loss", float(out.loss), "grad_norm", float(total_norm))


def masking_case(pre):
    """G3/G4: integer-exact masking vectors and the entity index-space table."""
    res = {}
    for seed in (0, 1, 1234):
        random.seed(seed)
        t_in, t_lab = pre.replace_mlm_tokens(tokens=list(range(1000, 1256)), vocab_len=28996)
        e_in, e_lab = pre.replace_mlm_tokens(tokens=[(7 * i) % 175094 for i in range(256)], vocab_len=175094)
        res[f"text_in_{seed}"] = np.array(t_in)
        res[f"text_lab_{seed}"] = np.array(t_lab)
        res[f"ent_in_{seed}"] = np.array(e_in)
        res[f"ent_lab_{seed}"] = np.array(e_lab)
        # NSP negatives: the two random.sample draws of _add_negative_nsp_samples (ref :88-98), same order
        import pandas as pd

        random.seed(seed)
        df = pd.DataFrame({"input_ids": [list(range(i * 10, i * 10 + 8)) for i in range(8)],
                           "attention_mask": [[1] * 8] * 8, "token_type_ids": [[0] * 4 + [1] * 4] * 8,
                           "masked_lm_labels": [[-100] * 4] * 8,
                           "ent_masked_lm_labels": [[i] * 4 for i in range(8)], "next_sentence_labels": [0] * 8})
        neg = pre._add_negative_nsp_samples(df, text_part_length=4)
        res[f"neg_input_ids_{seed}"] = np.array(neg["input_ids"].tolist())
        res[f"neg_ent_labels_{seed}"] = np.array(neg["ent_masked_lm_labels"].tolist())
        res[f"neg_nsp_{seed}"] = np.array(neg["next_sentence_labels"].tolist())
    np.savez_compressed(os.path.join(OUT, "masking.npz"), **res)
    print("masking: first positions seed 1234:", [i for i, l in enumerate(res["text_lab_1234"]) if l != -100][:8])


def splits_case(ft):
    """G7: the reference's cross-validation splitter (ref:stonkgs_finetuning.py:53-89) on small label frames: plain
    5-fold, the stratified cut to max_dataset_size followed by 5-fold, and n_splits = 1. Integer indices only."""
    import pandas as pd

    rng = np.random.RandomState(7)
    res = {}
    for name, n, kw in (("plain", 53, {}), ("cut", 90, {"max_dataset_size": 40}), ("single", 31, {"n_splits": 1}),
                        ("three", 20, {"n_splits": 3, "random_seed": 7})):
        labels = rng.randint(0, 3, n)
        df = pd.DataFrame({"input_ids": [[int(i)] for i in range(n)], "labels": labels})
        out = ft.get_train_test_splits(df, **kw)
        res[f"{name}_labels"] = labels.astype(np.int64)
        res[f"{name}_n"] = np.array(len(out))
        for i, d in enumerate(out):
            res[f"{name}_train_{i}"] = np.asarray(d["train_idx"], dtype=np.int64)
            res[f"{name}_test_{i}"] = np.asarray(d["test_idx"], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g7_splits.npz"), **res)
    print("splits: plain fold 0 test head:", res["plain_test_0"][:6].tolist())


def shape_true_case(name, sm, pre, cfg: orc.OracleConfig, B, seed):
    """G2 of SURVEY section 8c: the reference's forward / backward at the REAL depth, width and head count
    (12L / 768h / 12 heads / S = 512, V = 28 996; K = 4 096 keeps the table small). Weights are regenerated from the
    seed; stored are the batch, the loss terms, sampled outputs and per-tensor gradient norms + sampled gradient slices."""
    sd = orc.init_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    model = build_reference_model(sm, cfg, sd, tsv_rows)
    batch = make_batch(cfg, B, seed + 2, pre)
    out = model(**batch, return_dict=True)
    model.zero_grad()
    out.loss.backward()
    params = dict(model.named_parameters())
    arrays = {k: v.numpy() for k, v in batch.items()}
    with torch.no_grad():
        tl, el = out.prediction_logits
        lt = torch.nn.functional.cross_entropy(tl.reshape(-1, cfg.vocab_size), batch["masked_lm_labels"].reshape(-1))
        le = torch.nn.functional.cross_entropy(el.reshape(-1, cfg.kg_vocab_size), batch["ent_masked_lm_labels"].reshape(-1))
        ln = torch.nn.functional.cross_entropy(out.seq_relationship_logits, batch["next_sentence_labels"])
        arrays.update(loss=out.loss.numpy(), masked_lm_loss=lt.numpy(), ent_masked_lm_loss=le.numpy(),
                      next_sentence_loss=ln.numpy(), nsp_logits=out.seq_relationship_logits.numpy(),
                      pooler_output=out.pooler_output.numpy(),
                      hidden_states_s=out.hidden_states[:, ::37, ::11].numpy(),
                      text_logits_lab_s=tl[batch["masked_lm_labels"] != -100][:, ::97].numpy(),
                      ent_logits_lab_s=el[batch["ent_masked_lm_labels"] != -100][:, ::29].numpy())
        for sid in (100, 102, 103):
            arrays[f"special_{sid}"] = model.kg_backbone[sid].numpy()
    names = [k for k, p in params.items() if p.grad is not None]
    arrays["grad_norms"] = np.array([params[k].grad.double().norm().item() for k in names], dtype=np.float64)
    total_norm = torch.sqrt(sum((params[k].grad.double() ** 2).sum() for k in names))
    arrays["grad_norm"] = np.float32(total_norm.item())
    slices = {"bert.encoder.layer.0.attention.self.query.weight": (slice(None, None, 16), slice(None, None, 16)),
              "bert.encoder.layer.11.intermediate.dense.weight": (slice(None, None, 64), slice(None, None, 16)),
              "bert.encoder.layer.5.output.dense.weight": (slice(None, None, 16), slice(None, None, 64)),
              "bert.encoder.layer.6.attention.output.LayerNorm.weight": (slice(None),),
              "bert.embeddings.position_embeddings.weight": (slice(None, None, 8), slice(None, None, 16)),
              "cls.predictions.text_decoder.weight": (slice(None, None, 101), slice(None, None, 16)),
              "cls.predictions.entity_decoder.weight": (slice(None, None, 17), slice(None, None, 16))}
    for k, sl in slices.items():
        arrays["grad_s::" + k] = params[k].grad[sl].numpy()
    meta = {"config": {k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                    "num_attention_heads", "intermediate_size",
                                                    "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
            "B": B, "weight_seed": seed, "table_seed": seed + 1, "batch_seed": seed + 2, "table_std": 0.3,
            "weights_checksum": float(sum(v.double().abs().sum() for v in sd.values())),
            "table_checksum": float(tsv_rows.abs().sum()), "grad_names": names,
            "grad_slices": {k: [[x.start, x.stop, x.step] for x in sl] for k, sl in slices.items()},
            "grad_keys": list(slices), "dead_parameters": sorted(k for k, p in params.items() if p.requires_grad and p.grad is None),
            "torch": torch.__version__}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "loss", float(out.loss), "grad_norm", float(total_norm))


def _grad_envelope(params_fp32, params_bf16):
    """Per-tensor deviation of one backward pass from another: relative L2 error, cosine, norm ratio."""
    names, rel, cos, ratio = [], [], [], []
    for k, g in params_fp32.items():
        a, b = params_bf16[k].double().flatten(), g.double().flatten()
        nb = float(b.norm())
        names.append(k)
        rel.append(float((a - b).norm()) / (nb if nb >= 1e-5 else 1e-2))   # (key.bias: analytically zero, absolute scale)
        cos.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)))
        ratio.append(float(a.norm()) / (nb + 1e-30))
    return names, np.array(rel), np.array(cos), np.array(ratio)


def shape_true_bf16_case(name, sm, pre, cfg: orc.OracleConfig, B, seed):
    """G16 (round 4): the REFERENCE's own reduced-precision gradients at the real shape. Same weights, table and batch as
    g3_shapetrue (same seeds); the reference's forward runs once in fp32 and once under torch.autocast(bf16) - the
    mixed precision a CPU offers; the reference trains with fp16=True (ref:stonkgs_pretraining.py:178) - and both are
    back-propagated. Stored per gradient tensor: the relative L2 error, cosine and norm ratio of the autocast gradient
    against the fp32 one - the deviation the reference's own mixed-precision backward shows, which the HIP path's
    per-tensor gradient error is held to - plus both losses and the autocast gradients on g3's sampled slices."""
    sd = orc.init_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    model = build_reference_model(sm, cfg, sd, tsv_rows)
    batch = make_batch(cfg, B, seed + 2, pre)
    grads, losses = {}, {}
    for mode in ("fp32", "bf16_autocast"):
        model.zero_grad()
        if mode == "fp32":
            out = model(**batch, return_dict=True)
        else:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = model(**batch, return_dict=True)
        out.loss.float().backward()
        losses[mode] = float(out.loss)
        grads[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    names, rel, cos, ratio = _grad_envelope(grads["fp32"], grads["bf16_autocast"])
    arrays = {"loss_fp32": np.float64(losses["fp32"]), "loss_bf16_autocast": np.float64(losses["bf16_autocast"]),
              "grad_relerr_bf16": rel, "grad_cosine_bf16": cos, "grad_norm_ratio_bf16": ratio}
    with open(os.path.join(OUT, "g3_shapetrue.json")) as f:
        g3 = json.load(f)
    assert g3["grad_names"] == names, "g16 must list g3's gradient tensors in g3's order"
    for k, spec in g3["grad_slices"].items():
        sl = tuple(slice(a, b, c) for a, b, c in spec)
        arrays["grad_s_bf16::" + k] = grads["bf16_autocast"][k][sl].float().numpy()
    meta = {"same_case_as": "g3_shapetrue", "B": B, "weight_seed": seed, "table_seed": seed + 1, "batch_seed": seed + 2,
            "grad_names": names, "autocast": "torch.autocast('cpu', dtype=torch.bfloat16) around the forward",
            "summary": {"relerr_max": float(rel.max()), "relerr_median": float(np.median(rel)),
                        "cosine_min": float(cos.min())}, "torch": torch.__version__}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "loss fp32", losses["fp32"], "bf16", losses["bf16_autocast"], "| reference bf16 gradient error per tensor: max",
          rel.max(), "median", np.median(rel), "min cosine", cos.min())


def classification_bf16_case(name, sm, ft, pre, cfg: orc.OracleConfig, B, seed, num_labels):
    """G17 (round 4): BASELINE config 5 at the real depth - the reference's STonKGsForSequenceClassification
    (ref:stonkgs_finetuning.py:237-346) at 12L / 768h / 12 heads / S 512 on a ragged batch of three, forward + backward in
    fp32 and under torch.autocast(bf16). Stored: the batch, labels, loss and logits of both runs, the fp32 gradient norm of
    every tensor, sampled fp32 gradient slices, and the per-tensor error of the reference's own reduced-precision backward."""
    from transformers import BertConfig, BertForPreTraining, BertModel

    sd = orc.init_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    gw = torch.Generator().manual_seed(seed + 3)
    sd["classifier.weight"] = (torch.randn(num_labels, cfg.hidden_size, generator=gw) * 0.02).to(torch.bfloat16).float()
    sd["classifier.bias"] = (torch.randn(num_labels, generator=gw) * 0.02).to(torch.bfloat16).float()
    hf_cfg = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings,
                        type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager",
                        num_labels=num_labels)
    hf_cfg.update({"kg_vocab_size": cfg.kg_vocab_size})

    class RefCls(ft.STonKGsForSequenceClassification):  # forward is the reference's; only the hub-fetching init is not
        def __init__(self, c):
            BertForPreTraining.__init__(self, c)
            self.cls.predictions = sm.STonKGsELMPredictionHead(c)
            self.lm_backbone = BertModel(c)
            for p in self.lm_backbone.parameters():
                p.requires_grad = False
            self.lm_sep_id, self.lm_mask_id, self.lm_unk_id = 102, 103, 100
            self.num_labels = c.num_labels
            self.config = c
            self.bert = BertModel(c)
            self.dropout = torch.nn.Dropout(c.hidden_dropout_prob)
            self.classifier = torch.nn.Linear(c.hidden_size, c.num_labels)

    model = RefCls(hf_cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(("position_ids" in k or "decoder" in k) for k in missing), (missing, unexpected)
    model.eval()
    K = cfg.kg_vocab_size
    numeric_indices = [i for i in range(K + 3) if i not in (102, 103, 100)]
    model.kg_backbone = {i: torch.tensor(tsv_rows[r].numpy()) for r, i in enumerate(numeric_indices)}
    with torch.no_grad():
        for sid in (102, 103, 100):
            model.kg_backbone[sid] = model.lm_backbone(torch.tensor([[sid]]))[0][0][0]
    batch = make_batch(cfg, B, seed + 2, pre)
    labels = torch.tensor(np.random.RandomState(seed + 4).randint(0, num_labels, B))
    inputs = {k: batch[k] for k in ("input_ids", "attention_mask", "token_type_ids")}
    grads, res = {}, {}
    for mode in ("fp32", "bf16_autocast"):
        model.zero_grad()
        if mode == "fp32":
            out = model(**inputs, labels=labels, return_dict=True)
        else:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = model(**inputs, labels=labels, return_dict=True)
        out.loss.float().backward()
        res[mode] = (float(out.loss), out.logits.detach().float().numpy())
        grads[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    names, rel, cos, ratio = _grad_envelope(grads["fp32"], grads["bf16_autocast"])
    arrays = {k: v.numpy() for k, v in inputs.items()}
    arrays.update(labels=labels.numpy(), loss_fp32=np.float64(res["fp32"][0]), logits_fp32=res["fp32"][1],
                  loss_bf16_autocast=np.float64(res["bf16_autocast"][0]), logits_bf16_autocast=res["bf16_autocast"][1],
                  grad_norms=np.array([float(grads["fp32"][k].double().norm()) for k in names]),
                  grad_relerr_bf16=rel, grad_cosine_bf16=cos, grad_norm_ratio_bf16=ratio)
    L = cfg.num_hidden_layers
    slices = {"classifier.weight": (slice(None), slice(None)),
              "bert.pooler.dense.weight": (slice(None, None, 16), slice(None, None, 16)),
              f"bert.encoder.layer.{L - 1}.output.dense.weight": (slice(None, None, 16), slice(None, None, 64)),
              f"bert.encoder.layer.{L - 1}.attention.self.query.weight": (slice(None, None, 16), slice(None, None, 16)),
              "bert.encoder.layer.0.intermediate.dense.weight": (slice(None, None, 64), slice(None, None, 16)),
              "bert.embeddings.position_embeddings.weight": (slice(None, None, 8), slice(None, None, 16))}
    for k, sl in slices.items():
        arrays["grad_s::" + k] = grads["fp32"][k][sl].numpy()
        arrays["grad_s_bf16::" + k] = grads["bf16_autocast"][k][sl].float().numpy()
    meta = {"config": {k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                    "num_attention_heads", "intermediate_size",
                                                    "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
            "B": B, "weight_seed": seed, "table_seed": seed + 1, "batch_seed": seed + 2, "classifier_seed": seed + 3,
            "label_seed": seed + 4, "num_labels": num_labels, "table_std": 0.3, "grad_names": names,
            "grad_keys": list(slices), "grad_slices": {k: [[x.start, x.stop, x.step] for x in sl] for k, sl in slices.items()},
            "weights_checksum": float(sum(v.double().abs().sum() for k, v in sd.items() if not k.startswith("classifier"))),
            "table_checksum": float(tsv_rows.abs().sum()),
            "summary": {"relerr_max": float(rel.max()), "relerr_median": float(np.median(rel)), "cosine_min": float(cos.min())},
            "torch": torch.__version__}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "loss fp32", res["fp32"][0], "bf16", res["bf16_autocast"][0], "| reference bf16 gradient error per tensor: max",
          rel.max(), "median", np.median(rel), "min cosine", cos.min())


def curve_case(name, sm, pre, cfg: orc.OracleConfig, B, seed, steps, lr, n_batches):
    """G11/G12: the REFERENCE's loss curve and the reference's own mixed-precision envelope. The reference model (its own
    forward, HF BERT, torch AdamW, clip 1.0, linear schedule: ref:src/stonkgs/models/stonkgs_pretraining.py:171-223 ->
    hf:trainer.py:1780-1796) is trained twice from the same weights over the same batches, dropout off: once in fp32 and
    once with the forward under torch.autocast (bf16 - the mixed-precision path a CPU has; the reference trains with
    fp16=True on its GPUs, :178). Stored: the batches, both loss curves. |bf16 - fp32| per step is the deviation the
    reference's own reduced-precision training shows from its fp32 run; the HIP path is held to it."""
    from transformers import get_linear_schedule_with_warmup

    sd = orc.init_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    batches = [make_batch(cfg, B, seed + 10 + i, pre) for i in range(n_batches)]
    curves = {}
    for mode in ("fp32", "bf16_autocast"):
        model = build_reference_model(sm, cfg, sd, tsv_rows)
        train_params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(train_params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        sched = get_linear_schedule_with_warmup(opt, 0, steps)
        losses = []
        for i in range(steps):
            batch = batches[i % n_batches]
            model.zero_grad()
            if mode == "fp32":
                loss = model(**batch)[0]
            else:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    loss = model(**batch)[0]
            loss.float().backward()
            torch.nn.utils.clip_grad_norm_(train_params, 1.0)
            opt.step()
            sched.step()
            losses.append(float(loss))
            if i % 10 == 0 or i == steps - 1:
                print(name, mode, "step", i, "loss", losses[-1], flush=True)
        curves[mode] = np.array(losses, dtype=np.float64)
        del model, opt
    arrays = {"loss_fp32": curves["fp32"], "loss_bf16_autocast": curves["bf16_autocast"]}
    for i, b in enumerate(batches):
        for k, v in b.items():
            arrays[f"b{i}::{k}"] = v.numpy()
    d = curves["bf16_autocast"] - curves["fp32"]
    meta = {"config": {k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                    "num_attention_heads", "intermediate_size",
                                                    "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
            "B": B, "weight_seed": seed, "table_seed": seed + 1, "batch_seeds": [seed + 10 + i for i in range(n_batches)],
            "table_std": 0.3, "steps": steps, "learning_rate": lr, "n_batches": n_batches,
            "weights_checksum": float(sum(v.double().abs().sum() for v in sd.values())),
            "table_checksum": float(tsv_rows.abs().sum()),
            "reference_bf16_vs_fp32": {"max_abs": float(np.abs(d).max()), "rms": float(np.sqrt((d ** 2).mean())),
                                       "mean": float(d.mean())},
            "torch": torch.__version__}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "fp32", curves["fp32"][0], "->", curves["fp32"][-1], "| reference bf16-autocast vs fp32: max",
          np.abs(d).max(), "rms", np.sqrt((d ** 2).mean()), "mean", d.mean())


def import_reference_prepare_df():
    """The reference's own TSV reader, ref:src/stonkgs/models/kg_baseline_model.py:270-280, from the real module (the
    stub installed for importing stonkgs_model is bypassed). Its module-level imports of the experiment-tracking and
    trainer frameworks (absent here, used by the KG-baseline classes only) are given empty stand-in modules."""
    import importlib.util

    for mod, attrs in (("mlflow", {}), ("pytorch_lightning", {"LightningModule": torch.nn.Module, "LightningDataModule": object})):
        if mod not in sys.modules:
            m = types.ModuleType(mod)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[mod] = m
    const = sys.modules["stonkgs.constants"]
    for k in ("KG_BL_OUTPUT_DIR",):
        setattr(const, k, "/nonexistent/" + k)
    spec = importlib.util.spec_from_file_location("stonkgs.models._kg_baseline_real", REF + "/models/kg_baseline_model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.prepare_df


def table_case(prepare_df_ref):
    """G8 (row f3): node2vec TSVs as the reference reads them. g8_table.tsv - 120 named nodes x 128 dims, values written
    with 17 significant digits (pins the float parser), names of the kinds the real file holds; g8_numeric.tsv - an
    all-numeric first column, which pandas turns into an integer index. Expected: key list (with type) and fp64 values."""
    rng = np.random.RandomState(11)
    names = [f"HGNC:{i * 37 + 5}" for i in range(54)] + [f"CHEBI:{i * 101 + 3}" for i in range(30)] + \
            [f"MESH:D{i * 977 + 1:06d}" for i in range(28)] + ["p(HGNC:AKT1)", "a b", "GO:0006915", "7157", "1e5",
                                                               "NaN_node", "x" * 40, "FPLX:ERK"]
    assert len(set(names)) == 120     # (the reference's index construction needs K + 3 > 103, ref:stonkgs_model.py:126-127)
    vals = rng.randn(120, 128) * 0.3
    vals[0, :4] = [1e-310, -0.0, 3.0, 1e5]   # a subnormal, a signed zero, integers written without a fraction
    path = os.path.join(OUT, "g8_table.tsv")
    with open(path, "w") as fh:
        for n, v in zip(names, vals):
            fh.write(n + "\t" + "\t".join(repr(float(x)) if float(x) != int(x) or abs(x) > 1e15 else str(int(x)) for x in v) + "\n")
    d = prepare_df_ref(path)
    npath = os.path.join(OUT, "g8_numeric.tsv")
    ids = [900, 17, 100, 102, 5]
    nvals = rng.randn(5, 4)
    with open(npath, "w") as fh:
        for i, v in zip(ids, nvals):
            fh.write(str(i) + "\t" + "\t".join(repr(float(x)) for x in v) + "\n")
    dn = prepare_df_ref(npath)
    res = {"keys": np.array([str(k) for k in d.keys()]), "key_types": np.array([type(k).__name__ for k in d.keys()]),
           "values": np.stack([np.asarray(v, dtype=np.float64) for v in d.values()]),
           "value_dtype": np.array(str(next(iter(d.values())).dtype)),
           "num_keys": np.array([int(k) for k in dn.keys()], dtype=np.int64),
           "num_key_types": np.array([type(k).__name__ for k in dn.keys()]),
           "num_values": np.stack([np.asarray(v, dtype=np.float64) for v in dn.values()])}
    np.savez_compressed(os.path.join(OUT, "g8_table.npz"), **res)
    print("table: first key", res["keys"][0], res["key_types"][0], "| numeric keys", res["num_keys"].tolist(), res["num_key_types"][0])


def checkpoint_case(sm, pre, prepare_df_ref):
    """G9 (row f3): a checkpoint directory WRITTEN BY THE REFERENCE-SIDE MODEL with HF's own `save_pretrained`
    (config.json + model.safetensors, tied / dead aliases dropped as safetensors does) for the smallest shape the HIP
    path supports with one layer, on the 120-node table of G8; plus a local LM-backbone directory written by HF
    `BertModel.save_pretrained` (what `nlp_model_type=<dir>` reads); plus the reference's outputs on one batch."""
    import shutil

    from transformers import BertConfig, BertModel

    cfg = orc.OracleConfig(vocab_size=160, kg_vocab_size=120, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                           intermediate_size=256, max_position_embeddings=256)
    seed = 400
    sd = orc.init_state_dict(cfg, seed=seed)
    d = prepare_df_ref(os.path.join(OUT, "g8_table.tsv"))
    tsv_rows = torch.from_numpy(np.stack(list(d.values())))
    model = build_reference_model(sm, cfg, sd, tsv_rows)
    # the constructor the harness bypasses ends in post_init(), which ties cls.predictions.decoder.weight to the word
    # embeddings (tie_word_embeddings, hf BertForPreTraining._tied_weights_keys): a checkpoint of the reference has them equal
    model.tie_weights()
    assert model.cls.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight
    batch = make_batch(cfg, 3, seed + 2, pre)
    out = model(**batch, return_dict=True)
    ck = os.path.join(OUT, "g9_ref_checkpoint")
    shutil.rmtree(ck, ignore_errors=True)
    kg = model.kg_backbone
    # The writer: transformers 5.x refuses to serialise this model at all (its safetensors path finds the reference's
    # undeclared aliases decoder.text_bias / decoder.entity_bias and raises), so the directory is written the way the
    # transformers the reference pins (>= 4.6.1; the published stonkgs checkpoints are config.json + pytorch_model.bin,
    # ref:src/stonkgs/api/api.py:96-101) wrote it: config.save_pretrained + torch.save(model.state_dict()).
    os.makedirs(ck)
    model.config.save_pretrained(ck)
    torch.save(model.state_dict(), os.path.join(ck, "pytorch_model.bin"))
    # ... and the key set a safetensors writer keeps for the same model (safetensors' own shared-tensor handling)
    import tempfile

    from safetensors import safe_open
    from safetensors.torch import save_model

    with tempfile.TemporaryDirectory() as tmp:
        save_model(model, os.path.join(tmp, "model.safetensors"))
        with safe_open(os.path.join(tmp, "model.safetensors"), "pt") as f:
            keys = sorted(f.keys())
    bb = os.path.join(OUT, "g9_lm_backbone")
    shutil.rmtree(bb, ignore_errors=True)
    model.lm_backbone.save_pretrained(bb)   # HF's own writer for a plain BertModel: config.json + model.safetensors
    arrays = {k: v.numpy() for k, v in batch.items()}
    with torch.no_grad():
        arrays.update(loss=out.loss.numpy(), pooler_output=out.pooler_output.numpy(),
                      nsp_logits=out.seq_relationship_logits.numpy(),
                      ent_logits_lab=out.prediction_logits[1][batch["ent_masked_lm_labels"] != -100].numpy(),
                      table_row_7=kg[7].numpy(), table_row_101=kg[101].numpy(), table_row_104=kg[104].numpy(),
                      special_102=kg[102].numpy())
    np.savez_compressed(os.path.join(OUT, "g9_ref_checkpoint.npz"), **arrays)
    with open(os.path.join(OUT, "g9_ref_checkpoint.json"), "w") as f:
        json.dump({"weight_seed": seed, "safetensors_keys": keys, "state_dict_keys": sorted(model.state_dict().keys()),
                   "files": sorted(os.listdir(ck)),
                   "backbone_files": sorted(os.listdir(bb)), "idx_to_name_head": {str(i): str(model_name) for i, model_name in
                                                                                   list(zip([i for i in range(123) if i not in (100, 102, 103)], d.keys()))[98:104]},
                   "config": {k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                           "num_attention_heads", "intermediate_size",
                                                           "max_position_embeddings", "type_vocab_size", "layer_norm_eps")}},
                  f, indent=1, sort_keys=True)
    print("checkpoint:", len(keys), "tensors;", sorted(os.listdir(ck)), "loss", float(out.loss))


def unk_walk_case(pre, prepare_df_ref):
    """G10 (row f1, ref:src/stonkgs/models/stonkgs_for_embeddings.py:50-155): the reference's own row builder for embedding
    extraction - tokenised evidence | source walk [SEP] target walk [SEP], a walk of [UNK] ids for a node the pre-trained
    KG does not know, both halves masked by replace_mlm_tokens - run on a small local vocabulary file, the G8 node table
    and a random-walk TSV. Fixture: the three input files + the integer rows it yields under random.seed(5)."""
    pre.prepare_df = prepare_df_ref      # (the import harness stubbed it; the embeddings module binds it at import)
    emb = importlib.import_module("stonkgs.models.stonkgs_for_embeddings")
    # transformers 5.x dropped the `encode_plus` spelling the reference (written against 4.x) calls; in 4.x it is what
    # calling the tokenizer does for a single text, so the name is pointed at `__call__` for this run
    from transformers import BertTokenizer

    if not hasattr(BertTokenizer, "encode_plus"):
        BertTokenizer.encode_plus = BertTokenizer.__call__
    d = prepare_df_ref(os.path.join(OUT, "g8_table.tsv"))
    names = [str(k) for k in d.keys()]
    rng = np.random.RandomState(21)
    walks_path = os.path.join(OUT, "g10_walks.tsv")
    with open(walks_path, "w") as fh:
        for n in names[:100]:                     # the last 20 nodes have an embedding but no stored walk
            fh.write(n + "\t" + "\t".join(names[i] for i in rng.randint(0, len(names), 127)) + "\n")
    words = ["the", "of", "and", "in", "protein", "kinase", "binds", "phosphorylates", "inhibits", "activates", "cell",
             "expression", "receptor", "complex", "growth", "factor", "signal", "pathway", "increases", "decreases", "by",
             "to", "a", "is", "with", "##s", "##ed", "##ing", "##ation", "tumor", "akt", "##1", "mtor", "p53", "mdm", "##2",
             ".", ",", "(", ")", "-"]
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words
    # a local tokenizer directory (vocab.txt): the reference's `nlp_model_type` branch, BertTokenizer.from_pretrained(dir) -
    # its default branch, BertTokenizerFast(vocab_file=...), builds an empty vocabulary under transformers 5.x
    tok_dir = os.path.join(OUT, "g10_tokenizer")
    os.makedirs(tok_dir, exist_ok=True)
    with open(os.path.join(tok_dir, "vocab.txt"), "w") as fh:
        fh.write("\n".join(vocab) + "\n")
    rows = [(names[3], names[57], "AKT1 phosphorylates MDM2 and inhibits p53 expression in tumor cells."),
            (names[99], "not-a-node", "The receptor complex activates the mTOR signaling pathway."),
            ("unknown:1", names[110], "Growth factor binds to a kinase, (increasing) expression - unseenword."),
            (names[0], names[1], "")]
    random.seed(5)
    out = list(emb.preprocess_df_for_embeddings_iter(rows, embedding_name_to_vector_path=os.path.join(OUT, "g8_table.tsv"),
                                                     embedding_name_to_random_walk_path=walks_path,
                                                     nlp_model_type=tok_dir))
    res = {k: np.array([r[k] for r in out], dtype=np.int64) for k in out[0]}
    res["sources"] = np.array([r[0] for r in rows])
    res["targets"] = np.array([r[1] for r in rows])
    res["evidences"] = np.array([r[2] for r in rows])
    np.savez_compressed(os.path.join(OUT, "g10_embedding_rows.npz"), **res)
    print("embedding rows:", res["input_ids"].shape, "unk ids in row 1:", int((res["input_ids"][1, 256:] == 100).sum()),
          "text ids row 0 head:", res["input_ids"][0, :12].tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "splits":     # only (re)generate G7
        import_reference()                                 # (installs the stub modules the fine-tuning module needs too)
        splits_case(import_reference_finetuning())
        return
    torch.manual_seed(0)
    torch.set_num_threads(4)
    sm, pre = import_reference()
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only in ("curve", "curve_small"):   # round 3: the reference's loss curves, fp32 and bf16 autocast (minutes of CPU)
        torch.set_num_threads(8)
        if only == "curve":
            curve_case("g11_curve_shapetrue", sm, pre, orc.OracleConfig(kg_vocab_size=4096), B=2, seed=700, steps=60,
                       lr=1e-4, n_batches=6)
        else:
            curve_case("g12_curve_small", sm, pre,
                       orc.OracleConfig(vocab_size=512, kg_vocab_size=300, hidden_size=128, num_hidden_layers=2,
                                        num_attention_heads=2, intermediate_size=256, max_position_embeddings=256),
                       B=4, seed=800, steps=200, lr=1e-3, n_batches=6)
        return
    if only == "cls_losses":   # round 3: the regression / multi-label branches of the fine-tuning head
        ft = import_reference_finetuning()
        small = orc.OracleConfig(vocab_size=512, kg_vocab_size=300, hidden_size=128, num_hidden_layers=2,
                                 num_attention_heads=2, intermediate_size=256, max_position_embeddings=256)
        classification_case("g13_cls_regression_1d", sm, ft, pre, small, B=5, seed=310, num_labels=1, problem="regression_1d")
        classification_case("g14_cls_regression", sm, ft, pre, small, B=4, seed=320, num_labels=3, problem="regression")
        classification_case("g15_cls_multilabel", sm, ft, pre, small, B=6, seed=330, num_labels=3, problem="multi_label")
        return
    if only == "bf16_grads":   # round 4: the reference's own reduced-precision gradients at the real shape (minutes of CPU)
        torch.set_num_threads(8)
        shape_true_bf16_case("g16_shapetrue_bf16", sm, pre, orc.OracleConfig(kg_vocab_size=4096), B=2, seed=500)
        classification_bf16_case("g17_cls_shapetrue", sm, import_reference_finetuning(), pre,
                                 orc.OracleConfig(kg_vocab_size=1000), B=3, seed=900, num_labels=2)
        return
    if only == "cls":   # only g6 (its metadata keys were added after the first write)
        classification_case("g6_classification", sm, import_reference_finetuning(), pre,
                            orc.OracleConfig(vocab_size=512, kg_vocab_size=300, hidden_size=128, num_hidden_layers=2,
                                             num_attention_heads=2, intermediate_size=256, max_position_embeddings=256),
                            B=5, seed=300, num_labels=3)
        return
    if only in ("shape", "f3"):   # (re)generate only the round-2 cases
        if only == "shape":
            torch.set_num_threads(8)
            shape_true_case("g3_shapetrue", sm, pre, orc.OracleConfig(kg_vocab_size=4096), B=2, seed=500)
        else:
            pdf = import_reference_prepare_df()
            table_case(pdf)
            checkpoint_case(sm, pre, pdf)
            unk_walk_case(pre, pdf)
        return
    # G1: tiny, every tensor stored (pins the oracle op for op)
    model_case("g1_tiny", sm, pre, orc.OracleConfig(vocab_size=300, kg_vocab_size=150, hidden_size=64, num_hidden_layers=2,
                                                   num_attention_heads=4, intermediate_size=128,
                                                   max_position_embeddings=16), B=3, seed=100, full_logits=True)
    # G2: smallest shape the HIP path supports (head_dim 64, S = 256 = 128 + 128): pins GPU parity directly
    model_case("g2_hipsmall", sm, pre, orc.OracleConfig(vocab_size=512, kg_vocab_size=300, hidden_size=128,
                                                       num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                                       max_position_embeddings=256), B=3, seed=200, full_logits=False)
    masking_case(pre)
    # G6: fine-tuning head (BASELINE config 5) on the HIP-supported small shape, 3 relation classes, ragged batch of 5
    ft = import_reference_finetuning()
    classification_case("g6_classification", sm, ft, pre,
                        orc.OracleConfig(vocab_size=512, kg_vocab_size=300, hidden_size=128, num_hidden_layers=2,
                                         num_attention_heads=2, intermediate_size=256, max_position_embeddings=256),
                        B=5, seed=300, num_labels=3)
    splits_case(ft)
    # round 2: the shape-true case (12L / 768h / 12 heads / S = 512), TSV ingestion, a reference-written checkpoint
    torch.set_num_threads(8)
    shape_true_case("g3_shapetrue", sm, pre, orc.OracleConfig(kg_vocab_size=4096), B=2, seed=500)
    pdf = import_reference_prepare_df()
    table_case(pdf)
    checkpoint_case(sm, pre, pdf)
    unk_walk_case(pre, pdf)


if __name__ == "__main__":
    main()
