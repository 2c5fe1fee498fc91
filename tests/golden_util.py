"""Helpers shared by the parity tests: load a golden case and rebuild its weights / table from the seeds."""
import json
import os

import numpy as np
import torch

from oracle import stonkgs_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_KEYS = ("input_ids", "attention_mask", "token_type_ids", "masked_lm_labels", "ent_masked_lm_labels",
              "next_sentence_labels")


def load_case(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    cfg = orc.OracleConfig(**meta["config"])
    sd = orc.init_state_dict(cfg, seed=meta["weight_seed"])
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - meta["weights_checksum"]) <= 1e-9 * abs(chk), "torch CPU generator drifted: regenerate fixtures"
    g = torch.Generator().manual_seed(meta["table_seed"])
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * meta["table_std"]
    assert abs(float(tsv_rows.abs().sum()) - meta["table_checksum"]) <= 1e-9 * meta["table_checksum"]
    batch = {k: torch.from_numpy(arrays[k]) for k in BATCH_KEYS}
    return cfg, sd, tsv_rows, batch, arrays, meta


def load_curve_case(name):
    """A loss-curve case of oracle/make_golden.py `curve`: (cfg, weights, table rows, batches, reference fp32 curve,
    reference bf16-autocast curve, meta)."""
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    cfg = orc.OracleConfig(**meta["config"])
    sd = orc.init_state_dict(cfg, seed=meta["weight_seed"])
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - meta["weights_checksum"]) <= 1e-9 * abs(chk), "torch CPU generator drifted: regenerate fixtures"
    g = torch.Generator().manual_seed(meta["table_seed"])
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * meta["table_std"]
    assert abs(float(tsv_rows.abs().sum()) - meta["table_checksum"]) <= 1e-9 * meta["table_checksum"]
    batches = [{k: torch.from_numpy(arrays[f"b{i}::{k}"]) for k in BATCH_KEYS} for i in range(meta["n_batches"])]
    return cfg, sd, tsv_rows, batches, arrays["loss_fp32"], arrays["loss_bf16_autocast"], meta
