"""Host-side batch schema of the STonKGs pre-training step: the masking algorithm and NSP negatives of
ref:src/stonkgs/data/indra_for_pretraining.py (:33-77 replace_mlm_tokens, :80-126 _add_negative_nsp_samples,
:190-239 row assembly), plus the synthetic text-triple batches BASELINE.json's configs are defined on
(no tokenizer vocabulary, INDRA corpus or node2vec table exists offline).

Integer work: bit-exact with the reference for the same ``random`` state (tests/test_data.py pins it against
vectors produced by the reference itself)."""
from __future__ import annotations

import random
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

CLS_ID, SEP_ID, MASK_ID, UNK_ID, PAD_ID = 101, 102, 103, 100, 0


def replace_mlm_tokens(tokens: Sequence[int], vocab_len: int, mask_id: int = MASK_ID,
                       masked_tokens_percentage: float = 0.15, unmasked_label_id: int = -100):
    """15 % of the positions are chosen with random.sample; each becomes [MASK] (80 %), stays (10 %) or becomes a
    uniform random id in [0, vocab_len-1] (10 %); labels hold the original id there and -100 elsewhere.
    Same draws in the same order as ref:indra_for_pretraining.py:33-77."""
    mlm_input_tokens = list(tokens)
    mlm_labels = [unmasked_label_id] * len(mlm_input_tokens)
    candidate_pred_positions = random.sample(range(len(mlm_input_tokens)),
                                             int(len(mlm_input_tokens) * masked_tokens_percentage))
    for pos in candidate_pred_positions:
        if random.random() < 0.8:
            masked_token = mask_id
        elif random.random() < 0.5:
            masked_token = tokens[pos]
        else:
            masked_token = random.randint(0, vocab_len - 1)
        mlm_input_tokens[pos] = masked_token
        mlm_labels[pos] = tokens[pos]
    return mlm_input_tokens, mlm_labels


def add_negative_nsp_samples(rows: List[Dict[str, list]], nsp_negative_proportion: float = 0.25,
                             text_part_length: int = 256) -> List[Dict[str, list]]:
    """ref:indra_for_pretraining.py:80-126 on a list of row dicts: text half (+ its labels, mask, type ids) of row i
    joined with the entity half (+ its labels) of row j, NSP label 1."""
    k = int(len(rows) * nsp_negative_proportion)
    idx = random.sample(range(len(rows)), k)
    partner = random.sample(range(len(rows)), k)
    out = []
    for i, j in zip(idx, partner):
        t, e = rows[i], rows[j]
        out.append({"input_ids": t["input_ids"][:text_part_length] + e["input_ids"][text_part_length:],
                    "attention_mask": t["attention_mask"], "token_type_ids": t["token_type_ids"],
                    "masked_lm_labels": t["masked_lm_labels"], "ent_masked_lm_labels": e["ent_masked_lm_labels"],
                    "next_sentence_labels": 1})
    return out


def assemble_row(text_ids: Sequence[int], walk_source: Sequence[int], walk_target: Sequence[int], vocab_size: int,
                 kg_vocab_size: int, half: int = 256) -> Dict[str, list]:
    """One positive pre-training row (ref:indra_for_pretraining.py:190-239): masked text half padded to `half`,
    entity half = masked(source walk + [SEP] + target walk + [SEP]); NSP label 0."""
    n = len(text_ids)
    assert n <= half and len(walk_source) + len(walk_target) + 2 == half
    ent = list(walk_source) + [SEP_ID] + list(walk_target) + [SEP_ID]
    t_in, t_lab = replace_mlm_tokens(list(text_ids), vocab_size)
    e_in, e_lab = replace_mlm_tokens(ent, kg_vocab_size)
    pad = half - n
    return {"input_ids": t_in + [PAD_ID] * pad + e_in, "attention_mask": [1] * n + [0] * pad + [1] * half,
            "token_type_ids": [0] * half + [1] * half, "masked_lm_labels": t_lab + [-100] * pad,
            "ent_masked_lm_labels": e_lab, "next_sentence_labels": 0}


def collate(rows: List[Dict[str, list]]) -> Dict[str, torch.Tensor]:
    keys = ("input_ids", "attention_mask", "token_type_ids", "masked_lm_labels", "ent_masked_lm_labels",
            "next_sentence_labels")
    return {k: torch.tensor([r[k] for r in rows], dtype=torch.long) for k in keys}


def synthetic_batch(batch_size: int, vocab_size: int, kg_vocab_size: int, seq_len: int = 512, seed: int = 1234,
                    min_text: int = 32, nsp_positive_rate: float = 0.8) -> Dict[str, torch.Tensor]:
    """BASELINE config-2 batch: text ids uniform in the vocabulary with real length uniform in
    [min_text, half], entity half = two uniform walks of half/2-1 ids each closed by [SEP], masking by the
    reference's algorithm under random.seed(seed), NSP label 1 with probability 0.2 (= 25 % appended negatives)."""
    half = seq_len // 2
    rng = np.random.RandomState(seed)
    random.seed(seed)
    rows = []
    for _ in range(batch_size):
        n = int(rng.randint(min_text, half + 1))
        text = [CLS_ID] + [int(x) for x in rng.randint(min(1000, vocab_size // 2), vocab_size, n - 2)] + [SEP_ID]
        w = half // 2 - 1
        rows.append(assemble_row(text, [int(x) for x in rng.randint(0, kg_vocab_size, w)],
                                 [int(x) for x in rng.randint(0, kg_vocab_size, w)], vocab_size, kg_vocab_size, half))
    for r in rows:
        r["next_sentence_labels"] = int(rng.rand() >= nsp_positive_rate)
    return collate(rows)


def example_batch(vocab_size: int = 28996, kg_vocab_size: int = 1000, seq_len: int = 512, seed: int = 0):
    """BASELINE config 1: a 3-row batch shaped like the README's example_df (ref:README.md:115-132) with fixed
    synthetic ids (11/13/12 real text tokens)."""
    half = seq_len // 2
    rng = np.random.RandomState(seed)
    random.seed(seed)
    rows = []
    for n in (11, 13, 12):
        text = [CLS_ID] + [int(x) for x in rng.randint(min(1000, vocab_size // 2), vocab_size, n)] + [SEP_ID]
        w = half // 2 - 1
        rows.append(assemble_row(text, [int(x) for x in rng.randint(0, kg_vocab_size, w)],
                                 [int(x) for x in rng.randint(0, kg_vocab_size, w)], vocab_size, kg_vocab_size, half))
    return collate(rows)
