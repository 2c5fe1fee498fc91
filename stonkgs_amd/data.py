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
    """One positive pre-training row (ref:indra_for_pretraining.py:190-239): the text ids are padded to `half` FIRST
    (the tokenizer's padding="max_length") and the padded sequence is masked, so all `half` positions - padding
    included - are candidates and every row carries int(half * 0.15) = 38 text labels, as in the reference; entity
    half = masked(source walk + [SEP] + target walk + [SEP]); NSP label 0."""
    n = len(text_ids)
    assert n <= half and len(walk_source) + len(walk_target) + 2 == half
    pad = half - n
    ent = list(walk_source) + [SEP_ID] + list(walk_target) + [SEP_ID]
    t_in, t_lab = replace_mlm_tokens(list(text_ids) + [PAD_ID] * pad, vocab_size)
    e_in, e_lab = replace_mlm_tokens(ent, kg_vocab_size)
    return {"input_ids": t_in + e_in, "attention_mask": [1] * n + [0] * pad + [1] * half,
            "token_type_ids": [0] * half + [1] * half, "masked_lm_labels": t_lab,
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


class DeviceBatcher:
    """Per-step batch assembly and dynamic masking ON the device.

    The reference tokenises, assembles and masks every row once, offline (ref:indra_for_pretraining.py:190-239, :33-77),
    appends 25 % negative NSP rows (:80-126) and pickles the result (13.8 M rows); training then replays the same masks
    every epoch. Here the static part stops at the tokenised text ids and the (source, target) node pair of each
    statement; a step's ``input_ids`` / labels / NSP pairing are produced by two small kernels (csrc/data.hip) from a
    counter-based random stream keyed by (seed, step), so every step sees fresh masks and fresh negatives. Same schema
    and semantics as the reference's rows; the stream itself is restated bit for bit in oracle/masking_oracle.py.

    ``walks``: int64 [n_nodes, half/2 - 1] - the random-walk node ids of every KG node (the reference's
    ``random_walk_idx_dict``, a dict of lists, as one table)."""

    def __init__(self, walks: torch.Tensor, vocab_size: int, kg_vocab_size: int, half: int = 256,
                 nsp_negative_rate: float = 0.2, masked_tokens_percentage: float = 0.15, seed: int = 0,
                 device="cuda:0"):
        from . import _hip as hip

        hip.lib()  # fail loudly without the extension / a GPU
        self.hip = hip
        self.device = torch.device(device)
        self.walks = torch.as_tensor(walks).to(device=self.device, dtype=torch.long).contiguous()
        if self.walks.dim() != 2 or 2 * self.walks.shape[1] + 2 != half:
            raise ValueError(f"walks must be [n_nodes, {half // 2 - 1}]")
        self.vocab_size, self.kg_vocab_size, self.half = int(vocab_size), int(kg_vocab_size), int(half)
        self.k = int(half * masked_tokens_percentage)          # the reference's int(len * 0.15)
        self.negative_rate, self.seed = float(nsp_negative_rate), int(seed)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)

    def step_seed(self, step: int) -> int:
        return (self.seed * 1000003 + int(step) * 7919 + 12345) & 0xFFFFFFFF

    def __call__(self, text_ids, text_attention, source, target, step: int) -> Dict[str, torch.Tensor]:
        hip, dev, half = self.hip, self.device, self.half

        def prep(t):
            return torch.as_tensor(t).to(device=dev, dtype=torch.long).contiguous()

        text_ids, text_attention, source, target = prep(text_ids), prep(text_attention), prep(source), prep(target)
        B = text_ids.shape[0]
        if text_ids.shape != (B, half) or text_attention.shape != (B, half) or source.shape != (B,) or target.shape != (B,):
            raise ValueError("text_ids / text_attention must be [B, half]; source / target [B]")
        S = 2 * half
        raw = torch.empty(B, S, dtype=torch.long, device=dev)
        out = {"input_ids": torch.empty(B, S, dtype=torch.long, device=dev),
               "attention_mask": torch.empty(B, S, dtype=torch.long, device=dev),
               "token_type_ids": torch.empty(B, S, dtype=torch.long, device=dev),
               "masked_lm_labels": torch.empty(B, half, dtype=torch.long, device=dev),
               "ent_masked_lm_labels": torch.empty(B, half, dtype=torch.long, device=dev),
               "next_sentence_labels": torch.empty(B, dtype=torch.long, device=dev)}
        seed = self.step_seed(step)
        st = hip.stream_ptr()
        hip.call("stonk_assemble_rows", text_ids.data_ptr(), text_attention.data_ptr(), source.data_ptr(),
                 target.data_ptr(), self.walks.data_ptr(), self.walks.shape[0], self.walks.shape[1], raw.data_ptr(),
                 out["attention_mask"].data_ptr(), out["token_type_ids"].data_ptr(),
                 out["next_sentence_labels"].data_ptr(), B, S, half, SEP_ID, self.negative_rate, seed,
                 self.err.data_ptr(), st)
        hip.call("stonk_mlm_mask", raw.data_ptr(), out["input_ids"].data_ptr(), out["masked_lm_labels"].data_ptr(),
                 out["ent_masked_lm_labels"].data_ptr(), B, S, half, self.vocab_size, self.kg_vocab_size, MASK_ID, self.k,
                 self.k, seed ^ 0x5BD1E995, st)
        return out

    def check_errors(self) -> None:
        """One tiny D2H copy: raise KeyError if a (source, target) pair named a node without a walk."""
        if int(self.err.item()):
            self.err.zero_()
            raise KeyError("a source / target index is outside the random-walk table "
                           "(the reference raises KeyError on random_walk_idx_dict[...])")
