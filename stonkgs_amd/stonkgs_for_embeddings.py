"""Batched embedding extraction and inference on the HIP path.

Mirror of the two bs = 1 Python loops that call the hot path from outside training:

* ``get_stonkgs_embeddings`` (ref:src/stonkgs/models/stonkgs_for_embeddings.py:158-186): for every row of a
  pre-processed DataFrame, ``model(**row, return_dict=True).pooler_output[0]`` collected into a DataFrame with one
  ``embedding`` column of Python lists;
* ``infer`` / ``infer_iter`` (ref:src/stonkgs/api/api.py:308-336): for every pre-processed row,
  ``softmax(model(input_ids, attention_mask, token_type_ids).logits, dim=1)[0]``.

Same names, arguments and return layout; what changes is that rows are stacked into batches (``batch_size``, ragged last
batch allowed) and that the embedding path runs the encoder only (``STonKGsForPreTraining.encode``): the reference's loop
also evaluates both vocabulary-wide decoders and three cross-entropies per row and discards them.

``preprocess_df_for_embeddings`` / ``preprocess_df_for_embeddings_iter`` (ref:stonkgs_for_embeddings.py:26-155) turn
(source, target, evidence) triples into those rows: tokenised evidence | source walk [SEP] target walk [SEP], a walk of
[UNK] ids for a node the pre-trained KG does not know, both halves masked by ``replace_mlm_tokens`` - host-side integer
work, bit-exact with the reference's own function for the same ``random`` state (tests/golden/g10_embedding_rows.npz).
Everything is read from LOCAL files (the reference's defaults are hub / Zenodo downloads)."""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

from .stonkgs_model import STonKGsForPreTraining, STonKGsForSequenceClassification, SequenceClassifierOutput

_COLUMNS = ("input_ids", "attention_mask", "token_type_ids")


def _local_tokenizer(vocab_file_path=None, nlp_model_type=None):
    """BERT WordPiece tokenizer from a local ``vocab.txt`` (file, or directory holding one). The reference builds
    ``BertTokenizerFast(vocab_file=...)`` or ``BertTokenizer.from_pretrained(name)`` (ref:stonkgs_for_embeddings.py:91-97);
    a hub name cannot be resolved offline."""
    import os
    import shutil
    import tempfile

    from transformers import BertTokenizer

    if nlp_model_type is not None:
        if not os.path.isdir(nlp_model_type):
            raise FileNotFoundError(f"{nlp_model_type!r}: pass a local tokenizer directory (vocab.txt), hub names cannot be fetched")
        return BertTokenizer.from_pretrained(nlp_model_type)
    if vocab_file_path is None:
        raise ValueError("pass vocab_file_path (a BERT vocab.txt) or nlp_model_type (a local tokenizer directory)")
    # (`BertTokenizerFast(vocab_file=...)` as the reference spells it builds an EMPTY vocabulary under transformers 5.x:
    # the file is handed over as a one-file tokenizer directory instead, which every version reads)
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(str(vocab_file_path), os.path.join(tmp, "vocab.txt"))
        return BertTokenizer.from_pretrained(tmp)


def preprocess_df_for_embeddings_iter(rows: Iterable[Tuple[str, str, str]], *, embedding_name_to_vector_path=None,
                                      embedding_name_to_random_walk_path=None, vocab_file_path=None, nlp_model_type=None,
                                      sep_id: Optional[int] = None, unk_id: Optional[int] = None, tokenizer=None):
    """ref:src/stonkgs/models/stonkgs_for_embeddings.py:50-155, same names and defaults (sep 102, unk 100) except that
    every path must be given (local files). ``tokenizer``: an already built HF tokenizer (optional)."""
    from .data import replace_mlm_tokens
    from .stonkgs_model import prepare_df

    if embedding_name_to_vector_path is None or embedding_name_to_random_walk_path is None:
        raise ValueError("embedding_name_to_vector_path and embedding_name_to_random_walk_path must be local TSV files")
    sep_id = 102 if sep_id is None else sep_id
    unk_id = 100 if unk_id is None else unk_id
    # node name -> its row in the embedding TSV (quirk Q1's "preprocessing space"), and every node's walk as such rows
    row_of_node = {name: row for row, name in enumerate(prepare_df(embedding_name_to_vector_path))}
    n_nodes = len(row_of_node)
    walks = {node: [row_of_node[step] for step in walk]
             for node, walk in prepare_df(embedding_name_to_random_walk_path).items()}
    walk_len = len(next(iter(walks.values())))
    half = 2 * walk_len + 2                      # walk [SEP] walk [SEP]
    unknown = [unk_id] * walk_len                # a node without a walk: a walk of [UNK]s (ref :124-135)
    tok = tokenizer if tokenizer is not None else _local_tokenizer(vocab_file_path, nlp_model_type)
    n_tokens = len(tok.vocab)
    segments = [0] * half + [1] * half
    for source, target, evidence in rows:
        enc = tok(evidence, padding="max_length", truncation=True, max_length=half)
        entity_half = walks.get(source, unknown) + [sep_id] + walks.get(target, unknown) + [sep_id]
        # (the two masking calls in the reference's order: text first - they share Python's random stream)
        text_ids, text_labels = replace_mlm_tokens(tokens=list(enc["input_ids"]), vocab_len=n_tokens)
        entity_ids, entity_labels = replace_mlm_tokens(tokens=entity_half, vocab_len=n_nodes)
        yield {"input_ids": text_ids + entity_ids, "attention_mask": list(enc["attention_mask"]) + [1] * half,
               "token_type_ids": list(segments), "masked_lm_labels": text_labels, "ent_masked_lm_labels": entity_labels,
               "next_sentence_labels": 0}


def preprocess_df_for_embeddings(df, **kwargs):
    """ref:stonkgs_for_embeddings.py:26-47: DataFrame with source / target / evidence columns -> DataFrame of model rows."""
    import pandas as pd

    return pd.DataFrame(preprocess_df_for_embeddings_iter(rows=df[["source", "target", "evidence"]].values, **kwargs))


def _rows_of(data, indices: Optional[Sequence[int]]) -> List[dict]:
    """Accept a pandas DataFrame (the reference's input), a list of dicts, or a dict of equally long columns."""
    if hasattr(data, "iloc"):
        idx = range(len(data)) if indices is None else indices
        return [dict(data.iloc[i]) for i in idx]
    if isinstance(data, dict):
        n = len(data["input_ids"])
        idx = range(n) if indices is None else indices
        return [{k: v[i] for k, v in data.items()} for i in idx]
    rows = list(data)
    return rows if indices is None else [rows[i] for i in indices]


def _batches(rows: List[dict], batch_size: int) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]]:
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    for lo in range(0, len(rows), batch_size):
        chunk = rows[lo:lo + batch_size]
        cols = []
        for c in _COLUMNS:
            if all(c in r and r[c] is not None for r in chunk):
                cols.append(torch.as_tensor([list(map(int, r[c])) for r in chunk], dtype=torch.long))
            elif c == "input_ids":
                raise KeyError("every row needs input_ids")
            else:
                cols.append(None)
        yield tuple(cols)


def get_stonkgs_embeddings(preprocessed_df, pretrained_stonkgs_model_name: Optional[str] = None,
                           list_of_indices: Optional[List] = None, *, model: Optional[STonKGsForPreTraining] = None,
                           batch_size: int = 64):
    """ref:stonkgs_for_embeddings.py:158-186. Returns a DataFrame with one ``embedding`` column (lists of H floats),
    one row per entry of ``list_of_indices`` (default: all rows, in order). ``model`` lets a caller reuse a loaded model
    (the reference re-loads it on every call)."""
    import pandas as pd

    if model is None:
        model = (STonKGsForPreTraining.from_pretrained(pretrained_stonkgs_model_name) if pretrained_stonkgs_model_name
                 else STonKGsForPreTraining.from_default_pretrained())
    rows = _rows_of(preprocessed_df, list_of_indices)
    out: List[List[float]] = []
    for ids, am, tt in _batches(rows, batch_size):
        _, pooled = model.encode(ids, am, tt, pooled_only=True)
        out.extend(pooled.cpu().tolist())
    return pd.DataFrame({"embedding": out}, columns=["embedding"])


def infer_iter(model: STonKGsForSequenceClassification, data, batch_size: int = 64
               ) -> Iterable[Tuple[SequenceClassifierOutput, List[float]]]:
    """ref:api.py:318-336. Yields, per row, (prediction output with ``logits`` of shape [1, num_labels], class
    probabilities as a list) - the reference's per-row objects, computed a batch at a time."""
    was_training = model.training
    model.eval()
    try:
        for ids, am, tt in _batches(_rows_of(data, None), batch_size):
            with torch.no_grad():
                logits = model(input_ids=ids, attention_mask=am, token_type_ids=tt, return_dict=True).logits
            probs = torch.nn.functional.softmax(logits.float(), dim=1)
            for i in range(logits.shape[0]):
                yield (SequenceClassifierOutput(loss=None, logits=logits[i:i + 1].clone(), hidden_states=None,
                                                attentions=None), probs[i].tolist())
    finally:
        model.train(was_training)


def infer(model: STonKGsForSequenceClassification, data, batch_size: int = 64):
    """ref:api.py:308-315: (raw prediction outputs, probabilities), two lists with one entry per row."""
    raw_results, probabilities = [], []
    for r, p in infer_iter(model, data, batch_size):
        raw_results.append(r)
        probabilities.append(p)
    return raw_results, probabilities
