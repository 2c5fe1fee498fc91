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

Out of scope here, as in SURVEY section 8: turning (source, target, evidence) triples into ``input_ids`` needs the
BioBERT tokenizer, the node2vec table names and the random-walk file (``preprocess_df_for_embeddings``), none of which
can be fetched offline - these helpers take rows that already have ``input_ids`` / ``attention_mask`` /
``token_type_ids``."""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

from .stonkgs_model import STonKGsForPreTraining, STonKGsForSequenceClassification, SequenceClassifierOutput

_COLUMNS = ("input_ids", "attention_mask", "token_type_ids")


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
        _, pooled = model.encode(ids, am, tt)
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
