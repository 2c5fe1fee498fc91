"""Cross-validated fine-tuning around ``STonKGsForSequenceClassification``.

Mirror of the driver pieces of ref:src/stonkgs/models/stonkgs_finetuning.py that surround the hot path:

* ``get_train_test_splits`` (:53-89): deterministic 5-fold split (``KFold(shuffle=True, random_state=42)``) after an
  optional stratified, deterministic cut to ``max_dataset_size`` rows - the reference delegates both to scikit-learn,
  and so does this (same calls, same arguments; pinned against reference-made vectors in tests/golden/g7_splits.npz);
* ``INDRADataset`` (:92-110): the encodings + labels container;
* ``run_sequence_classification_cv`` (:403-484): per fold a FRESH model from the pre-trained weights, training with the
  Trainer's defaults of the reference run (batch 8, lr 5e-5, linear decay), prediction, arg-max, weighted F1; returns the
  per-fold scores and the predicted-labels frame the reference writes out.

Not mirrored: the mlflow logging, DeepSpeed switch and TSV/model dumps to pystow directories (control plane), and
``preprocess_fine_tuning_data`` (needs the tokenizer / embedding names / walks, unavailable offline) - rows arrive
pre-processed, as for the embedding helpers."""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .stonkgs_model import STonKGsForSequenceClassification
from .stonkgs_pretraining import Trainer, TrainingArguments

_COLUMNS = ("input_ids", "attention_mask", "token_type_ids")


def get_train_test_splits(train_data, type_column_name: str = "labels", random_seed: int = 42, n_splits: int = 5,
                          max_dataset_size: int = 100000) -> List[Dict[str, np.ndarray]]:
    """ref:stonkgs_finetuning.py:53-89. ``train_data``: DataFrame (or dict of columns) holding the label column.
    Returns ``[{"train_idx": ..., "test_idx": ...}, ...]``; ``n_splits == 1`` returns the first of five folds.
    Indices refer to the (possibly cut) data, exactly as in the reference."""
    from sklearn.model_selection import KFold, StratifiedShuffleSplit

    labels = np.asarray(train_data[type_column_name])
    n = len(labels)
    index = np.arange(n)
    if n > max_dataset_size:
        splitter = StratifiedShuffleSplit(n_splits=1, train_size=max_dataset_size, random_state=random_seed)
        for keep, _ in splitter.split(index.reshape(-1, 1), labels):
            index, labels = index[keep], labels[keep]
    skf = KFold(n_splits=5 if n_splits == 1 else n_splits, random_state=random_seed, shuffle=True)
    result = [{"train_idx": tr, "test_idx": te} for tr, te in skf.split(index.reshape(-1, 1), labels)]
    return [result[0]] if n_splits == 1 else result


class INDRADataset(torch.utils.data.Dataset):
    """ref:stonkgs_finetuning.py:92-110: dict of equally long columns + labels -> per-item dicts of tensors."""

    def __init__(self, encodings: Dict[str, Sequence], labels: Sequence[int]):
        self.encodings = encodings
        self.labels = list(labels)

    def __getitem__(self, idx):
        item = {k: torch.tensor(v[idx]) for k, v in self.encodings.items() if k in _COLUMNS}
        item["labels"] = torch.tensor(self.labels[idx])
        return item

    def __len__(self):
        return len(self.labels)


def weighted_f1_score(y_true: Sequence[int], y_pred: Sequence[int]) -> float:
    """``sklearn.metrics.f1_score(average="weighted")`` (ref:stonkgs_finetuning.py:463): per-class F1 weighted by the
    class's support among the true labels; a class that is never predicted scores 0."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    total, score = len(y_true), 0.0
    for c in np.unique(np.concatenate([y_true, y_pred])):
        tp = float(np.sum((y_true == c) & (y_pred == c)))
        fp = float(np.sum((y_true != c) & (y_pred == c)))
        fn = float(np.sum((y_true == c) & (y_pred != c)))
        f1 = 0.0 if tp == 0 else 2 * tp / (2 * tp + fp + fn)
        score += f1 * float(np.sum(y_true == c)) / total
    return score


def _collate(items: List[dict]) -> Dict[str, torch.Tensor]:
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


def predict_logits(model: STonKGsForSequenceClassification, dataset: INDRADataset, batch_size: int = 8) -> np.ndarray:
    """``trainer.predict(test_dataset).predictions`` (:446): eval-mode logits, a batch at a time (ragged last batch)."""
    was_training = model.training
    model.eval()
    out = []
    try:
        with torch.no_grad():
            for lo in range(0, len(dataset), batch_size):
                b = _collate([dataset[i] for i in range(lo, min(lo + batch_size, len(dataset)))])
                out.append(model(input_ids=b["input_ids"], attention_mask=b.get("attention_mask"),
                                 token_type_ids=b.get("token_type_ids"), return_dict=True).logits.float().cpu())
    finally:
        model.train(was_training)
    return torch.cat(out).numpy() if out else np.zeros((0, model.num_labels), dtype=np.float32)


def run_sequence_classification_cv(fine_tuning_data, labels: Optional[Sequence[int]] = None, *,
                                   model_factory: Callable[[int], STonKGsForSequenceClassification],
                                   class_column_name: str = "labels", epochs: int = 3, lr: float = 5e-5,
                                   batch_size: int = 8, gradient_accumulation: int = 1, n_splits: int = 5,
                                   random_seed: int = 42, max_dataset_size: int = 100000, seed: int = 42):
    """ref:stonkgs_finetuning.py:403-484. ``fine_tuning_data``: DataFrame / dict of the pre-processed columns
    (``input_ids``, ``attention_mask``, ``token_type_ids``) and, unless ``labels`` is given, the integer label column.
    ``model_factory(num_labels)`` must return a FRESH model from the pre-trained weights (the reference calls
    ``from_pretrained(model_path, num_labels=...)`` once per fold). Returns ``(f1_scores, result_df)``."""
    import pandas as pd

    cols = {k: list(fine_tuning_data[k]) for k in _COLUMNS if k in fine_tuning_data}
    y = np.asarray(labels if labels is not None else fine_tuning_data[class_column_name]).astype(np.int64)
    num_labels = int(len(np.unique(y)))
    splits = get_train_test_splits({"labels": y}, "labels", random_seed, n_splits, max_dataset_size)
    f1_scores, frames = [], []
    for idx, ind in enumerate(splits):
        model = model_factory(num_labels)
        tr_idx, te_idx = ind["train_idx"], ind["test_idx"]
        train_ds = INDRADataset({k: [v[i] for i in tr_idx] for k, v in cols.items()}, y[tr_idx].tolist())
        test_ds = INDRADataset({k: [v[i] for i in te_idx] for k, v in cols.items()}, y[te_idx].tolist())
        steps_per_epoch = max(1, math.ceil(len(train_ds) / (batch_size * gradient_accumulation)))
        args = TrainingArguments(learning_rate=lr, max_steps=epochs * steps_per_epoch,
                                 per_device_train_batch_size=batch_size, gradient_accumulation_steps=gradient_accumulation,
                                 seed=seed, logging_steps=max(1, steps_per_epoch))
        model.train()
        g = torch.Generator().manual_seed(seed + idx)
        trainer = Trainer(model, args)
        for _ in range(epochs):                       # RandomSampler order per epoch; the last batch may be ragged
            perm = torch.randperm(len(train_ds), generator=g).tolist()
            for lo in range(0, len(perm), batch_size):
                trainer.training_step(model, _collate([train_ds[i] for i in perm[lo:lo + batch_size]]))
        model.engine.check_errors()
        predicted = np.argmax(predict_logits(model, test_ds, batch_size), axis=1)
        f1_scores.append(weighted_f1_score(y[te_idx], predicted))
        frames.append(pd.DataFrame({"split": idx, "index": te_idx.tolist(), "predicted_label": predicted.tolist(),
                                    "true_label": y[te_idx].tolist()}))
    return f1_scores, pd.concat(frames, ignore_index=True)
