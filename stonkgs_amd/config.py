"""Local model configuration (the fields of HuggingFace ``BertConfig`` the STonKGs hot path reads, plus
``kg_vocab_size``).  The reference takes every hyper-parameter from the BioBERT hub config and only adds
``kg_vocab_size`` (ref:src/stonkgs/models/stonkgs_model.py:96-97); here the config is a local object / JSON file
(``config.json`` in HF layout) because nothing can be fetched."""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass


@dataclass
class STonKGsConfig:
    vocab_size: int = 28996               # BioBERT v1.1 (ref:src/stonkgs/models/protstonkgs_model.py:112)
    kg_vocab_size: int = 175094           # number of KG nodes (ref:notebooks/kg_component_check.ipynb:107)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    hidden_act: str = "gelu"
    initializer_range: float = 0.02
    model_type: str = "bert"
    num_labels: int = 2                   # fine-tuning head only (ref:stonkgs_finetuning.py:247)
    problem_type: str = None              # None -> inferred from num_labels / label dtype, as the reference does
    # The reference hands `self.bert(...).attentions` through (ref:stonkgs_model.py:256): None unless the HF config says
    # output_attentions. Here the attention probabilities never exist in memory (flash-style kernels keep a row's
    # running maximum and sum), so a config that asks for them is REFUSED rather than answered with None.
    output_attentions: bool = False

    @property
    def half_length(self) -> int:  # ref:stonkgs_model.py:52
        return self.max_position_embeddings // 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def validate_for_hip(self) -> None:
        """Shapes the gfx950 kernels are built for; anything else is refused loudly (no CPU fallback)."""
        if self.head_dim != 64 or self.hidden_size % self.num_attention_heads:
            raise ValueError(f"head_dim must be 64 (got hidden {self.hidden_size} / heads {self.num_attention_heads})")
        if self.hidden_size % 128 or self.intermediate_size % 128:
            raise ValueError("hidden_size and intermediate_size must be multiples of 128")
        if self.max_position_embeddings % 256:
            raise ValueError("max_position_embeddings must be a multiple of 256 (two halves of 128-row tiles)")
        if self.hidden_act != "gelu":
            raise ValueError("only the exact erf GELU of BERT is implemented")
        if self.type_vocab_size != 2:
            raise ValueError("type_vocab_size must be 2")
        if self.output_attentions:
            raise NotImplementedError(
                "output_attentions=True: the gfx950 attention kernels never materialise the [B, heads, S, S] probabilities "
                "(online softmax), so `attentions` cannot be returned; run the reference's CPU path for attention maps")

    def update(self, d: dict) -> None:
        for k, v in d.items():
            setattr(self, k, v)

    def to_dict(self) -> dict:
        return asdict(self)

    @classmethod
    def from_any(cls, obj) -> "STonKGsConfig":
        """Accept an STonKGsConfig, a dict, or any object with BertConfig-like attributes."""
        if isinstance(obj, cls):
            return obj
        src = obj if isinstance(obj, dict) else {k: getattr(obj, k) for k in cls.__dataclass_fields__ if hasattr(obj, k)}
        return cls(**{k: v for k, v in src.items() if k in cls.__dataclass_fields__})

    @classmethod
    def from_pretrained(cls, path: str) -> "STonKGsConfig":
        """Local directory (or config.json path) only: a hub NAME cannot be resolved offline."""
        f = os.path.join(path, "config.json") if os.path.isdir(path) else path
        if not os.path.exists(f):
            raise FileNotFoundError(
                f"{path!r} is not a local model directory; hub names such as 'dmis-lab/biobert-v1.1' cannot be fetched "
                "here - pass a directory holding config.json (and weights)")
        with open(f) as fh:
            return cls.from_any(json.load(fh))

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(self.to_dict(), fh, indent=2, sort_keys=True)
