"""Parameter storage for the HIP engine: every tensor of the model lives in ONE flat fp32 buffer per group
(trainable / frozen backbone), laid out for the hardware rather than for autograd:

* trainable group in BACKWARD-COMPLETION order (entity decoder first, embeddings last) so that gradient
  buckets for the RCCL all-reduce become ready front to back while backward is still running;
* a flat gradient buffer with the same offsets (``param.grad`` of every nn.Parameter is a view into it), which
  is what the fused clip+AdamW kernel and the all-reduce consume - one launch / a few large collectives
  instead of ~200 per-tensor ones;
* a bf16 mirror with the same offsets (refreshed by the AdamW kernel itself) that the MFMA GEMMs read, and
  bf16 W^T copies for the dgrad GEMMs;
* decoder weights are padded to a multiple of 128 rows so the vocab dimension tiles exactly
  (28996 -> 29056, 175094 -> 175104); the pad rows stay zero forever (zero grad -> zero AdamW update).

The nn.Module tree built on top only NAMES these views with the HuggingFace/STonKGs state-dict keys
 so that ``state_dict`` / ``load_state_dict`` / ``from_pretrained`` round-trip with the
reference's checkpoints, including its dead parameters (quirk Q4).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .config import STonKGsConfig

ALIGN = 256  # elements; keeps every slot 1 KiB aligned in fp32 and 512 B in bf16


def pad128(n: int) -> int:
    return (n + 127) // 128 * 128


class FlatStore:
    def __init__(self, specs: List[Tuple[str, Tuple[int, ...], Optional[int]]], device, trainable: bool):
        """specs: (name, logical shape, padded row count or None)."""
        self.index: Dict[str, Tuple[int, Tuple[int, ...], Tuple[int, ...]]] = {}
        off = 0
        for name, shape, pad_rows in specs:
            pshape = tuple(shape) if pad_rows is None else (pad_rows,) + tuple(shape[1:])
            n = 1
            for d in pshape:
                n *= d
            self.index[name] = (off, tuple(shape), pshape)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.device = device
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.bf16 = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device) if trainable else None
        self.wt: Dict[str, torch.Tensor] = {}  # name -> bf16 W^T [in, out_padded]

    def _view(self, buf, name, padded=False):
        off, shape, pshape = self.index[name]
        s = pshape if padded else shape
        n = 1
        for d in s:
            n *= d
        return buf[off:off + n].view(s)

    def view(self, name, padded=False):
        return self._view(self.data, name, padded)

    def grad_view(self, name, padded=False):
        return self._view(self.grad, name, padded)

    def bf16_view(self, name, padded=True):
        return self._view(self.bf16, name, padded)

    def span(self, name) -> Tuple[int, int]:
        off, _, pshape = self.index[name]
        n = 1
        for d in pshape:
            n *= d
        return off, off + (n + ALIGN - 1) // ALIGN * ALIGN


def _layer_specs(prefix: str, H: int, I: int):
    p = prefix
    return [
        (f"{p}.attention.self.qkv.weight", (3 * H, H), None),
        (f"{p}.attention.self.qkv.bias", (3 * H,), None),
        (f"{p}.attention.output.dense.weight", (H, H), None),
        (f"{p}.attention.output.dense.bias", (H,), None),
        (f"{p}.attention.output.LayerNorm.weight", (H,), None),
        (f"{p}.attention.output.LayerNorm.bias", (H,), None),
        (f"{p}.intermediate.dense.weight", (I, H), None),
        (f"{p}.intermediate.dense.bias", (I,), None),
        (f"{p}.output.dense.weight", (H, I), None),
        (f"{p}.output.dense.bias", (H,), None),
        (f"{p}.output.LayerNorm.weight", (H,), None),
        (f"{p}.output.LayerNorm.bias", (H,), None),
    ]


def pretraining_head_specs(cfg: STonKGsConfig):
    H = cfg.hidden_size
    return [
        ("cls.predictions.entity_decoder.weight", (cfg.kg_vocab_size, H), pad128(cfg.kg_vocab_size)),
        ("cls.predictions.text_decoder.weight", (cfg.vocab_size, H), pad128(cfg.vocab_size)),
        ("cls.predictions.transform.dense.weight", (H, H), None),
        ("cls.predictions.transform.dense.bias", (H,), None),
        ("cls.predictions.transform.LayerNorm.weight", (H,), None),
        ("cls.predictions.transform.LayerNorm.bias", (H,), None),
        ("cls.seq_relationship.weight", (2, H), None),
        ("cls.seq_relationship.bias", (2,), None),
    ]


def trainable_specs(cfg: STonKGsConfig, with_pretraining_heads: bool = True):
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    specs = (pretraining_head_specs(cfg) if with_pretraining_heads else []) + [
        ("bert.pooler.dense.weight", (H, H), None),
        ("bert.pooler.dense.bias", (H,), None),
    ]
    for i in reversed(range(L)):
        specs += _layer_specs(f"bert.encoder.layer.{i}", H, I)
    specs += [
        ("bert.embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H), None),
        ("bert.embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H), None),
        ("bert.embeddings.LayerNorm.weight", (H,), None),
        ("bert.embeddings.LayerNorm.bias", (H,), None),
    ]
    return specs


def backbone_specs(cfg: STonKGsConfig, n_layers: int):
    H, I = cfg.hidden_size, cfg.intermediate_size
    specs = [
        ("lm_backbone.embeddings.word_embeddings.weight", (cfg.vocab_size, H), None),
        ("lm_backbone.embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H), None),
        ("lm_backbone.embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H), None),
        ("lm_backbone.embeddings.LayerNorm.weight", (H,), None),
        ("lm_backbone.embeddings.LayerNorm.bias", (H,), None),
    ]
    for i in range(n_layers):
        specs += _layer_specs(f"lm_backbone.encoder.layer.{i}", H, I)
    specs += [("lm_backbone.pooler.dense.weight", (H, H), None), ("lm_backbone.pooler.dense.bias", (H,), None)]
    return specs


class _Node(nn.Module):
    """Naming-only container: the arithmetic is done by the HIP engine, never by module.forward."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("stonkgs_amd modules only name parameters; call the model's forward")


def _param(store: FlatStore, name: str, trainable: bool, rows: Optional[slice] = None) -> nn.Parameter:
    v = store.view(name)
    g = store.grad_view(name) if trainable else None
    if rows is not None:
        v = v[rows]
        g = g[rows] if g is not None else None
    p = nn.Parameter(v, requires_grad=trainable)
    if g is not None:
        p.grad = g
    return p


def _linear(store, prefix, trainable, bias=True):
    m = _Node()
    m.weight = _param(store, prefix + ".weight", trainable)
    if bias:
        m.bias = _param(store, prefix + ".bias", trainable)
    return m


def _bert_layer(store, prefix, H, trainable):
    layer = _Node()
    att = _Node()
    slf = _Node()
    for j, n in enumerate(("query", "key", "value")):
        lin = _Node()
        lin.weight = _param(store, prefix + ".attention.self.qkv.weight", trainable, slice(j * H, (j + 1) * H))
        lin.bias = _param(store, prefix + ".attention.self.qkv.bias", trainable, slice(j * H, (j + 1) * H))
        setattr(slf, n, lin)
    att.self = slf
    out = _Node()
    out.dense = _linear(store, prefix + ".attention.output.dense", trainable)
    out.LayerNorm = _linear(store, prefix + ".attention.output.LayerNorm", trainable)
    att.output = out
    layer.attention = att
    inter = _Node()
    inter.dense = _linear(store, prefix + ".intermediate.dense", trainable)
    layer.intermediate = inter
    o = _Node()
    o.dense = _linear(store, prefix + ".output.dense", trainable)
    o.LayerNorm = _linear(store, prefix + ".output.LayerNorm", trainable)
    layer.output = o
    return layer


def build_bert_tree(store: FlatStore, prefix: str, cfg: STonKGsConfig, n_layers: int, trainable: bool,
                    word_embeddings: Optional[nn.Parameter]) -> _Node:
    H = cfg.hidden_size
    bert = _Node()
    emb = _Node()
    we = _Node()
    we.weight = word_embeddings if word_embeddings is not None else _param(
        store, prefix + ".embeddings.word_embeddings.weight", trainable)
    emb.word_embeddings = we
    pe = _Node()
    pe.weight = _param(store, prefix + ".embeddings.position_embeddings.weight", trainable)
    emb.position_embeddings = pe
    te = _Node()
    te.weight = _param(store, prefix + ".embeddings.token_type_embeddings.weight", trainable)
    emb.token_type_embeddings = te
    emb.LayerNorm = _linear(store, prefix + ".embeddings.LayerNorm", trainable)
    bert.embeddings = emb
    enc = _Node()
    enc.layer = nn.ModuleList([_bert_layer(store, f"{prefix}.encoder.layer.{i}", H, trainable) for i in range(n_layers)])
    bert.encoder = enc
    pool = _Node()
    pool.dense = _linear(store, prefix + ".pooler.dense", trainable)
    bert.pooler = pool
    return bert
