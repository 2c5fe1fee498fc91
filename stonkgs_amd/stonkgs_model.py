"""MI355X-native ``STonKGsForPreTraining``: the reference's module contract on top of the HIP engine.

Mirror of ref:src/stonkgs/models/stonkgs_model.py (same class names, constructor arguments, ``forward`` signature,
return packing, attributes other code touches, state-dict keys), so that ``stonkgs.models.stonkgs_pretraining``'s
``Trainer(model=..., ...)`` loop, ``from_pretrained`` and the bs=1 inference helpers work unchanged - but no
arithmetic runs in torch: ``forward`` hands raw device pointers to ``libstonk_hip.so`` (see ``engine.py``).

Differences a user of the reference will notice (all forced by the offline, GPU-only setting):
  * ``nlp_model_type`` / ``from_pretrained`` take LOCAL directories, never hub names;
  * the model is built directly on the GPU (``.to(device)`` to the same device is a no-op);
  * in training mode the dense logits ([B,256,V] and [B,256,K] = 13 GB at B=64) are not materialised unless
    asked for (``model.materialize_logits = True``): loss and gradients are identical.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from functools import lru_cache
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _hip as hip
from .config import STonKGsConfig
from .engine import Engine
from .params import (FlatStore, _Node, _linear, _param, backbone_specs, build_bert_tree, pretraining_head_specs,
                     trainable_specs)

SEP_ID, MASK_ID, UNK_ID = 102, 103, 100  # BioBERT vocabulary (ref:stonkgs_model.py:116-118)
# state-dict keys a checkpoint may lack (see from_pretrained)
_DROPPED_ALIASES = ("position_ids", "cls.predictions.decoder.", "cls.predictions.bias", "cls.predictions.text_bias",
                    "cls.predictions.entity_bias", "bert.embeddings.word_embeddings.weight")
_FRESH_HEAD_PREFIXES = ("classifier.",)
_TERM_KEYS = ("masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss")


@dataclass
class BertForPreTrainingOutputWithPooling:
    """ref:stonkgs_model.py:30-34 (HF BertForPreTrainingOutput + pooler_output). Indexable like a ModelOutput."""

    loss: Optional[torch.Tensor] = None
    prediction_logits: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
    seq_relationship_logits: Optional[torch.Tensor] = None
    hidden_states: Optional[torch.Tensor] = None
    attentions: Optional[tuple] = None
    pooler_output: Optional[torch.Tensor] = None

    def to_tuple(self):
        return tuple(v for v in (self.loss, self.prediction_logits, self.seq_relationship_logits, self.hidden_states,
                                 self.attentions, self.pooler_output) if v is not None)

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else self.to_tuple()[k]


def prepare_df(embedding_path: str, sep: str = "\t") -> Dict[object, "object"]:
    """ref:src/stonkgs/models/kg_baseline_model.py:270-280: TSV (first column = node name, no header) ->
    {name: float64 vector}, in file order. Read with pandas like the reference, so that the parsed values (its C
    float parser) and the key types (an all-numeric name column becomes an integer index) are the reference's:
    pinned by tests/golden/g8_table.* made with the reference's own function."""
    import pandas as pd

    df = pd.read_csv(embedding_path, sep=sep, header=None, index_col=0)
    return dict(zip(df.index.tolist(), df.to_numpy()))   # (= {index: row.values for index, row in df.iterrows()})


class KGBackbone:
    """Dict-like view of the dense entity table: ``kg_backbone[i]`` is row i (a device tensor), KeyError outside
    the table - what ref:stonkgs_model.py:131-141 builds as a Python dict of 175 097 tensors."""

    def __init__(self, table: torch.Tensor):
        self.table = table

    def __getitem__(self, i: int) -> torch.Tensor:
        if not 0 <= int(i) < self.table.shape[0]:
            raise KeyError(i)
        return self.table[int(i)]

    def __len__(self):
        return self.table.shape[0]

    def __contains__(self, i):
        return 0 <= int(i) < self.table.shape[0]

    def keys(self):
        return range(self.table.shape[0])


class STonKGsELMPredictionHead(_Node):
    """Naming mirror of ref:stonkgs_model.py:37-73 (transform, split text/entity decoders, dead biases)."""

    def __init__(self, config: STonKGsConfig, store: FlatStore, device, trainable: bool = True):
        super().__init__()
        H = config.hidden_size
        tr = _Node()
        tr.dense = _linear(store, "cls.predictions.transform.dense", trainable)
        tr.LayerNorm = _linear(store, "cls.predictions.transform.LayerNorm", trainable)
        self.transform = tr
        self.text_decoder = _linear(store, "cls.predictions.text_decoder", trainable, bias=False)
        self.entity_decoder = _linear(store, "cls.predictions.entity_decoder", trainable, bias=False)
        self.half_length = config.max_position_embeddings // 2
        # dead parameters (quirk Q4): declared, serialised, never used by forward, never updated
        self.bias = nn.Parameter(torch.zeros(config.vocab_size, device=device), requires_grad=False)
        self.text_bias = nn.Parameter(torch.zeros(config.vocab_size, device=device), requires_grad=False)
        self.entity_bias = nn.Parameter(torch.zeros(config.kg_vocab_size, device=device), requires_grad=False)
        dec = _Node()
        dec.bias = self.bias
        dec.text_bias = self.text_bias
        dec.entity_bias = self.entity_bias
        self.decoder = dec


class _StepFunction(torch.autograd.Function):
    """Makes ``loss.backward()`` (HF Trainer / accelerate) drive the engine's hand-written backward."""

    @staticmethod
    def forward(ctx, anchor, model, loss):
        ctx.model = model
        return loss.clone()

    @staticmethod
    def backward(ctx, dloss):
        model = ctx.model
        model._prepare_grads_for_autograd()
        model.engine.backward(float(dloss), model._segment_hook)
        model._reattach_grads()
        model._external_step_pending = True   # whoever called loss.backward() steps the masters with its own optimizer
        return None, None, None


class STonKGsForPreTraining(nn.Module):
    _TRAIN_PRETRAINING_HEADS = True

    def __init__(self, config=None, nlp_model_type: Optional[str] = None, kg_embedding_dict_path: Optional[str] = None,
                 *, kg_embeddings: Optional[torch.Tensor] = None, kg_names=None, device=None, seed: int = 0,
                 backbone_layers: Optional[int] = None):
        """:param config: STonKGsConfig / dict / BertConfig-like (the reference ignores its ``config`` and reads the hub;
            here it is honoured unless ``nlp_model_type`` is a local directory holding config.json)
        :param nlp_model_type: LOCAL directory of the LM backbone (config.json + pytorch_model.bin|model.safetensors)
        :param kg_embedding_dict_path: TSV of node2vec embeddings (prepare_df format)
        :param kg_embeddings: alternative to the TSV: [K, H] tensor in TSV row order (synthetic tables)"""
        super().__init__()
        if not torch.cuda.is_available():
            raise hip.StonkHipError("STonKGsForPreTraining needs an MI355X: the hot path has no CPU fallback")
        self._device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if nlp_model_type is not None and os.path.isdir(nlp_model_type):
            cfg = STonKGsConfig.from_pretrained(nlp_model_type)
            if config is not None:  # keep run-time knobs (dropout) of an explicit config
                c2 = STonKGsConfig.from_any(config)
                cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob = c2.hidden_dropout_prob, c2.attention_probs_dropout_prob
        elif config is not None:
            cfg = STonKGsConfig.from_any(config)
        else:
            raise ValueError("pass a local config, or nlp_model_type = local directory with config.json "
                             "(hub names cannot be fetched offline)")
        names = None
        if kg_embedding_dict_path is not None:
            d = prepare_df(kg_embedding_dict_path)  # ref:stonkgs_model.py:93
            names = list(d.keys())
            import numpy as np

            kg_embeddings = torch.from_numpy(np.stack(list(d.values())))
        if kg_embeddings is not None:
            cfg.update({"kg_vocab_size": int(kg_embeddings.shape[0])})  # ref:stonkgs_model.py:97
            if kg_embeddings.shape[1] != cfg.hidden_size:
                raise ValueError("KG embedding width must equal hidden_size")
        self.config = cfg
        cfg.validate_for_hip()
        dev = self._device
        n_bb = cfg.num_hidden_layers if backbone_layers is None else backbone_layers
        self._store = FlatStore(self._head_specs(cfg) + trainable_specs(cfg, self._TRAIN_PRETRAINING_HEADS), dev,
                                trainable=True)
        # the fine-tuning subclass inherits the MLM/ELM/NSP heads (they stay in its checkpoints) but never uses them:
        # there they live in a frozen side store - no gradient buffer, no optimizer state, no all-reduce bytes
        heads = self._store if self._TRAIN_PRETRAINING_HEADS else FlatStore(pretraining_head_specs(cfg), dev, False)
        self._heads_store = heads
        self._bb_store = FlatStore(backbone_specs(cfg, n_bb), dev, trainable=False)
        # ---- module tree (names only)
        dead_we = nn.Parameter(torch.zeros(cfg.vocab_size, cfg.hidden_size, device=dev), requires_grad=False)
        self.bert = build_bert_tree(self._store, "bert", cfg, cfg.num_hidden_layers, True, dead_we)
        cls = _Node()
        cls.predictions = STonKGsELMPredictionHead(cfg, heads, dev, self._TRAIN_PRETRAINING_HEADS)
        cls.predictions.decoder.weight = dead_we  # tied: cls.predictions.decoder.weight <-> word_embeddings (both dead)
        cls.seq_relationship = _linear(heads, "cls.seq_relationship", self._TRAIN_PRETRAINING_HEADS)
        self.cls = cls
        self.lm_backbone = build_bert_tree(self._bb_store, "lm_backbone", cfg, n_bb, False, None)
        self.lm_sep_id, self.lm_mask_id, self.lm_unk_id = SEP_ID, MASK_ID, UNK_ID
        self.engine = Engine(cfg, self._store, self._bb_store, n_bb, dev)
        self.materialize_logits: Optional[bool] = None  # None = "only when not training"
        self._segment_hook = None
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        # ---- init weights, KG table
        self._init_weights(seed)
        if nlp_model_type is not None and os.path.isdir(nlp_model_type):
            sd = _load_weights_file(nlp_model_type)
            if sd is not None:
                sd = {("lm_backbone." + k[len("bert."):] if k.startswith("bert.") else "lm_backbone." + k): v
                      for k, v in sd.items() if not k.startswith("cls.")}
                self.load_state_dict(sd, strict=False, _refresh=False)
        K = cfg.kg_vocab_size
        if kg_embeddings is None:
            g = torch.Generator().manual_seed(seed + 1)
            kg_embeddings = torch.randn(K, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
        self._kg_rows = kg_embeddings.to(torch.float32).to(dev)  # fp64 -> fp32 RN, as ref:stonkgs_model.py:193-200
        numeric_indices = [i for i in range(K + 3) if i not in (SEP_ID, MASK_ID, UNK_ID)]
        names = names if names is not None else [f"node{r}" for r in range(K)] if K <= 4096 else None
        self.kg_idx_to_name = dict(zip(numeric_indices, names)) if names is not None else _LazyNames(numeric_indices)
        self._numeric_indices = torch.tensor(numeric_indices, device=dev)
        self._grad_views = {n: p.grad for n, p in self.named_parameters() if p.requires_grad}
        self.refresh()

    # -------------------------------------------------------------- construction helpers
    @staticmethod
    def _head_specs(cfg):
        """Extra trainable tensors a subclass puts in FRONT of the flat buffer (their gradients are final first)."""
        return []

    def _init_weights(self, seed: int) -> None:
        """BERT init (hf _init_weights: N(0, initializer_range) matrices/embeddings, zero bias, unit LayerNorm)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        std = self.config.initializer_range
        with torch.no_grad():
            for store in {id(x): x for x in (self._store, self._bb_store, self._heads_store)}.values():
                for name, (off, shape, _) in store.index.items():
                    v = store.view(name)
                    if "LayerNorm.weight" in name:
                        v.fill_(1.0)
                    elif name.endswith(".bias") or "LayerNorm.bias" in name:
                        v.zero_()
                    else:
                        v.copy_((torch.randn(shape, generator=g) * std).to(v.device))

    def refresh(self) -> None:
        """Recompute everything derived from the fp32 masters: bf16 mirrors, W^T copies, the entity table with its
        three LM special-token rows (quirks Q1/Q2). Called after init / load_state_dict."""
        eng = self.engine
        eng.refresh_derived(bf16_mirror=True)
        K, H = self.config.kg_vocab_size, self.config.hidden_size
        table = torch.zeros(K + 3, H, dtype=torch.float32, device=self._device)
        table[self._numeric_indices] = self._kg_rows  # TSV row r -> model index numeric_indices[r]  (Q1)
        eng.kg_table = table
        sv = eng.special_vectors()  # (Q2)
        for sid, vec in sv.items():
            if sid < K + 3:
                table[sid] = vec
        self.kg_backbone = KGBackbone(table)
        self._mark_synced()

    # -------------------------------------------------------------- masters <-> derived copies
    # The GEMMs read bf16 mirrors and bf16 W^T copies of the fp32 master weights; the fused optimizer refreshes them itself.
    # Anything ELSE that writes the masters is noticed here as far as torch lets it be noticed:
    #  * in-place writes THROUGH the parameters (a torch optimizer's step, `p.copy_` / `p.add_` under no_grad,
    #    `load_state_dict`) bump the version counter the parameter views share with their flat buffer;
    #  * an autograd-driven backward sets `_external_step_pending`: whoever called loss.backward() is about to step the
    #    masters, possibly through `p.data` - which has a version counter of its OWN and is invisible above. The flag stays
    #    set until a version change is seen (the ordinary optimizer) or `refresh()` is called: while it is set, every forward
    #    re-derives the copies (0.5 ms), so a `.data`-writing optimizer is never read stale;
    #  * a write through `p.data` that no backward preceded (weight surgery before eval / encode) CANNOT be seen:
    #    call `model.refresh()` after it.
    def _mark_synced(self, clear_pending: bool = True) -> None:
        self._synced_versions = (self._store.data._version, self._bb_store.data._version,
                                 self._heads_store.data._version)
        if clear_pending:
            self._external_step_pending = False

    def _sync_derived(self) -> None:
        """Called at the top of every forward: bring the bf16 mirrors / W^T copies (and, if the frozen backbone was
        written, the entity table's LM special rows) up to date with externally modified master weights."""
        v = (self._store.data._version, self._bb_store.data._version, self._heads_store.data._version)
        if v == self._synced_versions and not self._external_step_pending:
            return
        self._wait_params()
        changed = v != self._synced_versions
        if v[1] != self._synced_versions[1]:
            self.refresh()              # backbone changed: special vectors of the entity table too (quirk Q2)
        else:
            self.engine.refresh_derived(bf16_mirror=True)
            self._mark_synced(clear_pending=changed)   # (no version change yet: the external step may still come via .data)

    def zero_grad(self, set_to_none: bool = True) -> None:
        """nn.Module.zero_grad on the flat gradient buffer: the engine's backward ACCUMULATES into it (+= / atomics), so
        dropping the `.grad` attributes alone would leave last step's gradients in place. The buffer is zeroed for both
        values of `set_to_none`; with True the attributes are dropped as torch does and re-attached by the next backward."""
        self._wait_params()
        self._store.grad.zero_()
        for p in self.parameters():
            if p.requires_grad:
                p.grad = None if set_to_none else p.grad
        if not set_to_none:
            self._reattach_grads(force=True)

    def _prepare_grads_for_autograd(self) -> None:
        """torch semantics for `loss.backward()`: a parameter whose `.grad` is None starts from zero (an optimizer's
        zero_grad(set_to_none=True) only drops the attributes), one that still has its view accumulates."""
        params = [(n, p) for n, p in nn.Module.named_parameters(self) if p.requires_grad]
        dropped = [n for n, p in params if p.grad is None]
        if not dropped:
            return
        if len(dropped) == len(params):
            self._store.grad.zero_()
        else:
            for n in dropped:
                self._grad_views[n].zero_()

    # -------------------------------------------------------------- nn.Module plumbing
    @property
    def device(self) -> torch.device:
        return self._device

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self._device))
        if probe.device != self._device or probe.dtype != torch.float32:
            raise NotImplementedError("construct STonKGsForPreTraining on its target GPU; parameters are views of flat "
                                      "HBM buffers and cannot be moved or cast")
        return self

    # Accessors order the caller's stream after an optimizer step still running on the optimizer stream
    # (Engine.wait_params): whatever they hand out is final for work enqueued on the current stream.
    def _wait_params(self) -> None:
        eng = self.__dict__.get("engine")
        if eng is not None:
            eng.wait_params()

    def state_dict(self, *args, **kwargs):
        self._wait_params()
        return super().state_dict(*args, **kwargs)

    def named_parameters(self, *args, **kwargs):
        self._wait_params()
        return super().named_parameters(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False, _refresh: bool = True):
        self._wait_params()
        res = super().load_state_dict(state_dict, strict=strict)
        if _refresh:
            self.refresh()
        return res

    def _reattach_grads(self, force: bool = False) -> None:
        for name, p in nn.Module.named_parameters(self):
            if p.requires_grad and (force or p.grad is None):
                p.grad = self._grad_views[name]

    def named_grad_views(self) -> Dict[str, torch.Tensor]:
        """name -> view into the flat gradient buffer (stable across zero_grad(set_to_none=True))."""
        self._wait_params()
        return self._grad_views

    # -------------------------------------------------------------- loaders
    @classmethod
    def from_pretrained(cls, path: str, **kwargs) -> "STonKGsForPreTraining":
        """Local directory in HF layout (config.json + weights). ``kwargs`` go to the constructor, as in HF."""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path!r}: only local checkpoints can be loaded (no network)")
        cfg = STonKGsConfig.from_pretrained(path)
        model = cls(cfg, **kwargs)
        sd = _load_weights_file(path)
        if sd is None:
            raise FileNotFoundError(f"no pytorch_model.bin / model.safetensors under {path!r}")
        missing, unexpected = model.load_state_dict(sd, strict=False)
        # what HF's from_pretrained tolerates for this architecture: buffers that are not parameters here, the tied / dead
        # aliases a safetensors checkpoint drops (quirk Q4: word_embeddings <-> cls.predictions.decoder.weight,
        # cls.predictions.bias <-> decoder.bias, text_bias / entity_bias <-> decoder.*), and the heads the checkpoint's
        # architecture does not have - `STonKGsForSequenceClassification.from_pretrained(pretraining_dir, num_labels=n)`
        # (ref:stonkgs_finetuning.py:404-407) keeps the freshly initialised classifier, as HF does, and says so
        fresh = [k for k in missing if k.startswith(_FRESH_HEAD_PREFIXES)]
        bad = [k for k in missing if k not in fresh and not any(t in k for t in _DROPPED_ALIASES)]
        if bad:
            raise KeyError(f"checkpoint misses {bad[:5]}...")
        if fresh:
            import warnings

            warnings.warn(f"Some weights of {cls.__name__} were not initialized from the checkpoint at {path} and are "
                          f"newly initialized: {fresh}. You should probably TRAIN this model on a down-stream task.")
        return model

    @classmethod
    @lru_cache(maxsize=32)
    def from_default_pretrained(cls, **kwargs) -> "STonKGsForPreTraining":
        """ref:stonkgs_model.py:143-147 fetches 'stonkgs/stonkgs-150k' from the hub; offline, the checkpoint must be
        provided locally through $STONKGS_PRETRAINED_DIR."""
        path = os.environ.get("STONKGS_PRETRAINED_DIR")
        if not path:
            raise FileNotFoundError("set STONKGS_PRETRAINED_DIR to a local copy of stonkgs/stonkgs-150k")
        return cls.from_pretrained(path, **kwargs)

    def save_pretrained(self, path: str, safe_serialization: bool = False) -> None:
        """HF layout: config.json + pytorch_model.bin, or model.safetensors with ``safe_serialization=True`` (what newer
        HF versions write by default; `from_pretrained` reads either)."""
        self.config.save_pretrained(path)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file

            save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(path, "pytorch_model.bin"))

    # -------------------------------------------------------------- forward
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, masked_lm_labels=None,
                ent_masked_lm_labels=None, next_sentence_labels=None, return_dict=None, head_mask=None):
        """ref:stonkgs_model.py:149-258. ``head_mask`` is accepted and ignored (quirk Q8)."""
        cfg = self.config
        dev = self._device
        if input_ids is None:
            raise ValueError("input_ids is required")
        if getattr(cfg, "output_attentions", False):   # (set after construction: the constructor refuses it already)
            cfg.validate_for_hip()

        def prep(t):
            if t is None:
                return None
            t = torch.as_tensor(t)
            if t.device != dev or t.dtype != torch.long or not t.is_contiguous():
                t = t.to(device=dev, dtype=torch.long).contiguous()
            return t

        input_ids, attention_mask, token_type_ids = prep(input_ids), prep(attention_mask), prep(token_type_ids)
        mlm, elm, nsp = prep(masked_lm_labels), prep(ent_masked_lm_labels), prep(next_sentence_labels)
        if input_ids.dim() != 2 or input_ids.shape[1] != cfg.max_position_embeddings:
            raise ValueError(f"input_ids must be [B, {cfg.max_position_embeddings}] (text half | entity half)")
        have_labels = mlm is not None and elm is not None and nsp is not None
        training = self.training
        self._sync_derived()
        need_bwd = have_labels and torch.is_grad_enabled() and training
        dense = self.materialize_logits if self.materialize_logits is not None else not training
        # (hidden states leave only in the dataclass: a tuple-returning training forward may run the encoder unpadded)
        out = self.engine.forward(input_ids, attention_mask, token_type_ids, mlm if have_labels else None,
                                  elm if have_labels else None, nsp if have_labels else None, training, dense, need_bwd,
                                  want_hidden=bool(return_dict) or not need_bwd)
        total_loss = None
        if have_labels:
            total_loss = out["loss"]
            self.last_loss_terms = tuple(out[k].clone() for k in _TERM_KEYS)
            total_loss = _StepFunction.apply(self._anchor, self, total_loss) if need_bwd else total_loss.clone()
        prediction_scores = (out.get("text_logits"), out.get("ent_logits"))
        nsp_logits = out["nsp_logits"].clone()
        if not return_dict:
            output = (prediction_scores, nsp_logits)
            return ((total_loss,) + output) if total_loss is not None else output
        return BertForPreTrainingOutputWithPooling(
            loss=total_loss, prediction_logits=prediction_scores, seq_relationship_logits=nsp_logits,
            hidden_states=out["hidden_states"].float(), attentions=None, pooler_output=out["pooler_output"].clone())


    # -------------------------------------------------------------- forward-only encoder (embedding extraction)
    @torch.no_grad()
    def encode(self, input_ids, attention_mask=None, token_type_ids=None, pooled_only: bool = False):
        """(sequence_output bf16 [B, S, H], pooler_output fp32 [B, H]) without the pre-training heads: what
        ``model(**row, return_dict=True).pooler_output`` costs in ref:stonkgs_for_embeddings.py:179 minus the two
        vocabulary-wide decoders, whose logits that caller throws away. Dropout is off (the reference's
        ``from_pretrained`` models are in eval mode); any batch size. ``pooled_only``: the caller reads the pooled vector
        only (embedding extraction does): the encoder runs on the rows that are live keys or position 0, its last layer on
        position 0 alone behind the QKV projection (Engine.unpad), and the first element returned is None."""
        cfg = self.config
        dev = self._device

        def prep(t):
            if t is None:
                return None
            t = torch.as_tensor(t)
            if t.device != dev or t.dtype != torch.long or not t.is_contiguous():
                t = t.to(device=dev, dtype=torch.long).contiguous()
            return t

        input_ids, attention_mask, token_type_ids = prep(input_ids), prep(attention_mask), prep(token_type_ids)
        if input_ids.dim() != 2 or input_ids.shape[1] != cfg.max_position_embeddings:
            raise ValueError(f"input_ids must be [B, {cfg.max_position_embeddings}] (text half | entity half)")
        self._sync_derived()
        seq_out, pooled = self.engine.encode(input_ids, attention_mask, token_type_ids, False, None,
                                             (None, None) if pooled_only else None)
        self.engine.check_errors()
        B = input_ids.shape[0]
        if pooled_only:
            return None, pooled.clone()
        return seq_out.view(B, cfg.max_position_embeddings, cfg.hidden_size).clone(), pooled.clone()

    # -------------------------------------------------------------- fused training path (no autograd)
    def forward_backward(self, inputs: Dict[str, torch.Tensor], gscale: float = 1.0, on_segment_done=None):
        """forward + hand-written backward in one call: what ``Trainer.training_step`` does through
        ``loss = model(**inputs)[0]; loss.backward()`` (hf:trainer.py:1892-1963), minus the autograd graph and the
        host sync on ``dloss``. Gradients are accumulated into the flat buffer; returns the (detached) loss."""
        dev = self._device
        t = {k: (v if (torch.is_tensor(v) and v.device == dev and v.dtype == torch.long and v.is_contiguous())
                 else torch.as_tensor(v).to(device=dev, dtype=torch.long).contiguous()) for k, v in inputs.items()}
        self._sync_derived()
        out = self.engine.forward(t["input_ids"], t.get("attention_mask"), t.get("token_type_ids"),
                                  t["masked_lm_labels"], t["ent_masked_lm_labels"], t["next_sentence_labels"],
                                  self.training, False, True, want_hidden=False)
        terms = out["loss_terms"].clone()          # [total, text MLM, entity MLM, NSP]: one copy out of the workspace
        loss, self.last_loss_terms = terms[0], (terms[1], terms[2], terms[3])
        self.engine.backward(gscale, on_segment_done)
        return loss


@dataclass
class SequenceClassifierOutput:
    """hf:modeling_outputs.SequenceClassifierOutput as returned by ref:stonkgs_finetuning.py:340-345."""

    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits, self.hidden_states, self.attentions) if v is not None)[k]


class _ClsStepFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, loss):
        ctx.model = model
        return loss.clone()

    @staticmethod
    def backward(ctx, dloss):
        model = ctx.model
        model._prepare_grads_for_autograd()
        model.engine.backward_cls(float(dloss), model._segment_hook)
        model._reattach_grads()
        model._external_step_pending = True
        return None, None, None


class STonKGsForSequenceClassification(STonKGsForPreTraining):
    """Fine-tuning model of BASELINE config 5 (relation-type etc. classification): mirror of
    ref:src/stonkgs/models/stonkgs_finetuning.py:237-346. Same frozen LM backbone + KG table front, same encoder,
    pooled [CLS] -> dropout -> Linear(hidden, num_labels) -> CrossEntropyLoss. Like the reference it inherits from the
    pre-training class (its MLM/ELM heads stay in the state dict, unused). The three loss branches of the reference
    (:328-338) run on the HIP path: CrossEntropyLoss, MSELoss (regression) and BCEWithLogitsLoss (multi-label)."""

    _TRAIN_PRETRAINING_HEADS = False

    def __init__(self, config, **kwargs):
        cfg = STonKGsConfig.from_any(config) if config is not None else None
        if "num_labels" in kwargs:  # from_pretrained(model_path, num_labels=n) as the reference calls it (:404-407)
            n = kwargs.pop("num_labels")
            if cfg is not None:
                cfg.num_labels = n
        super().__init__(cfg, **kwargs)
        self.num_labels = self.config.num_labels
        self.dropout = nn.Dropout(self.config.hidden_dropout_prob)  # naming only; the engine applies it
        self.classifier = _linear(self._store, "classifier", True)
        self._grad_views = {n: p.grad for n, p in self.named_parameters() if p.requires_grad}

    @staticmethod
    def _head_specs(cfg):
        return [("classifier.weight", (cfg.num_labels, cfg.hidden_size), None), ("classifier.bias", (cfg.num_labels,), None)]

    def _prep(self, t):
        if t is None:
            return None
        t = torch.as_tensor(t)
        if t.device != self._device or t.dtype != torch.long or not t.is_contiguous():
            t = t.to(device=self._device, dtype=torch.long).contiguous()
        return t

    def _labels_and_mode(self, labels):
        """ref:stonkgs_finetuning.py:316-338: infer `problem_type` once from num_labels and the label dtype, as the
        reference does, and bring the labels to what the loss kernels read. Returns (labels on the device, loss mode):
        None = CrossEntropyLoss on int64 [B]; hip.LOSS_MSE / LOSS_MSE_BROADCAST (regression: MSELoss on
        logits.view(-1, num_labels) - with num_labels = 1 and 1-D labels torch broadcasts [B,1] x [B] to [B,B], and the
        reference inherits that) / hip.LOSS_BCE (multi-label: BCEWithLogitsLoss) on fp32 labels."""
        if labels is None:
            return None, None
        lt = torch.as_tensor(labels)
        pt = self.config.problem_type
        if pt is None:
            if self.num_labels == 1:
                pt = "regression"
            elif self.num_labels > 1 and lt.dtype in (torch.long, torch.int):
                pt = "single_label_classification"
            else:
                pt = "multi_label_classification"
            self.config.problem_type = pt
        if pt == "single_label_classification":
            return self._prep(lt.reshape(-1)), None
        if pt not in ("regression", "multi_label_classification"):
            raise ValueError(f"unknown problem_type {pt!r}")
        f = lt.to(device=self._device, dtype=torch.float32).contiguous()
        n, C = f.numel(), self.num_labels
        if pt == "regression" and C == 1 and f.dim() == 1 and n > 1:
            return f, hip.LOSS_MSE_BROADCAST
        B = n // max(C, 1)
        if n != B * C or (f.dim() > 1 and f.shape[-1] != C):
            raise ValueError(f"labels of shape {tuple(f.shape)} do not match logits [B, {C}]")
        return f.view(B, C), hip.LOSS_MSE if pt == "regression" else hip.LOSS_BCE

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, labels=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        if input_ids is None:
            raise ValueError("input_ids is required")
        lab, mode = self._labels_and_mode(labels)
        ids, am, tt = self._prep(input_ids), self._prep(attention_mask), self._prep(token_type_ids)
        training = self.training
        self._sync_derived()
        need_bwd = lab is not None and torch.is_grad_enabled() and training
        out = self.engine.forward_cls(ids, am, tt, lab, self.num_labels, training, need_bwd, mode)
        loss = None
        if lab is not None:
            loss = _ClsStepFunction.apply(self._anchor, self, out["loss"]) if need_bwd else out["loss"].clone()
        logits = out["logits"].clone()
        if not return_dict:
            return ((loss,) + (logits,)) if loss is not None else (logits,)
        return SequenceClassifierOutput(loss=loss, logits=logits, hidden_states=None, attentions=None)

    def forward_backward(self, inputs, gscale: float = 1.0, on_segment_done=None):
        lab, mode = self._labels_and_mode(inputs.get("labels"))
        self._sync_derived()
        out = self.engine.forward_cls(self._prep(inputs["input_ids"]), self._prep(inputs.get("attention_mask")),
                                      self._prep(inputs.get("token_type_ids")), lab, self.num_labels,
                                      self.training, True, mode)
        loss = out["loss"].clone()
        self.engine.backward_cls(gscale, on_segment_done)
        return loss


class _LazyNames:
    """kg_idx_to_name for synthetic tables too large to name eagerly."""

    def __init__(self, numeric_indices):
        self._idx = {i: r for r, i in enumerate(numeric_indices)}

    def __getitem__(self, i):
        return f"node{self._idx[i]}"

    def keys(self):
        return self._idx.keys()

    def __len__(self):
        return len(self._idx)


def _load_weights_file(path: str) -> Optional[Dict[str, torch.Tensor]]:
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file

        return load_file(st)
    pb = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pb):
        return torch.load(pb, map_location="cpu", weights_only=True)
    return None
