"""Pre-training driver for the MI355X-native STonKGs: mirror of ref:src/stonkgs/models/stonkgs_pretraining.py
(``pretrain_stonkgs`` -> ``TrainingArguments`` -> ``Trainer(model, args, train_dataset).train()``) with the HF
``Trainer`` inner loop (hf:trainer.py:1892-1963 training_step, :1780-1796 clip / optimizer / scheduler / zero_grad)
restated for one process per GPU:

  forward + hand-written backward (engine)  ->  bucketed RCCL all-reduce of the flat gradient buffer, launched while
  backward is still producing earlier layers' gradients  ->  ONE fused kernel: global-norm clip + AdamW + bf16
  weight refresh + gradient zeroing  ->  W^T refresh for the next step's dgrad GEMMs.

Defaults are the reference's: AdamW(lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 0), max_grad_norm 1.0,
linear decay to zero over max_steps with no warm-up, gradient_accumulation_steps 1.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import asdict, dataclass, field
from typing import Callable, Dict, Iterable, List, Optional

import torch

from . import _hip as hip
from .data import collate, synthetic_batch
from .params import FlatStore


@dataclass
class TrainingArguments:
    """The subset of hf TrainingArguments the reference sets (ref:stonkgs_pretraining.py:171-193) or relies on."""

    output_dir: str = "stonkgs_pretraining_out"
    per_device_train_batch_size: int = 8
    max_steps: int = 200
    learning_rate: float = 1e-4
    gradient_accumulation_steps: int = 1
    max_grad_norm: float = 1.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    weight_decay: float = 0.0
    warmup_steps: int = 0
    logging_steps: int = 100
    save_steps: int = 5000
    save_total_limit: int = 5
    seed: int = 42
    ddp_bucket_mb: float = 64.0
    # the last `ddp_tail_mb` of the gradient buffer (what backward finishes last) in buckets of >= `ddp_tail_bucket_mb`: the
    # final bucket's collective is exposed in full, so it should be small (one encoder layer = 28 MB fp32)
    ddp_tail_mb: float = 100.0
    ddp_tail_bucket_mb: float = 24.0
    # clip + AdamW + W^T refresh (and the tail of the gradient all-reduce) on their own stream, beside the next step's
    # frozen-backbone forward; parameters read through the model's accessors or after torch.cuda.synchronize() are final
    optimizer_overlap: bool = True
    # queue the NEXT batch's frozen-backbone forward beside the current step's encoder forward (it depends on token ids and
    # frozen weights only); needs the next batch: `Trainer.train` peeks it, `training_step(..., next_inputs=)` takes it
    prefetch_backbone: bool = True
    # run the gradient collectives even when the process group has a single rank (hardware rehearsal of the N > 1 path)
    ddp_force_collectives: bool = False
    # ZeRO-2-style optimizer sharding (the reference can run DeepSpeed ZeRO-2: ref:stonkgs_pretraining.py:174-175): every
    # gradient bucket is reduce-SCATTERED instead of all-reduced, each rank keeps Adam's m / v for - and updates - its
    # 1/world piece of every bucket only, and the updated fp32 parameters are all-gathered (same bytes on the wire as the
    # all-reduce; AdamW's 30 B/param of HBM traffic and the optimizer state shrink by the world size)
    shard_optimizer: bool = False
    # sum(g^2) for the clip coefficient per gradient bucket as backward finalises it (beside matrix work), instead of one
    # pass over the whole buffer between backward and AdamW. OFF by default: measured on one GPU (round 4, tools/ab_step.py
    # args:bucket_grad_norm, interleaved in one process) the step is 0.33 ms LONGER with it (28.30 against 27.97 ms) - the
    # 0.26-ms serial read goes, but a notification per segment orders the weight-gradient stream behind the main stream
    # once per layer and a 974-MB read runs beside the GEMMs. Kept for N > 1, where the partial of a bucket can only be
    # taken behind its collective and the alternative is the same serial pass after the LAST collective.
    bucket_grad_norm: bool = False
    # who runs the gradient collectives: "torch" = torch.distributed's process group (backend nccl = RCCL), "stonk" = the C
    # ABI's own communicator (stonk_comm_*: RCCL on a library-owned stream with event hand-off; torch.distributed, if
    # initialised, only carries the 128-byte RCCL id once)
    comm_backend: str = "torch"
    # refresh the bf16 W^T copies (read by backward's dgrad launches only) behind the "parameters final" event, beside the
    # next step's forward, instead of in front of it. OFF: measured (round 4, tools/ab_step.py args:defer_wt_refresh) 27.22
    # against 27.16 ms - the 0.2-ms transpose is then paid beside the forward's first GEMMs instead of before them
    defer_wt_refresh: bool = False


def linear_schedule_lr(base_lr: float, step: int, max_steps: int, warmup: int = 0) -> float:
    """hf get_linear_schedule_with_warmup: learning rate used by optimizer step number `step` (0-based)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (max_steps - step) / max(1, max_steps - warmup))


class FusedAdamW:
    """clip_grad_norm_ + AdamW + zero_grad over the model's flat fp32 buffers in two kernel launches - or, with `spans`
    (optimizer sharding: the [lo, hi) pieces of the flat buffer this rank owns), the same per piece with m / v stored for
    the owned pieces only."""

    def __init__(self, store: FlatStore, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 spans: Optional[List[tuple]] = None):
        self.store = store
        self.betas, self.eps, self.weight_decay, self.max_grad_norm = betas, eps, weight_decay, max_grad_norm
        # HF Trainer decays every parameter except biases and LayerNorm weights (hf:trainer.py get_decay_parameter_names);
        # the reference trains with weight_decay = 0, so this only matters to a caller who sets it
        self.decayed = [name for name in store.index if name.endswith(".weight") and "LayerNorm" not in name]
        self.spans = None if spans is None else [(int(lo), int(hi)) for lo, hi in spans if hi > lo]
        n_state = store.numel if self.spans is None else sum(hi - lo for lo, hi in self.spans)
        dev = store.data.device
        self.m = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        # partial sums + ticket of the fixed-order grad-norm reduction (zeroed once; the kernel leaves it ready)
        self.norm_ws = torch.zeros(int(hip.lib().stonk_sumsq_workspace_floats()), dtype=torch.float32, device=dev)
        self.step_count = 0
        self._decay_spans = None   # device [n, 2] int64 of the decayed tensors' [lo, hi) offsets (built on first use)

    def _pieces(self):
        """(offset in the flat buffers, offset in m / v, length) of every piece this rank updates."""
        if self.spans is None:
            return [(0, 0, self.store.numel)]
        out, acc = [], 0
        for lo, hi in self.spans:
            out.append((lo, acc, hi - lo))
            acc += hi - lo
        return out

    def accumulate_grad_norm_sq(self) -> None:
        """gnorm_sq = sum of squares of the gradient over this rank's pieces (the whole buffer without sharding) - in a
        fixed order, so that replicas holding the same gradients get the same bits."""
        s, st = self.store, hip.stream_ptr()
        self.gnorm_sq.zero_()
        for off, _, n in self._pieces():
            hip.call("stonk_sumsq_f32", s.grad.data_ptr() + 4 * off, n, self.gnorm_sq.data_ptr(), self.norm_ws.data_ptr(),
                     self.norm_ws.numel(), st)

    def _decay_table(self):
        if self._decay_spans is None:
            s = self.store
            rows = []
            for name in self.decayed:
                off, _, pshape = s.index[name]
                n = 1
                for d in pshape:
                    n *= d
                rows.append((off, off + n))
            rows.sort()
            self._decay_spans = torch.tensor(rows, dtype=torch.int64, device=s.data.device).reshape(-1, 2)
        return self._decay_spans

    def apply_update(self, lr: float, grad_scale: float = 1.0) -> None:
        """clip (from gnorm_sq) + AdamW + bf16 mirror + gradient zeroing over this rank's pieces. Decoupled weight decay
        (p *= 1 - lr * wd on the decayed tensors only: torch.optim.AdamW's order) happens inside the same kernel, from a
        device table of the decayed tensors' spans."""
        s, st = self.store, hip.stream_ptr()
        b1, b2 = self.betas
        wd = float(self.weight_decay)
        tab = self._decay_table() if wd else None
        for off, soff, n in self._pieces():
            hip.call("stonk_adamw_step", s.data.data_ptr() + 4 * off, s.grad.data_ptr() + 4 * off,
                     self.m.data_ptr() + 4 * soff, self.v.data_ptr() + 4 * soff, s.bf16.data_ptr() + 2 * off, n, lr, b1, b2,
                     self.eps, wd, 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count, self.gnorm_sq.data_ptr(),
                     self.max_grad_norm, grad_scale, hip.ptr(tab), 0 if tab is None else tab.shape[0], off, st)

    def step(self, lr: float, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        self.accumulate_grad_norm_sq()
        self.apply_update(lr, grad_scale)

    def last_grad_norm(self, grad_scale: float = 1.0) -> float:
        torch.cuda.synchronize(self.gnorm_sq.device)   # the optimizer may have run on its own stream
        return float(self.gnorm_sq.sqrt().item()) * grad_scale

    def zero_grad(self) -> None:
        self.store.grad.zero_()

    def state_dict(self, sync=None):
        """Adam's state in the flat buffer's layout, whatever the sharding: with a sharded optimizer the pieces are
        all-gathered through `sync` first (a collective: every rank calls it), so a checkpoint written under one world
        size loads under another."""
        if self.spans is None:
            return {"step": self.step_count, "m": self.m.cpu(), "v": self.v.cpu()}
        full = {}
        for key, src in (("m", self.m), ("v", self.v)):
            buf = torch.zeros(self.store.numel, dtype=torch.float32, device=src.device)
            for off, soff, n in self._pieces():
                buf[off:off + n].copy_(src[soff:soff + n])
            sync.gather_params(buf)
            full[key] = buf.cpu()
        return {"step": self.step_count, **full}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        if sd["m"].numel() != self.store.numel:
            raise ValueError("optimizer state does not match this model's flat parameter layout")
        for key, dst in (("m", self.m), ("v", self.v)):
            src = sd[key]
            for off, soff, n in self._pieces():
                dst[soff:soff + n].copy_(src[off:off + n])


def plan_buckets(segment_ends: List[int], bucket_elems: int, tail_elems: int = 0, tail_bucket_elems: int = 0) -> List[tuple]:
    """Cut the flat gradient buffer into contiguous [lo, hi) buckets whose boundaries are segment ends (a segment =
    what one backward notification finalises), each at least `bucket_elems` long except possibly the last.

    `tail_elems` / `tail_bucket_elems` (round 4, from tools/overlap_budget.py's timeline): the last `tail_elems` elements of
    the buffer - what backward finishes LAST - are cut into buckets of at least `tail_bucket_elems` instead. The collective
    of the final bucket cannot start before backward ends, so its duration is exposed whatever the bandwidth: with uniform
    64 MB buckets that was an 85 MB bucket (the last three layers), with a tapered tail it is one layer."""
    buckets, lo = [], 0
    total = segment_ends[-1] if segment_ends else 0
    for end in segment_ends:
        want = min(tail_bucket_elems, bucket_elems) if (tail_bucket_elems and total - end < tail_elems) else bucket_elems
        if end - lo >= want:
            buckets.append((lo, end))
            lo = end
    if segment_ends and lo < total:
        if buckets and tail_bucket_elems and total - lo < tail_bucket_elems // 4:   # (a sliver: the embeddings' few tensors)
            buckets[-1] = (buckets[-1][0], total)
        else:
            buckets.append((lo, total))
    return buckets


class GradSynchronizer:
    """Data-parallel gradient averaging over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" on CPU tests).

    The flat gradient buffer is laid out in backward-completion order (params.py), so when the engine reports
    "segment X done" every byte before X's end is final: the bucket ending there is all-reduced immediately, on
    RCCL's own stream, while the compute stream continues with the next layer's backward. xGMI is point-to-point
    (7 links/GPU), so few large buckets (default 64 MB) beat DDP's 25 MB default. Averaging is folded into the
    optimizer kernel's grad_scale (sum here, 1/world there).

    `shard=True` (TrainingArguments.shard_optimizer; ZeRO-2 as the reference can run it through DeepSpeed,
    ref:stonkgs_pretraining.py:174-175): every bucket is cut into `world` equal pieces, rank r owns piece r of EVERY bucket
    (so its share becomes final bucket by bucket, like the buckets themselves), the collective is a reduce-scatter into the
    owned piece, the optimizer updates the owned pieces only (`owned_spans`) and `gather_params` all-gathers the updated
    fp32 parameters bucket by bucket: the same bytes on the wire as the all-reduce, 1/world of AdamW's HBM traffic and of
    Adam's state. Buckets end at tensor boundaries, which are multiples of 256 elements (params.ALIGN): any world size
    that divides 64 cuts them into 16-byte aligned pieces."""

    def __init__(self, grad: torch.Tensor, segments: Dict[str, int], bucket_mb: float = 64.0, group=None,
                 force: bool = False, shard: bool = False, comm=None, tail_mb: float = 0.0, tail_bucket_mb: float = 0.0):
        """`force`: issue the collectives even in a one-rank group (a sum over one rank: the values do not change) - lets
        a one-GPU box execute the RCCL path, its stream ordering and the kernel routing that goes with it."""
        import torch.distributed as dist

        self.dist = dist
        self.grad = grad
        self.group = group
        # `comm`: a stonkgs_amd.comm.StonkComm - the collectives then go through the C ABI's own communicator (RCCL on the
        # library's stream, event hand-off) instead of torch.distributed
        self.comm = comm
        if comm is not None:
            self.world, self.rank = comm.world, comm.rank
            self.active = self.world > 1 or force
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self.active = self.world > 1 or (force and dist.is_initialized())
        self.segment_end = dict(segments)  # notification name -> end offset in the flat buffer
        ends = sorted(set(segments.values()))
        self.buckets = plan_buckets(ends, int(bucket_mb * (1 << 20) / 4), int(tail_mb * (1 << 20) / 4),
                                    int(tail_bucket_mb * (1 << 20) / 4))
        self.shard = bool(shard) and self.active
        if self.shard:
            bad = [(lo, hi) for lo, hi in self.buckets if (hi - lo) % (4 * self.world)]
            if bad:
                raise ValueError(f"shard_optimizer: world size {self.world} does not cut bucket {bad[0]} into 16-byte "
                                 "aligned pieces (tensors start at multiples of 256 elements: use a world size dividing 64)")
            # gloo (CPU tests, several ranks on one card) has no reduce-scatter: an all-reduce leaves the sum in the owned
            # piece as well - same result, more bytes, never a measurement
            self._has_rs = comm is not None or dist.get_backend(group) == "nccl"
        self._next = 0
        self._works = []
        self._norm_parts = None    # per-bucket sums of squares of the FINAL gradient (enable_bucket_norm)
        self.norm_on = True        # (a per-step switch for A/B runs: Trainer copies TrainingArguments.bucket_grad_norm here)

    # ---- the gradient norm, bucket by bucket (round 4). The clip coefficient needs sum(g^2) over the whole buffer: a 974-MB
    # read that sat between the end of backward and AdamW (0.26 ms of the step's serial tail). A bucket's gradients are
    # final long before that - when backward reports the segment ending there (after the bucket's collective at N > 1) - so
    # its partial sum is taken THEN, on the weight-gradient stream (N = 1) or on a stream of its own behind the collective,
    # beside matrix work that leaves HBM idle. Partials are summed in bucket order: the same bits on every replica.
    def enable_bucket_norm(self) -> None:
        dev = self.grad.device
        if dev.type != "cuda" or self._norm_parts is not None:
            return
        nws = int(hip.lib().stonk_sumsq_workspace_floats())
        self._norm_parts = torch.zeros(len(self.buckets), dtype=torch.float32, device=dev)
        self._norm_ws = torch.zeros(len(self.buckets), nws, dtype=torch.float32, device=dev)   # (one per bucket: launches overlap)
        self._norm_stream = torch.cuda.Stream(device=dev) if self.active else None

    def _bucket_norm(self, b: int, work) -> None:
        lo, hi = self.buckets[b]
        if self.shard:
            piece = (hi - lo) // self.world
            lo, hi = lo + self.rank * piece, lo + (self.rank + 1) * piece
        args = (self.grad.data_ptr() + 4 * lo, hi - lo, self._norm_parts.data_ptr() + 4 * b, self._norm_ws[b].data_ptr(),
                self._norm_ws.shape[1])
        if work is None:
            hip.call("stonk_sumsq_f32", *args, hip.stream_ptr())
            return
        self._norm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._norm_stream):
            work.wait()                       # the norm stream is ordered behind the collective, nobody else waits here
            hip.call("stonk_sumsq_f32", *args, hip.stream_ptr())

    def take_grad_norm_sq(self, out: torch.Tensor) -> bool:
        """After `finish()`: `out[0]` = sum of the bucket partials (this rank's pieces when sharded), in bucket order.
        False when the partials are not kept (CPU tensors): the caller then reads the buffer itself."""
        if self._norm_parts is None or not self.norm_on:
            return False
        if self._norm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._norm_stream)
        torch.sum(self._norm_parts, dim=0, keepdim=True, out=out)
        self._norm_parts.zero_()
        return True

    def owned_spans(self) -> Optional[List[tuple]]:
        """[lo, hi) pieces of the flat buffers this rank reduces into and updates; None when the optimizer is replicated."""
        if not self.shard:
            return None
        out = []
        for lo, hi in self.buckets:
            piece = (hi - lo) // self.world
            out.append((lo + self.rank * piece, lo + (self.rank + 1) * piece))
        return out

    def _launch(self, lo: int, hi: int):
        d = self.dist
        if self.comm is not None:
            from .comm import _Work

            if self.shard:
                piece = (hi - lo) // self.world
                self.comm.reduce_scatter(self.grad[lo + self.rank * piece: lo + (self.rank + 1) * piece], self.grad[lo:hi])
            else:
                self.comm.all_reduce(self.grad[lo:hi])
            return _Work(self.comm)
        if self.shard and self._has_rs:
            piece = (hi - lo) // self.world
            mine = self.grad[lo + self.rank * piece: lo + (self.rank + 1) * piece]   # in place: output = own slice of input
            return d.reduce_scatter_tensor(mine, self.grad[lo:hi], op=d.ReduceOp.SUM, group=self.group, async_op=True)
        return d.all_reduce(self.grad[lo:hi], op=d.ReduceOp.SUM, group=self.group, async_op=True)

    def _bucket_final(self) -> None:
        b = self._next
        work = None
        if self.active:
            work = self._launch(*self.buckets[b])
            self._works.append(work)
        if self._norm_parts is not None and self.norm_on:
            self._bucket_norm(b, work)
        self._next += 1

    def on_segment_done(self, name: str) -> None:
        if not self.active and (self._norm_parts is None or not self.norm_on):
            return
        end = self.segment_end.get(name)
        if end is None:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][1] <= end:
            self._bucket_final()

    def finish(self) -> float:
        """Flush remaining buckets, wait for all of them; returns the factor that turns the sum into the mean."""
        if self.active or (self._norm_parts is not None and self.norm_on):
            while self._next < len(self.buckets):
                self._bucket_final()
            for w in self._works:
                w.wait()
        self._works, self._next = [], 0
        return 1.0 / self.world

    def all_reduce_scalar(self, t: torch.Tensor) -> None:
        """Sum a small tensor over the ranks (the sharded grad-norm: every rank holds the squares of its pieces)."""
        if self.shard and self.comm is not None:
            self.comm.all_reduce(t)
            self.comm.wait()
        elif self.shard:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def gather_params(self, flat: torch.Tensor) -> None:
        """Sharded optimizer: every rank has updated its piece of every bucket of `flat` (the fp32 parameters, or any
        buffer with the gradient's layout); all-gather the pieces in place, bucket by bucket."""
        if not self.shard:
            return
        d, works = self.dist, []
        for lo, hi in self.buckets:
            piece = (hi - lo) // self.world
            mine = flat[lo + self.rank * piece: lo + (self.rank + 1) * piece]
            if self.comm is not None:
                self.comm.all_gather(flat[lo:hi], mine)
                continue
            if self._has_rs:
                works.append(d.all_gather_into_tensor(flat[lo:hi], mine, group=self.group, async_op=True))
            else:   # gloo: list form, into the bucket's own pieces
                outs = [flat[lo + r * piece: lo + (r + 1) * piece] for r in range(self.world)]
                works.append(d.all_gather(outs, mine.clone(), group=self.group, async_op=True))
        for w in works:
            w.wait()
        if self.comm is not None:
            self.comm.wait()


def segment_ends_for(model) -> Dict[str, int]:
    """Map the engine's backward notifications to end offsets in the flat gradient buffer."""
    store, cfg = model._store, model.config
    seg = {}
    if "classifier.bias" in store.index:
        seg["classifier.bias"] = store.span("classifier.bias")[1]
    for name in ("cls.predictions.entity_decoder.weight", "cls.predictions.text_decoder.weight", "bert.pooler.dense.bias"):
        if name in store.index:
            seg[name] = store.span(name)[1]
    for i in range(cfg.num_hidden_layers):
        seg[f"bert.encoder.layer.{i}"] = store.span(f"bert.encoder.layer.{i}.output.LayerNorm.bias")[1]
    seg["bert.embeddings"] = store.numel
    return seg


class Trainer:
    """``Trainer(model=model, args=training_args, train_dataset=ds).train()`` as the reference calls it
    (ref:stonkgs_pretraining.py:215-223). `train_dataset`: sequence of row dicts with the six schema columns, or any
    iterable of already collated batches."""

    def __init__(self, model, args: Optional[TrainingArguments] = None, train_dataset=None,
                 data_collator: Callable = collate):
        import torch.distributed as dist

        self.model = model
        self.args = args or TrainingArguments()
        self.train_dataset = train_dataset
        self.data_collator = data_collator
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        comm = None
        if self.args.comm_backend == "stonk":
            from .comm import StonkComm

            comm = StonkComm(self.rank, self.world, model.device.index or 0)
        elif self.args.comm_backend != "torch":
            raise ValueError("comm_backend must be 'torch' or 'stonk'")
        self.comm = comm
        self.sync = GradSynchronizer(model._store.grad, segment_ends_for(model), self.args.ddp_bucket_mb,
                                     force=self.args.ddp_force_collectives, shard=self.args.shard_optimizer, comm=comm,
                                     tail_mb=self.args.ddp_tail_mb, tail_bucket_mb=self.args.ddp_tail_bucket_mb)
        self.optimizer = FusedAdamW(model._store, (self.args.adam_beta1, self.args.adam_beta2), self.args.adam_epsilon,
                                    self.args.weight_decay, self.args.max_grad_norm, spans=self.sync.owned_spans())
        model.engine.comm_overlap = self.sync.active   # (see Engine.comm_overlap)
        if self.args.bucket_grad_norm:
            self.sync.enable_bucket_norm()
        self.global_step = 0
        self._micro = 0
        self._next_cache = None
        self.log_history: List[dict] = []

    # hf:trainer.py:1892-1963 (+ the optimizer half of :1780-1796 when the accumulation window closes)
    def _on_device(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The batch as contiguous int64 device tensors; a batch handed over earlier as `next_inputs` is converted once."""
        cached = self._next_cache
        if cached is not None and cached[0] is batch:
            return cached[1]
        dev = self.model.device
        ints = ("input_ids", "attention_mask", "token_type_ids", "masked_lm_labels", "ent_masked_lm_labels",
                "next_sentence_labels")   # (anything else - a classification head's labels - is the model's to interpret)
        return {k: (v if (k not in ints or (torch.is_tensor(v) and v.device == dev and v.dtype == torch.long and
                                            v.is_contiguous()))
                    else torch.as_tensor(v).to(device=dev, dtype=torch.long).contiguous()) for k, v in batch.items()}

    def training_step(self, model, inputs: Dict[str, torch.Tensor], num_items_in_batch=None, *,
                      next_inputs: Optional[Dict] = None) -> torch.Tensor:
        """hf:trainer.py training_step(model, inputs, num_items_in_batch) - the third argument is accepted and unused, as
        the reference's loss takes no such count. `next_inputs` (keyword, optional): the batch of the NEXT call - its
        frozen-backbone forward is queued now, beside this step's encoder forward (TrainingArguments.prefetch_backbone)."""
        model.train()
        inputs = self._on_device(inputs)
        self._next_cache = None
        if next_inputs is not None and self.args.prefetch_backbone:
            nxt = self._on_device(next_inputs)
            self._next_cache = (next_inputs, nxt)
            model.engine.next_input_ids = nxt["input_ids"]
        gas = self.args.gradient_accumulation_steps
        self._micro += 1
        last = self._micro % gas == 0
        self.sync.norm_on = bool(self.args.bucket_grad_norm)
        hook = self.sync.on_segment_done if last else None
        with model.engine.block("K1-K15 forward + backward"):
            loss = model.forward_backward(inputs, gscale=1.0 / gas, on_segment_done=hook)
        if last:
            lr = linear_schedule_lr(self.args.learning_rate, self.global_step, self.args.max_steps, self.args.warmup_steps)
            with model.engine.optimizer_stream(self.args.optimizer_overlap), model.engine.block("K16-K17 all-reduce wait + AdamW"):
                scale = self.sync.finish()
                opt = self.optimizer
                opt.step_count += 1
                if not self.sync.take_grad_norm_sq(opt.gnorm_sq):   # (bucket partials taken during backward)
                    opt.accumulate_grad_norm_sq()           # replicated: the whole buffer; sharded: this rank's pieces ...
                self.sync.all_reduce_scalar(opt.gnorm_sq)   # ... summed over the ranks (a no-op when replicated)
                opt.apply_update(lr, grad_scale=scale)
                if self.sync.shard:
                    model._store.grad.zero_()               # (the kernel zeroed the owned pieces only)
                    self.sync.gather_params(model._store.data)
                # sharded: the bf16 mirror of the gathered pieces too; the W^T copies follow behind "parameters final"
                model.engine.refresh_derived(bf16_mirror=self.sync.shard, transposes=not self.args.defer_wt_refresh)
            if self.args.defer_wt_refresh:
                model.engine.refresh_wt_deferred()
            self.global_step += 1
        return loss

    def _batches(self) -> Iterable[Dict[str, torch.Tensor]]:
        ds, bs = self.train_dataset, self.args.per_device_train_batch_size
        if ds is None:
            raise ValueError("Trainer.train() needs a train_dataset")
        if isinstance(ds, (list, tuple)) and ds and isinstance(ds[0], dict) and not torch.is_tensor(ds[0]["input_ids"]):
            g = torch.Generator().manual_seed(self.args.seed)
            while True:
                perm = torch.randperm(len(ds), generator=g).tolist()
                shard = perm[self.rank::self.world]  # DistributedSampler-style sharding
                for i in range(0, len(shard) - bs + 1, bs):
                    yield self.data_collator([ds[j] for j in shard[i:i + bs]])
        else:
            while True:
                for b in ds:
                    yield b

    def train(self, resume_from_checkpoint: Optional[str] = None):
        if resume_from_checkpoint:
            self.load_checkpoint(resume_from_checkpoint)
        it = iter(self._batches())
        # hf:trainer.py resume semantics (ignore_data_skip=False): the batches the checkpointed run consumed are drawn
        # and discarded, so the resumed run continues on the data - and, with the dropout counter restored by
        # load_checkpoint, on the random masks - the uninterrupted run would have seen
        for _ in range(self.global_step * self.args.gradient_accumulation_steps):
            next(it)
        t0 = time.time()
        batch = next(it)
        while self.global_step < self.args.max_steps:
            upcoming = next(it)   # (known one step ahead: its frozen-backbone forward is queued beside this step)
            loss = self.training_step(self.model, batch, next_inputs=upcoming)
            batch = upcoming
            if self._micro % self.args.gradient_accumulation_steps == 0:
                if self.global_step % self.args.logging_steps == 0 or self.global_step == self.args.max_steps:
                    self.model.engine.check_errors()
                    self.log_history.append({"step": self.global_step, "loss": float(loss), "time": time.time() - t0})
                if self.args.save_steps and self.global_step % self.args.save_steps == 0:
                    self.save_checkpoint()   # (every rank: a sharded optimizer gathers its state; rank 0 writes)
        return {"global_step": self.global_step, "training_loss": float(loss)}

    # checkpoint / resume (ref:stonkgs_pretraining.py:185-186,196-223: save_steps, save_total_limit, resume)
    def save_checkpoint(self) -> str:
        """Rank 0 writes. With a SHARDED optimizer this is a collective - every rank must call it (each holds 1/world of
        Adam's state, gathered here into the flat layout a checkpoint carries whatever the sharding) - and `Trainer.train`
        does; with replicated state the other ranks return at once and copy nothing."""
        self.model.engine.wait_params()
        d = os.path.join(self.args.output_dir, f"checkpoint-{self.global_step}")
        sharded = bool(getattr(self.sync, "shard", False))
        if self.rank != 0 and not sharded:
            return d
        opt_state = self.optimizer.state_dict(self.sync)
        if self.rank != 0:
            return d
        self.model.save_pretrained(d)
        torch.save(opt_state, os.path.join(d, "optimizer.pt"))
        with open(os.path.join(d, "trainer_state.json"), "w") as fh:
            json.dump({"global_step": self.global_step, "log_history": self.log_history, "args": asdict(self.args),
                       "dropout_counter": int(self.model.engine.seed_base)}, fh)
        ckpts = sorted((c for c in os.listdir(self.args.output_dir) if c.startswith("checkpoint-")),
                       key=lambda c: int(c.split("-")[1]))
        for old in ckpts[:-self.args.save_total_limit]:
            import shutil

            shutil.rmtree(os.path.join(self.args.output_dir, old))
        return d

    def load_checkpoint(self, d: str) -> None:
        from .stonkgs_model import _load_weights_file

        self.model.load_state_dict(_load_weights_file(d), strict=False)
        self.optimizer.load_state_dict(torch.load(os.path.join(d, "optimizer.pt"), weights_only=True))
        with open(os.path.join(d, "trainer_state.json")) as fh:
            state = json.load(fh)
        self.global_step = state["global_step"]
        if "dropout_counter" in state:   # (hf restores its RNG states from rng_state.pth; ours is one counter)
            self.model.engine.seed_base = int(state["dropout_counter"])

    def save_model(self, output_dir: Optional[str] = None) -> None:
        self.model.save_pretrained(output_dir or self.args.output_dir)


def get_last_checkpoint(folder: str) -> Optional[str]:
    if not os.path.isdir(folder):
        return None
    c = [x for x in os.listdir(folder) if x.startswith("checkpoint-") and x.split("-")[1].isdigit()]
    return os.path.join(folder, max(c, key=lambda x: int(x.split("-")[1]))) if c else None


def pretrain_stonkgs(model, train_dataset, batch_size: int = 8, deepspeed: bool = False, fp16: bool = True, lr: float = 1e-4,
                     dataloader_num_workers: int = 2, gradient_accumulation_steps: int = 1, logging_steps: int = 100,
                     max_steps: int = 10000, overwrite_output_dir: bool = False, save_limit: int = 5, save_steps: int = 5000,
                     training_dir: str = "pretraining"):
    """ref:stonkgs_pretraining.py:103-230 without mlflow / hub: build args, resume if a checkpoint exists (raise if
    the directory is non-empty without one, as the reference does :203-207), train, save.

    The reference's keyword arguments keep their names, defaults and meaning (`:103-120`); what they map to here:
    `deepspeed` -> `TrainingArguments.shard_optimizer` (the reference's DeepSpeed configuration is ZeRO stage 2: optimizer
    state and gradients partitioned over the ranks - the sharded optimizer of this build, no DeepSpeed involved);
    `fp16` is accepted and has no effect: the step always computes in bf16 with fp32 master weights and fp32 accumulation,
    which needs no loss scaling (BASELINE config 2; `fp16=False` would ask for an fp32 path this build does not have and
    is refused); `save_limit` -> `save_total_limit`; `save_steps` as is; `dataloader_num_workers` is accepted and unused
    (batches are assembled on the device or handed over as tensors: there is no worker pool to size). The model and the
    dataset are passed in (the reference builds them from hub names / a pickled frame: out of scope)."""
    if not fp16:
        raise ValueError("fp16=False asks for a full-precision training step; this build computes in bf16 (fp32 master "
                         "weights and accumulation) and has no fp32 path")
    del dataloader_num_workers
    args = TrainingArguments(output_dir=training_dir, per_device_train_batch_size=batch_size, max_steps=max_steps,
                             learning_rate=lr, gradient_accumulation_steps=gradient_accumulation_steps,
                             logging_steps=logging_steps, save_steps=save_steps, save_total_limit=save_limit,
                             shard_optimizer=bool(deepspeed))
    last = None
    if os.path.isdir(training_dir) and not overwrite_output_dir:
        last = get_last_checkpoint(training_dir)
        if last is None and os.listdir(training_dir):
            raise ValueError(f"Output directory ({training_dir}) already exists and is not empty.")
    trainer = Trainer(model=model, args=args, train_dataset=train_dataset)
    result = trainer.train(resume_from_checkpoint=last)
    trainer.save_model()
    return trainer, result
