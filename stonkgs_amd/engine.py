"""HIP execution engine for the STonKGs hot path: schedules the C-ABI kernels (include/stonk_hip.h) for
forward, backward and the derived-weight refresh.  No torch arithmetic on the data path: torch tensors are
device memory, ``torch.cuda.current_stream()`` is the stream every launcher is given.

Call structure mirrors SURVEY.md section 3.2 (ref:src/stonkgs/models/stonkgs_model.py:149-258):
  F1 frozen LM backbone on the text half  ->  F2 KG gather + concat + embeddings LayerNorm  ->  F3 encoder layers
  ->  F4 pooler / NSP  ->  F5 head transform  ->  F6 label-sparse decoders + fused cross-entropy
and the mirrored backward.  Activations needed by backward are kept in a persistent workspace (HBM is 288 GB;
B=64 needs ~12 GB), nothing is recomputed except attention probabilities.
"""
from __future__ import annotations

import contextlib

import math
import os
import time
from typing import Callable, Dict, List, Optional

import torch

from . import _hip as hip
from .config import STonKGsConfig
from .params import FlatStore, pad128

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32

ERR_BITS = {1: "entity id outside the KG table (the reference raises KeyError, stonkgs_model.py:185)",
            2: "token_type_id outside [0, type_vocab_size)", 4: "text token id outside the LM vocabulary",
            8: "MLM/ELM label outside the decoder's class range", 16: "NSP label outside [0, 2)"}


class GemmTimer:
    """Optional per-launch HIP-event timing of the GEMM kernel (bench.py roofline leg)."""

    def __init__(self):
        self.records = []  # (kind, start_event, end_event, M, N, K, m_dev, k_dev)
        self.spans = {}    # name -> [(start_event, end_event)]: stream-order spans (the encoder's forward / backward)

    def kinds(self):
        return sorted({r[0] for r in self.records})

    def span_seconds(self, name) -> float:
        torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in self.spans.get(name, [])) * 1e-3

    def summarize(self, kind=None):
        """Algorithmic FLOPs use the rows / contraction length that exist at run time (device-side counts of the
        label-sparse decoders), not the launch capacity. kind: "tn_a4" / "tn" (weight-gradient kernels: four-wave 256x256, written-out loop /
        128x128), "nt", or None = all."""
        torch.cuda.synchronize()
        tot_t = tot_f = 0.0
        n = 0
        for k, s, e, M, N, K, m_dev, k_dev in self.records:
            if kind is not None and k != kind:
                continue
            n += 1
            tot_t += s.elapsed_time(e) * 1e-3
            if m_dev is not None:
                M = min(M, int(m_dev.item()))
            if k_dev is not None:
                K = min(K, int(k_dev.item()))
            tot_f += 2.0 * M * N * K
        return {"launches": n, "seconds": tot_t, "flops": tot_f}


class Engine:
    def __init__(self, cfg: STonKGsConfig, store: FlatStore, backbone: FlatStore, n_backbone_layers: int, device):
        cfg.validate_for_hip()
        hip.lib()  # fail loudly, now, if the extension is missing
        self.cfg = cfg
        self.P = store
        self.BB = backbone
        self.n_bb = n_backbone_layers
        self.device = device
        self.ws: Dict[str, torch.Tensor] = {}
        self.kg_table: Optional[torch.Tensor] = None  # fp32 [K+3, H]
        self.err = torch.zeros(1, dtype=I32, device=device)
        self.gemm_timer: Optional[GemmTimer] = None
        self.saved = None
        self.seed_base = 0x5710
        # Weight-gradient GEMMs are off the critical path of backward (nothing downstream reads dW until the optimizer):
        # they are launched on a second HIP stream and overlap the dgrad / LayerNorm / attention chain on the main one.
        self.overlap_wgrad = True
        # Set by the trainer when gradients are all-reduced while backward runs (world size > 1): RCCL's kernels then hold
        # some CUs for the length of a collective, and a PERSISTENT one-workgroup-per-CU kernel with a static tile split
        # would wait for them (its late workgroups own a share of the tiles). Backward's persistent launches on the main
        # stream (the dgrad GEMMs of every layer, `_kernel(site, True)`) are then launched STONK_GEMM_DISPATCHED: the
        # same four-wave kernel with one work item per workgroup, so the hardware dispatcher hands out the tiles (one GPU,
        # nothing beside it: +0.9 ms per step against the persistent grids; the 128x128 kernel used here before: +1.7 ms).
        self.comm_overlap = False
        # ... with TWO work items per workgroup (STONK_GEMM_DISPATCHED2, round 3): every second tile boundary keeps its
        # prefetch; measured on one GPU with nothing beside it (tools/ab_step.py dispatched_pairs engine.comm_overlap=1)
        self.dispatched_pairs = True
        self.decoder_dgrad_256 = True
        # LayerNorm backward: dgamma / dbeta partial sums added on the weight-gradient stream (stonk_layernorm_bwd_reduce) instead
        # of behind the kernel on the main one. Measured: +0.16 ms (26.87 against 26.71, tools/ab_step.py ln_reduce_side) - the
        # 26 launches of ~7 us it takes off the main chain cost less there than the event record / wait pairs that order them: off
        self.ln_reduce_side = False
        self._ln_done = {}
        # weight gradients whose row count is a multiple of 128 only (the text decoder's 29 056 = 113.5 x 256) on the
        # written-out 256x256 kernel as well (its last row tile is half empty, the buffers' range checks drop what it
        # adds): correct, and no change to the step - 26.76 against 26.77 ms interleaved (tools/ab_step.py tn_ragged) - so off
        self.tn_ragged = False
        # the label-sparse decoders' dgrad on gemm_a4.hip (fp32 atomics over a K split): measured SLOWER than the eight-wave
        # kernel at these shapes - rows 58 / 350 KB apart, 737 against 612 us (entity) and 349 against 148 (text),
        # tools/decoder_probe.py - so off; the decoders' FORWARD (fp16 logits) does run there: 687 against 888, 128 / 168
        self.decoder_dgrad_a4 = False
        self.decoder_fwd_a4 = True     # (False: the eight-wave kernel, as before round 4)
        # Which of the library's three NT kernels runs a launch is the LIBRARY's choice (STONK_GEMM_AUTO: from shape and
        # epilogue, stonk_gemm_nt_bf16) - except where the engine knows what the library cannot: that an all-reduce is
        # running beside backward (comm_overlap -> the dynamically scheduled 128x128 kernel for the persistent launches).
        # `kernel_for` lets a tool pin a site for an A/B (tools/ab_step.py): "qkv", "attn_out", "ffn_up", "ffn_down",
        # "dgrad_gelu", "dgrad_resid", "dgrad_attn_out", "dgrad_head" -> hip.GEMM_*.
        self.kernel_for: Dict[str, int] = {}
        self.f16_logits = True      # label-sparse decoder logits in fp16 (False: fp32, 4 more bytes of HBM traffic per logit)
        # Unpadded trainable encoder (csrc/unpad.hip): rows that are neither live keys, nor labelled, nor position 0 are
        # dropped before the embeddings LayerNorm - no loss term and no gradient reads them - and every per-row kernel,
        # GEMM and the attention run on the packed rows. Used by the training paths that hand out neither hidden states
        # nor dense logits (`forward_backward`, `forward(..., return_dict=False)` in training mode); costs one host wait
        # per step for the packed row count, taken while the frozen backbone's forward is already queued.
        self.unpad = True
        # ... and, in the packed layout, the LAST layer's feed-forward block, the pooler and the head transform run on the
        # READ rows only (labelled positions + position 0: ~77 of a sequence's ~410 rows) - they are row-wise, and nothing
        # reads the last layer's output at any other row. Attention and its projections still see every row (keys).
        self.prune_last_ffn = True
        # ... and so does the last layer's attention block behind the QKV projection: the row plan puts a sequence's read
        # rows first, the attention kernels compute the first rows of every sequence as queries only (`q_offsets`; keys and
        # values: every row), and the output projection + LayerNorm run on the gathered read rows.
        self.prune_last_attn = True
        # attention backward with the dQ and the dK / dV kernels on two streams, delta by a small kernel in front of them
        # (stonk_attention_bwd_phases). Measured: 30.33 against 30.05 ms per step - both kernels are VALU-bound and fill the
        # chip on their own, side by side they only share it. Off; kept as a switch for tools/ab_step.py.
        self.attn_bwd_two_streams = False
        self._astream: Optional[torch.cuda.Stream] = None
        # The frozen backbone's forward depends on the batch's token ids and on frozen weights only: given a hint of the NEXT
        # batch (`next_input_ids`, set by the trainer) it is queued on a stream of its own at the start of the current step and
        # runs beside the current step's encoder forward, where no weight-gradient stream competes for the CUs.
        self.next_input_ids: Optional[torch.Tensor] = None
        self._prefetch: Optional[dict] = None
        self._bb_stream: Optional[torch.cuda.Stream] = None
        self._pref_par = 0
        # what ran: [rows, padded rows they stand for, encoder passes, sum over sequences of rows^2, of S^2] - bench.py prices
        # the step on the FLOPs actually executed (linear layers ~ rows, attention ~ rows^2 per sequence)
        # (sixth: rows of the last layer's feed-forward block / pooler / head; seventh: query rows x key rows of the last
        # layer's attention, summed over sequences)
        self.rows_executed = [0, 0, 0, 0, 0, 0, 0]
        self._plan_host: Optional[torch.Tensor] = None
        self._wstream: Optional[torch.cuda.Stream] = None
        # The optimizer (grad-norm, AdamW, W^T refresh: ~2 ms of HBM-bound work) runs on a third stream; the next step's
        # frozen-backbone forward reads none of what it writes and starts beside it. `wait_params()` orders the current
        # stream after the last parameter write - called before the first trainable weight is read and by every accessor.
        self._opt_stream: Optional[torch.cuda.Stream] = None
        self._params_ready: Optional[torch.cuda.Event] = None
        self._wt_ready: Optional[torch.cuda.Event] = None   # W^T copies refreshed behind the parameters (refresh_wt_deferred)
        # tools/step_marks.py: a list to collect (name, host time, event on the current stream, was the optimizer's
        # "parameters final" event already complete) at a few points of the step; None (default) = nothing is recorded
        self.marks: Optional[list] = None
        self._wgrad_done: Dict[int, torch.cuda.Event] = {}   # layer parity -> side-stream event after its last wgrad
        self._wt_desc = None   # (device table, entries, tiles) of the batched W^T refresh
        # development switches, read ONCE here: the 128x128 weight-gradient kernel everywhere / the CU share of the
        # side-stream weight gradients
        self.tn_v1 = bool(os.environ.get("STONK_TN_V1"))
        self.tn_cus = int(os.environ.get("STONK_TN_CUS", "160"))
        # roctx ranges around the blocks of SURVEY section 2.3 (K1 backbone ... K16 optimizer), so that a
        # `rocprofv3 --marker-trace --kernel-trace` timeline reads by block. Off unless STONK_ROCTX=1 (read once, here).
        self.roctx = bool(os.environ.get("STONK_ROCTX"))
        self.tn_min_k = 16384     # contraction length (rows) from which the four-wave kernel takes a weight gradient
        # CU share of a weight gradient of fewer than 16 256x256 tiles (the 768 x 768 ones: 9 tiles); 0 = tn_cus. 80 CUs'
        # worth (8 K splits instead of 17: half the float atomics for the same K loops) measured 29.81 against 29.98 ms per
        # step - inside the box-to-box noise, and each such launch then takes longer on fewer CUs (the dominant kernel's
        # average launch 246 -> 260 us): left at 0, kept as a switch for tools/ab_step.py
        self.tn_cus_small = 0
        self.tn_min_tiles = 36    # 128x128 tiles of an output from which the four-wave kernel takes the gradient: 36 = the
                                  # 768 x 768 ones too (35.67 against 35.79 ms per step with 100, tools/sweep_engine_int.py)

    # ------------------------------------------------------------------ plumbing
    def buf(self, name: str, shape, dtype=BF16, zero=False) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= d
        t = self.ws.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(max(n, 1), dtype=dtype, device=self.device)
            self.ws[name] = t
        return t[:n].view(shape)

    @contextlib.contextmanager
    def optimizer_stream(self, enabled: bool = True):
        """Run the body on the optimizer stream, ordered after everything enqueued so far on the current stream; on
        exit the engine remembers the event that marks the parameters as final (see `wait_params`)."""
        if not enabled:
            yield
            return
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.device)
        self._opt_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._opt_stream):
            yield
            ev = torch.cuda.Event()
            ev.record(self._opt_stream)
        self._params_ready = ev

    @contextlib.contextmanager
    def block(self, name: str):
        """roctx range (torch.cuda.nvtx maps to roctx on ROCm) around one block of the step; a no-op unless `roctx`."""
        if not self.roctx:
            yield
            return
        torch.cuda.nvtx.range_push(name)
        try:
            yield
        finally:
            torch.cuda.nvtx.range_pop()

    def mark(self, name: str) -> None:
        if self.marks is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        pr = self._params_ready
        self.marks.append((name, time.perf_counter(), ev, None if pr is None else pr.query()))

    def wait_params(self) -> None:
        """Order the current stream after the last optimizer step, if it ran on the optimizer stream (no host
        synchronisation). The event is kept until the next step replaces it: callers on different streams each wait."""
        ev = self._params_ready
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _span_begin(self):
        """(GemmTimer only) start of a stream-order span on the current stream."""
        if self.gemm_timer is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _span_end(self, name, start) -> None:
        if start is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.gemm_timer.spans.setdefault(name, []).append((start, e))

    def ln_ws(self) -> torch.Tensor:
        """Partial-sum workspace of the LayerNorm backward kernels, sized by the library's own query for the largest row
        count it can be asked for (every token of the largest batch seen so far)."""
        n = int(hip.lib().stonk_layernorm_bwd_workspace_floats(1 << 30, self.cfg.hidden_size))
        return self.buf("ln.ws", (n,), F32)

    def ln_bwd(self, dy, x, mean, rstd, gamma, dx, dx_drop, dgamma, dbeta, rows, H, flags, p_in, seed_in, p_out, seed_out):
        """stonk_layernorm_bwd on the current stream. With the weight gradients on their own stream (`overlap_wgrad`) the sum
        of the per-workgroup dgamma / dbeta partials can go there too (`ln_reduce_side`; off - measured slower, see __init__):
        nothing on the main chain reads the two vectors. The partials then need a workspace of their own until that launch
        has run: a ring of four, each slot guarded by the event of the reduce that last read it."""
        st = hip.stream_ptr()
        args = (hip.ptr(dy), hip.ptr(x), hip.ptr(mean), hip.ptr(rstd), hip.ptr(gamma), hip.ptr(dx), hip.ptr(dx_drop),
                hip.ptr(dgamma), hip.ptr(dbeta), rows, H)
        if not (self.ln_reduce_side and self.overlap_wgrad) or dgamma is None or rows == 0:
            ws = self.ln_ws()
            hip.call("stonk_layernorm_bwd", *args, flags, p_in, seed_in, p_out, seed_out, ws.data_ptr(), ws.numel(), st)
            return
        n = int(hip.lib().stonk_layernorm_bwd_workspace_floats(1 << 30, self.cfg.hidden_size))
        slot = self._ln_slot = (getattr(self, "_ln_slot", -1) + 1) % 4
        ws = self.buf(f"ln.ws.ring{slot}", (n,), F32)
        last = self._ln_done.get(slot)
        if last is not None:
            torch.cuda.current_stream().wait_event(last)   # (the reduce of four LayerNorms ago: done long since)
        hip.call("stonk_layernorm_bwd", *args, flags | hip.LN_DEFER_REDUCE, p_in, seed_in, p_out, seed_out, ws.data_ptr(),
                 ws.numel(), st)
        if self._wstream is None:
            self._wstream = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record()
        self._wstream.wait_event(ready)
        with torch.cuda.stream(self._wstream):
            hip.call("stonk_layernorm_bwd_reduce", ws.data_ptr(), rows, H, hip.ptr(dgamma), hip.ptr(dbeta), hip.stream_ptr())
            done = torch.cuda.Event()
            done.record()
        self._ln_done[slot] = done

    def check_errors(self) -> None:
        """Raise for any flag the kernels set (one tiny D2H copy; call where a sync is acceptable)."""
        e = int(self.err.item())
        if e:
            self.err.zero_()
            msgs = [m for b, m in ERR_BITS.items() if e & b]
            if e & 1:
                raise KeyError("; ".join(msgs))
            raise IndexError("; ".join(msgs))

    def gemm(self, A, B, C, M, N, K, flags=0, bias=None, resid=None, aux=None, alpha=1.0, split_k=1, m_dev=None,
             k_dev=None, drop_p=0.0, seed=0, kernel=hip.GEMM_AUTO):
        st = hip.stream_ptr()
        timed = self.gemm_timer is not None
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.call("stonk_gemm_nt_bf16", A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0),
                 M, N, K, flags, hip.ptr(bias), hip.ptr(resid), 0 if resid is None else resid.stride(0), hip.ptr(aux),
                 0 if aux is None else aux.stride(0), alpha, split_k, hip.ptr(m_dev), hip.ptr(k_dev), drop_p,
                 seed & 0xFFFFFFFF, kernel, st)
        if timed:
            e1.record()
            self.gemm_timer.records.append(("nt", e0, e1, M, N, K, m_dev, k_dev))

    def _split_k(self, M, N, K, side_stream=False) -> int:
        # 128x128 tiles, two workgroups co-resident per CU = 512 slots on 256 CUs: the sweep in tools/sweep_wgrad.py
        # is fastest when tiles x split fills ONE co-resident wave without a tail (432-480 workgroups)
        tiles = ((M + 127) // 128) * (N // 128)
        # the large weight gradients (FFN up / down, fused QKV) go to the four-wave 256x256 kernel, which splits K itself
        # (split_k = 0): -1.4 ms per step in an interleaved A/B; 768 x 768 (36 tiles) and the text decoder's
        # gradient (29 056 rows: not a multiple of 256) stay on the 128x128 kernel. Beside the dgrad chain (second
        # stream) the kernel is held to 160 CUs' worth of workgroups (split_k = -160): another -1.5 ms
        # ... and so does the entity decoder's 175 104 x 768 gradient (2052 unsplit tiles, device-side token count, 5.7 GB
        # operand extent - the kernel re-bases its buffer resources per K tile): 863 us against 966 alone, -0.27 ms in the step
        if (tiles >= self.tn_min_tiles and K >= self.tn_min_k and (M % 256 == 0 or self.tn_ragged) and N % 256 == 0
                and not self.tn_v1):
            if side_stream and self.tn_cus_small and tiles < 64:   # (tiles counts 128x128 ones: 64 = 16 of 256x256)
                return -self.tn_cus_small
            return -self.tn_cus if side_stream else 0
        return max(1, min(32, 480 // tiles, K // 64))

    def wgrad(self, dy, x, dW, db, M_out, N_in, T, k_dev=None, alpha=1.0):
        """dW[M_out, N_in] += dy[T, M_out]^T . x[T, N_in];  db[M_out] += colsum(dy)   (fp32 atomics, split-K).
        With `overlap_wgrad` the launch goes to the second stream, ordered after everything enqueued so far on the
        current one; a GemmTimer brackets it with events on THAT stream, in the same launch configuration."""
        side = self.overlap_wgrad
        split = self._split_k(M_out, N_in, T, side)
        timed = self.gemm_timer is not None
        if side:
            if self._wstream is None:
                self._wstream = torch.cuda.Stream(device=self.device)
            ready = torch.cuda.Event()
            ready.record()                       # dy (and x) are complete on the main stream at this point
            self._wstream.wait_event(ready)
        with (torch.cuda.stream(self._wstream) if side else contextlib.nullcontext()):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            hip.call("stonk_gemm_tn_bf16", dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dW.data_ptr(),
                     dW.stride(0), hip.ptr(db), M_out, N_in, T, alpha, split, hip.ptr(k_dev), hip.stream_ptr())
            if timed:
                e1.record()
                self.gemm_timer.records.append(("tn_a4" if split <= 0 else "tn", e0, e1, M_out, N_in, T, None, k_dev))

    def transpose(self, x, rows, cols, name, colsum=None, rows_dev=None):
        rpad = (rows + 63) // 64 * 64
        out = self.buf(name, (cols, rpad))
        hip.call("stonk_transpose_bf16", x.data_ptr(), x.stride(0), out.data_ptr(), rpad, rows, cols, hip.ptr(colsum),
                 hip.ptr(rows_dev), hip.stream_ptr())
        return out

    def _kernel(self, site: str, persistent_in_backward: bool = False) -> int:
        """NT kernel of a launch site: a tool's pin, else the dispatcher-scheduled form for a backward launch that
        would otherwise be persistent while a collective holds CUs, else the library's own choice."""
        k = self.kernel_for.get(site)
        if k is not None:
            return k
        if persistent_in_backward and self.comm_overlap:
            return hip.GEMM_DISPATCHED2 if self.dispatched_pairs else hip.GEMM_DISPATCHED
        return hip.GEMM_AUTO

    def seed(self, layer: int, site: int) -> int:
        return (self.seed_base * 0x9E3779B1 + layer * 64 + site) & 0xFFFFFFFF

    # ------------------------------------------------------------------ derived weights
    def refresh_derived(self, bf16_mirror: bool = True, transposes: bool = True) -> None:
        """bf16 mirrors (if the optimizer has not just written them) and W^T copies for dgrad. `transposes=False`: the caller
        refreshes the W^T copies itself, later (`refresh_wt_deferred`)."""
        st = hip.stream_ptr()
        if bf16_mirror:
            for s in (self.P, self.BB):
                hip.call("stonk_cast_f32_to_bf16", s.data.data_ptr(), s.bf16.data_ptr(), s.numel, st)
        if transposes:
            self.wait_wt()   # (a deferred refresh still in flight writes the same copies)
            self._refresh_wt(st)

    def refresh_wt_deferred(self) -> None:
        """The W^T copies BEHIND the "parameters final" event (round 4): only backward's dgrad launches read them, ten
        milliseconds into the next step, so the 0.2-ms transpose of 0.49 GB runs on the optimizer stream beside the next
        step's forward instead of between AdamW and it; `backward` waits for `_wt_ready`."""
        stream = self._opt_stream if self._opt_stream is not None and self._params_ready is not None else None
        if stream is None:
            self._refresh_wt(hip.stream_ptr())
            return
        with torch.cuda.stream(stream):
            self._refresh_wt(hip.stream_ptr())
            ev = torch.cuda.Event()
            ev.record(stream)
        self._wt_ready = ev

    def wait_wt(self) -> None:
        ev = self._wt_ready
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _refresh_wt(self, st) -> None:
        if self._wt_desc is None or self._wt_desc[3] != self.P.bf16.data_ptr():   # (the table holds raw addresses)
            self._wt_desc = self._build_wt_table()
        desc, n, tiles, _ = self._wt_desc
        hip.call("stonk_transpose_bf16_batched", desc.data_ptr(), n, tiles, st)

    def _build_wt_table(self):
        """Descriptor table of the batched W^T refresh (stonk_transpose_bf16_batched): one entry per dgrad weight - source =
        its slice of the bf16 mirror (the optimizer has just written it; bit-identical to casting the fp32 master), destination
        = the [in, out_padded] copy. Addresses are stable for the life of the store, so the table is built once."""
        import struct

        entries, first = [], 0
        for name, (off, shape, pshape) in self.P.index.items():
            if len(shape) != 2 or not name.endswith(".weight") or shape[0] < 64:
                continue
            if "embeddings" in name or "pooler" in name or "seq_relationship" in name:
                continue
            rows, cols = shape
            rpad = pshape[0] if pshape[0] % 64 == 0 else (pshape[0] + 63) // 64 * 64
            wt = self.P.wt.get(name)
            if wt is None:
                wt = torch.zeros(cols, rpad, dtype=BF16, device=self.device)
                self.P.wt[name] = wt
            src = self.P.bf16_view(name, padded=False)
            # the kernel moves 16-byte vectors and cannot validate a device-side table: what it assumes is checked here
            if cols % 8 or src.data_ptr() % 16 or wt.data_ptr() % 16 or rpad % 8 or wt.shape[1] < (rows + 63) // 64 * 64:
                raise ValueError(f"{name}: [{rows}, {cols}] cannot take the batched W^T refresh (needs cols % 8 == 0, "
                                 "16-byte aligned slices and a destination of roundup64(rows) columns)")
            col_tiles = (cols + 63) // 64
            entries.append(struct.pack("<QQqqqiiii", src.data_ptr(), wt.data_ptr(), cols, rpad, rows, cols, first, col_tiles, 0))
            first += ((rows + 63) // 64) * col_tiles
        raw = torch.frombuffer(bytearray(b"".join(entries)), dtype=torch.uint8).to(self.device)
        return raw, len(entries), first, self.P.bf16.data_ptr()

    # ------------------------------------------------------------------ one BERT layer
    def layer_fwd(self, S: FlatStore, prefix: str, x, B, seq, mask, p_hid, p_att, lidx, save: Optional[dict],
                  T: Optional[int] = None, cu=None, rd: Optional[dict] = None):
        """One BERT layer on T rows. Padded layout: T = B * seq, `mask` = attention_mask [B, seq]. Packed layout (`cu` =
        sequence offsets of stonk_unpad_plan): T = the packed row count rounded up to 64, `mask` = one word per row.
        `rd` (last layer of the packed layout): the feed-forward block runs on the READ rows only - gathered after the
        attention block's LayerNorm - and the layer's output has rd["T"] rows."""
        cfg = self.cfg
        H, I, NH = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
        cap = B * seq
        T = cap if T is None else T
        st = hip.stream_ptr()
        tag = prefix if save is not None else "tmp"
        w = S.bf16_view
        f = S.view
        qkv = self.buf(f"{tag}.qkv", (cap, 3 * H))
        self.gemm(x, w(prefix + ".attention.self.qkv.weight"), qkv, T, 3 * H, H, flags=hip.EPI_BIAS,
                  bias=f(prefix + ".attention.self.qkv.bias"), kernel=self._kernel("qkv"))
        # (zeroed when allocated: in the packed layout the rows between the last sequence and T are never written by the
        # attention kernel and must stay finite for the projections that run over them)
        ctx = self.buf(f"{tag}.ctx", (cap, H), zero=True)
        lse = self.buf(f"{tag}.lse", (B, NH, seq), F32)
        qattn = rd is not None and rd.get("attn", False)   # queries = the read rows only (a sequence's first rows)
        qoff = rd["offsets"] if qattn else None
        hip.call("stonk_attention_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H,
                 hip.ptr(mask), hip.ptr(cu), hip.ptr(qoff), ctx.data_ptr(), H, lse.data_ptr(), B, NH, seq, 64,
                 1.0 / math.sqrt(64.0), p_att, self.seed(lidx, 1), st)
        Ta, a_in, res = T, ctx, x                  # rows / input / residual of the output projection
        if qattn:
            Ta = rd["T"]
            a_in, res = self.buf(f"{tag}.ctxrd", (cap, H)), self.buf(f"{tag}.xrd", (cap, H))
            for src, dst in ((ctx, a_in), (x, res)):   # (rows from the count up to the next multiple of 128 are zero-filled)
                hip.call("stonk_gather_rows_bf16", src.data_ptr(), H, rd["rows"].data_ptr(), rd["cnt"].data_ptr(),
                         dst.data_ptr(), H, H, cap, st)
        s1 = self.buf(f"{tag}.s1", (cap, H))
        fl = hip.EPI_BIAS | hip.EPI_RESID | (hip.EPI_DROPOUT if p_hid > 0 else 0)
        self.gemm(a_in, w(prefix + ".attention.output.dense.weight"), s1, Ta, H, H, flags=fl,
                  bias=f(prefix + ".attention.output.dense.bias"), resid=res, drop_p=p_hid, seed=self.seed(lidx, 2),
                  kernel=self._kernel("attn_out"))
        h1 = self.buf(f"{tag}.h1", (cap, H))
        st1 = self.buf(f"{tag}.st1", (2, cap), F32)
        hip.call("stonk_layernorm_fwd", s1.data_ptr(), f(prefix + ".attention.output.LayerNorm.weight").data_ptr(),
                 f(prefix + ".attention.output.LayerNorm.bias").data_ptr(), h1.data_ptr(), st1[0].data_ptr(),
                 st1[1].data_ptr(), Ta, H, cfg.layer_norm_eps, 0, 0.0, 0, st)
        hf, Tf = h1, Ta                            # input rows of the feed-forward block
        if rd is not None and not qattn:
            hf, Tf = self.buf(f"{tag}.h1rd", (cap, H)), rd["T"]
            hip.call("stonk_gather_rows_bf16", h1.data_ptr(), H, rd["rows"].data_ptr(), rd["cnt"].data_ptr(), hf.data_ptr(),
                     H, H, cap, st)            # (rows from the count up to the next multiple of 128 are zero-filled)
        g = self.buf(f"{tag}.g", (cap, I))
        u = self.buf(f"{tag}.u", (cap, I)) if save is not None else None
        fl = hip.EPI_BIAS | hip.EPI_GELU | ((hip.EPI_SAVE_PREACT | hip.EPI_AUX_GRAD) if save is not None else 0)
        self.gemm(hf, w(prefix + ".intermediate.dense.weight"), g, Tf, I, H, flags=fl,
                  bias=f(prefix + ".intermediate.dense.bias"), aux=u, kernel=self._kernel("ffn_up"))
        s2 = self.buf(f"{tag}.s2", (cap, H))
        fl = hip.EPI_BIAS | hip.EPI_RESID | (hip.EPI_DROPOUT if p_hid > 0 else 0)
        self.gemm(g, w(prefix + ".output.dense.weight"), s2, Tf, H, I, flags=fl, bias=f(prefix + ".output.dense.bias"),
                  resid=hf, drop_p=p_hid, seed=self.seed(lidx, 3), kernel=self._kernel("ffn_down"))
        y = self.buf(f"{prefix}.y" if save is not None else f"tmp.y{lidx & 1}", (cap, H))
        st2 = self.buf(f"{tag}.st2", (2, cap), F32)
        hip.call("stonk_layernorm_fwd", s2.data_ptr(), f(prefix + ".output.LayerNorm.weight").data_ptr(),
                 f(prefix + ".output.LayerNorm.bias").data_ptr(), y.data_ptr(), st2[0].data_ptr(), st2[1].data_ptr(), Tf,
                 H, cfg.layer_norm_eps, 0, 0.0, 0, st)
        if save is not None:
            save[prefix] = dict(x=x, qkv=qkv, ctx=ctx, ctx_in=a_in, lse=lse, s1=s1, h1=hf, st1=st1, g=g, u=u, s2=s2, st2=st2)
        return y

    def layer_bwd(self, prefix: str, dy, B, seq, mask, p_hid, p_att, lidx, sv, T: Optional[int] = None, cu=None,
                  rows: Optional[int] = None, rd: Optional[dict] = None):
        """dy: bf16 [T,H] gradient of the layer output. Returns the gradient of the layer input. Packed layout: `cu`,
        per-row `mask`, T = packed rows rounded up to 64, `rows` = the packed rows that belong to a sequence. `rd`: the
        feed-forward block ran on the read rows (dy has rd["T"] rows); its input gradient is scattered back to all rows."""
        cfg = self.cfg
        H, I, NH = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
        cap = B * seq
        T = cap if T is None else T
        st = hip.stream_ptr()
        P = self.P
        g_ = P.grad_view
        f = P.view
        wt = P.wt
        # dY buffers alternate by layer parity: the side stream may still be reading layer i+1's while layer i writes;
        # before reusing the buffers of layer i+2 the main stream waits for that layer's last weight-gradient GEMM
        par = lidx & 1
        ev = self._wgrad_done.pop(par, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        # ---- LN2 backward: ds2 (residual branch) and df (through the FFN-output dropout)
        ds2 = self.buf(f"b.ds2.{par}", (cap, H))
        df = self.buf(f"b.df.{par}", (cap, H)) if p_hid > 0 else None
        Tf = T if rd is None else rd["T"]          # rows the feed-forward block ran on
        self.ln_bwd(dy, sv["s2"], sv["st2"][0], sv["st2"][1], f(prefix + ".output.LayerNorm.weight"), ds2, df,
                    g_(prefix + ".output.LayerNorm.weight"), g_(prefix + ".output.LayerNorm.bias"), Tf, H, 0, 0.0, 0, p_hid,
                    self.seed(lidx, 3))
        if df is None:
            df = ds2
        # ---- FFN down: wgrad, bias grad, dgrad fused with GELU'
        self.wgrad(df, sv["g"], g_(prefix + ".output.dense.weight"), g_(prefix + ".output.dense.bias"), H, I, Tf)
        du = self.buf(f"b.du.{par}", (cap, I))
        self.gemm(df, wt[prefix + ".output.dense.weight"], du, Tf, I, H, flags=hip.EPI_GELU_BWD | hip.EPI_AUX_GRAD,
                  aux=sv["u"], kernel=self._kernel("dgrad_gelu", True))
        # ---- FFN up
        self.wgrad(du, sv["h1"], g_(prefix + ".intermediate.dense.weight"), g_(prefix + ".intermediate.dense.bias"), I, H,
                   Tf)
        qattn = rd is not None and rd.get("attn", False)
        Ta = Tf if qattn else T                    # rows the attention block's projection / LayerNorm ran on
        dh1 = self.buf("b.dh1", (cap, H))
        if rd is None or qattn:
            self.gemm(du, wt[prefix + ".intermediate.dense.weight"], dh1, Tf, H, I, flags=hip.EPI_RESID, resid=ds2,
                      kernel=self._kernel("dgrad_resid", True))
        else:   # gradient of the gathered rows, scattered into an otherwise zero gradient of the attention block's output
            dh1rd = self.buf("b.dh1rd", (cap, H))
            self.gemm(du, wt[prefix + ".intermediate.dense.weight"], dh1rd, Tf, H, I, flags=hip.EPI_RESID, resid=ds2,
                      kernel=self._kernel("dgrad_resid", True))
            dh1[:T].zero_()
            hip.call("stonk_scatter_rows_bf16", dh1rd.data_ptr(), H, rd["rows"].data_ptr(), rd["cnt"].data_ptr(),
                     dh1.data_ptr(), H, H, st)
        # ---- LN1 backward
        ds1 = self.buf(f"b.ds1.{par}", (cap, H))
        da = self.buf(f"b.da.{par}", (cap, H)) if p_hid > 0 else None
        self.ln_bwd(dh1, sv["s1"], sv["st1"][0], sv["st1"][1], f(prefix + ".attention.output.LayerNorm.weight"), ds1, da,
                    g_(prefix + ".attention.output.LayerNorm.weight"), g_(prefix + ".attention.output.LayerNorm.bias"), Ta, H, 0,
                    0.0, 0, p_hid, self.seed(lidx, 2))
        if da is None:
            da = ds1
        # ---- attention output projection
        self.wgrad(da, sv["ctx_in"], g_(prefix + ".attention.output.dense.weight"),
                   g_(prefix + ".attention.output.dense.bias"), H, H, Ta)
        dctx = self.buf("b.dctx", (cap, H))
        if not qattn:
            self.gemm(da, wt[prefix + ".attention.output.dense.weight"], dctx, T, H, H,
                      kernel=self._kernel("dgrad_attn_out", True))
        else:   # the read rows' gradients go back to their packed rows: into zeros (the kernels read whole query tiles)
            dctxrd, ds1rd = self.buf("b.dctxrd", (cap, H)), ds1
            self.gemm(da, wt[prefix + ".attention.output.dense.weight"], dctxrd, Ta, H, H,
                      kernel=self._kernel("dgrad_attn_out", True))
            ds1 = self.buf("b.ds1full", (cap, H))
            for src, dst in ((dctxrd, dctx), (ds1rd, ds1)):
                dst[:T].zero_()
                hip.call("stonk_scatter_rows_bf16", src.data_ptr(), H, rd["rows"].data_ptr(), rd["cnt"].data_ptr(),
                         dst.data_ptr(), H, H, st)
        # ---- attention core
        qkv = sv["qkv"]
        dqkv = self.buf(f"b.dqkv.{par}", (cap, 3 * H))
        delta = self.buf("b.delta", (B, NH, seq), F32)
        qoff = rd["offsets"] if qattn else None
        if qattn:
            # dQ of the rows that were not queries (the kernel does not write them) feeds the projection's gradients; their dK and
            # dV are the dK/dV kernel's to write (every key row of a sequence), so only the dQ third is cleared: 40 of 122 MB
            dqkv[:T, :H].zero_()
        # (packed layout: the rows between the last sequence and T are not the attention kernel's to write, and the weight
        # gradient below contracts over all T rows - backward_encoder zeroes them in both dqkv buffers once per step)
        aargs = (qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, hip.ptr(mask), hip.ptr(cu),
                 hip.ptr(qoff), sv["ctx"].data_ptr(), H, dctx.data_ptr(), H, sv["lse"].data_ptr(), delta.data_ptr(),
                 dqkv.data_ptr(), dqkv.data_ptr() + 2 * H, 3 * H, dqkv.data_ptr() + 4 * H, B, NH, seq, 64,
                 1.0 / math.sqrt(64.0), p_att, self.seed(lidx, 1))
        if not self.attn_bwd_two_streams:
            hip.call("stonk_attention_bwd", *aargs, st)
        else:
            if self._astream is None:
                self._astream = torch.cuda.Stream(device=self.device)
            hip.call("stonk_attention_bwd_phases", hip.ATTN_BWD_DELTA, *aargs, st)
            fork = torch.cuda.Event()
            fork.record()
            self._astream.wait_event(fork)
            with torch.cuda.stream(self._astream):
                hip.call("stonk_attention_bwd_phases", hip.ATTN_BWD_DKV, *aargs, hip.stream_ptr())
                join = torch.cuda.Event()
                join.record()
            hip.call("stonk_attention_bwd_phases", hip.ATTN_BWD_DQ, *aargs, st)
            torch.cuda.current_stream().wait_event(join)
        # ---- QKV projection
        self.wgrad(dqkv, sv["x"], g_(prefix + ".attention.self.qkv.weight"), g_(prefix + ".attention.self.qkv.bias"),
                   3 * H, H, T)
        dx = self.buf(f"b.dx{lidx & 1}", (cap, H))
        self.gemm(dqkv, wt[prefix + ".attention.self.qkv.weight"], dx, T, H, 3 * H, flags=hip.EPI_RESID, resid=ds1,
                  kernel=self._kernel("dgrad_resid", True))
        if self._wstream is not None and self.overlap_wgrad:
            done = torch.cuda.Event()
            done.record(self._wstream)
            self._wgrad_done[par] = done
        return dx

    # ------------------------------------------------------------------ frozen backbone
    def backbone_fwd(self, input_ids, ld_ids, B, seq, training: bool, mask=None):
        cfg = self.cfg
        H = cfg.hidden_size
        p_hid = cfg.hidden_dropout_prob if training else 0.0  # quirk Q6: dropout is live inside the frozen LM
        p_att = cfg.attention_probs_dropout_prob if training else 0.0
        f = self.BB.view
        x = self.buf("bb.x", (B * seq, H))
        hip.call("stonk_text_embed_ln_fwd", input_ids.data_ptr(), ld_ids,
                 f("lm_backbone.embeddings.word_embeddings.weight").data_ptr(),
                 f("lm_backbone.embeddings.position_embeddings.weight").data_ptr(),
                 f("lm_backbone.embeddings.token_type_embeddings.weight").data_ptr(),
                 f("lm_backbone.embeddings.LayerNorm.weight").data_ptr(),
                 f("lm_backbone.embeddings.LayerNorm.bias").data_ptr(), x.data_ptr(), B, seq, H, cfg.vocab_size,
                 cfg.layer_norm_eps, hip.LN_DROPOUT if p_hid > 0 else 0, p_hid, self.seed(100, 0), self.err.data_ptr(),
                 hip.stream_ptr())
        for i in range(self.n_bb):
            x = self.layer_fwd(self.BB, f"lm_backbone.encoder.layer.{i}", x, B, seq, mask, p_hid, p_att, 101 + i, None)
        return x

    def prefetch_backbone(self, input_ids, training: bool) -> None:
        """Queue the frozen backbone's forward for a FUTURE batch on the backbone stream (ordered after everything queued so
        far on the current stream: the ids are there, the previous prefetch's consumer has read its buffer). The result is
        kept in one of two buffers until `encode` is called with the same ids."""
        cfg = self.cfg
        H, S, half = cfg.hidden_size, cfg.max_position_embeddings, cfg.half_length
        B = input_ids.shape[0]
        if self._bb_stream is None:
            self._bb_stream = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record()
        self._bb_stream.wait_event(ready)
        self._pref_par ^= 1
        out = self.buf(f"bb.pref{self._pref_par}", (B * half, H))
        with torch.cuda.stream(self._bb_stream):
            # dropout masks are a function of (step counter, layer, site, element): the prefetched forward draws the masks
            # of the step that will CONSUME it (the counter advances once per encode), so a run that prefetches and one
            # that does not - or a resumed one - see the same masks
            self.seed_base += 1
            try:
                x = self.backbone_fwd(input_ids, S, B, half, training)
            finally:
                self.seed_base -= 1
            out.copy_(x)
            done = torch.cuda.Event()
            done.record()
        self._prefetch = dict(ids=input_ids, ptr=input_ids.data_ptr(), ver=input_ids._version, shape=tuple(input_ids.shape),
                              out=out, done=done, training=training)

    def _take_prefetched(self, input_ids, training: bool):
        """The prefetched backbone output for exactly these ids (same storage, same version, same mode), or None. Either
        way the current stream is ordered after the prefetch: its scratch buffers are the inline forward's too."""
        pf, self._prefetch = self._prefetch, None
        if pf is None:
            return None
        torch.cuda.current_stream().wait_event(pf["done"])
        hit = (pf["ptr"] == input_ids.data_ptr() and pf["ver"] == input_ids._version and
               pf["shape"] == tuple(input_ids.shape) and pf["training"] == training)
        return pf["out"] if hit else None

    def special_vectors(self) -> Dict[int, torch.Tensor]:
        """Quirk Q2: kg_backbone[sid] = lm_backbone([[sid]])[0][0][0], eval mode. A 1-token sequence is run as a
        128-token sequence whose only unmasked key is position 0 - identical arithmetic for that position."""
        ids = torch.zeros(3, 128, dtype=torch.long, device=self.device)
        ids[:, 0] = torch.tensor([102, 103, 100], device=self.device)
        mask = torch.zeros(3, 128, dtype=torch.long, device=self.device)
        mask[:, 0] = 1
        out = self.backbone_fwd(ids, 128, 3, 128, False, mask=mask).view(3, 128, -1)[:, 0].float()
        return {102: out[0].clone(), 103: out[1].clone(), 100: out[2].clone()}

    # ------------------------------------------------------------------ forward
    def _plan_rows(self, attention_mask, mlm_labels, ent_labels, B, S, half, keep: bool):
        """Launch the row plan of the unpadded encoder (csrc/unpad.hip) on the current stream and start the copy of the
        packed row count to the host; returns (plan dict, event). The caller queues the frozen backbone's forward - which
        does not depend on the plan - and only then waits for the event, so the GPU has work while the host learns the
        count. `keep`: the plan belongs to a forward whose backward will read it (its maps live in buffers of their own,
        which a forward-only call in between - evaluation, embedding extraction - does not touch)."""
        n = B * S
        u = "u" if keep else "ut"
        row_of_pos = self.buf(f"{u}.row_of_pos", (n,), I32)
        pos_of_row = self.buf(f"{u}.pos_of_row", (n,), I32)
        row_mask = self.buf(f"{u}.row_mask", (n,), torch.int64)
        offs = self.buf(f"{u}.offsets", (2 * (B + 1),), I32)     # [sequence offsets | read-row offsets]: one copy to the host
        cu, cu_rd = offs[:B + 1], offs[B + 1:]
        read_rows = self.buf(f"{u}.read_rows", (n,), I32)
        read_of_pos = self.buf(f"{u}.read_of_pos", (n,), I32)
        ws = self.buf("u.ws", (int(hip.lib().stonk_unpad_workspace_ints(B)),), I32)
        hip.call("stonk_unpad_plan", attention_mask.data_ptr(), hip.ptr(mlm_labels), hip.ptr(ent_labels), B, S, half,
                 row_of_pos.data_ptr(), pos_of_row.data_ptr(), cu.data_ptr(), row_mask.data_ptr(), read_rows.data_ptr(),
                 read_of_pos.data_ptr(), cu_rd.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream_ptr())
        if self._plan_host is None or self._plan_host.numel() < 2 * (B + 1):
            self._plan_host = torch.empty(2 * (B + 1), dtype=I32).pin_memory()
        self._plan_host[:2 * (B + 1)].copy_(offs, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return dict(row_of_pos=row_of_pos, pos_of_row=pos_of_row, cu=cu, row_mask=row_mask, read_rows=read_rows,
                    read_of_pos=read_of_pos, cu_rd=cu_rd), ev

    def encode(self, input_ids, attention_mask, token_type_ids, training: bool, save: dict, unpad_labels=None):
        """F1-F4: frozen backbone, KG gather + embeddings LayerNorm, encoder layers, pooler. Shared by the pre-training
        and the sequence-classification models (ref:stonkgs_model.py:178-212, ref:stonkgs_finetuning.py:277-310).
        `unpad_labels`: None = padded layout (every position a row: callers that hand out hidden states or dense
        logits); a (text_labels, entity_labels) pair (either may be None) = packed layout, see `Engine.unpad`."""
        cfg = self.cfg
        H, S, half = cfg.hidden_size, cfg.max_position_embeddings, cfg.half_length
        B = input_ids.shape[0]
        cap = B * S
        st = hip.stream_ptr()
        P = self.P
        f = P.view
        self.seed_base += 1
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        p_att = cfg.attention_probs_dropout_prob if training else 0.0
        plan = ev = None
        if unpad_labels is not None and attention_mask is not None and self.unpad and B > 0:
            plan, ev = self._plan_rows(attention_mask, unpad_labels[0], unpad_labels[1], B, S, half, keep=save is not None)
        # F1 frozen backbone (no attention mask: quirk Q5; always padded - its padding positions ARE attended)
        with self.block("K1 frozen backbone fwd"):
            text_hidden = self._take_prefetched(input_ids, training)
            hit = text_hidden is not None
            if not hit:
                text_hidden = self.backbone_fwd(input_ids, S, B, half, training)
            # The next batch's backbone forward, if the trainer named the batch. After a hit it is queued HERE, before the
            # wait for the optimizer: it then runs beside AdamW (HBM-bound) and the encoder forward below. After a miss the
            # inline forward's scratch buffers are still to be read by the embedding kernel: queued behind that kernel.
            hint, self.next_input_ids = self.next_input_ids, None
            if hint is not None and hit:
                self.prefetch_backbone(hint, training)
                hint = None
        self.wait_params()   # everything above read frozen weights only; from here on the trainable ones
        T, rows, cu, mask, rd = cap, cap, None, attention_mask, None
        if plan is not None:
            self.mark("plan_wait")
            ev.synchronize()                       # (a backbone forward - this batch's or the next one's - is queued: the GPU has work)
            self.mark("plan_known")
            host = self._plan_host[:2 * (B + 1)].tolist()
            offs, n_read = host[:B + 1], host[2 * B + 1]
            rows = offs[B]
            T = min(cap, (rows + 63) // 64 * 64)   # whole 64-row K tiles for the weight gradients; the tail rows are zeros
            cu, mask = plan["cu"], plan["row_mask"]
            sq = sum((offs[i + 1] - offs[i]) ** 2 for i in range(B))
            if self.prune_last_ffn:
                rd = dict(rows=plan["read_rows"], cnt=plan["cu_rd"][B:B + 1], n=n_read, offsets=plan["cu_rd"],
                          T=min(cap, (n_read + 63) // 64 * 64), attn=self.prune_last_attn)
        else:
            sq = B * S * S
        Th = T if rd is None else rd["T"]          # rows of the sequence output (and of everything the heads run on)
        sq_last = sq
        if rd is not None and rd["attn"]:
            sq_last = sum((host[B + 2 + i] - host[B + 1 + i]) * (offs[i + 1] - offs[i]) for i in range(B))
        for i, v in enumerate((T, cap, 1, sq, B * S * S, Th, sq_last)):
            self.rows_executed[i] += v
        # F2 gather + concat + embeddings LayerNorm
        # (a forward-only call - save is None - works in scratch buffers of its own throughout: it may run between a
        # training forward and its backward without touching what that backward reads)
        e = "e" if save is not None else "et"
        sum0 = self.buf(f"{e}.sum0", (cap, H))
        x = self.buf(f"{e}.x0", (cap, H))
        st0 = self.buf(f"{e}.st0", (2, cap), F32)
        hip.call("stonk_joint_embed_ln_fwd", input_ids.data_ptr(), hip.ptr(token_type_ids), text_hidden.data_ptr(),
                 self.kg_table.data_ptr(), f("bert.embeddings.position_embeddings.weight").data_ptr(),
                 f("bert.embeddings.token_type_embeddings.weight").data_ptr(),
                 f("bert.embeddings.LayerNorm.weight").data_ptr(), f("bert.embeddings.LayerNorm.bias").data_ptr(),
                 sum0.data_ptr(), x.data_ptr(), st0[0].data_ptr(), st0[1].data_ptr(), B, S, half, H,
                 self.kg_table.shape[0], cfg.type_vocab_size, cfg.layer_norm_eps, hip.LN_DROPOUT if p_hid > 0 else 0,
                 p_hid, self.seed(200, 0), self.err.data_ptr(), 0 if plan is None else plan["pos_of_row"].data_ptr(),
                 T if plan is not None else 0, st)
        if hint is not None:   # (after an inline backbone forward: behind the kernel above, the last reader of its scratch)
            self.prefetch_backbone(hint, training)
        # F3 encoder
        self.mark("encoder_fwd_begin")
        span = self._span_begin()
        last = cfg.num_hidden_layers - 1
        for i in range(cfg.num_hidden_layers):
            with self.block(f"K4-K8 encoder layer {i} fwd"):
                x = self.layer_fwd(P, f"bert.encoder.layer.{i}", x, B, S, mask, p_hid, p_att, i, save, T=T, cu=cu,
                                   rd=rd if i == last else None)
        self._span_end("encoder_fwd", span)
        seq_out = x                                # [Th rows]: every packed row, or the read rows only
        # F4 pooler (fp32 master weights) on position 0 of every sequence
        pooled = self.buf(f"{e}.pooled", (B, H), F32)
        first_rows = None if plan is None else (cu if rd is None else plan["cu_rd"])
        if plan is None:
            first, ld_first = seq_out, S * H
        else:   # packed: position 0 of sequence b is row cu[b] (among the read rows: read_offsets[b])
            nb = self.buf(f"{e}.nb", (1,), I32)
            nb.fill_(B)
            first, ld_first = self.buf(f"{e}.first", ((B + 127) // 128 * 128, H)), H
            hip.call("stonk_gather_rows_bf16", seq_out.data_ptr(), H, first_rows.data_ptr(), nb.data_ptr(), first.data_ptr(),
                     H, H, first.shape[0], st)
        hip.call("stonk_small_linear_fwd", first.data_ptr(), ld_first, f("bert.pooler.dense.weight").data_ptr(),
                 f("bert.pooler.dense.bias").data_ptr(), pooled.data_ptr(), B, H, H, hip.SMALL_TANH, st)
        if save is not None:   # (None: forward-only callers - embedding extraction, batched inference)
            save.update(B=B, attention_mask=mask, token_type_ids=token_type_ids, sum0=sum0, st0=st0,
                        seq_out=seq_out, pooled=pooled, p_hid=p_hid, p_att=p_att, T=T, rows=rows, plan=plan,
                        first=first, ld_first=ld_first, rd=rd, Th=Th, first_rows=first_rows,
                        head_map=None if plan is None else (plan["row_of_pos"] if rd is None else plan["read_of_pos"]))
        return seq_out, pooled

    def forward(self, input_ids, attention_mask, token_type_ids, mlm_labels, ent_labels, nsp_labels, training: bool,
                dense_logits: bool, need_backward: bool, want_hidden: bool = True):
        """`want_hidden=False` (with labels, without dense logits): nobody will read `hidden_states`, so the trainable
        encoder may run on the packed rows (`Engine.unpad`); `hidden_states` is then None."""
        cfg = self.cfg
        H, S, half = cfg.hidden_size, cfg.max_position_embeddings, cfg.half_length
        B = input_ids.shape[0]
        cap_rows = B * S
        st = hip.stream_ptr()
        P = self.P
        f, w = P.view, P.bf16_view
        save: dict = {}
        have_labels = mlm_labels is not None and ent_labels is not None and nsp_labels is not None
        packed = have_labels and not dense_logits and not want_hidden
        seq_out, pooled = self.encode(input_ids, attention_mask, token_type_ids, training, save,
                                      (mlm_labels, ent_labels) if packed else None)
        T, plan = save["Th"], save["plan"]          # rows of the sequence output: all packed rows, or the read rows
        row_of_pos = save["head_map"]
        nsp = self.buf("h.nsp", (B, 2), F32)
        hip.call("stonk_small_linear_fwd", pooled.data_ptr(), H, f("cls.seq_relationship.weight").data_ptr(),
                 f("cls.seq_relationship.bias").data_ptr(), nsp.data_ptr(), B, 2, H, hip.SMALL_X_F32, st)
        # F5 head transform: dense + GELU, LayerNorm
        gt = self.buf("h.gt", (cap_rows, H))
        ut = self.buf("h.ut", (cap_rows, H))
        self.gemm(seq_out, w("cls.predictions.transform.dense.weight"), gt, T, H, H,
                  flags=hip.EPI_BIAS | hip.EPI_GELU | hip.EPI_SAVE_PREACT,
                  bias=f("cls.predictions.transform.dense.bias"), aux=ut)
        t = self.buf("h.t", (cap_rows, H))
        stt = self.buf("h.stt", (2, cap_rows), F32)
        hip.call("stonk_layernorm_fwd", gt.data_ptr(), f("cls.predictions.transform.LayerNorm.weight").data_ptr(),
                 f("cls.predictions.transform.LayerNorm.bias").data_ptr(), t.data_ptr(), stt[0].data_ptr(),
                 stt[1].data_ptr(), T, H, cfg.layer_norm_eps, 0, 0.0, 0, st)
        out = dict(hidden_states=seq_out.view(B, S, H) if plan is None else None, pooler_output=pooled, nsp_logits=nsp)
        heads = (("text", "cls.predictions.text_decoder.weight", cfg.vocab_size, 0, mlm_labels),
                 ("ent", "cls.predictions.entity_decoder.weight", cfg.kg_vocab_size, half, ent_labels))
        # F6 label-sparse decoders + fused softmax cross-entropy (value and logits-gradient in one sweep)
        if have_labels:
            acc = self.buf("l.acc", (8,), F32)  # [text_sum, ent_sum, nsp_sum, nsp_cnt, loss x4]
            acc.zero_()
            cnts = self.buf("l.cnt", (2,), I32)
            cap = B * half
            for hi, (nm, wname, N, off, labels) in enumerate(heads):
                npad = pad128(N)
                rows = self.buf(f"l.{nm}.rows", (cap,), I32)
                tg = self.buf(f"l.{nm}.tg", (cap,), I32)
                cnt = cnts[hi:hi + 1]
                hip.call("stonk_label_compact", labels.data_ptr(), cap, half, S, off, rows.data_ptr(), tg.data_ptr(),
                         cnt.data_ptr(), hip.ptr(row_of_pos), st)
                hs = self.buf(f"l.{nm}.hs", (cap, H))
                hip.call("stonk_gather_rows_bf16", t.data_ptr(), H, rows.data_ptr(), cnt.data_ptr(), hs.data_ptr(), H, H,
                         cap, st)
                # logits of the labelled rows only, in fp16 (11 significant bits; the softmax arithmetic stays fp32): the
                # decoder GEMM writes and the cross-entropy reads 2 bytes per logit instead of 4 - both are HBM-bound on them
                f16 = self.f16_logits
                logits = self.buf(f"l.{nm}.logits", (cap, npad), torch.float16 if f16 else F32)
                self.gemm(hs, w(wname), logits, cap, npad, H, flags=hip.EPI_OUT_F16 if f16 else hip.EPI_OUT_F32, m_dev=cnt,
                          kernel=hip.GEMM_AUTO if self.decoder_fwd_a4 else hip.GEMM_WAVE8)
                dl = self.buf(f"l.{nm}.dl", (cap, npad)) if need_backward else None
                hip.call("stonk_softmax_xent_f16_fwd_bwd" if f16 else "stonk_softmax_xent_fwd_bwd", logits.data_ptr(), npad,
                         N, npad, tg.data_ptr(), cnt.data_ptr(), acc[hi:hi + 1].data_ptr(), hip.ptr(dl), npad, 1.0, cap,
                         self.err.data_ptr(), st)
                save[nm] = dict(rows=rows, cnt=cnt, hs=hs, dl=dl)
            dnsp = self.buf("l.dnsp", (B, 2), F32) if need_backward else None
            hip.call("stonk_nsp_xent_fwd_bwd", nsp.data_ptr(), nsp_labels.data_ptr(), B, 2, acc[2:4].data_ptr(),
                     hip.ptr(dnsp), 1.0, self.err.data_ptr(), st)
            hip.call("stonk_loss_finalize", acc[0:1].data_ptr(), cnts[0:1].data_ptr(), acc[1:2].data_ptr(),
                     cnts[1:2].data_ptr(), acc[2:4].data_ptr(), acc[4:8].data_ptr(), st)
            out.update(loss=acc[4], masked_lm_loss=acc[5], ent_masked_lm_loss=acc[6], next_sentence_loss=acc[7],
                       loss_terms=acc[4:8])
            save["dnsp"] = dnsp
        # F7 dense logits (what the reference always materialises: stonkgs_model.py:70-71)
        if dense_logits:
            cap = B * half
            full = self.buf("d.cnt", (1,), I32)
            full.fill_(cap)
            for nm, wname, N, off, _ in heads:
                npad = pad128(N)
                rows = self.buf(f"d.{nm}.rows", (cap,), I32)
                rows.copy_((torch.arange(cap, device=self.device, dtype=I32) // half) * S + off +
                           torch.arange(cap, device=self.device, dtype=I32) % half)
                hs = self.buf("d.hs", (cap, H))
                hip.call("stonk_gather_rows_bf16", t.data_ptr(), H, rows.data_ptr(), full.data_ptr(), hs.data_ptr(), H, H,
                         cap, st)
                # fresh tensor: handed to the caller, must not alias the workspace
                logits = torch.empty(cap, npad, dtype=F32, device=self.device)
                self.gemm(hs, w(wname), logits, cap, npad, H, flags=hip.EPI_OUT_F32)
                out[f"{nm}_logits"] = logits[:, :N].view(B, half, N)
        if need_backward:
            save.update(gt=gt, ut=ut, t=t, stt=stt)
            self.saved = save
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, gscale: float = 1.0, on_segment_done: Optional[Callable[[str], None]] = None) -> None:
        """Accumulate d(loss * gscale)/d(param) into the flat gradient buffer. `on_segment_done(name)` fires as
        soon as the gradients of a contiguous region of the flat buffer are final (DP all-reduce overlap)."""
        sv = self.saved
        if sv is None:
            raise RuntimeError("backward() without a training forward (labels are required)")
        self.saved = None
        self.wait_wt()
        cfg = self.cfg
        H, S, half = cfg.hidden_size, cfg.max_position_embeddings, cfg.half_length
        B = sv["B"]
        T, cap_rows = sv["Th"], B * S               # the heads ran on the sequence output's rows
        st = hip.stream_ptr()
        P = self.P
        f, g_, wt = P.view, P.grad_view, P.wt
        cap = B * half
        notify = self._make_notify(on_segment_done)
        # ---- decoders (label-sparse): dHs = dlogits . W ; dW += dlogits^T . Hs
        dt = self.buf("b.dt", (cap_rows, H))[:T]
        dt.zero_()
        for nm, wname, N in (("ent", "cls.predictions.entity_decoder.weight", cfg.kg_vocab_size),
                             ("text", "cls.predictions.text_decoder.weight", cfg.vocab_size)):
            npad = pad128(N)
            h = sv[nm]
            # few output tiles (count/128 x H/128), very long contraction (the vocabulary): split-K into fp32
            dhs = self.buf("b.dhs32", (cap, H), F32)
            # (labelled rows of this head <= read rows, known to the host in the packed layout: the split-K atomics and the
            # scatter touch no row beyond them)
            dhs[:cap if sv["rd"] is None else min(cap, (sv["rd"]["n"] + 255) // 256 * 256)].zero_()
            # the persistent 256x256 kernel with 8 splits (<= 30 live tiles x 8 = one round of the CUs) against 128x128
            # tiles with 16: 734 us against 951 for the entity decoder (tools/bench_decoder_dgrad.py)
            # (not beside a running all-reduce - the entity decoder's bucket is in flight when the text decoder's dgrad is
            # launched: a persistent launch would wait for the CUs RCCL holds, tools/hog_test.py)
            # (the written-out four-wave kernel on 256x192 tiles with as many K shares as fill the CUs once - it counts the
            # live tiles itself, split_k is an upper bound - is the slower one here: see __init__)
            if self.decoder_dgrad_a4 and cap >= 1024 and H % 192 == 0 and (npad // 64) % 2 == 0 and not (self.comm_overlap and nm != "ent"):
                self.gemm(h["dl"], wt[wname], dhs, cap, H, npad, flags=hip.EPI_OUT_F32_ATOMIC, split_k=64, m_dev=h["cnt"],
                          alpha=gscale, kernel=hip.GEMM_ASM4_192)
            elif (self.decoder_dgrad_256 and npad >= 16384 and cap >= 1024 and H % 256 == 0
                    and not (self.comm_overlap and nm != "ent")):
                self.gemm(h["dl"], wt[wname], dhs, cap, H, npad, flags=hip.EPI_OUT_F32_ATOMIC, split_k=8, m_dev=h["cnt"],
                          alpha=gscale, kernel=hip.GEMM_WAVE8)
            else:
                self.gemm(h["dl"], wt[wname], dhs, cap, H, npad, flags=hip.EPI_OUT_F32_ATOMIC,
                          split_k=max(1, min(16, npad // 2048)), m_dev=h["cnt"], alpha=gscale)
            hip.call("stonk_scatter_rows_f32_to_bf16", dhs.data_ptr(), H, h["rows"].data_ptr(), h["cnt"].data_ptr(),
                     dt.data_ptr(), H, H, st)
            self.wgrad(h["dl"], h["hs"], g_(wname, padded=True), None, npad, H, cap, k_dev=h["cnt"], alpha=gscale)
            notify(wname)
        # ---- head transform backward
        dgt = self.buf("b.dgt", (cap_rows, H))
        self.ln_bwd(dt, sv["gt"], sv["stt"][0], sv["stt"][1], f("cls.predictions.transform.LayerNorm.weight"), dgt, None,
                    g_("cls.predictions.transform.LayerNorm.weight"), g_("cls.predictions.transform.LayerNorm.bias"), T, H, 0,
                    0.0, 0, 0.0, 0)
        dut = self.buf("b.dut", (cap_rows, H))
        hip.call("stonk_gelu_bwd_bf16", dgt.data_ptr(), sv["ut"].data_ptr(), dut.data_ptr(), T * H, st)
        self.wgrad(dut, sv["seq_out"], g_("cls.predictions.transform.dense.weight"),
                   g_("cls.predictions.transform.dense.bias"), H, H, T)
        dseq = self.buf("b.dseq", (cap_rows, H))
        self.gemm(dut, wt["cls.predictions.transform.dense.weight"], dseq, T, H, H, kernel=self._kernel("dgrad_head", True))
        # ---- NSP + pooler (fp32), pooler gradient lands on position 0 of d(sequence_output)
        dnsp = sv["dnsp"]
        if gscale != 1.0:
            hip.call("stonk_scale_f32", dnsp.data_ptr(), dnsp.numel(), gscale, st)
        dpooled = self.buf("b.dpooled", (B, H), F32)
        hip.call("stonk_small_linear_bwd", dnsp.data_ptr(), 0, sv["pooled"].data_ptr(), H,
                 f("cls.seq_relationship.weight").data_ptr(), g_("cls.seq_relationship.weight").data_ptr(),
                 g_("cls.seq_relationship.bias").data_ptr(), dpooled.data_ptr(), 0, 0, B, 2, H, hip.SMALL_X_F32, st)
        self.backward_encoder(dpooled, dseq, sv, notify)

    def _make_notify(self, hook):
        """Gradients of a segment are final once the SIDE stream has run its weight-gradient GEMMs: the DP hook (RCCL
        all-reduce) is therefore issued under the side stream, which it then orders itself after."""
        if hook is None:
            return lambda name: None
        if not self.overlap_wgrad:
            return hook

        def notify(name):
            if self._wstream is None:
                hook(name)
                return
            ready = torch.cuda.Event()
            ready.record()                      # bias / LayerNorm gradients written by main-stream kernels
            self._wstream.wait_event(ready)
            with torch.cuda.stream(self._wstream):
                hook(name)
        return notify

    def join_wgrad(self) -> None:
        """Main stream waits for every outstanding weight-gradient GEMM (before the optimizer reads the gradients)."""
        if self._wstream is not None:
            torch.cuda.current_stream().wait_stream(self._wstream)
        self._wgrad_done.clear()

    def backward_encoder(self, dpooled, dseq, sv, notify) -> None:
        """Pooler (tanh) backward into position 0 of d(sequence_output), encoder layers last to first, embeddings."""
        cfg = self.cfg
        H, S = cfg.hidden_size, cfg.max_position_embeddings
        B = sv["B"]
        T, rows, plan, rd, first_rows = sv["T"], sv["rows"], sv["plan"], sv["rd"], sv["first_rows"]
        cap = B * S
        st = hip.stream_ptr()
        f, g_ = self.P.view, self.P.grad_view
        if plan is None:
            acc, ld_acc = dseq, S * H
        else:   # packed: position 0 of sequence b is row first_rows[b] - its gradient rows are gathered, added to, scattered back
            nb = self.buf("e.nb", (1,), I32)
            nb.fill_(B)
            acc, ld_acc = self.buf("b.dfirst", ((B + 127) // 128 * 128, H)), H
            hip.call("stonk_gather_rows_bf16", dseq.data_ptr(), H, first_rows.data_ptr(), nb.data_ptr(), acc.data_ptr(), H,
                     H, acc.shape[0], st)
        hip.call("stonk_small_linear_bwd", dpooled.data_ptr(), sv["pooled"].data_ptr(), sv["first"].data_ptr(),
                 sv["ld_first"], f("bert.pooler.dense.weight").data_ptr(), g_("bert.pooler.dense.weight").data_ptr(),
                 g_("bert.pooler.dense.bias").data_ptr(), 0, acc.data_ptr(), ld_acc, B, H, H, hip.SMALL_TANH, st)
        if plan is not None:
            hip.call("stonk_scatter_rows_bf16", acc.data_ptr(), H, first_rows.data_ptr(), nb.data_ptr(), dseq.data_ptr(), H,
                     H, st)
        notify("bert.pooler.dense.bias")
        # ---- encoder layers, last to first
        dy = dseq
        cu = None if plan is None else plan["cu"]
        last = cfg.num_hidden_layers - 1
        if plan is not None and rows < T:   # what an earlier step left in the rows no sequence owns must not reach dW
            for par in (0, 1):
                self.buf(f"b.dqkv.{par}", (cap, 3 * H))[rows:T].zero_()
        span = self._span_begin()
        for i in reversed(range(cfg.num_hidden_layers)):
            prefix = f"bert.encoder.layer.{i}"
            with self.block(f"K15 encoder layer {i} bwd"):
                dy = self.layer_bwd(prefix, dy, B, S, sv["attention_mask"], sv["p_hid"], sv["p_att"], i, sv[prefix], T=T,
                                    cu=cu, rows=rows, rd=rd if i == last else None)
            notify(prefix)
        if span is not None and self._wstream is not None:   # (timing only) the span ends when the layers' weight gradients have
            torch.cuda.current_stream().wait_stream(self._wstream)
        self._span_end("encoder_bwd", span)
        # ---- embeddings LayerNorm, position / token-type embeddings
        dsum = self.buf("b.dsum0", (cap, H))
        p_hid = sv["p_hid"]
        self.ln_bwd(dy, sv["sum0"], sv["st0"][0], sv["st0"][1], f("bert.embeddings.LayerNorm.weight"), dsum, None,
                    g_("bert.embeddings.LayerNorm.weight"), g_("bert.embeddings.LayerNorm.bias"), T, H,
                    hip.LN_DROPOUT if p_hid > 0 else 0, p_hid, self.seed(200, 0), 0.0, 0)
        hip.call("stonk_embed_grad", dsum.data_ptr(), hip.ptr(sv["token_type_ids"]),
                 g_("bert.embeddings.position_embeddings.weight").data_ptr(),
                 g_("bert.embeddings.token_type_embeddings.weight").data_ptr(), B, S, H, cfg.type_vocab_size,
                 0 if plan is None else plan["row_of_pos"].data_ptr(), st)
        notify("bert.embeddings")
        self.mark("backward_end")
        self.join_wgrad()

    # ------------------------------------------------------------------ sequence classification head (config 5)
    def forward_cls(self, input_ids, attention_mask, token_type_ids, labels, num_labels: int, training: bool,
                    need_backward: bool, loss_mode: Optional[int] = None):
        """pooled -> dropout -> Linear(H, num_labels) -> loss (ref:stonkgs_finetuning.py:310-338). `loss_mode` None:
        CrossEntropyLoss on int64 labels [B]; hip.LOSS_MSE / LOSS_MSE_BROADCAST / LOSS_BCE: the regression and
        multi-label branches on fp32 labels."""
        cfg = self.cfg
        H = cfg.hidden_size
        B = input_ids.shape[0]
        st = hip.stream_ptr()
        f = self.P.view
        save: dict = {}
        # (the classification head hands out no per-position output: the encoder runs on the packed rows, training or not)
        seq_out, pooled = self.encode(input_ids, attention_mask, token_type_ids, training, save, (None, None))
        p = cfg.hidden_dropout_prob if training else 0.0
        dropped = pooled
        if p > 0:
            dropped = self.buf("c.dropped", (B, H), F32)
            hip.call("stonk_dropout_f32", pooled.data_ptr(), dropped.data_ptr(), B * H, p, self.seed(300, 0), st)
        logits = self.buf("c.logits", (B, num_labels), F32)
        hip.call("stonk_small_linear_fwd", dropped.data_ptr(), H, f("classifier.weight").data_ptr(),
                 f("classifier.bias").data_ptr(), logits.data_ptr(), B, num_labels, H, hip.SMALL_X_F32, st)
        out = dict(logits=logits, pooler_output=pooled,
                   hidden_states=seq_out.view(B, cfg.max_position_embeddings, H) if save["plan"] is None else None)
        if labels is not None:
            dl = self.buf("c.dl", (B, num_labels), F32) if need_backward else None
            loss = self.buf("c.loss", (1,), F32)
            if loss_mode is None:
                acc = self.buf("c.acc", (2,), F32)
                acc.zero_()
                hip.call("stonk_nsp_xent_fwd_bwd", logits.data_ptr(), labels.data_ptr(), B, num_labels, acc.data_ptr(),
                         hip.ptr(dl), 1.0, self.err.data_ptr(), st)
                hip.call("stonk_ratio_f32", acc[0:1].data_ptr(), acc[1:2].data_ptr(), loss.data_ptr(), st)
            else:
                hip.call("stonk_elementwise_loss_fwd_bwd", logits.data_ptr(), labels.data_ptr(), B, num_labels, loss_mode,
                         loss.data_ptr(), hip.ptr(dl), 1.0, st)
            out["loss"] = loss[0]
            if need_backward:
                save.update(dl=dl, dropped=dropped, p_cls=p, num_labels=num_labels)
                self.saved = save
        return out

    def backward_cls(self, gscale: float = 1.0, on_segment_done: Optional[Callable[[str], None]] = None) -> None:
        sv = self.saved
        if sv is None:
            raise RuntimeError("backward_cls() without a training forward (labels are required)")
        self.saved = None
        self.wait_wt()
        cfg = self.cfg
        H, S = cfg.hidden_size, cfg.max_position_embeddings
        B, C = sv["B"], sv["num_labels"]
        st = hip.stream_ptr()
        f, g_ = self.P.view, self.P.grad_view
        notify = self._make_notify(on_segment_done)
        dl = sv["dl"]
        if gscale != 1.0:
            hip.call("stonk_scale_f32", dl.data_ptr(), dl.numel(), gscale, st)
        dpooled = self.buf("b.dpooled", (B, H), F32)
        hip.call("stonk_small_linear_bwd", dl.data_ptr(), 0, sv["dropped"].data_ptr(), H, f("classifier.weight").data_ptr(),
                 g_("classifier.weight").data_ptr(), g_("classifier.bias").data_ptr(), dpooled.data_ptr(), 0, 0, B, C, H,
                 hip.SMALL_X_F32, st)
        if sv["p_cls"] > 0:  # same (seed, index) mask as the forward
            hip.call("stonk_dropout_f32", dpooled.data_ptr(), dpooled.data_ptr(), B * H, sv["p_cls"], self.seed(300, 0), st)
        notify("classifier.bias")
        dseq = self.buf("b.dseq", (B * S, H))
        dseq[:sv["Th"]].zero_()  # only position 0 of every sequence receives a gradient (from the pooler)
        self.backward_encoder(dpooled, dseq, sv, notify)
