"""The autograd bridge: `loss = model(**batch)[0]; loss.backward()` driven by somebody else's loop and optimizer - what
the reference does through HF Trainer (ref:src/stonkgs/models/stonkgs_pretraining.py:215-223 -> hf:trainer.py
training_step / clip_grad_norm_ / optimizer.step / model.zero_grad).

The engine's GEMMs read bf16 mirrors and W^T copies of the fp32 masters and its backward accumulates into a flat
gradient buffer; both must follow an EXTERNAL optimizer and an external zero_grad."""
import numpy as np
import pytest
import torch

from oracle import stonkgs_oracle as orc
from tests.golden_util import load_case

pytestmark = pytest.mark.gpu


def _build(cfg, sd, tsv_rows):
    from stonkgs_amd.config import STonKGsConfig
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    c = STonKGsConfig(**{k: getattr(cfg, k) for k in ("vocab_size", "kg_vocab_size", "hidden_size", "num_hidden_layers",
                                                      "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "layer_norm_eps")},
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = STonKGsForPreTraining(c, kg_embeddings=tsv_rows)
    model.load_state_dict(sd, strict=False)
    return model


def _oracle_steps(cfg, sd, tsv_rows, batches, lr, max_steps):
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    osd = {k: v.clone() for k, v in sd.items()}
    state = orc.AdamState()
    return [float(orc.train_step(osd, cfg, table, b, state, base_lr=lr, max_steps=max_steps)["loss"]) for b in batches], osd


@pytest.mark.parametrize("set_to_none", [True, False])
def test_torch_optimizer_and_zero_grad_drive_the_model(hip, set_to_none):
    """Four steps with torch.optim.AdamW + clip_grad_norm_ + model.zero_grad(): the loss must follow the oracle's curve
    (it would stay on the initial weights' loss if the bf16 copies were stale, and the gradient norm would grow step by
    step if the flat buffer kept accumulating)."""
    from stonkgs_amd.data import synthetic_batch

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    lr, steps = 2e-3, 4
    batches = [synthetic_batch(3, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=300 + i, min_text=16)
               for i in range(steps)]
    ref_losses, osd = _oracle_steps(cfg, sd, tsv_rows, batches, lr, 200)
    model = _build(cfg, sd, tsv_rows)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: max(0.0, (200 - s) / 200))
    losses, norms = [], []
    for b in batches:
        loss = model(**b)[0]
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, 1.0)))
        opt.step()
        sched.step()
        model.zero_grad(set_to_none=set_to_none)
        losses.append(float(loss))
    model.engine.check_errors()
    print("external optimizer:", losses, "oracle:", ref_losses, "grad norms:", norms)
    np.testing.assert_allclose(losses, ref_losses, atol=1e-2)
    assert ref_losses[0] - ref_losses[-1] > 0.3               # the weights did move: a stale-copy run would not follow
    p = dict(model.named_parameters())
    for k in ("bert.encoder.layer.1.intermediate.dense.weight", "cls.predictions.text_decoder.weight"):
        got, ref = p[k].detach().cpu() - sd[k], osd[k] - sd[k]
        assert torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item() > 0.9, k
    # the optimizer's own zero_grad(set_to_none=True) only drops the attributes: the next backward starts from zero too
    opt.zero_grad(set_to_none=True)
    loss = model(**batches[0])[0]
    loss.backward()
    g1 = float(torch.nn.utils.clip_grad_norm_(params, 1e9))
    opt.zero_grad(set_to_none=True)
    loss = model(**batches[0])[0]
    loss.backward()
    g2 = float(torch.nn.utils.clip_grad_norm_(params, 1e9))
    assert g2 == pytest.approx(g1, rel=1e-3)
    # ... while keeping the attributes accumulates, as torch does
    loss = model(**batches[0])[0]
    loss.backward()
    g3 = float(torch.nn.utils.clip_grad_norm_(params, 1e9))
    assert g3 == pytest.approx(2 * g1, rel=1e-2)


def test_fused_trainer_and_external_optimizer_agree(hip):
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    batches = [synthetic_batch(3, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=310 + i, min_text=16)
               for i in range(3)]
    fused = _build(cfg, sd, tsv_rows)
    tr = Trainer(fused, TrainingArguments(max_steps=200, learning_rate=1e-3, per_device_train_batch_size=3))
    lf = [float(tr.training_step(fused, b)) for b in batches]
    ext = _build(cfg, sd, tsv_rows)
    ext.train()
    params = [p for p in ext.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: max(0.0, (200 - s) / 200))
    le = []
    for b in batches:
        loss = ext(**b)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        sched.step()
        ext.zero_grad()
        le.append(float(loss))
    np.testing.assert_allclose(le, lf, atol=3e-3)


def test_hf_trainer_two_steps(hip, tmp_path):
    """The real `transformers.Trainer` (as ref:stonkgs_pretraining.py:215-223 builds it: model, TrainingArguments, dataset)
    for two optimizer steps against the fused Trainer on the same batches."""
    transformers = pytest.importorskip("transformers")
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_pretraining import Trainer as FusedTrainer
    from stonkgs_amd.stonkgs_pretraining import TrainingArguments as FusedArguments

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    b0 = synthetic_batch(4, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=320, min_text=16)
    b1 = synthetic_batch(4, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=321, min_text=16)
    rows = [{k: v[i] for k, v in b.items()} for b in (b0, b1) for i in range(4)]      # dataset order = batch order

    class Rows(torch.utils.data.Dataset):
        def __len__(self):
            return len(rows)

        def __getitem__(self, i):
            return rows[i]

    model = _build(cfg, sd, tsv_rows)
    args = transformers.TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=4, max_steps=2,
                                          learning_rate=1e-3, logging_steps=1, save_strategy="no", report_to=[],
                                          remove_unused_columns=False, dataloader_pin_memory=False, seed=0,
                                          lr_scheduler_type="linear", warmup_steps=0, weight_decay=0.0, max_grad_norm=1.0)

    class InOrder(transformers.Trainer):     # the default sampler shuffles; the comparison needs the same batches
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    trainer = InOrder(model=model, args=args, train_dataset=Rows())
    trainer.train()
    hf_losses = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    fused = _build(cfg, sd, tsv_rows)
    ft = FusedTrainer(fused, FusedArguments(max_steps=2, learning_rate=1e-3, per_device_train_batch_size=4))
    fl = [float(ft.training_step(fused, b)) for b in (b0, b1)]
    print("HF Trainer losses", hf_losses, "fused", fl)
    assert len(hf_losses) == 2
    np.testing.assert_allclose(hf_losses, fl, atol=5e-3)
    a, b = dict(model.named_parameters()), dict(fused.named_parameters())
    for k in ("bert.encoder.layer.0.output.dense.weight", "cls.predictions.entity_decoder.weight", "bert.pooler.dense.bias"):
        d0 = (a[k].detach() - sd[k].cuda()).flatten()
        d1 = (b[k].detach() - sd[k].cuda()).flatten()
        assert torch.nn.functional.cosine_similarity(d0, d1, dim=0).item() > 0.9, k


def test_weight_decay_skips_biases_and_layernorm_like_hf(hip):
    """`TrainingArguments.weight_decay` (0 in the reference's run): the fused optimizer decays what HF Trainer decays - every
    parameter except biases and LayerNorm weights (hf:trainer.py get_decay_parameter_names) - checked against the oracle,
    whose grouping is pinned against torch.optim.AdamW in tests/test_oracle_golden.py. lr and decay are large on purpose."""
    from stonkgs_amd.data import synthetic_batch
    from stonkgs_amd.stonkgs_pretraining import Trainer, TrainingArguments

    cfg, sd, tsv_rows, _, _, _ = load_case("g2_hipsmall")
    batches = [synthetic_batch(3, cfg.vocab_size, cfg.kg_vocab_size, cfg.max_position_embeddings, seed=330 + i, min_text=16)
               for i in range(2)]
    model = _build(cfg, sd, tsv_rows)
    lr, wd = 1e-3, 10.0            # lr * wd = 1 % shrink per step: well above what two Adam steps of lr move a tensor's norm
    tr = Trainer(model, TrainingArguments(max_steps=200, learning_rate=lr, weight_decay=wd, per_device_train_batch_size=3))
    for b in batches:
        tr.training_step(model, b)
    with torch.no_grad():
        table = orc.build_kg_table(tsv_rows, orc.special_vectors(sd, cfg))
    osd = {k: v.clone() for k, v in sd.items()}
    state = orc.AdamState()
    for b in batches:
        orc.train_step(osd, cfg, table, b, state, base_lr=lr, max_steps=200, weight_decay=wd)
    p = dict(model.named_parameters())

    def shrink(t, k):
        return float(t.norm() / sd[k].norm())

    for k in ("bert.encoder.layer.0.output.dense.weight", "cls.predictions.entity_decoder.weight",
              "bert.embeddings.position_embeddings.weight", "cls.predictions.transform.dense.weight"):      # decayed
        got, ref = shrink(p[k].detach().cpu(), k), shrink(osd[k], k)
        assert abs(got - ref) < 3e-3 and 0.97 < ref < 0.99, (k, got, ref)
    for k in ("bert.encoder.layer.0.output.LayerNorm.weight", "cls.predictions.transform.LayerNorm.weight",
              "bert.embeddings.LayerNorm.weight"):                                                        # not decayed
        got, ref = shrink(p[k].detach().cpu(), k), shrink(osd[k], k)
        assert abs(got - ref) < 3e-3 and abs(ref - 1.0) < 5e-3, (k, got, ref)
    k = "bert.encoder.layer.1.intermediate.dense.bias"   # a bias: values ~0.02, moved by Adam only (<= 2 lr per element)
    assert float((p[k].detach().cpu() - sd[k]).abs().max()) <= 2.05 * lr
