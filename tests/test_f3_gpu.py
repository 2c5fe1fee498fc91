"""Row f3 (checkpoint compatibility) on the GPU, against artefacts the REFERENCE side produced (oracle/make_golden.py f3):
  * `kg_embedding_dict_path=<TSV>` read with the reference's prepare_df semantics (ref:src/stonkgs/models/
    kg_baseline_model.py:270-280, used at ref:stonkgs_model.py:93), names -> kg_idx_to_name, rows -> entity table (Q1);
  * `nlp_model_type=<local dir>`: the frozen LM backbone from a directory HF's BertModel.save_pretrained wrote
    (ref:stonkgs_model.py:107);
  * `from_pretrained(<dir the reference-side model wrote>)` in HF layout (ref:src/stonkgs/api/api.py:104-112), also as
    safetensors with the alias keys a safetensors writer drops, and a pre-training checkpoint loaded into the fine-tuning
    class with a fresh classifier (ref:src/stonkgs/models/stonkgs_finetuning.py:404-407)."""
import json
import os
import shutil
import warnings

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu
BATCH_KEYS = ("input_ids", "attention_mask", "token_type_ids", "masked_lm_labels", "ent_masked_lm_labels",
              "next_sentence_labels")


@pytest.fixture(scope="module")
def g9():
    gold = dict(np.load(os.path.join(GOLDEN, "g9_ref_checkpoint.npz")))
    meta = json.load(open(os.path.join(GOLDEN, "g9_ref_checkpoint.json")))
    return gold, meta, {k: torch.from_numpy(gold[k]) for k in BATCH_KEYS}


def _check_against_reference(model, gold, batch):
    model.eval()
    with torch.no_grad():
        out = model(**batch, return_dict=True)
    model.engine.check_errors()
    assert abs(float(out.loss) - float(gold["loss"])) < 1e-2
    assert torch.allclose(out.pooler_output.cpu(), torch.from_numpy(gold["pooler_output"]), atol=2e-2)
    el = out.prediction_logits[1][batch["ent_masked_lm_labels"].cuda() != -100].cpu()
    ref = torch.from_numpy(gold["ent_logits_lab"])
    assert float((el - ref).norm() / ref.norm()) < 3e-2


def test_reference_written_checkpoint_with_tsv_table(hip, g9):
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    gold, meta, batch = g9
    model = STonKGsForPreTraining.from_pretrained(os.path.join(GOLDEN, "g9_ref_checkpoint"),
                                                  kg_embedding_dict_path=os.path.join(GOLDEN, "g8_table.tsv"))
    assert model.config.kg_vocab_size == 120 and model.config.num_hidden_layers == 1
    # names in file order on the model's index space (quirk Q1: 100 / 102 / 103 stay free)
    for idx, name in meta["idx_to_name_head"].items():
        assert model.kg_idx_to_name[int(idx)] == name
    assert 100 not in model.kg_idx_to_name.keys() and len(model.kg_idx_to_name) == 120
    for e in (7, 101, 104):        # fp64 TSV value -> fp32 round-to-nearest, bit for bit (ref:stonkgs_model.py:193-200)
        assert torch.equal(model.kg_backbone[e].cpu(), torch.from_numpy(gold[f"table_row_{e}"].astype(np.float32)))
    assert torch.allclose(model.kg_backbone[102].cpu(), torch.from_numpy(gold["special_102"]), atol=2e-2)
    _check_against_reference(model, gold, batch)
    # every tensor of the reference's state dict arrived (dead parameters included)
    sd_ref = torch.load(os.path.join(GOLDEN, "g9_ref_checkpoint", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    sd = model.state_dict()
    for k, v in sd_ref.items():
        if "position_ids" not in k:
            assert torch.equal(sd[k].cpu(), v), k


def test_safetensors_checkpoint_without_alias_keys(hip, g9, tmp_path):
    from safetensors.torch import save_file

    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    gold, meta, batch = g9
    sd_ref = torch.load(os.path.join(GOLDEN, "g9_ref_checkpoint", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    kept = meta["safetensors_keys"]          # what safetensors' shared-tensor handling keeps for the reference model
    assert set(meta["state_dict_keys"]) - set(kept)      # (it does drop aliases)
    shutil.copy(os.path.join(GOLDEN, "g9_ref_checkpoint", "config.json"), tmp_path / "config.json")
    save_file({k: sd_ref[k].clone().contiguous() for k in kept}, str(tmp_path / "model.safetensors"), metadata={"format": "pt"})
    tab = dict(np.load(os.path.join(GOLDEN, "g8_table.npz")))
    model = STonKGsForPreTraining.from_pretrained(str(tmp_path), kg_embeddings=torch.from_numpy(tab["values"]))
    _check_against_reference(model, gold, batch)
    # a checkpoint that lacks a LIVE tensor is still refused
    save_file({k: sd_ref[k].clone().contiguous() for k in kept if k != "bert.pooler.dense.weight"},
              str(tmp_path / "model.safetensors"), metadata={"format": "pt"})
    with pytest.raises(KeyError):
        STonKGsForPreTraining.from_pretrained(str(tmp_path), kg_embeddings=torch.from_numpy(tab["values"]))


def test_local_backbone_directory(hip, g9):
    """nlp_model_type = local directory: config and frozen LM weights come from it (ref:stonkgs_model.py:96,107); with the
    trainable weights loaded on top the model reproduces the reference's outputs."""
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    gold, meta, batch = g9
    model = STonKGsForPreTraining(None, nlp_model_type=os.path.join(GOLDEN, "g9_lm_backbone"),
                                  kg_embedding_dict_path=os.path.join(GOLDEN, "g8_table.tsv"))
    assert model.config.hidden_size == 128 and model.config.kg_vocab_size == 120
    sd_ref = torch.load(os.path.join(GOLDEN, "g9_ref_checkpoint", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    bb = {k: v for k, v in model.state_dict().items() if k.startswith("lm_backbone.")}
    assert bb
    for k, v in bb.items():
        assert torch.equal(v.cpu(), sd_ref[k]), k           # the backbone came from the directory alone
    assert torch.allclose(model.kg_backbone[102].cpu(), torch.from_numpy(gold["special_102"]), atol=2e-2)
    model.load_state_dict({k: v for k, v in sd_ref.items() if not k.startswith("lm_backbone.")}, strict=False)
    _check_against_reference(model, gold, batch)


def test_pretraining_checkpoint_into_the_finetuning_class(hip, g9):
    """`STonKGsForSequenceClassification.from_pretrained(pretraining_dir, num_labels=n)` - the reference's only fine-tuning
    entry: the classifier is missing from the checkpoint, stays freshly initialised, and a warning says so."""
    from stonkgs_amd.stonkgs_model import STonKGsForSequenceClassification

    gold, meta, batch = g9
    tab = dict(np.load(os.path.join(GOLDEN, "g8_table.npz")))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = STonKGsForSequenceClassification.from_pretrained(os.path.join(GOLDEN, "g9_ref_checkpoint"), num_labels=3,
                                                                 kg_embeddings=torch.from_numpy(tab["values"]))
    assert any("newly initialized" in str(x.message) and "classifier.weight" in str(x.message) for x in w)
    assert model.num_labels == 3 and tuple(model.classifier.weight.shape) == (3, 128)
    assert float(model.classifier.weight.abs().sum()) > 0
    model.eval()
    with torch.no_grad():
        out = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                    token_type_ids=batch["token_type_ids"], labels=torch.tensor([0, 2, 1]), return_dict=True)
    assert out.logits.shape == (3, 3) and torch.isfinite(out.loss)
    sd_ref = torch.load(os.path.join(GOLDEN, "g9_ref_checkpoint", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    assert torch.equal(model.state_dict()["bert.encoder.layer.0.output.dense.weight"].cpu(),
                       sd_ref["bert.encoder.layer.0.output.dense.weight"])
