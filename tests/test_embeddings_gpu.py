"""GPU parity of the batched embedding-extraction / inference helpers - the callers either side
of the hot path that the reference runs as bs = 1 Python loops (ref:stonkgs_for_embeddings.py:158-186, ref:api.py:308-336):
against the reference-made golden vectors (G2 pooler_output, G6 logits), against the full pre-training forward, and
batched vs one row at a time."""
import numpy as np
import pytest
import torch

from tests.golden_util import load_case
from tests.test_finetune_gpu import _build_cls, _g6
from tests.test_model_gpu import _build, _rel

pytestmark = pytest.mark.gpu


def test_get_stonkgs_embeddings_matches_reference_pooler_output(hip):
    import pandas as pd

    from stonkgs_amd.stonkgs_for_embeddings import get_stonkgs_embeddings

    cfg, sd, tsv_rows, batch, gold, meta = load_case("g2_hipsmall")
    model = _build(cfg, sd, tsv_rows)
    model.eval()
    B = batch["input_ids"].shape[0]
    # the reference's input: a pre-processed DataFrame whose rows also carry the (unused) label columns
    df = pd.DataFrame({k: [v[i].tolist() for i in range(B)] for k, v in batch.items()
                       if k != "next_sentence_labels"})
    df["next_sentence_labels"] = batch["next_sentence_labels"].tolist()
    emb = get_stonkgs_embeddings(df, model=model, batch_size=2)          # ragged last batch when B is odd
    assert list(emb.columns) == ["embedding"] and len(emb) == B
    got = torch.tensor(emb["embedding"].tolist())
    assert got.shape == (B, cfg.hidden_size)
    assert _rel(got, gold["pooler_output"]) < 2e-2                        # vs the REFERENCE's pooler_output
    # the full forward of the pre-training model (every position a row, both decoders) gives the same vector up to the
    # rounding the packed layout moves around (the extraction runs on the live rows; its last layer on position 0 alone)
    assert model.engine.rows_executed[5] < model.engine.rows_executed[1] // 4       # the extraction did run packed
    with torch.no_grad():
        full = model(**batch, return_dict=True).pooler_output
    assert _rel(got, full.cpu()) < 5e-3
    # one row at a time == batched, and list_of_indices selects / orders rows
    one = get_stonkgs_embeddings(df, model=model, list_of_indices=[B - 1, 0], batch_size=1)
    torch.testing.assert_close(torch.tensor(one["embedding"].tolist()), got[[B - 1, 0]], rtol=0, atol=0)
    model.engine.check_errors()


def test_encode_rejects_bad_shapes_and_unknown_entities(hip):
    cfg, sd, tsv_rows, batch, gold, meta = load_case("g2_hipsmall")
    model = _build(cfg, sd, tsv_rows)
    with pytest.raises(ValueError):
        model.encode(batch["input_ids"][:, :100])
    bad = batch["input_ids"].clone()
    bad[0, cfg.max_position_embeddings - 1] = cfg.kg_vocab_size + 50      # not in the table: the reference raises KeyError
    with pytest.raises(KeyError):
        model.encode(bad, batch["attention_mask"], batch["token_type_ids"])


def test_infer_matches_reference_logits_and_softmax(hip):
    from stonkgs_amd.stonkgs_for_embeddings import infer, infer_iter

    cfg, sd, rows, gold, meta = _g6()
    model = _build_cls(cfg, sd, rows, meta["num_labels"])
    B = gold["input_ids"].shape[0]
    data = [{k: gold[k][i].tolist() for k in ("input_ids", "attention_mask", "token_type_ids")} for i in range(B)]
    model.train()                                                        # infer() must not depend on the mode it finds
    raw, probs = infer(model, data, batch_size=2)
    assert model.training
    assert len(raw) == B and len(probs) == B and all(r.logits.shape == (1, meta["num_labels"]) for r in raw)
    logits = torch.cat([r.logits for r in raw]).float().cpu()
    assert _rel(logits, gold["logits"]) < 3e-2                            # vs the REFERENCE's logits
    ref_p = torch.softmax(torch.from_numpy(gold["logits"]).float(), dim=1)
    np.testing.assert_allclose(np.array(probs), ref_p.numpy(), atol=2e-2)
    assert all(abs(sum(p) - 1.0) < 1e-5 for p in probs)
    # batched == row by row
    one = [p for _, p in infer_iter(model, data, batch_size=1)]
    np.testing.assert_allclose(np.array(one), np.array(probs), rtol=0, atol=1e-6)
