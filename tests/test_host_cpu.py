"""CPU (no GPU): host-side logic of the product - masking / batch schema against reference-made vectors, config,
learning-rate schedule, bucket planning, the C-ABI library's exported symbols, loud failure without a GPU."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

from stonkgs_amd import _hip
from stonkgs_amd import data as D
from stonkgs_amd.config import STonKGsConfig
from stonkgs_amd.stonkgs_pretraining import linear_schedule_lr, plan_buckets
from tests.golden_util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_masking_bit_exact_with_reference_vectors():
    gold = dict(np.load(GOLDEN + "/masking.npz"))
    for seed in (0, 1, 1234):
        random.seed(seed)
        t_in, t_lab = D.replace_mlm_tokens(list(range(1000, 1256)), 28996)
        e_in, e_lab = D.replace_mlm_tokens([(7 * i) % 175094 for i in range(256)], 175094)
        assert t_in == gold[f"text_in_{seed}"].tolist() and t_lab == gold[f"text_lab_{seed}"].tolist()
        assert e_in == gold[f"ent_in_{seed}"].tolist() and e_lab == gold[f"ent_lab_{seed}"].tolist()
        random.seed(seed)
        rows = [{"input_ids": list(range(i * 10, i * 10 + 8)), "attention_mask": [1] * 8,
                 "token_type_ids": [0] * 4 + [1] * 4, "masked_lm_labels": [-100] * 4, "ent_masked_lm_labels": [i] * 4,
                 "next_sentence_labels": 0} for i in range(8)]
        neg = D.add_negative_nsp_samples(rows, text_part_length=4)
        assert [r["input_ids"] for r in neg] == gold[f"neg_input_ids_{seed}"].tolist()
        assert [r["ent_masked_lm_labels"] for r in neg] == gold[f"neg_ent_labels_{seed}"].tolist()
        assert [r["next_sentence_labels"] for r in neg] == gold[f"neg_nsp_{seed}"].tolist()


def test_empty_and_short_sequences_mask_nothing():
    random.seed(0)
    assert D.replace_mlm_tokens([], 10) == ([], [])
    inp, lab = D.replace_mlm_tokens([5, 6, 7], 10)  # int(3 * 0.15) = 0 positions
    assert inp == [5, 6, 7] and lab == [-100] * 3


def test_synthetic_batch_schema():
    b = D.synthetic_batch(6, 28996, 175094, 512, seed=3)
    assert b["input_ids"].shape == (6, 512) and b["masked_lm_labels"].shape == (6, 256)
    assert (b["attention_mask"][:, 256:] == 1).all() and (b["token_type_ids"][:, 256:] == 1).all()
    assert ((b["ent_masked_lm_labels"] != -100).sum(1) == 38).all()  # int(256 * 0.15)
    assert (b["input_ids"][:, 383] == 102).sum() >= 4 and b["input_ids"].max() < 175094
    # the reference masks the PADDED text sequence (ref:indra_for_pretraining.py:195-218): every row has
    # int(256 * 0.15) = 38 text labels too, and a padded position can be one of them (label = [PAD] = 0)
    assert ((b["masked_lm_labels"] != -100).sum(1) == 38).all()
    pad = b["attention_mask"][:, :256] == 0
    lab_on_pad = b["masked_lm_labels"][pad]
    assert ((lab_on_pad == -100) | (lab_on_pad == 0)).all() and (lab_on_pad == 0).any()
    untouched = pad & (b["masked_lm_labels"] == -100)
    assert (b["input_ids"][:, :256][untouched] == 0).all()
    b2 = D.synthetic_batch(6, 28996, 175094, 512, seed=3)
    assert all(torch.equal(b[k], b2[k]) for k in b)
    ex = D.example_batch()
    assert ex["input_ids"].shape == (3, 512) and ex["attention_mask"][:, :256].sum(1).tolist() == [13, 15, 14]


def test_config_validation_and_roundtrip(tmp_path):
    c = STonKGsConfig()
    c.validate_for_hip()
    assert c.half_length == 256 and c.head_dim == 64
    with pytest.raises(ValueError):
        STonKGsConfig(hidden_size=64, num_attention_heads=4).validate_for_hip()
    c.update({"kg_vocab_size": 123})
    c.save_pretrained(str(tmp_path))
    assert STonKGsConfig.from_pretrained(str(tmp_path)).kg_vocab_size == 123
    with pytest.raises(FileNotFoundError):
        STonKGsConfig.from_pretrained("dmis-lab/biobert-v1.1")  # hub names cannot be resolved offline


def test_linear_schedule_matches_hf_lambda():
    for step in (0, 1, 100, 199, 200, 250):
        assert linear_schedule_lr(1e-4, step, 200) == pytest.approx(1e-4 * max(0.0, (200 - step) / 200))
    assert linear_schedule_lr(1.0, 5, 100, warmup=10) == pytest.approx(0.5)


def test_plan_buckets_covers_buffer_exactly():
    ends = [100, 150, 160, 400, 410, 1000]
    b = plan_buckets(ends, 200)
    assert b[0][0] == 0 and b[-1][1] == 1000
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    assert all(hi in ends for _, hi in b)
    assert plan_buckets([10], 100) == [(0, 10)] and plan_buckets([], 10) == []
    # tapered tail: the last 300 elements in buckets of >= 100 - the final bucket (exposed in full) is small; a sliver joins
    # its predecessor; everything before the tail keeps the large buckets
    ends = [400, 800, 900, 1000, 1100, 1200, 1210]
    t = plan_buckets(ends, 400, tail_elems=320, tail_bucket_elems=100)
    assert t == [(0, 400), (400, 800), (800, 900), (900, 1000), (1000, 1100), (1100, 1210)], t
    assert plan_buckets(ends, 400) == [(0, 400), (400, 800), (800, 1200), (1200, 1210)]


def test_library_exports_every_declared_symbol():
    """include/stonk_hip.h, the ctypes table and the built .so agree (no compute call: no GPU here)."""
    header = open(os.path.join(ROOT, "include", "stonk_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|void\*) (stonk_\w+)\(", header, flags=re.M))
    assert declared == set(_hip.exported_symbols())
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _hip.lib().stonk_abi_version() == 5
    assert _hip.lib().stonk_sumsq_workspace_floats() == 257   # (256 partial-sum slots + the ticket word)
    assert _hip.lib().stonk_layernorm_bwd_workspace_floats(32768, 768) == 1024 * 2 * 768   # (no GPU touched: a size query)
    assert _hip.lib().stonk_layernorm_bwd_workspace_floats(0, 768) == 0


def test_ctypes_constants_mirror_the_flag_header():
    """stonkgs_amd/_hip.py repeats the values of csrc/stonk_flags.h (what include/stonk_hip.h hands a C caller): every
    `#define STONK_<NAME> <value>` that has a Python twin must agree with it, and the kernel selectors must all have one."""
    text = open(os.path.join(ROOT, "stonkgs_amd", "csrc", "stonk_flags.h")).read()
    defs = {}
    for name, value in re.findall(r"^#define STONK_(\w+) +(\(1 << \d+\)|\d+)", text, flags=re.M):
        defs[name] = eval(value)
    assert len(defs) > 20
    twins = {n: getattr(_hip, n) for n in defs if hasattr(_hip, n)}
    assert twins and all(twins[n] == defs[n] for n in twins), {n: (twins[n], defs[n]) for n in twins if twins[n] != defs[n]}
    for n in defs:
        if n.startswith("GEMM_"):
            assert n in twins, n


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    lib = _hip.lib()
    assert lib.stonk_gemm_nt_bf16(0, 0, 0, 0, 0, 0, 1, 128, 64, 0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0.0, 0, 0, 0) == -1
    assert lib.stonk_gemm_nt_bf16(16, 64, 16, 64, 16, 128, 1, 128, 64, 0, 0, 0, 0, 0, 0, 1.0, 1, 0, 0, 0.0, 0, 9, 0) == -1   # unknown kernel
    assert lib.stonk_layernorm_fwd(16, 16, 16, 16, 0, 0, 4, 7, 1e-12, 0, 0.0, 0, 0) == -2
    assert lib.stonk_attention_fwd(16, 16, 16, 192, 0, 0, 0, 16, 64, 0, 1, 1, 100, 64, 0.125, 0.0, 0, 0) == -2
    assert lib.stonk_attention_fwd(16, 16, 16, 192, 0, 0, 0, 16, 64, 0, 1, 1, 4224, 64, 0.125, 0.0, 0, 0) == -2    # > 4096 keys: refused
    assert lib.stonk_attention_fwd(16, 16, 16, 192, 0, 16, 0, 16, 64, 0, 1, 1, 512, 64, 0.125, 0.0, 0, 0) == -1
    assert lib.stonk_attention_fwd(16, 16, 16, 192, 16, 0, 16, 16, 64, 0, 1, 1, 512, 64, 0.125, 0.0, 0, 0) == -1  # query limits need the packed layout  # packed rows need a mask
    # the gradient-exchange entry points: a null communicator / buffer, an unknown dtype
    assert lib.stonk_comm_allreduce_async(0, 16, 4, 0, 0) == -1 and lib.stonk_comm_wait(0, 0) == -1
    assert lib.stonk_comm_reduce_scatter_async(0, 16, 16, 4, 0, 0) == -1 and lib.stonk_comm_allgather_async(0, 16, 16, 4, 7, 0) == -1
    assert lib.stonk_comm_init(None, 1, 0, 0, 0) == -1 and lib.stonk_comm_destroy(0) == -1 and lib.stonk_comm_unique_id(0) == -1
    assert not lib.stonk_comm_stream(0)
    with pytest.raises(_hip.StonkHipError):
        _hip.check(-2, "x")


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_model_fails_loudly_without_gpu():
    from stonkgs_amd.stonkgs_model import STonKGsForPreTraining

    with pytest.raises(_hip.StonkHipError):
        STonKGsForPreTraining(STonKGsConfig())


def test_output_attentions_is_refused_not_answered_with_none():
    """ref:stonkgs_model.py:256 returns `outputs.attentions` (None unless the HF config asks for them). The gfx950 attention
    never materialises the probabilities, so a config that asks is refused where the config meets the kernels - also when
    it comes in as a dict / BertConfig-like object (the flag must not be dropped on the way)."""
    cfg = STonKGsConfig.from_any({"output_attentions": True, "hidden_size": 768})
    assert cfg.output_attentions is True
    with pytest.raises(NotImplementedError, match="output_attentions"):
        cfg.validate_for_hip()
    STonKGsConfig().validate_for_hip()


def test_embedding_helper_batching_is_host_only_logic():
    """Row collection / batching of stonkgs_for_embeddings (f2): DataFrame, list-of-dicts and dict-of-columns inputs,
    ragged last batch, optional columns, index selection - no GPU needed."""
    import pandas as pd

    from stonkgs_amd.stonkgs_for_embeddings import _batches, _rows_of

    rows = [{"input_ids": [i, i + 1, i + 2], "attention_mask": [1, 1, 0], "token_type_ids": [0, 0, 1], "extra": 7}
            for i in range(5)]
    df = pd.DataFrame(rows)
    for data in (df, rows, {k: [r[k] for r in rows] for k in rows[0]}):
        got = _rows_of(data, [4, 0])
        assert [list(r["input_ids"]) for r in got] == [[4, 5, 6], [0, 1, 2]]
    sizes = [b[0].shape[0] for b in _batches(_rows_of(df, None), 2)]
    assert sizes == [2, 2, 1]
    ids, am, tt = next(_batches(_rows_of(df, None), 4))
    assert ids.dtype == torch.long and ids.tolist()[3] == [3, 4, 5] and am.tolist()[0] == [1, 1, 0] and tt is not None
    ids, am, tt = next(_batches([{"input_ids": [1, 2]}], 8))
    assert am is None and tt is None
    with pytest.raises(KeyError):
        next(_batches([{"attention_mask": [1]}], 1))
    with pytest.raises(ValueError):
        next(_batches(rows, 0))


def test_cv_splits_match_reference_vectors_and_weighted_f1_matches_sklearn():
    """f4: get_train_test_splits against vectors made by the REFERENCE's own function (tests/golden/g7_splits.npz,
    oracle/make_golden.py splits) and the weighted F1 against scikit-learn, which the reference calls."""
    from sklearn.metrics import f1_score

    from stonkgs_amd.stonkgs_finetuning import INDRADataset, get_train_test_splits, weighted_f1_score

    gold = dict(np.load(GOLDEN + "/g7_splits.npz"))
    for name, kw in (("plain", {}), ("cut", {"max_dataset_size": 40}), ("single", {"n_splits": 1}),
                     ("three", {"n_splits": 3, "random_seed": 7})):
        out = get_train_test_splits({"labels": gold[f"{name}_labels"]}, **kw)
        assert len(out) == int(gold[f"{name}_n"])
        for i, d in enumerate(out):
            assert np.array_equal(d["train_idx"], gold[f"{name}_train_{i}"]), (name, i)
            assert np.array_equal(d["test_idx"], gold[f"{name}_test_{i}"]), (name, i)
    rng = np.random.RandomState(0)
    for _ in range(20):
        k = rng.randint(2, 6)
        yt, yp = rng.randint(0, k, 50), rng.randint(0, k, 50)
        assert weighted_f1_score(yt, yp) == pytest.approx(f1_score(yt, yp, average="weighted"), abs=1e-12)
    assert weighted_f1_score([0, 0, 1], [0, 0, 0]) == pytest.approx(f1_score([0, 0, 1], [0, 0, 0], average="weighted"))
    ds = INDRADataset({"input_ids": [[1, 2], [3, 4]], "attention_mask": [[1, 1], [1, 0]], "other": [0, 0]}, [1, 0])
    assert len(ds) == 2 and set(ds[1]) == {"input_ids", "attention_mask", "labels"} and int(ds[0]["labels"]) == 1


def test_pretrain_refuses_a_non_empty_output_dir_without_checkpoint(tmp_path):
    """ref:stonkgs_pretraining.py:203-207: an existing, non-empty training_dir with no checkpoint in it raises unless
    overwrite_output_dir; get_last_checkpoint picks the highest step (host logic: no model is touched before the check)."""
    from stonkgs_amd.stonkgs_pretraining import get_last_checkpoint, pretrain_stonkgs

    (tmp_path / "stale.txt").write_text("x")
    with pytest.raises(ValueError):
        pretrain_stonkgs(object(), [], training_dir=str(tmp_path))
    assert get_last_checkpoint(str(tmp_path)) is None and get_last_checkpoint(str(tmp_path / "missing")) is None
    for n in (2, 10, 4):
        (tmp_path / f"checkpoint-{n}").mkdir()
    (tmp_path / "checkpoint-final").mkdir()   # not a step directory
    assert get_last_checkpoint(str(tmp_path)).endswith("checkpoint-10")


def test_pretrain_driver_takes_the_reference_keywords(tmp_path, monkeypatch):
    """ref:stonkgs_pretraining.py:103-120: batch_size, deepspeed, fp16, lr, dataloader_num_workers,
    gradient_accumulation_steps, logging_steps, max_steps, overwrite_output_dir, save_limit, save_steps, training_dir - same
    names and defaults; `deepspeed=True` reaches the sharded optimizer, `save_steps` / `save_limit` the checkpoint cadence,
    `fp16=False` (an fp32 step) is refused. Host logic only: the Trainer is replaced by a recorder."""
    import inspect

    from stonkgs_amd import stonkgs_pretraining as sp

    sig = inspect.signature(sp.pretrain_stonkgs)
    want = dict(batch_size=8, deepspeed=False, fp16=True, lr=1e-4, dataloader_num_workers=2, gradient_accumulation_steps=1,
                logging_steps=100, max_steps=10000, overwrite_output_dir=False, save_limit=5, save_steps=5000)
    for k, v in want.items():
        assert sig.parameters[k].default == v, k
    seen = {}

    class Recorder:
        def __init__(self, model, args, train_dataset):
            seen["args"] = args

        def train(self, resume_from_checkpoint=None):
            seen["resume"] = resume_from_checkpoint
            return {}

        def save_model(self):
            seen["saved"] = True

    monkeypatch.setattr(sp, "Trainer", Recorder)
    out = tmp_path / "run"
    sp.pretrain_stonkgs(object(), [], batch_size=4, deepspeed=True, lr=3e-5, max_steps=77, save_steps=11, save_limit=2,
                        logging_steps=5, gradient_accumulation_steps=3, training_dir=str(out))
    a = seen["args"]
    assert (a.shard_optimizer, a.save_steps, a.save_total_limit, a.max_steps, a.learning_rate, a.logging_steps,
            a.per_device_train_batch_size, a.gradient_accumulation_steps) == (True, 11, 2, 77, 3e-5, 5, 4, 3)
    assert seen["resume"] is None and seen["saved"]
    with pytest.raises(ValueError, match="fp16=False"):
        sp.pretrain_stonkgs(object(), [], fp16=False, training_dir=str(out))
    # a non-empty directory: refused, unless overwrite_output_dir (which also ignores checkpoints, as the reference :197-201)
    out.mkdir(exist_ok=True)
    (out / "checkpoint-3").mkdir()
    sp.pretrain_stonkgs(object(), [], training_dir=str(out))
    assert seen["resume"].endswith("checkpoint-3")
    sp.pretrain_stonkgs(object(), [], training_dir=str(out), overwrite_output_dir=True)
    assert seen["resume"] is None


def test_bench_contract_constants():
    """bench.py reports BASELINE.json's metric under that exact name, defaults to one GPU and a short run, and the README
    generator finds every artefact it quotes (no GPU: only the module's constants and argument defaults are touched)."""
    import importlib
    import json
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert f'"metric": "{metric}"' in open(os.path.join(ROOT, "bench.py")).read()
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert a.gpus == 1 and 0 < a.steps <= 50 and 0 < a.warmup <= 10 and a.batch == 64
    assert bench.PEAK_BF16_TFLOPS == 2500.0
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_final_bench.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "encoder_path"):
        assert key in line, key
    assert line["metric"] == metric and line["config"]["workload"] and line["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "alone", "timing"} <= set(line["roofline"])
    alone = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_roofline_alone.json")))["roofline"]
    assert alone["alone"]["frac"] > alone["frac"]   # the in-step figure is the headline; the alone one is a labelled extra
    assert {"value", "unit", "cores", "kind", "sample", "at_8_threads"} <= set(line["cpu_baseline"])


def test_bench_spawns_ranks_and_refuses_a_mismatched_launcher(monkeypatch):
    """`bench.py --gpus N` outside a launcher starts N ranks it owns (stonkgs_amd/launch.py: no GPU touched by the parent,
    a wall limit, the ranks' exit code relayed); inside a launcher whose WORLD_SIZE differs from --gpus it refuses to report."""
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    launch = importlib.import_module("stonkgs_amd.launch")
    calls = {}

    def fake_run_ranks(world, argv, timeout, env=None, relay=False, **kw):
        calls.update(world=world, argv=list(argv), timeout=timeout, relay=relay)
        return launch.RankResult(7, False, [""] * world, [""] * world, [7] + [0] * (world - 1))

    monkeypatch.setattr(launch, "run_ranks", fake_run_ranks)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("STONK_DIST_BACKEND", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7   # the ranks' exit code is relayed
    assert calls["world"] == 4 and calls["relay"] and 0 < calls["timeout"] < 3600
    assert calls["argv"][0] == sys.executable and calls["argv"][1].endswith("bench.py")
    assert calls["argv"][-4:] == ["--gpus", "4", "--steps", "2"]
    env = launch.rank_env(2, 4, 12345)
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == \
        ("2", "2", "4", "127.0.0.1", "12345")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["GLOO_SOCKET_IFNAME"] == "lo"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:   # fewer GPUs than ranks: loud failure, no 1-rank run under an N-GPU label
        bench.main()
    assert e.value.code == 2
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2


def test_embedding_row_builder_matches_the_reference():
    """Row f1 (ref:src/stonkgs/models/stonkgs_for_embeddings.py:50-155): (source, target, evidence) -> model rows, against
    rows the REFERENCE's own function yielded (oracle/make_golden.py f3: local tokenizer directory, the G8 node table, a
    random-walk TSV, random.seed(5)): tokenised text, walks in TSV-row index space, [UNK] walks for unknown nodes, masks,
    labels - every integer equal."""
    import random

    import pandas as pd

    from stonkgs_amd.stonkgs_for_embeddings import preprocess_df_for_embeddings, preprocess_df_for_embeddings_iter

    gold = dict(np.load(GOLDEN + "/g10_embedding_rows.npz"))
    rows = list(zip(gold["sources"].tolist(), gold["targets"].tolist(), gold["evidences"].tolist()))
    kw = dict(embedding_name_to_vector_path=GOLDEN + "/g8_table.tsv", embedding_name_to_random_walk_path=GOLDEN + "/g10_walks.tsv")
    random.seed(5)
    out = list(preprocess_df_for_embeddings_iter(rows, nlp_model_type=GOLDEN + "/g10_tokenizer", **kw))
    for k in ("input_ids", "attention_mask", "token_type_ids", "masked_lm_labels", "ent_masked_lm_labels", "next_sentence_labels"):
        assert np.array_equal(np.array([r[k] for r in out]), gold[k]), k
    assert (gold["input_ids"][1, 256 + 128:256 + 255] == 100).sum() > 100     # the unknown target's walk is [UNK] ids
    # the vocab-file spelling and the DataFrame wrapper give the same rows
    random.seed(5)
    df = preprocess_df_for_embeddings(pd.DataFrame(rows, columns=["source", "target", "evidence"]),
                                      vocab_file_path=GOLDEN + "/g10_tokenizer/vocab.txt", **kw)
    assert df["input_ids"].tolist() == gold["input_ids"].tolist() and list(df.columns)[:3] == ["input_ids", "attention_mask", "token_type_ids"]
    with pytest.raises(FileNotFoundError):
        next(preprocess_df_for_embeddings_iter(rows, nlp_model_type="dmis-lab/biobert-v1.1", **kw))


def test_unpad_plan_restatement_keeps_exactly_what_the_loss_reads():
    """oracle/masking_oracle.unpad_plan (the checker of csrc/unpad.hip): live keys, labelled positions and position 0 are
    kept, nothing else - a sequence's READ rows (labelled + position 0) first, then its other rows, each group in position
    order; a sequence without live keys keeps everything; the maps invert each other."""
    import numpy as np

    from oracle import masking_oracle as mo

    B, S = 4, 16
    half = S // 2
    am = np.ones((B, S), dtype=np.int64)
    am[0, 3:half] = 0
    am[1, :] = 0
    am[2, 0] = 0                    # position 0 masked: still kept (the pooler reads it)
    tl = np.full((B, half), -100)
    el = np.full((B, half), -100)
    tl[0, 5] = 17                   # a labelled padding position
    el[0, 2] = 4                    # a labelled entity position (s = 10)
    rop, por, cu, rm, rr, rofp, ro = mo.unpad_plan(am, tl, el, read=True)
    order0 = [0, 5, 10] + [1, 2] + [8, 9] + list(range(11, S))       # read rows first, then the other kept rows
    assert cu.tolist() == [0, len(order0), len(order0) + S, len(order0) + 2 * S, len(order0) + 3 * S]
    assert por[:len(order0)].tolist() == order0 and rop[3] == -1 and rop[5] == 1 and rop[1] == 3
    assert rm[:len(order0)].tolist() == [1, 0, 1] + [1] * (len(order0) - 3)   # the labelled pad row is a query, never a key
    total = int(cu[-1])
    assert (rop[por[:total]] == np.arange(total)).all() and (por[total:] == -1).all()
    assert rm[cu[2]] == 0 and rop[2 * S] == cu[2]
    assert ro.tolist() == [0, 3, 4, 5, 6] and rr[:6].tolist() == [0, 1, 2, cu[1], cu[2], cu[3]]
    assert rofp[5] == 1 and rofp[10] == 2 and rofp[1] == -1 and (rr[6:] == -1).all()
    for b in range(B):                                                  # a sequence's read rows are its first rows
        assert (rr[ro[b]:ro[b + 1]] == cu[b] + np.arange(ro[b + 1] - ro[b])).all()


def test_asan_host_build_of_the_launchers():
    """`make asan`: the launchers' HOST code (argument validation, launch geometry, descriptor handling) under
    AddressSanitizer - the device code is built as usual and nothing is launched (no GPU here; GPU ASAN is not available on
    this pool). The argument-validation and symbol tests above run again in a child process against that build, with the
    sanitizer runtime preloaded; any host-side out-of-bounds access or use-after-free in a launcher aborts the child."""
    import shutil
    import subprocess
    import sys

    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    r = subprocess.run(["make", "-C", ROOT, "asan", "-j8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "build", "asan", "libstonk_hip_asan.so")
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"],
                        capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no sanitizer runtime")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", STONK_HIP_LIB=lib)
    child = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_cpu.py"), "-q", "-x", "-k",
                            "exports_every or rejected", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                           timeout=600)
    assert child.returncode == 0 and "2 passed" in child.stdout, child.stdout[-2000:] + child.stderr[-2000:]
    assert "AddressSanitizer" not in child.stderr


def test_written_out_kernels_use_no_scratch():
    """The hand-laid-out register files of the written-out GEMM kernels and the attention kernels must not spill: a spill
    keeps every result right and costs 10-20 % of a launch (round 4: a few more scalars alive across the K loop put every
    256-wide gemm_a4 instance into scratch). Read from the built objects' code-object metadata (tools/kernel_resources.py);
    the weight-gradient kernel's 272 bytes (reloaded once per work item, outside the loop) are its known state."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    csrc = os.path.join(root, "stonkgs_amd", "csrc")
    import subprocess

    # (objects are build products, not in the history: a no-op when __graft_entry__.build() has run)
    r = subprocess.run(["make", "-C", root, "-j8"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    a4 = kr.kernel_resources(os.path.join(csrc, "gemm_a4.o"))
    assert len(a4) >= 17 and all("gemm_a4_kernel" in k["name"] for k in a4)
    bad = [(k["name"], k["scratch"]) for k in a4 if k["scratch"] or k.get("vgpr_spill", 0)]
    assert not bad, bad
    tn = kr.kernel_resources(os.path.join(csrc, "gemm_tn_a4.o"))
    assert [k["scratch"] <= 272 for k in tn] == [True]
    attn = kr.kernel_resources(os.path.join(csrc, "attention.o"))
    bad = [(k["name"], k["scratch"]) for k in attn if k["scratch"] or k.get("vgpr_spill", 0)]
    assert len(attn) == 13 and not bad, bad
