"""GPU: the whole hot path (M3AETransformerSS.infer -> vqa_head -> BCE -> backward -> AdamW) through the C ABI
against (a) the CPU oracle on the same seeded inputs and (b) the committed reference fixtures.

Tolerances (stated per north_star): parity mode (fp32) logits rtol 1e-3 vs the reference fixtures, gradient norms
rtol 2e-3; perf mode (bf16 storage, fp32 accumulate / statistics / softmax) is held to bf16-appropriate bounds:
logits atol 0.05 (|logits| <= 0.7), loss rtol 2e-3, global grad-norm rtol 3e-2."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from m3ae_amd import synth  # noqa: E402
from m3ae_amd.modules import M3AETransformerSS  # noqa: E402
from oracle import m3ae_oracle as O  # noqa: E402
from oracle_util import (large1_batch, large1_config, finetune_vqa_rad_config, full_batch, load_golden, make_sd, oracle_cfg, tiny_batch,  # noqa: E402
                         tiny_config)


def to_dev(batch, dev="cuda"):
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev)
        elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(dev) for t in v]
        else:
            out[k] = v
    return out


def build(cfg, dtype):
    m = M3AETransformerSS(cfg)
    synth.fill_deterministic(m)
    m.finalize("cuda", dtype)
    m.eval()
    return m


def grad_report(m, g, rtol, floor):
    names, ref = g["grad_names"].tolist(), g["grad_norm"]
    gn = float(g["global_grad_norm"])
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n, r in zip(names, ref):
        mine = params[n].grad.double().norm().item()
        err = abs(mine - r)
        if err > rtol * r + floor * gn:
            rel = err / (r + 1e-30)
            if rel > worst[0]:
                worst = (rel, f"{n}: {mine:.6e} vs {r:.6e}")
    return worst


def test_tiny_fp32_parity_against_reference_fixture_and_oracle():
    cfg = tiny_config(compute_dtype="fp32")
    m = build(cfg, torch.float32)
    g = load_golden("tiny_vqa.npz")
    b = to_dev(tiny_batch())
    out = m.infer(b)
    np.testing.assert_allclose(out["multi_modal_text_feats"].detach().cpu().numpy(), g["text_feats"], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(out["multi_modal_image_feats"].detach().cpu().numpy(), g["image_feats"], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(out["multi_modal_cls_feats"].detach().cpu().numpy(), g["cls_feats"], rtol=1e-3, atol=1e-5)
    m.store.zero_grad()
    m.set_task()
    ret = m(b)
    np.testing.assert_allclose(ret["vqa_logits"].detach().cpu().numpy(), g["logits"], rtol=1e-3, atol=1e-5)
    loss = ret["vqa_loss"]
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * float(g["loss"])
    loss.backward()
    worst = grad_report(m, g, 2e-3, 1e-6)
    assert worst[0] == 0.0, worst
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad::"):
            ref = g[k]
            np.testing.assert_allclose(params[k[6:]].grad.cpu().numpy(), ref, rtol=5e-3,
                                       atol=1e-4 * np.abs(ref).max() + 1e-9, err_msg=k)
    # the oracle agrees with the fixture on the same inputs (checked on CPU in test_oracle_golden); spot-check here
    sd = make_sd(cfg)
    lo, logits_o, _ = O.training_loss(sd, oracle_cfg(cfg), tiny_batch())
    np.testing.assert_allclose(ret["vqa_logits"].detach().cpu().numpy(), logits_o.numpy(), rtol=1e-3, atol=1e-5)


def test_tiny_bf16_perf_mode_within_bf16_bounds():
    cfg = tiny_config(compute_dtype="bf16")
    m = build(cfg, torch.bfloat16)
    g = load_golden("tiny_vqa.npz")
    b = to_dev(tiny_batch())
    m.store.zero_grad()
    m.set_task()
    ret = m(b)
    logits = ret["vqa_logits"].detach().float().cpu().numpy()
    assert np.abs(logits - g["logits"]).max() < 0.05, np.abs(logits - g["logits"]).max()
    loss = ret["vqa_loss"]
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"])
    loss.backward()
    names, ref = g["grad_names"].tolist(), g["grad_norm"]
    params = dict(m.named_parameters())
    mine = np.array([params[n].grad.double().norm().item() for n in names])
    gn = np.sqrt((mine ** 2).sum())
    assert abs(gn - float(g["global_grad_norm"])) < 3e-2 * float(g["global_grad_norm"]), (gn, float(g["global_grad_norm"]))
    big = ref > 1e-3 * ref.max()
    rel = np.abs(mine[big] - ref[big]) / ref[big]
    assert rel.max() < 0.15, (rel.max(), np.array(names)[big][rel.argmax()])


def test_tiny_pretrain_objectives_fp32():
    cfg = tiny_config(compute_dtype="fp32", loss_names={"mlm": 1, "mim": 1, "itm": 1, "vqa": 0, "cls": 0, "irtr": 0},
                      mim_layer=1, mim_decoder_hidden_size=128, mim_decoder_num_layers=2, mim_decoder_num_heads=2)
    m = build(cfg, torch.float32)
    g = load_golden("tiny_pretrain.npz")
    b = to_dev(tiny_batch(pretrain=True))
    from m3ae_amd.modules import objectives
    r = objectives.compute_mlm(m, b)
    np.testing.assert_allclose(r["mlm_logits"].detach().cpu().numpy(), g["mlm_logits"], rtol=1e-3, atol=1e-4)
    assert abs(r["mlm_loss"].item() - float(g["mlm_loss"])) < 1e-4 * float(g["mlm_loss"])
    r = objectives.compute_mim(m, b)
    np.testing.assert_allclose(r["mim_logits"].detach().cpu().numpy(), g["mim_pred"], rtol=1e-3, atol=1e-4)
    assert abs(r["mim_loss"].item() - float(g["mim_loss"])) < 1e-4
    r = objectives.compute_itm(m, b, torch.tensor([1.0, 1.0]))  # un-swapped images, as in the fixture
    np.testing.assert_allclose(r["itm_logits"].detach().cpu().numpy(), g["itm_logits"], rtol=1e-3, atol=1e-5)
    # the ITM negatives swapped in by the reference's selection rule for labels [1, 0] (objectives.py:85-93)
    r = objectives.compute_itm(m, b, torch.tensor([1.0, 0.0]))
    np.testing.assert_allclose(r["itm_logits"].detach().cpu().numpy(), g["itm_swapped_logits"], rtol=1e-3, atol=1e-5)
    assert abs(r["itm_loss"].item() - float(g["itm_swapped_loss"])) < 1e-4
    # one whole pre-training step (MLM + MIM + swapped ITM): loss and per-parameter gradient norms of the reference
    b["itm_labels"] = torch.tensor([1.0, 0.0])
    m.store.zero_grad()
    loss = m.training_step(b)
    loss.backward()
    assert abs(loss.item() - float(g["step_loss"])) < 1e-4 * float(g["step_loss"])
    worst = grad_report(m, g, rtol=2e-3, floor=1e-5)
    assert worst[0] == 0.0, worst
    gn = math.sqrt(sum(p.grad.double().pow(2).sum().item() for n, p in m.named_parameters() if p.grad is not None))
    assert abs(gn - float(g["global_grad_norm"])) < 1e-3 * float(g["global_grad_norm"])


@pytest.mark.parametrize("cross_rule", ["fused", "default"])
def test_tiny_pretrain_step_bf16_against_reference_gradients(cross_rule):
    """configs[3] heads in perf mode (bf16 storage): step loss and gradient norms of the reference's fp32 run within the
    stated bf16 bounds (loss rtol 2e-3... here 1e-2 on a 3-term loss, global gradient norm 5 %, large per-parameter norms 15 %).
    cross_rule: "fused" = the cross-attention sub-blocks on the fused training path (conftest lowers the batch threshold),
    "default" = the product's rule (ops.XATTN_TRAIN_MIN_BATCH = 96: this batch takes the composition) -- the path the
    reference's own run scripts (per-GPU batch 8 / 32) take stays pinned to the reference fixtures."""
    from m3ae_amd import ops
    if cross_rule == "default":
        ops.XATTN_TRAIN_MIN_BATCH = 96          # the autouse fixture restores it
    cfg = tiny_config(compute_dtype="bf16", loss_names={"mlm": 1, "mim": 1, "itm": 1, "vqa": 0, "cls": 0, "irtr": 0},
                      mim_layer=1, mim_decoder_hidden_size=128, mim_decoder_num_layers=2, mim_decoder_num_heads=2)
    m = build(cfg, torch.bfloat16)
    g = load_golden("tiny_pretrain.npz")
    b = to_dev(tiny_batch(pretrain=True))
    b["itm_labels"] = torch.tensor([1.0, 0.0])
    m.store.zero_grad()
    loss = m.training_step(b)
    loss.backward()
    assert abs(loss.item() - float(g["step_loss"])) < 1e-2 * float(g["step_loss"])
    names, ref = g["grad_names"].tolist(), g["grad_norm"]
    params = dict(m.named_parameters())
    mine = np.array([params[n].grad.double().norm().item() for n in names])
    gn = np.sqrt((mine ** 2).sum())
    assert abs(gn - float(g["global_grad_norm"])) < 5e-2 * float(g["global_grad_norm"]), (gn, float(g["global_grad_norm"]))
    big = ref > 1e-2 * ref.max()
    rel = np.abs(mine[big] - ref[big]) / ref[big]
    assert rel.max() < 0.15, (rel.max(), np.array(names)[big][rel.argmax()])


def test_full_size_fp32_logits_within_rtol_1e3_of_reference():
    """configs[1] dims, B = 2, parity mode: north_star's "logits within rtol 1e-3 of the reference"."""
    cfg = finetune_vqa_rad_config(compute_dtype="fp32")
    m = build(cfg, torch.float32)
    g = load_golden("full_vqa.npz")
    b = to_dev(full_batch())
    m.store.zero_grad()
    m.set_task()
    ret = m(b)
    np.testing.assert_allclose(ret["vqa_logits"].detach().cpu().numpy(), g["logits"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(ret["multi_modal_cls_feats"].detach().cpu().numpy(), g["cls_feats"], rtol=1e-3, atol=1e-5)
    loss = ret["vqa_loss"]
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * float(g["loss"])
    loss.backward()
    worst = grad_report(m, g, 2e-3, 1e-6)
    assert worst[0] == 0.0, worst
    gn = m.store.grad.double().norm().item()
    assert abs(gn - float(g["global_grad_norm"])) < 1e-3 * float(g["global_grad_norm"])


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_large_tower_dims_reduced_depth_against_reference(mode):
    """configs[4] tower dimensions on the HIP path: one ViT-L/16 block (width 1024, 16 heads, 512 x 512 -> 1025 image tokens:
    K / V no longer fit one LDS tile of the attention kernels), one RoBERTa-large layer (1024 / 16 heads / 4096), one
    co-attention layer pair with 1025 image tokens on the long side (the fused cross-attention covers <= 640 and falls back
    to the composition here), against the reference-generated reduced-depth fixture (oracle/make_golden.py large1).
    fp32: north_star's rtol 1e-3 on logits, 2e-3 on per-parameter gradient norms; bf16: the stated perf-mode bounds."""
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    m = build(large1_config(compute_dtype=mode), dtype)
    g = load_golden("large1_vqa.npz")
    b = to_dev(large1_batch())
    m.store.zero_grad()
    m.set_task()
    ret = m(b)
    loss = ret["vqa_loss"]
    logits = ret["vqa_logits"].detach().float().cpu().numpy()
    if mode == "fp32":
        np.testing.assert_allclose(logits, g["logits"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(ret["multi_modal_cls_feats"].detach().cpu().numpy(), g["cls_feats"], rtol=1e-3, atol=1e-5)
        assert abs(loss.item() - float(g["loss"])) < 1e-4 * float(g["loss"])
    else:
        assert np.abs(logits - g["logits"]).max() < 0.05
        assert abs(loss.item() - float(g["loss"])) < 3e-3 * float(g["loss"])
    loss.backward()
    gn = m.store.grad.double().norm().item()
    if mode == "fp32":
        worst = grad_report(m, g, 2e-3, 1e-6)
        assert worst[0] == 0.0, worst
        assert abs(gn - float(g["global_grad_norm"])) < 1e-3 * float(g["global_grad_norm"])
    else:
        assert abs(gn - float(g["global_grad_norm"])) < 5e-2 * float(g["global_grad_norm"]), (gn, float(g["global_grad_norm"]))


# Observed on MI355X at round 4 (tools/parity_full.py, profiles/r04_parity_full_size_observed.log), configs[1] dimensions, B = 2,
# eval mode, bf16 MFMA path against the reference fixture: max |dlogits| 0.0222 / 0.0232 (fused cross-attention / composition;
# |logits| <= 1.91), loss relative error 1.0e-5 / 2.2e-4, global gradient norm 1.0e-3 / 1.7e-3, the 404 per-parameter gradient
# norms above 1 % of the largest: worst 2.6e-3 / 5.5e-3, median over all 663: 1.2e-3 / 1.7e-3.  Bounds = 2 x the observed values.
BF16_FULL_SIZE_BOUNDS = {"max_abs_dlogits": 0.046, "loss_rel_err": 4.5e-4, "global_grad_norm_rel_err": 3.4e-3,
                         "max_rel_err_large_param_grad_norms": 1.1e-2, "median_rel_err_param_grad_norms": 3.5e-3}


@pytest.mark.parametrize("cross_rule", ["fused", "default"])
def test_full_size_bf16_observed_errors_within_twice_the_recorded_ones(cross_rule):
    """The path bench.py times (bf16 storage, MFMA kernels, fused cross-attention under either training rule) at FULL size
    against the reference fixture: the observed errors are printed (pytest -s) and held to 2 x the values recorded above --
    a 10 % regression of a fused kernel does not pass."""
    from m3ae_amd import ops
    from m3ae_amd.parity import parity_report
    if cross_rule == "default":
        ops.XATTN_TRAIN_MIN_BATCH = 96          # the autouse fixture restores it
    m = build(finetune_vqa_rad_config(compute_dtype="bf16"), torch.bfloat16)
    rep = parity_report(m, load_golden("full_vqa.npz"), to_dev(full_batch()))
    print(f"\n[bf16 full-size parity, cross-attention rule {cross_rule}] " + ", ".join(
        f"{k} {rep[k]:.3e}" for k in BF16_FULL_SIZE_BOUNDS) + f"; worst large parameter: {rep['worst_large_param']}")
    for k, bound in BF16_FULL_SIZE_BOUNDS.items():
        assert rep[k] <= bound, (k, rep[k], bound, rep)


@pytest.mark.parametrize("cross_rule", ["fused", "default"])
def test_full_size_bf16_step_and_optimizer(cross_rule):
    """configs[1] perf mode: one full training step (fwd + bwd + fused AdamW); loss close to the fp32 reference,
    parameters move, shadows stay in sync.  cross_rule "default": the product's batch rule for the fused cross-attention
    training path (B = 2 < 96: the composition), "fused": threshold 0 (conftest)."""
    from m3ae_amd import ops
    if cross_rule == "default":
        ops.XATTN_TRAIN_MIN_BATCH = 96          # the autouse fixture restores it
    cfg = finetune_vqa_rad_config(compute_dtype="bf16")
    m = build(cfg, torch.bfloat16)
    g = load_golden("full_vqa.npz")
    b = to_dev(full_batch())
    m.store.zero_grad()
    loss = m.training_step(b)
    assert abs(loss.item() - float(g["loss"])) < 3e-3 * float(g["loss"])
    loss.backward()
    gn = m.store.grad.double().norm().item()
    assert abs(gn - float(g["global_grad_norm"])) < 5e-2 * float(g["global_grad_norm"]), (gn, float(g["global_grad_norm"]))
    before = m.store.flat.clone()
    m.store.adamw_step(max_steps=1000, lr_factor=1.0)
    delta = (m.store.flat - before)[: m.store.trainable_end]
    assert torch.isfinite(delta).all() and delta.abs().max().item() > 0
    w = m.vqa_head[3].weight
    assert torch.equal(w.m3ae_c, w.data.to(torch.bfloat16))
    assert torch.equal(w.m3ae_t, w.data.to(torch.bfloat16).t().contiguous())


T5_DIMS = dict(d_model=512, d_kv=64, d_ff=2048, num_layers=2, num_decoder_layers=2, num_heads=8)


def _build_t5(mode, dtype, vocab=1100, dims=None):
    from m3ae_amd.modules import T5VQA_MMEncoderInput
    m = T5VQA_MMEncoderInput(tiny_config(compute_dtype=mode), t5_vocab=vocab, t5_dims=dims or T5_DIMS)
    m.unfreeze_top_layers(4, 4)  # main_t5_m3ae.py:30
    synth.fill_deterministic(m)
    m.finalize("cuda", dtype)
    m.eval()
    return m


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_tiny_t5_generative_head_against_reference_fixture(mode):
    """configs[2] path: frozen M3AE -> [prefix | projected CLS | zero pad to 512] -> T5 encoder -> teacher-forced decoder
    -> tied LM head -> CE, forward + backward, vs the fixture captured from the reference's T5VQA_MMEncoderInput."""
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    m = _build_t5(mode, dtype)
    g = load_golden("tiny_t5.npz")
    b = to_dev(tiny_batch())
    b["t5_labels"] = torch.from_numpy(g["labels"]).cuda()
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == sorted(
        n for n in g["trainable_names"].tolist() if not n.startswith("feature_projection"))
    m.store.zero_grad()
    m.current_tasks = ["vqa"]  # training branch (an empty task list selects generation, as in the reference)
    out = m(b)
    loss = out["vqa_loss"]
    logits = out["vqa_logits"].detach().float().cpu().numpy()
    if mode == "fp32":
        np.testing.assert_allclose(logits, g["logits"], rtol=1e-3, atol=1e-5)
        assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    else:
        assert np.abs(logits - g["logits"]).max() < 0.05 * max(1.0, np.abs(g["logits"]).max())
        assert abs(loss.item() - float(g["loss"])) < 5e-3 * float(g["loss"])
    m.training_step(b)["loss"].backward()
    params = dict(m.named_parameters())
    tol = 2e-3 if mode == "fp32" else 8e-2
    ref_total = float(np.sqrt((g["grad_norm"] ** 2).sum()))
    for n, r in zip(g["grad_names"].tolist(), g["grad_norm"]):
        mine = params[n].grad.double().norm().item()
        assert abs(mine - r) <= tol * r + 1e-3 * tol * ref_total, (n, mine, r)
    m.store.adamw_step(max_steps=100, lr_factor=1.0)
    assert torch.isfinite(m.store.flat).all()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_full_t5_small_generative_head_against_reference_fixture(mode):
    """The head architecture the reference hard-codes (t5-small at full depth: 6 + 6 layers, 8 heads, d_ff 2048, vocabulary
    32128; m3ae_t5_mm_encoder_input.py:26-27), forward + backward, against the fixture captured from the reference
    (oracle/make_golden.py t5small): loss, logits at every 64th column, their row-wise logsumexp, all 72 gradient norms."""
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    m = _build_t5(mode, dtype, vocab=32128, dims=dict(T5_DIMS, num_layers=6, num_decoder_layers=6))
    g = load_golden("t5small_full.npz")
    b = to_dev(tiny_batch())
    b["t5_labels"] = torch.from_numpy(g["labels"]).cuda()
    m.store.zero_grad()
    m.current_tasks = ["vqa"]
    out = m(b)
    loss = out["vqa_loss"]
    logits = out["vqa_logits"].detach().float()
    lse = torch.logsumexp(logits.double(), -1).cpu().numpy()
    sub = logits[:, :, ::64].cpu().numpy()
    if mode == "fp32":
        np.testing.assert_allclose(sub, g["logits_stride64"], rtol=1e-3, atol=2e-5)
        np.testing.assert_allclose(lse, g["logits_lse"], rtol=1e-5)
        assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    else:
        assert np.abs(sub - g["logits_stride64"]).max() < 0.05 * max(1.0, np.abs(g["logits_stride64"]).max())
        np.testing.assert_allclose(lse, g["logits_lse"], rtol=5e-3)
        assert abs(loss.item() - float(g["loss"])) < 5e-3 * float(g["loss"])
    m.training_step(b)["loss"].backward()
    params = dict(m.named_parameters())
    tol = 2e-3 if mode == "fp32" else 8e-2
    ref_total = float(np.sqrt((g["grad_norm"] ** 2).sum()))
    assert len(g["grad_names"]) == 72
    # noise floor: with these weights the attention is nearly uniform and the q / k gradients are ~3e-4 of the total norm;
    # bf16 rounding of the activations alone is worth 1e-3 of it
    floor = (2e-6 if mode == "fp32" else 1e-3) * ref_total
    for n, r in zip(g["grad_names"].tolist(), g["grad_norm"]):
        mine = params[n].grad.double().norm().item()
        assert abs(mine - r) <= tol * r + floor, (n, mine, r)


@pytest.mark.parametrize("which", ["base", "large"])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_t5_base_dimensions_against_hf_fixture(mode, which):
    """BASELINE configs[2] names a T5-BASE head: d_model 768, 12 heads of 64, d_ff 3072; configs[4] a T5-LARGE head: d_model
    1024, 16 heads of 64, d_ff 4096 (2 + 2 layers here, every dimension exercised).  The reference's wrapper hard-wires
    t5-small's width, so the fixtures (oracle/make_golden.py t5base / t5large) come from the class it instantiates -- HF
    T5ForConditionalGeneration -- with the deterministic weights under the reference's names and its unfreeze recipe: encoder
    output, logits, loss and all 38 gradient norms, fp32 and bf16."""
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    DM, DFF, NH = (768, 3072, 12) if which == "base" else (1024, 4096, 16)
    tag = "t5base_dims" if which == "base" else "t5large_dims"
    dims = dict(d_model=DM, d_kv=64, d_ff=DFF, num_layers=2, num_decoder_layers=2, num_heads=NH)
    m = _build_t5(mode, dtype, vocab=1100, dims=dims)
    g = load_golden(tag + ".npz")
    x = synth.det_normal(tag + ".inputs_embeds", (2, 24, DM), std=0.5).cuda().to(dtype)
    labels = torch.from_numpy(g["labels"]).cuda()
    labels = labels.masked_fill(labels == 0, -100)
    m.store.zero_grad()
    out = m.t5(x, labels)
    logits = out.logits.detach().float().cpu().numpy()
    enc = out.encoder_last_hidden_state.detach().float().cpu().numpy()[:, :6]
    if mode == "fp32":
        np.testing.assert_allclose(enc, g["enc_out"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(logits, g["logits"], rtol=1e-3, atol=2e-5)
        assert abs(out.loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    else:
        assert np.abs(logits - g["logits"]).max() < 0.05 * max(1.0, np.abs(g["logits"]).max())
        assert abs(out.loss.item() - float(g["loss"])) < 5e-3 * float(g["loss"])
    out.loss.backward()
    params = dict(m.named_parameters())
    tol = 2e-3 if mode == "fp32" else 8e-2
    ref_total = float(np.sqrt((g["grad_norm"] ** 2).sum()))
    floor = (2e-6 if mode == "fp32" else 1e-3) * ref_total
    assert len(g["grad_names"]) == 38
    for n, r in zip(g["grad_names"].tolist(), g["grad_norm"]):
        mine = params[n].grad.double().norm().item()
        assert abs(mine - r) <= tol * r + floor, (n, mine, r)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_training_mode_dropout_is_seeded_and_active(mode):
    """model.train() with drop_rate = 0.1 (the reference's fine-tuning setting, m3ae/config.py:74): the step is
    reproducible from `ops.set_dropout_seed`, differs across seeds and from eval(), and gradients stay finite."""
    from m3ae_amd import ops
    cfg = tiny_config(compute_dtype=mode, drop_rate=0.1)
    m = build(cfg, torch.float32 if mode == "fp32" else torch.bfloat16)
    b = to_dev(tiny_batch())
    m.set_task()

    def step(seed, train=True):
        m.train(train)
        m.store.zero_grad()
        ops.set_dropout_seed(seed)
        ret = m(b)
        ret["vqa_loss"].backward()
        return ret["vqa_loss"].item(), m.store.grad.clone()

    l1, g1 = step(11)
    l2, g2 = step(11)
    l3, g3 = step(12)
    le, ge = step(11, train=False)
    assert np.isfinite(l1) and torch.isfinite(g1).all()
    # same masks -> same step, up to the summation order of the fp32 atomics (loss reduction, split-K wgrad)
    assert abs(l1 - l2) <= 1e-6 * abs(l1) and (g1 - g2).double().norm().item() <= 1e-5 * g1.double().norm().item()
    assert abs(l1 - l3) > 1e-5 * abs(l1) and (g1 - g3).double().norm().item() > 1e-3 * g1.double().norm().item()
    assert abs(l1 - le) > 1e-5 * abs(le)
    # dropout perturbs, it does not destroy: loss within 20% of the eval loss, gradient norm within 2x
    assert abs(l1 - le) < 0.2 * abs(le), (l1, le)
    r = (g1.double().norm() / ge.double().norm()).item()
    assert 0.5 < r < 2.0, r


def test_trainer_shim_fit_checkpoint_resume(tmp_path):
    """the sacred-free entry point on the run_scripts grammar -- accumulation, schedule, validation,
    checkpoints with the reference's state_dict names, resume."""
    from m3ae_amd import trainer
    from m3ae_amd.modules import state_dict_spec
    tiny = ("image_size=64 hidden_size=128 num_heads=2 num_top_layer=2 input_image_embed_size=128 "
            "input_text_embed_size=128 vocab_size=1000 vit_width=128 vit_layers=3 text_hidden=128 text_layers=2 "
            "text_heads=2 text_inter=512").split()
    argv = (["with", "data_root=synthetic", "num_gpus=1", "num_nodes=1", "task_finetune_vqa_vqa_rad", "clip16",
             "text_roberta", "per_gpu_batchsize=4", "batch_size=8", "max_steps=6", "learning_rate=0.0005",
             "synthetic_train_samples=32", "synthetic_val_samples=8", f"log_dir={tmp_path}", "seed=3"] + tiny)
    out = trainer.run(argv)
    assert out["global_step"] == 6
    hist = out["history"]
    assert hist[0][0] == 1 and np.isfinite([h[1] for h in hist]).all()
    cfg = trainer.config_mod.parse_cli(argv)
    run_dir = os.path.join(str(tmp_path), f'{cfg["exp_name"]}-seed3-from_', "checkpoints")
    ck = torch.load(os.path.join(run_dir, "last.ckpt"), map_location="cpu", weights_only=False)
    spec = state_dict_spec(cfg)
    assert set(ck["state_dict"]) == set(spec) and all(tuple(ck["state_dict"][k].shape) == tuple(spec[k]) for k in spec)
    assert ck["global_step"] == 6 and "optimizer_flat" not in ck  # finetune runs save weights only (main.py:42)
    # the saved weights reproduce the trained model's logits in a fresh module
    m = M3AETransformerSS(cfg)
    m.load_state_dict(ck["state_dict"], strict=False)
    m.finalize("cuda", torch.bfloat16)
    m.eval()
    m.set_task()
    b = to_dev(synth.synthetic_batch(4, text_len=32, image_size=64, vocab_size=1000, rank=0))
    l1 = m(b)["vqa_logits"].float()
    assert torch.isfinite(l1).all()
    # training moved the weights away from the initial ones
    m0 = build(cfg, torch.bfloat16)
    m0.set_task()
    assert (m0(b)["vqa_logits"].float() - l1).abs().max().item() > 1e-3
    # resume: the step counter (and with it the LR schedule) continues
    out2 = trainer.run(argv[:-len(tiny)] + tiny + [f"resume_from={os.path.join(run_dir, 'last.ckpt')}", "max_steps=8"])
    assert out2["global_step"] == 8


def _decoder_cfg(mode):
    return tiny_config(compute_dtype=mode, image_size=64, hidden_size=768, num_heads=12, num_top_layer=1,
                       input_image_embed_size=128, input_text_embed_size=128, vocab_size=1000, vit_width=128,
                       vit_layers=2, text_hidden=128, text_layers=1, text_heads=2, text_inter=512,
                       mm_encoder_inputs_include_cls_feats=True, mm_encoder_inputs_include_imagetext_feats=False)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_tiny_decoder_generative_head_against_reference_fixture(mode):
    """DecoderModel (frozen M3AE -> decoder head -> CE, greedy search) through the C ABI against the
    fixture captured from the reference's m3ae_decoder.py (tests/golden/tiny_decoder.npz)."""
    from m3ae_amd.modules import DecoderModel
    g = load_golden("tiny_decoder.npz")
    cfg = _decoder_cfg(mode)
    m = DecoderModel(cfg, vocab_size=1200)
    ref_dec = {n: s for n, s in zip(g["state_names"].tolist(), g["state_shapes"].tolist()) if n.startswith("decoder.")}
    mine_dec = {n: str(tuple(v.shape)) for n, v in m.state_dict().items() if n.startswith("decoder.")}
    assert mine_dec == ref_dec  # the M3AE part of the state_dict is checked in test_host_logic
    synth.fill_deterministic(m)
    m.finalize("cuda", torch.float32 if mode == "fp32" else torch.bfloat16)
    m.eval()
    b = to_dev(synth.synthetic_batch(2, text_len=32, image_size=64, vocab_size=1000, rank=0))
    b["decoder_tokens"] = torch.from_numpy(g["tokens"]).cuda()
    enc = m.features(b)
    tol = dict(rtol=1e-3, atol=1e-5) if mode == "fp32" else dict(rtol=5e-2, atol=2e-2)
    np.testing.assert_allclose(enc.float().cpu().numpy(), g["cls"], **tol)
    m.store.zero_grad()
    out = m.training_step(b)
    loss = out["loss"]
    lt = 1e-5 if mode == "fp32" else 2e-3
    assert abs(loss.item() - float(g["loss"])) < lt * float(g["loss"]), (loss.item(), float(g["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    rt = 2e-3 if mode == "fp32" else 5e-2
    for n, r in zip(g["grad_names"].tolist(), g["grad_norm"]):
        mine = params[n].grad.double().norm().item()
        assert abs(mine - r) <= rt * r + 1e-9, (n, mine, r)
    # the reference's dead layers: present in the state_dict, outside the optimizer
    for n in g["trainable_nograd"].tolist():
        assert params[n].grad is None and not params[n].requires_grad
    if mode == "bf16":
        # perf mode: the head runs on the bf16 MFMA GEMMs (96-wide heads: fp32 attention kernels on fp32 copies)
        from m3ae_amd import ops
        tin = torch.from_numpy(g["tokens"][:, :-1]).cuda()
        feats = torch.from_numpy(g["cls"]).cuda()
        logits = m.decoder(tin.masked_fill(tin == 102, 0), (tin != 102) & (tin != 0), feats)
        assert logits.dtype == torch.bfloat16 and ops.last_gemm_path().startswith("mfma")
        ref = g["logits"]
        assert np.abs(logits.detach().float().cpu().numpy() - ref).max() < 0.03 * max(1.0, np.abs(ref).max())
        m.decoder.max_len = 16
        a, c = m.decoder.search_path(feats), m.decoder.search_path(feats, use_cache=False)
        np.testing.assert_array_equal(a.cpu().numpy(), c.cpu().numpy())   # cached rows = full-prefix passes, bf16 too
        assert (a.cpu().numpy() == g["greedy"]).mean() > 0.9              # and the reference's tokens up to bf16 ties
    if mode == "fp32":
        m.current_tasks = []
        logits = m.decoder(torch.from_numpy(g["tokens"][:, :-1]).cuda().masked_fill(
            torch.from_numpy(g["tokens"][:, :-1]).cuda() == 102, 0),
            (torch.from_numpy(g["tokens"][:, :-1]).cuda() != 102) & (torch.from_numpy(g["tokens"][:, :-1]).cuda() != 0),
            torch.from_numpy(g["cls"]).cuda())
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=1e-3, atol=5e-5)
        m.decoder.max_len = 16
        feats = torch.from_numpy(g["cls"]).cuda()
        greedy = m.decoder.search_path(feats)                       # one decoder row per step (key / value cache)
        np.testing.assert_array_equal(greedy.cpu().numpy(), g["greedy"])
        np.testing.assert_array_equal(m.decoder.search_path(feats, use_cache=False).cpu().numpy(), g["greedy"])
        # the cached step reproduces the full-prefix logits at every position (m3ae_decoder.py:141-182 re-runs the prefix)
        toks = torch.from_numpy(g["tokens"][:, :5]).cuda()
        layer = m.decoder.dec_layers[m.decoder.num_layers - 1]
        enc_kv = layer.cross_kv(feats)
        cache = torch.empty((toks.shape[0], 16, 2 * 768), dtype=torch.float32, device="cuda")
        for t in range(toks.shape[1]):
            step = m.decoder.step_logits(toks[:, t:t + 1], t, enc_kv, cache)
            full = m.decoder(toks[:, :t + 1], None, feats)[:, -1]
            np.testing.assert_allclose(step.cpu().numpy(), full.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_trainer_shim_decoder_head_entry_point(tmp_path):
    """main_decoder_m3ae.py path: frozen M3AE + decoder head for a few optimizer steps; only the live parameters move."""
    from m3ae_amd import trainer
    small = ("image_size=64 hidden_size=768 num_heads=12 num_top_layer=1 input_image_embed_size=128 "
             "input_text_embed_size=128 vocab_size=1000 vit_width=128 vit_layers=2 text_hidden=128 text_layers=1 "
             "text_heads=2 text_inter=512").split()
    argv = (["with", "data_root=synthetic", "num_gpus=1", "num_nodes=1", "task_finetune_vqa_vqa_rad", "clip16",
             "text_roberta", "per_gpu_batchsize=4", "batch_size=4", "max_steps=4", "learning_rate=0.001",
             "synthetic_train_samples=16", "synthetic_val_samples=4", f"log_dir={tmp_path}", "seed=5",
             "mm_encoder_inputs_include_cls_feats=True", "mm_encoder_inputs_include_imagetext_feats=True"] + small)
    out = trainer.run(argv, head="decoder")
    assert out["global_step"] == 4
    losses = [h[1] for h in out["history"]]
    assert np.isfinite(losses).all() and losses[0] > 5.0  # ~ln(30522) at the start


def test_pretraining_caption_tables_feed_the_three_objectives(tmp_path):
    """SURVEY 8f-2 remainder on the GPU: ROCO + MedICaT caption tables (prepro/make_arrow.py schema) through the prefetching
    datamodule -- captions tokenised, MLM fields from the collator, `false_image_0` negatives drawn per sample and
    normalised on the GPU -- into one MLM + MIM + ITM training step (configs[3] recipe at the tiny size)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from arrow_util import CollatorTokenizer, write_caption_split
    from m3ae_amd import data

    class Tok(CollatorTokenizer):   # + the dataset-side call: ids from a word hash, RoBERTa specials
        def __call__(self, text, padding="max_length", truncation=True, max_length=32, **kw):
            ids = [0] + [4 + (sum(ord(ch) * (i + 1) for i, ch in enumerate(w)) % 900) for w in text.lower().split()][: max_length - 2] + [2]
            mask = [1] * len(ids)
            ids += [1] * (max_length - len(ids))
            mask += [0] * (max_length - len(mask))
            return {"input_ids": ids, "attention_mask": mask}

    root = str(tmp_path / "pre")
    n = write_caption_split(root, "medicat", "train", 6, seed=3) + write_caption_split(root, "roco", "train", 6, seed=50)
    for name in ("medicat", "roco"):
        write_caption_split(root, name, "val", 2, seed=77)
    cfg = tiny_config(compute_dtype="bf16", data_root=root, per_gpu_batchsize=4, num_workers=2, seed=1,
                      datasets=["medicat", "roco"], draw_false_image=1,
                      loss_names={"mlm": 1, "mim": 1, "itm": 1, "vqa": 0, "cls": 0, "irtr": 0}, mim_layer=1,
                      mim_decoder_hidden_size=128, mim_decoder_num_layers=2, mim_decoder_num_heads=2)
    dm = data.ArrowDataModule(cfg, 0, 1, torch.device("cuda", 0), tokenizer=Tok("roberta"))
    assert dm.train_samples == n and dm.mlm_collator is not None
    batches = list(dm.train_batches(0))
    assert sum(b["text_ids"].shape[0] for b in batches) == n
    b0 = batches[0]
    for k in ("image", "false_image_0"):
        assert b0[k][0].shape == (4, 3, 64, 64) and b0[k][0].dtype == torch.float32 and b0[k][0].is_cuda
    assert not torch.equal(b0["image"][0], b0["false_image_0"][0])
    assert b0["text_ids_mlm"].shape == (4, 32) and (b0["text_labels_mlm"] != -100).any()
    m = build(cfg, torch.bfloat16)
    m.train()
    m.store.zero_grad()
    loss = m.training_step(b0)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(m.store.grad).all() and m.store.grad.abs().max().item() > 0
    g_itm = m.itm_head.fc.weight.grad
    assert g_itm is not None and g_itm.abs().max().item() > 0


def test_input_pipeline_device_tail_and_trainer_on_arrow_data(tmp_path):
    """uint8 NHWC upload + ToTensor / Normalize on the GPU is bit-equal to the reference's torch
    arithmetic; the prefetching datamodule feeds the trainer shim end to end from an arrow file."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from arrow_util import HashTokenizer, write_split
    from m3ae_amd import data, trainer
    u8 = torch.randint(0, 256, (3, 40, 56, 3), dtype=torch.uint8, device="cuda")
    out = data.normalize_on_device(u8)
    mean = torch.tensor(synth.CLIP_MEAN, device="cuda").view(1, 3, 1, 1)
    std = torch.tensor(synth.CLIP_STD, device="cuda").view(1, 3, 1, 1)
    # ToTensor + Normalize (transform.py:65-66) as the reference runs them: on the CPU (IEEE divisions; torch's GPU
    # `div(scalar)` multiplies by the reciprocal instead and differs in the last bit)
    ref = (u8.cpu().permute(0, 3, 1, 2).float().div(255) - mean.cpu()) / std.cpu()
    assert torch.equal(out.cpu(), ref)
    root = str(tmp_path / "arrows")
    n_train = write_split(root, "train", 12)
    write_split(root, "val", 4, seed=100)
    tok = HashTokenizer()
    cfg = tiny_config(compute_dtype="bf16", data_root=root, per_gpu_batchsize=4, num_workers=3, seed=1)
    dm = data.ArrowDataModule(cfg, 0, 1, torch.device("cuda", 0), tokenizer=tok)
    assert dm.train_samples == n_train
    batches = list(dm.train_batches(0))
    assert sum(b["text_ids"].shape[0] for b in batches) == n_train and len(batches) == (n_train + 3) // 4
    b0 = batches[0]
    assert b0["image"][0].shape == (4, 3, 64, 64) and b0["image"][0].dtype == torch.float32
    assert b0["text_labels"].eq(-100).all() and b0["text_ids"].is_cuda
    # same sample through the dataset directly == through the prefetching loader (order of epoch 0 is seeded)
    order = dm._indices(dm.train_set, 0, True)
    s = dm.train_set[order[0]]
    want = (torch.from_numpy(s["image_u8"].copy()).permute(2, 0, 1).float().div(255) - mean[0].cpu()) / std[0].cpu()
    assert torch.equal(b0["image"][0][0].cpu(), want) and b0["text_ids"][0].tolist() == s["input_ids"]
    tiny = ("image_size=64 hidden_size=128 num_heads=2 num_top_layer=2 input_image_embed_size=128 "
            "input_text_embed_size=128 vocab_size=1000 vit_width=128 vit_layers=3 text_hidden=128 text_layers=2 "
            "text_heads=2 text_inter=512").split()
    argv = (["with", f"data_root={root}", "num_gpus=1", "num_nodes=1", "task_finetune_vqa_vqa_rad", "clip16",
             "text_roberta", "per_gpu_batchsize=4", "batch_size=8", "max_steps=3", "num_workers=2",
             f"log_dir={tmp_path}", "seed=2"] + tiny)
    out = trainer.run(argv, tokenizer=tok)
    assert out["global_step"] == 3 and np.isfinite(out["test"])


@pytest.mark.parametrize("len_offset", [0, 1])
def test_t5_generate_matches_oracle_and_third_party(len_offset):
    """`T5ForConditionalGeneration.generate` (decoder on the GPU kernels, fp32) against the CPU restatement
    for both length conventions, and for len_offset=1 against the third-party sequences of the fixture."""
    from m3ae_amd.modules.t5 import T5ForConditionalGeneration
    from m3ae_amd.param_store import ParamStore, group_hparams_decoder, param_group_of_decoder
    from oracle_util import canon_generated, gen_t5_weights
    g = load_golden("tiny_t5_generate.npz")
    m = T5ForConditionalGeneration(dict(d_model=512, d_kv=64, d_ff=2048, num_layers=2, num_decoder_layers=2,
                                        num_heads=8), 1100)
    sd = gen_t5_weights({"t5." + k: v for k, v in m.state_dict().items()})
    m.load_state_dict({k[3:]: v for k, v in sd.items()})
    cfg = tiny_config(compute_dtype="fp32")
    ParamStore(m, cfg, "cuda", torch.float32, m.weight_units, group_fn=param_group_of_decoder,
               hparams_fn=group_hparams_decoder)
    m.eval()
    enc = torch.from_numpy(g["enc"]).cuda()
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    # the encoder-side key / value cache of generate(): same logits, bit for bit, as the decoder run on `enc` itself
    prefix = torch.tensor([[0, 5, 9, 700]] * enc.shape[0], device="cuda")
    kv = m.decoder.cross_kv(enc)
    assert len(kv) == 2 and kv[0].shape == (enc.shape[0] * enc.shape[1], 2 * 8 * 64)
    assert torch.equal(m.next_token_logits(enc, prefix), m.next_token_logits(enc, prefix, kv))
    # the self-attention key / value cache (HF past_key_values; m3ae_t5_mm_encoder_input.py:209-227): feeding the prefix one
    # token at a time gives, at every position, the logits of the full-prefix pass -- same kernels, same reduction order
    cache = m.decoder.new_self_cache(enc.shape[0], 8, enc.dtype, enc.device)
    for t in range(prefix.shape[1]):
        step = m.next_token_logits_cached(enc, prefix[:, t:t + 1], kv, cache, t)
        full = m.next_token_logits(enc, prefix[:, :t + 1], kv)
        assert torch.equal(step, full), (t, float((step - full).abs().max()))
    for eos in g["eos_ids"].tolist():
        mine = m.generate(enc, num_beams=4, max_length=8, eos_token_id=eos, len_offset=len_offset)
        with torch.no_grad():
            ref = O.t5_beam_search(sd_cpu, torch.from_numpy(g["enc"]), 8, num_beams=4, max_length=8, eos_id=eos,
                                   len_offset=len_offset)
        assert mine.cpu().tolist() == ref.tolist(), eos
        if len_offset == 1:
            assert canon_generated(mine.cpu().tolist(), eos) == canon_generated(g[f"seq_{eos}"].tolist(), eos), eos


def test_trainer_shim_pretrain_objectives(tmp_path):
    """configs[3] recipe through the entry point: MLM + MIM + ITM on synthetic pre-training batches (three `infer`
    passes per step, masked image stream, vocabulary GEMM), a few optimizer steps."""
    from m3ae_amd import trainer
    tiny = ("image_size=64 hidden_size=128 num_heads=2 num_top_layer=2 input_image_embed_size=128 "
            "input_text_embed_size=128 vocab_size=1000 vit_width=128 vit_layers=3 text_hidden=128 text_layers=2 "
            "text_heads=2 text_inter=512 mim_decoder_hidden_size=128 mim_decoder_num_layers=2 mim_decoder_num_heads=2 "
            "mim_layer=1 max_text_len=32").split()
    argv = (["with", "data_root=synthetic", "num_gpus=1", "num_nodes=1", "task_pretrain_m3ae", "clip16", "text_roberta",
             "per_gpu_batchsize=4", "batch_size=4", "max_steps=3", "learning_rate=0.0005", "synthetic_train_samples=16",
             "synthetic_val_samples=4", f"log_dir={tmp_path}", "seed=7", "precision=32"] + tiny)
    out = trainer.run(argv)
    assert out["global_step"] == 3
    assert np.isfinite([h[1] for h in out["history"]]).all()


def test_full_size_bench_batch_properties():
    """Size-independent properties at the BENCH's per-GPU batch scale (configs[1] dims, bf16, B = 64, the large-tile
    kernels): forward determinism bit for bit, sample independence (a sample's logits do not depend on its batch mates
    or its position), loss / gradient linearity in the loss scale, and agreement of the first two samples with the
    reference fixture."""
    cfg = finetune_vqa_rad_config(compute_dtype="bf16")
    m = build(cfg, torch.bfloat16)
    B = 64
    b = to_dev(synth.synthetic_batch(B, text_len=32, image_size=384, rank=0))
    m.set_task()
    with torch.no_grad():
        l1 = m(b)["vqa_logits"].float()
        l2 = m(b)["vqa_logits"].float()
    assert torch.equal(l1, l2)
    perm = torch.arange(B - 1, -1, -1, device="cuda")
    bp = dict(b)
    bp["image"] = [b["image"][0][perm]]
    for k in ("text_ids", "text_masks", "text_labels"):
        bp[k] = b[k][perm]
    bp["vqa_labels"] = [b["vqa_labels"][i] for i in perm.tolist()]
    bp["vqa_scores"] = [b["vqa_scores"][i] for i in perm.tolist()]
    with torch.no_grad():
        lp = m(bp)["vqa_logits"].float()
    assert torch.equal(lp[perm], l1)        # every output row is reduced in the same order wherever the sample sits
    # the fixture's two samples are the first two of this batch (same generator, rank 0): bf16 bound on the logits
    g = load_golden("full_vqa.npz")
    b2 = to_dev(full_batch())
    # (the prefix property of synth.synthetic_batch is asserted, not assumed: a silent change would skip the check)
    assert torch.equal(b2["text_ids"], b["text_ids"][:2]) and torch.equal(b2["image"][0], b["image"][0][:2])
    assert np.abs(l1[:2].cpu().numpy() - g["logits"]).max() < 0.05
    # linearity: d(2 L) = 2 dL (fp32 atomics in the split-K wgrad: tolerance, not bit equality)
    m.store.zero_grad()
    loss = m(b)["vqa_loss"]
    loss.backward()
    g1 = m.store.grad.clone()
    m.store.zero_grad()
    (2.0 * m(b)["vqa_loss"]).backward()
    g2 = m.store.grad
    rel = ((g2 - 2.0 * g1).double().norm() / (2.0 * g1).double().norm()).item()
    assert rel < 1e-3, rel
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0


@pytest.mark.parametrize("task", ["vqa", "pretrain"])
def test_two_stream_schedule_equals_the_single_stream_step(task):
    """M3AETransformerSS runs its text half (RoBERTa tower + the text half of every fusion layer) on a second HIP stream
    (modules/m3ae_module.py::_fusion_two_streams).  Same kernels, same dropout seeds, different interleaving: the forward is
    bit-identical to the single-stream schedule and the gradients agree up to the order of the fp32 atomics -- five times in a
    row, with the flat gradient buffer read right after backward() (the end-of-backward join is what makes that read safe)."""
    from m3ae_amd import ops
    if task == "vqa":
        cfg = finetune_vqa_rad_config(compute_dtype="bf16")
        b = to_dev(synth.synthetic_batch(8, text_len=32, image_size=384, rank=0))
    else:
        cfg = tiny_config(compute_dtype="bf16", drop_rate=0.1,
                          loss_names={"mlm": 1, "mim": 1, "itm": 1, "vqa": 0, "cls": 0, "irtr": 0},
                          mim_layer=1, mim_decoder_hidden_size=128, mim_decoder_num_layers=2, mim_decoder_num_heads=2)
        b = to_dev(tiny_batch(pretrain=True))
        b["itm_labels"] = torch.tensor([1.0, 0.0])
    m = build(cfg, torch.bfloat16)
    m.set_task()

    def step(two):
        m.two_streams = two
        m.train()
        m.store.zero_grad()
        ops.set_dropout_seed(5)
        torch.manual_seed(3)            # the MIM masking noise of the pre-training step
        if task == "vqa":
            ret = m(b)
            loss, feats = ret["vqa_loss"], ret["multi_modal_cls_feats"].float().clone()
        else:
            loss, feats = m.training_step(b), None
        loss.backward()
        return loss.item(), m.store.grad.clone(), feats

    l0, g0, f0 = step(False)
    assert len(m.store.streams) == 0
    for _ in range(5):
        l1, g1, f1 = step(True)
        assert len(m.store.streams) == 2
        assert abs(l1 - l0) <= 1e-6 * abs(l0), (l0, l1)
        if f0 is not None:
            assert torch.equal(f0, f1)
        rel = ((g1 - g0).double().norm() / g0.double().norm()).item()
        assert rel <= 1e-5, rel
    m.two_streams = type(m).two_streams


def test_step_on_the_high_priority_launch_stream_equals_the_default_stream_step():
    """bench.py and trainer.py issue the step on ops.launch_stream() (high priority: the image half's workgroups are dispatched ahead
    of the text half's on the normal-priority side stream).  Same kernels, same seeds: forward bit-identical, gradients equal up to
    the order of the fp32 atomics; the caller's stream is restored."""
    from m3ae_amd import ops
    cfg = finetune_vqa_rad_config(compute_dtype="bf16")
    b = to_dev(synth.synthetic_batch(8, text_len=32, image_size=384, rank=0))
    m = build(cfg, torch.bfloat16)
    m.set_task()

    def step():
        m.train()
        m.store.zero_grad()
        ops.set_dropout_seed(11)
        ret = m(b)
        ret["vqa_loss"].backward()
        torch.cuda.synchronize()
        return ret["vqa_loss"].item(), m.store.grad.clone(), ret["multi_modal_cls_feats"].float().clone()

    l0, g0, f0 = step()
    prev = ops.use_launch_stream()
    try:
        assert torch.cuda.current_stream().priority == -1 and torch.cuda.current_stream() == ops.launch_stream()
        for _ in range(3):
            l1, g1, f1 = step()
            assert abs(l1 - l0) <= 1e-6 * abs(l0) and torch.equal(f0, f1)
            assert ((g1 - g0).double().norm() / g0.double().norm()).item() <= 1e-5
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(prev)
    assert torch.cuda.current_stream() == prev


def test_optimizer_in_backward_equals_the_step_after_backward():
    """`FlatGradReducer(update_in_backward=True)` without data parallelism (an opt-in of bench.py: --optimizer-in-backward): AdamW runs bucket by
    bucket on its own stream as backward completes the buckets.  Same kernels over the same ranges with the same
    hyper-parameters: after four steps at configs[1] dimensions (two HIP streams, dropout on) the fp32 masters, both moments,
    the bf16 shadows and a transposed weight copy equal those of the plain zero_grad / backward / adamw_step loop up to
    the run-to-run noise of the fp32 atomics, and most buckets were updated before backward ended."""
    from m3ae_amd import ops
    from m3ae_amd.ddp import FlatGradReducer
    cfg = finetune_vqa_rad_config(compute_dtype="bf16")
    b = to_dev(synth.synthetic_batch(4, text_len=32, image_size=384, rank=0))

    def run(in_backward):
        m = build(cfg, torch.bfloat16)
        m.set_task()
        m.train()
        red = FlatGradReducer(m.store, update_in_backward=in_backward)
        assert red.world == 1
        red.attach()
        early = []
        try:
            for step in range(4):
                m.store.zero_grad()
                ops.set_dropout_seed(21 + step)
                if in_backward:
                    red.arm_update(max_steps=20)
                m(b)["vqa_loss"].backward()
                red.finish()
                if in_backward:
                    early.append(red.updated_in_backward)
                else:
                    m.store.adamw_step(max_steps=20)
        finally:
            red.detach()
        st = m.store
        wt = st._t_bufs[3][3].float().clone()
        return st.flat.clone(), st.exp_avg.clone(), st.exp_avg_sq.clone(), st.shadow.float().clone(), wt, early, red.nb

    f0, m0, v0, s0, t0, _, _ = run(False)
    f1, m1, v1, s1, t1, early, nb = run(True)
    assert early[0] == 0 and min(early[1:]) >= nb - 3, (early, nb)      # all but the last two (+ glue) under backward
    init = build(cfg, torch.bfloat16).store.flat
    moved = (f0 - init).double().norm().item()
    assert (f1 - f0).double().norm().item() <= 1e-3 * moved
    assert (m1 - m0).double().norm().item() <= 1e-3 * m0.double().norm().item()      # (atomics noise measured: 1.5e-4)
    assert (v1 - v0).double().norm().item() <= 1e-3 * v0.double().norm().item()
    assert (s1 - s0).abs().max().item() <= 1e-2 and (t1 - t0).abs().max().item() <= 1e-2


def test_transposed_weight_copies_equal_the_shadows_after_an_optimizer_step():
    """The dgrad operands (`weight.m3ae_t`, [K][N] bf16) are rebuilt from the bf16 shadows by ONE table-driven launch
    (`m3ae_transpose_bf16_batched`, 64 x 64 tiles, 16 B per lane; edge tiles and the 50265-row table element-wise): bit-equal to
    the transposed shadow for every weight unit of the full-size model, after load and after an optimizer step."""
    import time
    cfg = finetune_vqa_rad_config(compute_dtype="bf16")
    m = build(cfg, torch.bfloat16)
    st = m.store

    def check():
        n = 0
        for u, rows, cols, t in st._t_bufs:
            off = (u.data.data_ptr() - st.flat.data_ptr()) // 4
            sh = st.shadow[off:off + rows * cols].view(rows, cols)
            assert torch.equal(t, sh.t()), (rows, cols)
            assert torch.equal(sh.float(), st.flat[off:off + rows * cols].view(rows, cols).to(torch.bfloat16).float())
            n += 1
        return n

    assert check() >= 100
    assert any(r % 64 or c % 64 for _, r, c, _ in st._t_bufs)      # the edge path is exercised
    b = to_dev(synth.synthetic_batch(2, text_len=32, image_size=384, rank=0))
    m.set_task()
    m.train()
    st.zero_grad()
    m(b)["vqa_loss"].backward()
    st.adamw_step(max_steps=10)
    check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        st.sync_shadows(cast=False)
    torch.cuda.synchronize()
    print(f"transposed copies of {len(st._t_bufs)} units: {(time.perf_counter() - t0) * 1e5:.0f} us per pass")


def test_t5_head_training_mode_dropout_is_seeded_and_active():
    """configs[2] in train() mode: every HF-T5 dropout site (embeddings, attention weights, sub-layer outputs, inside the
    feed-forward, final norm) plus the frozen M3AE's own: reproducible from the seed, different across seeds / from eval."""
    from m3ae_amd import ops
    m = _build_t5("bf16", torch.bfloat16)
    g = load_golden("tiny_t5.npz")
    b = to_dev(tiny_batch())
    b["t5_labels"] = torch.from_numpy(g["labels"]).cuda()

    def step(seed, train=True):
        m.train(train)
        m.store.zero_grad()
        ops.set_dropout_seed(seed)
        loss = m.training_step(b)["loss"]
        loss.backward()
        return loss.item(), m.store.grad.clone()

    l1, g1 = step(21)
    l2, g2 = step(21)
    l3, g3 = step(22)
    le, ge = step(21, train=False)
    assert np.isfinite(l1) and torch.isfinite(g1).all()
    assert abs(l1 - l2) <= 1e-6 * abs(l1) and (g1 - g2).double().norm().item() <= 1e-5 * g1.double().norm().item()
    assert abs(l1 - l3) > 1e-6 * abs(l1) and (g1 - g3).double().norm().item() > 1e-3 * g1.double().norm().item()
    assert abs(l1 - le) > 1e-6 * abs(le) and abs(l1 - le) < 0.2 * abs(le)
    assert abs(le - float(g["loss"])) < 3e-3 * float(g["loss"])   # eval mode is still the reference fixture


def test_decoder_head_training_mode_dropout():
    """DecoderModel in train() mode: embedding / attention-weight / dropout1..3 sites active and seeded."""
    from m3ae_amd import ops
    from m3ae_amd.modules import DecoderModel
    g = load_golden("tiny_decoder.npz")
    m = DecoderModel(_decoder_cfg("bf16"), vocab_size=1200)
    synth.fill_deterministic(m)
    m.finalize("cuda", torch.bfloat16)
    b = to_dev(synth.synthetic_batch(2, text_len=32, image_size=64, vocab_size=1000, rank=0))
    b["decoder_tokens"] = torch.from_numpy(g["tokens"]).cuda()

    def step(seed, train=True):
        m.train(train)
        m.store.zero_grad()
        ops.set_dropout_seed(seed)
        loss = m.training_step(b)["loss"]
        loss.backward()
        return loss.item(), m.store.grad.clone()

    l1, g1 = step(31)
    l2, g2 = step(31)
    l3, _ = step(32)
    le, _ = step(31, train=False)
    assert np.isfinite(l1) and torch.isfinite(g1).all()
    assert abs(l1 - l2) <= 1e-6 * abs(l1) and (g1 - g2).double().norm().item() <= 1e-5 * g1.double().norm().item()
    assert abs(l1 - l3) > 1e-7 * abs(l1) and abs(l1 - le) > 1e-7 * abs(le)
    assert abs(le - float(g["loss"])) < 2e-3 * float(g["loss"])


@pytest.mark.parametrize("task", ["vqa", "pretrain"])
def test_grad_reducer_hooks_on_the_real_backward(task, monkeypatch):
    """The data-parallel reducer against the REAL backward's report pattern (single process: the collective is replaced
    by a recorder).  Learning step: no early launch; afterwards buckets are released during backward, each exactly once,
    and never before the last in-place contribution of their parameters -- including the pre-training step, where every
    encoder parameter receives three contributions (MLM / MIM / ITM passes)."""
    import torch.distributed as dist
    from m3ae_amd import ops
    from m3ae_amd.ddp import FlatGradReducer
    if task == "vqa":
        cfg = tiny_config(compute_dtype="bf16")
        b = to_dev(tiny_batch())
    else:
        cfg = tiny_config(compute_dtype="bf16", loss_names={"mlm": 1, "mim": 1, "itm": 1, "vqa": 0, "cls": 0, "irtr": 0},
                          mim_layer=1, mim_decoder_hidden_size=128, mim_decoder_num_layers=2, mim_decoder_num_heads=2)
        b = to_dev(tiny_batch(pretrain=True))
    m = build(cfg, torch.bfloat16)
    m.train()
    red = FlatGradReducer(m.store, bucket_bytes=256 << 10)
    red.world = 2                                   # pretend: exercise the hook path in one process
    calls = []

    class H:
        def wait(self):
            pass

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        if t.numel() > 1:   # (the 1-element all-reduce of finish() is the collective "late contribution" flag)
            calls.append((t.data_ptr(), t.numel(), float(t.double().abs().sum().item())))
        return H()

    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    red.attach()
    try:
        for step in range(3):
            m.store.zero_grad()
            calls.clear()
            ops.set_dropout_seed(5)
            loss = m.training_step(b)
            loss.backward()
            early = len(calls)
            snap = {c[0]: c[2] for c in calls}      # |grad| of each bucket at the moment it was released
            red.finish()
            assert len(calls) == red.nb              # every bucket exactly once
            if step == 0:
                assert early == 0 and red.expected is not None
                if task == "pretrain":
                    n = m.store.names
                    w = m.language_encoder.encoder.layer[0].intermediate.dense.weight
                    assert red.expected[id(w)] == 3, red.expected[id(w)]
            else:
                assert early >= red.nb // 2, (early, red.nb)
                # a bucket released early already held its FINAL gradient
                for bi in range(red.nb):
                    a, e = red.bounds[bi], red.bounds[bi + 1]
                    ptr = m.store.grad[a:e].data_ptr()
                    if ptr in snap:
                        final = float(m.store.grad[a:e].double().abs().sum().item())
                        assert abs(snap[ptr] - final) <= 1e-9 * max(final, 1.0), (bi, snap[ptr], final)
                if task == "vqa":
                    # backward order (round 3): the text tower runs SECOND in the forward pass, so its backward -- and the
                    # word-embedding table's gradient, the largest single bucket of the real model -- completes in the first
                    # part of backward, and the bucket of the image tower's first parameters (the small tail bucket) is the last
                    # one the hooks release
                    order = red.finish_order[:early]
                    emb = red.bucket_of[id(m.language_encoder.embeddings.word_embeddings.weight)]
                    conv = red.bucket_of[id(m.vision_encoder.visual.conv1.weight)]
                    assert emb in order and conv in order
                    assert order.index(emb) < order.index(conv), (order.index(emb), order.index(conv))
                    text_last = max(order.index(red.bucket_of[id(p)]) for n, p in m.language_encoder.named_parameters()
                                    if p.requires_grad and red.bucket_of.get(id(p)) in order)
                    assert text_last < order.index(conv)
    finally:
        red.detach()
        from m3ae_amd import ops as _ops
        assert _ops.NT_NO_PERSISTENT is False   # detach() restored the launch policy attach() had set


def test_grad_reducer_over_rccl_single_rank_group():
    """The reducer on the REAL backend (`nccl` == RCCL) in a one-rank group on the GPU: the bucket all-reduces are issued
    asynchronously from the backward hooks on RCCL's stream, `finish()` joins them, and three optimizer steps give the
    losses of the same three steps without any reducer.  (A one-rank sum is the identity: what this covers is the API /
    stream-ordering path bench.py and the trainer use for N > 1, which the gloo tests cannot.)"""
    import torch.distributed as dist
    from m3ae_amd import ops
    from m3ae_amd.ddp import FlatGradReducer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = tiny_config(compute_dtype="bf16")
        b = to_dev(tiny_batch())

        def run(with_reducer, in_backward=False):
            m = build(cfg, torch.bfloat16)
            m.train()
            red = None
            if with_reducer:
                red = FlatGradReducer(m.store, bucket_bytes=128 << 10, update_in_backward=in_backward)
                red.world = 2                      # take the hook path; the group itself has one rank
                red.attach()
            losses = []
            try:
                for step in range(3):
                    m.store.zero_grad()
                    ops.set_dropout_seed(11 + step)
                    if in_backward:                # AdamW bucket by bucket behind each bucket's all-reduce, under backward
                        red.arm_update(max_steps=10, grad_scale=1.0)
                    loss = m.training_step(b)
                    loss.backward()
                    if red is not None:
                        early = sum(red.launched)
                        red.finish()
                        assert (early == 0) if step == 0 else (early >= red.nb // 2), (step, early, red.nb)
                        if in_backward:
                            assert (red.updated_in_backward == 0) if step == 0 else (red.updated_in_backward >= red.nb // 2 - 1)
                    if not in_backward:
                        m.store.adamw_step(max_steps=10, grad_scale=1.0)
                    losses.append(loss.item())
            finally:
                if red is not None:
                    red.detach()
            return losses, m.store.flat.detach().clone(), m.store.shadow.detach().clone()

        l0, p0, s0 = run(False)
        l1, p1, s1 = run(True)
        l2, p2, s2 = run(True, in_backward=True)
        # fp32 atomics in the split reductions: run-to-run differences of a few ulp are expected, nothing more
        assert np.allclose(l0, l1, rtol=1e-5, atol=0), (l0, l1)
        assert torch.allclose(p0, p1, rtol=0, atol=2e-5), float((p0 - p1).abs().max())
        assert np.allclose(l0, l2, rtol=1e-5, atol=0), (l0, l2)
        assert torch.allclose(p0, p2, rtol=0, atol=2e-5), float((p0 - p2).abs().max())
        assert torch.allclose(s0.float(), s2.float(), rtol=0, atol=1e-3)
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.parametrize("mode,buckets", [("fp32", "fp32"), ("bf16", "fp32"), ("bf16", "bf16")])
def test_two_virtual_ranks_equal_the_unsplit_global_batch_step(mode, buckets):
    """Data-parallel semantics end to end (main.py:59-63: Lightning DDP averages the ranks' gradients): ONE batch of 4 is
    split into two "virtual ranks" of 2 that run one after the other through the reducer's real hook path -- buckets
    released from the wgrad-completion hooks, SUM collective, 1 / world in the AdamW kernel -- with a summing stand-in for
    the collective that plays the other rank; after three optimizer steps the parameters equal those of three steps on the
    un-split batch of 4 (BCE `mean` over the batch: the mean of two half-batch means).
    buckets = "bf16": the gradient buckets travel as bf16 (FlatGradReducer(grad_dtype="bf16"): every rank's addend rounded once,
    the sum rounded once): the reduced gradient within 1e-2 relative L2 of the un-split batch's (stated tolerance, ddp.py), the
    parameter vector after three AdamW steps within 10 % of the distance it moved."""
    from m3ae_amd import ops
    from m3ae_amd.ddp import FlatGradReducer
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    cfg = tiny_config(compute_dtype=mode)
    full = to_dev(synth.synthetic_batch(4, text_len=32, image_size=64, vocab_size=1000, rank=0))

    def half(i):
        h = {}
        for k, v in full.items():
            if isinstance(v, torch.Tensor):
                h[k] = v[2 * i:2 * i + 2]
            elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
                h[k] = [t[2 * i:2 * i + 2] for t in v]
            elif isinstance(v, list):
                h[k] = v[2 * i:2 * i + 2]
            else:
                h[k] = v
        return h

    # reference: the un-split global batch, no reducer
    m0 = build(cfg, dtype)
    g0 = None
    for step in range(3):
        m0.store.zero_grad()
        m0.training_step(full).backward()
        if step == 0:
            g0 = m0.store.grad.clone()           # the un-split batch's gradient at the common starting point
        m0.store.adamw_step(max_steps=10, grad_scale=1.0)
    # two virtual ranks
    m1 = build(cfg, dtype)
    other = torch.zeros_like(m1.store.grad)     # what the "other rank" contributes to each bucket
    phase = {"record": True}
    released = []

    class Done:
        def wait(self):
            return None

    def collective(t):
        if buckets == "bf16":
            assert t.dtype == torch.bfloat16
            off = (t.data_ptr() - red._stage.data_ptr()) // 2
        else:
            off = (t.data_ptr() - m1.store.grad.data_ptr()) // 4
        released.append(off)
        if phase["record"]:
            other[off:off + t.numel()].copy_(t)     # rank 0's bucket, as released from its hooks
        else:
            t.add_(other[off:off + t.numel()].to(t.dtype))      # rank 1: SUM over the two ranks lands in place
        return Done()

    red = FlatGradReducer(m1.store, bucket_bytes=64 << 10, tail_bytes=8 << 10, collective=collective, world=2, grad_dtype=buckets)
    red.attach()
    try:
        for step in range(3):
            for r in (0, 1):
                phase["record"] = r == 0
                released.clear()
                m1.store.zero_grad()
                m1.training_step(half(r)).backward()
                early = len(released)
                red.finish()
                assert len(released) == red.nb
                if step > 0:
                    assert early >= red.nb // 2, (early, red.nb)    # the hook path really released buckets during backward
            if step == 0:   # the reduced gradient (SUM over the two ranks) x 1 / world against the un-split batch's
                gerr = ((m1.store.grad * 0.5 - g0).double().norm() / g0.double().norm()).item()
                assert gerr < (1e-2 if buckets == "bf16" else (1e-5 if mode == "fp32" else 2e-3)), gerr
            m1.store.adamw_step(max_steps=10, grad_scale=red.grad_scale)
    finally:
        red.detach()
    assert red.grad_scale == 0.5
    a, b = m0.store.flat[: m0.store.trainable_end], m1.store.flat[: m1.store.trainable_end]
    moved = (a - build(cfg, dtype).store.flat[: m0.store.trainable_end]).abs().max().item()
    assert moved > 1e-4
    # fp32: same arithmetic up to the summation order of the batch reduction; bf16: per-sample activations are identical
    # (every row is reduced in the same order wherever it sits), the split changes only fp32 accumulation order
    tol = 2e-6 if mode == "fp32" else 2e-5
    if buckets == "fp32":
        assert (a - b).abs().max().item() < tol + 1e-3 * moved, ((a - b).abs().max().item(), moved)
    else:
        # bf16 buckets: AdamW's update is m / sqrt(v) -- a parameter whose gradient is rounding noise can move by lr either way,
        # so single elements are not comparable; the parameter vector as a whole follows the fp32-bucket run (relative L2 of the
        # difference against the distance moved), and the reduced gradient itself was held to 1e-2 above
        a0 = build(cfg, dtype).store.flat[: m0.store.trainable_end]
        drift = ((a - b).double().norm() / (a - a0).double().norm()).item()
        assert drift < 0.1, drift
    # buckets were released in descending offset order inside every optimizer group (reverse execution order), and the bucket
    # that holds a group's first parameters is a small one
    sizes = red.bucket_bytes_list()
    assert min(sizes) <= (8 << 10) + 64 * 4 * 8 and max(sizes) <= (64 << 10) + max(p.numel() for p in m1.parameters()) * 4


def test_configure_optimizers_returns_a_torch_optimizer_and_scheduler():
    """m3ae_module.py:372-373 / m3ae_utils.py:240-242: `([optimizer], [{"scheduler", "interval": "step"}])`.  A
    Lightning-style loop over the returned objects (optimizer.step(); scheduler.step(); optimizer.zero_grad()) gives the
    parameters of ParamStore.adamw_step with its built-in schedule; param groups carry the reference's six
    (lr, weight_decay) pairs; state_dict() / load_state_dict() round-trip the moments into a fresh model."""
    cfg = tiny_config(compute_dtype="bf16")
    b = to_dev(tiny_batch())
    g = load_golden("tiny_vqa.npz")

    class StubTrainer:
        max_steps = 20

    m = build(cfg, torch.bfloat16)
    m.trainer_ref = StubTrainer()
    opts, scheds = m.configure_optimizers()
    opt, sched = opts[0], scheds[0]["scheduler"]
    assert isinstance(opt, torch.optim.Optimizer) and scheds[0]["interval"] == "step"
    assert isinstance(sched, torch.optim.lr_scheduler.LRScheduler)
    np.testing.assert_allclose([pg["initial_lr"] for pg in opt.param_groups], g["group_lr"], rtol=1e-6)
    np.testing.assert_allclose([pg["weight_decay"] for pg in opt.param_groups], g["group_wd"])
    names = {id(p): n for n, p in m.named_parameters()}
    mine = {names[id(p)]: gi for gi, pg in enumerate(opt.param_groups) for p in pg["params"] if id(p) in names}
    from m3ae_amd.param_store import NEVER_USED
    for n, gi in zip(g["group_names"].tolist(), g["group_index"].tolist()):
        if n in mine:
            assert mine[n] == gi, n
        else:   # the 6 tensors that never receive a gradient carry no optimizer state here
            assert any(n == u or n.endswith("." + u) for u in NEVER_USED), n
    ref = build(cfg, torch.bfloat16)
    lrs = []
    for step in range(4):
        lrs.append([pg["lr"] for pg in opt.param_groups])
        opt.zero_grad()
        m.training_step(b).backward()
        opt.step()
        sched.step()
        ref.store.zero_grad()
        ref.training_step(b).backward()
        ref.store.adamw_step(max_steps=20)
    # (fp32 atomics in the split reductions: run-to-run differences of a few ulp in the gradients, nothing more)
    assert torch.allclose(m.store.flat, ref.store.flat, rtol=0, atol=1e-6), float((m.store.flat - ref.store.flat).abs().max())
    warm = int(20 * cfg["warmup_steps"])
    from oracle import m3ae_oracle as O_
    for step, row in enumerate(lrs):
        f = O_.poly_lr_factor(step, warm, 20, cfg["learning_rate"], cfg["end_lr"], cfg["decay_power"])
        np.testing.assert_allclose(row, np.array(g["group_lr"]) * f, rtol=1e-6, atol=1e-15)
    # state round trip into a fresh model: the next step is identical
    sd_opt, sd_sched = opt.state_dict(), sched.state_dict()
    assert len(sd_opt["state"]) > 100 and all("exp_avg" in v for v in sd_opt["state"].values())
    m2 = build(cfg, torch.bfloat16)
    m2.load_state_dict(m.state_dict())
    m2.store.sync_shadows()
    m2.trainer_ref = StubTrainer()
    o2, s2 = m2.configure_optimizers()
    o2[0].load_state_dict(sd_opt)
    s2[0]["scheduler"].load_state_dict(sd_sched)
    for mm_, oo, ss in ((m, opt, sched), (m2, o2[0], s2[0]["scheduler"])):
        oo.zero_grad()
        mm_.training_step(b).backward()
        oo.step()
        ss.step()
    assert torch.allclose(m.store.flat, m2.store.flat, rtol=0, atol=1e-7)


def test_bench_line_contract():
    """bench.py as the driver runs it (a child process, defaults except a smaller batch and fewer steps): exactly one JSON
    line with the contract's keys, the roofline object of the dominant kernel measured live, finite positive numbers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600,
                       cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["per_gpu_batch"] == 64 and d["config"]["global_batch"] == 64
    assert d["value"] > 0 and abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0 < r["achieved"] < r["peak"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    assert d["cpu_baseline"] is None          # switched off above; the default run carries it
    assert d["secondary"] is None             # likewise (configs[2] line, a child process in the default run)
    sm = d["step_ms"]
    assert sm["min"] <= sm["p10"] <= sm["median"] <= sm["p90"] <= sm["max"] and sm["min"] > 0
    x = d["cross_attention_fwd"]
    assert x["path"].startswith("m3ae_xattn_fwd") and x["tflop"] == round(17.922 * 64 / 1e3, 3)
    x = d["cross_attention_fwd"]
    assert x["batch"] == 64 and 0 < x["frac"] < 1


def test_graphed_step_equals_eager_and_draws_fresh_dropout_masks_per_replay():
    """m3ae_amd/graph.py (ABI 3: dropout salt + AdamW hyper-parameters in device memory): a step captured as ONE hipGraph and
    replayed equals the eager steps in eval mode (same losses; parameters equal up to the order of the fp32 atomics) and, in
    train mode, every replay draws NEW dropout masks although the captured launches carry frozen seeds."""
    from m3ae_amd import ops
    from m3ae_amd.graph import GraphedStep

    def eager(m, b):
        m.store.zero_grad()
        loss = m.training_step(b)
        loss.backward()
        m.store.adamw_step(max_steps=50)
        return loss.detach()

    from m3ae_amd.modules.objectives import build_vqa_targets
    b = to_dev(tiny_batch())
    # a captured step must not build tensors on the host: the targets are a static device input of the graph, like the batch
    b["vqa_targets"] = build_vqa_targets(b, tiny_config()["vqa_label_size"], "cuda")
    m1 = build(tiny_config(compute_dtype="bf16"), torch.bfloat16)
    le = [eager(m1, b).item() for _ in range(4)]
    m2 = build(tiny_config(compute_dtype="bf16"), torch.bfloat16)
    lg = [eager(m2, b).item()]
    gs = GraphedStep(m2, b, max_steps=50)
    lg += [gs.step().item() for _ in range(3)]
    torch.cuda.synchronize()
    assert m2.store.step_count == m1.store.step_count == 4
    for a, c in zip(le, lg):
        assert abs(a - c) <= 2e-3 * abs(a), (le, lg)
    # AdamW normalises every gradient element: where the gradient is rounding noise (the order of the fp32 atomics differs between
    # two runs) an element moves by +-lr either way, so the two parameter vectors are compared in L2 against the distance moved
    moved = (m1.store.flat - build(tiny_config(compute_dtype="bf16"), torch.bfloat16).store.flat).double().norm().item()
    assert (m1.store.flat - m2.store.flat).double().norm().item() <= 0.1 * moved
    # train mode: fresh masks per replay (the loss of the same batch differs from replay to replay)
    m3 = build(tiny_config(compute_dtype="bf16"), torch.bfloat16)
    m3.train(True)
    eager(m3, b)
    gs3 = GraphedStep(m3, b, max_steps=10 ** 7)   # warm-up of 10 % of the steps: the learning rate stays ~0, only the masks change
    lt = [gs3.step().item() for _ in range(4)]
    assert len({round(x, 5) for x in lt}) >= 3, lt
    assert ops.DROPOUT_SALT is None and m3.store.hyper_dev is None   # capture restores the eager configuration


def test_dropout_salt_changes_the_mask_and_null_keeps_it():
    """ABI 3: m3ae_dropout with a device salt: salt = 0 is NOT the unsalted mask's... any salt value gives a valid mask of the right
    keep rate, different salts give different masks, and the same salt reproduces the mask (forward / backward consistency)."""
    import ctypes as C
    from m3ae_amd import _lib, ops
    rows, cols, p, seed = 64, 768, 0.1, 4242
    base = ops.dropout_keep_mask(rows, cols, p, seed)
    masks = []
    for sv in (1, 2, 1):
        ops.DROPOUT_SALT = torch.tensor([sv], dtype=torch.int32, device="cuda")
        try:
            masks.append(ops.dropout_keep_mask(rows, cols, p, seed))
        finally:
            ops.DROPOUT_SALT = None
    assert torch.equal(masks[0], masks[2]) and not torch.equal(masks[0], masks[1]) and not torch.equal(masks[0], base)
    for mk in masks:
        assert 0.88 < mk.float().mean().item() < 0.92
    assert torch.equal(base, ops.dropout_keep_mask(rows, cols, p, seed))
