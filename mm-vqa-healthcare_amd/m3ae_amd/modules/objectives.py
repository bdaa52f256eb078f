"""Loss arithmetic of the reference's m3ae/modules/objectives.py (the string metrics / logging around it are out
of scope).  Each function returns the same `ret` keys the reference does for the tensors it computes."""
import torch

from .. import ops


def build_vqa_targets(batch, label_size, device):
    """objectives.py:188-197 without the per-element host->device writes: one index_put from host lists.
    A pre-built device tensor `batch["vqa_targets"]` is used as is (bench: inputs resident in HBM)."""
    if "vqa_targets" in batch:
        return batch["vqa_targets"]
    rows, cols, vals = [], [], []
    for i, (ls, ss) in enumerate(zip(batch["vqa_labels"], batch["vqa_scores"])):
        for l, s in zip(ls, ss):
            rows.append(i)
            cols.append(l)
            vals.append(s)
    t = torch.zeros(len(batch["vqa_labels"]), label_size, dtype=torch.float32)
    if rows:
        t[rows, cols] = torch.tensor(vals, dtype=torch.float32)
    return t.to(device, non_blocking=True)


def compute_vqa_m3ae(pl_module, batch, test=False):
    """objectives.py:176-201."""
    infer = pl_module.infer(batch, mask_text=False, mask_image=False)
    vqa_logits = pl_module.vqa_head_forward(infer["multi_modal_cls_feats"])
    vqa_targets = build_vqa_targets(batch, pl_module.hparams.config["vqa_label_size"], vqa_logits.device)
    vqa_loss = ops.bce_with_logits_loss(vqa_logits, vqa_targets)
    return {
        "vqa_loss": vqa_loss,
        "vqa_logits": vqa_logits,
        "vqa_targets": vqa_targets,
        "vqa_labels": batch.get("vqa_labels"),
        "vqa_scores": batch.get("vqa_scores"),
        "multi_modal_cls_feats": infer["multi_modal_cls_feats"],
    }


def compute_mlm(pl_module, batch):
    """objectives.py:14-23."""
    infer = pl_module.infer(batch, mask_text=True, mask_image=False)
    mlm_logits = pl_module.mlm_head(infer["multi_modal_text_feats"])
    mlm_labels = infer["text_labels"]
    mlm_loss = ops.cross_entropy(mlm_logits, mlm_labels)
    return {"mlm_loss": mlm_loss, "mlm_logits": mlm_logits, "mlm_labels": mlm_labels, "mlm_ids": infer["text_ids"]}


def mim_targets(pl_module, images):
    """objectives.py:52-56 from the IMAGE (patchify + per-patch standardisation with the unbiased variance + 1e-6 in one
    kernel); a function of the input only (no parameters, no gradient)."""
    cfg = pl_module.hparams.config
    return ops.mim_targets(images, cfg["patch_size"], cfg["norm_pix_loss"])


def compute_mim(pl_module, batch):
    """objectives.py:41-62."""
    infer = pl_module.infer(batch, mask_text=False, mask_image=True)
    layer_idx = pl_module.hparams.config["mim_layer"]
    feats = infer["multi_modal_image_feats"] if layer_idx == -1 else infer[f"multi_modal_image_feats_{layer_idx}"]
    full = pl_module.mim_head(feats, infer["mim_ids_restore"], keep_cls=True)   # [B, L + 1, D]
    target = mim_targets(pl_module, batch["image_0" if "image_0" in batch else "image"][0])   # the image infer() used
    mim_loss = ops.mim_loss(full, target, infer["mim_masks"])   # masked per-patch MSE, class row skipped inside
    return {"mim_loss": mim_loss, "mim_logits": full[:, 1:, :], "mim_labels": target}


def compute_itm(pl_module, batch, itm_labels=None):
    """objectives.py:79-101.  `itm_labels` (0/1 per sample) may be supplied for reproducibility; default is the
    reference's random half/half permutation."""
    n = batch["text_ids"].shape[0]
    dev = batch["text_ids"].device
    if itm_labels is None:
        pos_len = n // 2
        itm_labels = torch.cat([torch.ones(pos_len), torch.zeros(n - pos_len)])[torch.randperm(n)]
    itm_labels = itm_labels.to(dev)
    sel = itm_labels.view(n, 1, 1, 1).bool()
    itm_images = [torch.where(sel, ti, fi) for ti, fi in zip(batch["image"], batch["false_image_0"])]
    batch = dict(batch)
    batch["image"] = itm_images
    infer = pl_module.infer(batch, mask_text=False, mask_image=False)
    itm_logits = pl_module.itm_head(infer["multi_modal_cls_feats"])
    itm_loss = ops.cross_entropy(itm_logits, itm_labels.long())
    return {"itm_loss": itm_loss, "itm_logits": itm_logits, "itm_labels": itm_labels}
