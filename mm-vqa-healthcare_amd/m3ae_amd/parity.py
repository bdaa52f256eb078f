"""Observed error of the bf16 MFMA path (what bench.py times) against the reference fixture at FULL size (configs[1] dimensions,
B = 2, eval mode): tests/golden/full_vqa.npz holds the reference's logits, loss and 663 per-parameter gradient norms
(oracle/make_golden.py from the unmodified reference modules).  Used by tests/test_gpu_model.py (asserts at 2 x the observed
values) and by bench.py's "parity" block (reports them next to the throughput)."""
import numpy as np
import torch


def parity_report(model, golden, batch):
    """One forward + backward of `model` on `batch`; errors against the fixture arrays in `golden`."""
    model.store.zero_grad()
    model.set_task()
    ret = model(batch)
    logits = ret["vqa_logits"].detach().float().cpu().numpy()
    ref_logits = golden["logits"]
    loss = ret["vqa_loss"]
    out = {
        "max_abs_dlogits": float(np.abs(logits - ref_logits).max()),
        "max_abs_logits_ref": float(np.abs(ref_logits).max()),
        "rms_dlogits": float(np.sqrt(((logits - ref_logits) ** 2).mean())),
        "loss": float(loss.item()),
        "loss_ref": float(golden["loss"]),
    }
    out["loss_rel_err"] = abs(out["loss"] - out["loss_ref"]) / out["loss_ref"]
    if "cls_feats" in golden:
        cf = ret["multi_modal_cls_feats"].detach().float().cpu().numpy()
        out["max_abs_dcls_feats"] = float(np.abs(cf - golden["cls_feats"]).max())
    loss.backward()
    torch.cuda.synchronize()
    names, ref = golden["grad_names"].tolist(), golden["grad_norm"]
    params = dict(model.named_parameters())
    mine = np.array([params[n].grad.double().norm().item() for n in names])
    gn_ref = float(golden["global_grad_norm"])
    gn = float(np.sqrt((mine ** 2).sum()))
    rel = np.abs(mine - ref) / np.maximum(ref, 1e-30)
    big = ref > 1e-2 * ref.max()
    out.update({
        "global_grad_norm": gn, "global_grad_norm_ref": gn_ref, "global_grad_norm_rel_err": abs(gn - gn_ref) / gn_ref,
        "n_param_grad_norms": int(len(names)), "n_large": int(big.sum()),
        "max_rel_err_large_param_grad_norms": float(rel[big].max()), "worst_large_param": names[int(np.argmax(np.where(big, rel, -1)))],
        "median_rel_err_param_grad_norms": float(np.median(rel)),
        "p99_rel_err_param_grad_norms": float(np.quantile(rel, 0.99)),
    })
    return out
