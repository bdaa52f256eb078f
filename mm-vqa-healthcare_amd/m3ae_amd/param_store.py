"""Flat parameter / gradient / optimizer-state storage laid out for MI355X (288 GB HBM3E: everything resident).

All trainable parameters live in ONE fp32 buffer ordered by optimizer group (m3ae_utils.py:135-204's six groups),
each tensor 64-element aligned, with `param.data` and `param.grad` re-pointed to views of the flat buffers:
  * AdamW is one fused kernel launch per group segment (6 launches instead of ~670 per-tensor updates);
  * the gradient all-reduce operates on contiguous byte ranges of the flat gradient buffer (m3ae_amd/ddp.py);
  * query/key/value weights of one attention block are adjacent, so the packed [3D, D] (or [2D, D]) projection the
    kernels want is a zero-copy view (PackedParam) while state_dict names stay the reference's;
  * in bf16 mode the AdamW kernel also writes the bf16 shadow the forward GEMMs read (same offsets), and each GEMM
    weight unit gets a transposed bf16 copy [K, N] so dgrad is the same K-contiguous "NT" MFMA kernel.
Parameters that never receive a gradient (CLIP text-tower leftovers, RoBERTa pooler) keep their names and
values but are left out of the gradient / optimizer / all-reduce buffers.
"""
import ctypes as C
import math

import torch

from . import _lib

ALIGN = 64

NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight"]
HEAD_NAMES = ["mlm_head", "mim_head", "itm_head", "vqa_head", "cls_head", "irtr_head"]
NEVER_USED = ("vision_encoder.positional_embedding", "vision_encoder.token_embedding.weight",
              "vision_encoder.ln_final.weight", "vision_encoder.ln_final.bias",
              "language_encoder.pooler.dense.weight", "language_encoder.pooler.dense.bias")


def param_group_of(name):
    """Index 0..5 of the reference's six AdamW groups (m3ae_utils.py:135-204), by substring match on the name."""
    nd = any(k in name for k in NO_DECAY)
    hd = any(k in name for k in HEAD_NAMES)
    mm = "multi_modal" in name
    if not hd and not mm:
        return 1 if nd else 0
    if hd and not mm:
        return 3 if nd else 2
    if mm and not hd:
        return 5 if nd else 4
    return -1


def group_hparams(cfg):
    lr, wd = cfg["learning_rate"], cfg["weight_decay"]
    lh, lm = cfg["lr_multiplier_head"], cfg["lr_multiplier_multi_modal"]
    return [(lr, wd), (lr, 0.0), (lr * lh, wd), (lr * lh, 0.0), (lr * lm, wd), (lr * lm, 0.0)]


class PackedParam:
    """Zero-copy [sum(N_i), K] view over adjacent parameters (q|k|v weights or biases) of the flat buffers."""

    def __init__(self, members):
        self.members = list(members)
        m0 = self.members[0]
        rows = sum(m.shape[0] for m in self.members)
        self.shape = (rows,) + tuple(m0.shape[1:])
        self.m3ae_t = None
        self._check_adjacent()

    def _check_adjacent(self):
        off = 0
        m0 = self.members[0]
        for m in self.members:
            if m.data_ptr() != m0.data_ptr() + off * m0.element_size():
                raise _lib.M3AEHipError("PackedParam members are not adjacent: apply ParamStore before forward")
            off += m.numel()

    def _view(self, t0):
        stride = (self.shape[1], 1) if len(self.shape) == 2 else (1,)
        return t0.as_strided(self.shape, stride)

    @property
    def data(self):
        return self._view(self.members[0].data)

    @property
    def m3ae_c(self):
        m0 = self.members[0]
        return self._view(getattr(m0, "m3ae_c", m0.data))

    @property
    def requires_grad(self):
        return all(m.requires_grad for m in self.members)

    @property
    def grad(self):
        g0 = self.members[0].grad
        return None if g0 is None else self._view(g0)


def param_group_of_decoder(name):
    """The two groups of m3ae_t5_utils.set_schedule_decoder (:308-333): no-decay by substring, one learning rate."""
    nd = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight"]
    return 1 if any(k in name for k in nd) else 0


def group_hparams_decoder(cfg):
    lr, wd = cfg["learning_rate"], cfg["weight_decay"]
    return [(lr, wd), (lr, 0.0)] + [(lr, 0.0)] * 4


class ParamStore:
    def __init__(self, module, cfg, device, compute_dtype=torch.bfloat16, weight_units=None, frozen=(),
                 group_fn=param_group_of, hparams_fn=group_hparams):
        self.module, self.cfg, self.device, self.compute_dtype = module, cfg, device, compute_dtype
        # the HIP streams that carry this module's backward work when there is more than one (M3AETransformerSS runs its text
        # half on a second stream): consumers of the flat gradient buffer that run INSIDE a backward pass (ddp.FlatGradReducer)
        # make their own stream wait for the others
        self.streams = set()
        self.hparams_fn = hparams_fn
        named = [(n, p) for n, p in module.named_parameters()]
        self.names = {id(p): n for n, p in named}
        # 0..5 optimizer groups; 6..11 = the same classes for parameters without gradient (frozen / never used), kept
        # apart so that weights stay adjacent to weights and biases to biases (PackedParam views) there too
        groups = [[] for _ in range(12)]
        for n, p in named:
            gi = group_fn(n)
            unused = any(n == u or n.endswith("." + u) for u in NEVER_USED) or any(n.startswith(f) for f in frozen)
            if unused or gi < 0 or not p.requires_grad:
                groups[6 + max(gi, 0)].append((n, p))
            else:
                groups[gi].append((n, p))
        self.groups = groups
        # offsets
        self.offset, self.segments = {}, []
        off = 0
        for gi, g in enumerate(groups):
            start = off
            for n, p in g:
                self.offset[id(p)] = off
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.segments.append((start, off))
        self.total = off
        self.trainable_end = self.segments[5][1]
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.trainable_end, dtype=torch.float32, device=device)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=device) \
            if compute_dtype == torch.bfloat16 else None
        for gi, g in enumerate(groups):
            for n, p in g:
                o = self.offset[id(p)]
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                if gi < 6:
                    p.grad = self.grad[o:o + p.numel()].view(p.shape)
                else:
                    p.requires_grad_(False)
                    p.grad = None
                if self.shadow is not None:
                    p.m3ae_c = self.shadow[o:o + p.numel()].view(p.shape)
        self.step_count = 0
        # weight units: GEMM weights that need a transposed bf16 copy for dgrad
        self.units = list(weight_units() if callable(weight_units) else (weight_units or []))
        self._t_bufs = []
        if self.shadow is not None:
            for u in self.units:
                shp = u.shape
                rows, cols = shp[0], int(math.prod(shp[1:]))
                t = torch.empty((cols, rows), dtype=torch.bfloat16, device=device)
                u.m3ae_t = t
                self._t_bufs.append((u, rows, cols, t))
            # job table for the one-launch transposer
            import numpy as np
            tab = np.zeros((len(self._t_bufs), 5), dtype=np.int64)
            first = 0
            for i, (u, rows, cols, t) in enumerate(self._t_bufs):
                # source = the unit's bf16 shadow (same element offset as its fp32 master in the flat buffer)
                off = (u.data.data_ptr() - self.flat.data_ptr()) // 4
                assert 0 <= off < self.total
                tab[i] = (self.shadow.data_ptr() + 2 * off, t.data_ptr(), rows, cols, first)
                first += ((rows + 63) // 64) * ((cols + 63) // 64)
            self._t_tiles = first
            self._t_jobs = torch.from_numpy(tab).to(device)
        self.sync_shadows()

    # ---- shadows -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sync_shadows(self, cast=True):
        """Refresh bf16 shadows from the fp32 masters (after load_state_dict / manual edits), and the transposed
        copies of every GEMM weight unit.  After an optimizer step only the transposes are needed (cast=False)."""
        if self.shadow is None:
            return
        L = _lib.lib()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if cast:
            _lib.check(L.m3ae_cast(C.c_void_p(self.flat.data_ptr()), C.c_void_p(self.shadow.data_ptr()), self.total,
                                   _lib.F32, _lib.BF16, s), "m3ae_cast")
        if self._t_bufs:
            _lib.check(L.m3ae_transpose_bf16_batched(C.c_void_p(self._t_jobs.data_ptr()), len(self._t_bufs),
                                                     self._t_tiles, s), "m3ae_transpose_bf16_batched")

    # ---- optimizer -----------------------------------------------------------------------------------------
    def zero_grad(self):
        from . import ops as _ops
        _lib.check(_lib.lib().m3ae_zero(C.c_void_p(self.grad.data_ptr()), self.grad.numel() * 4, _ops._stream()), "m3ae_zero")

    def lr_factor(self, step, max_steps):
        """transformers get_polynomial_decay_schedule_with_warmup (m3ae_utils.py:232-238), lr_init-relative."""
        cfg = self.cfg
        warm = cfg["warmup_steps"]
        if isinstance(warm, float):
            warm = int(max_steps * warm)
        lr_init, lr_end, power = cfg["learning_rate"], cfg["end_lr"], cfg["decay_power"]
        if step < warm:
            return float(step) / float(max(1, warm))
        if power == "cosine":  # transformers get_cosine_schedule_with_warmup (m3ae_utils.py:225-230), half a cosine
            progress = float(step - warm) / float(max(1, max_steps - warm))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
        if step > max_steps:
            return lr_end / lr_init
        pct = 1 - (step - warm) / (max_steps - warm)
        return ((lr_init - lr_end) * pct ** power + lr_end) / lr_init

    def begin_update(self, max_steps=None, grad_scale=1.0, lr_factor=None, group_lrs=None, betas=(0.9, 0.98), eps=1e-8,
                     group_wds=None):
        """The host part of an optimizer step: per-group learning rates (built-in schedule unless given), step count.  Returns
        the hyper-parameter record `adamw_range` takes; `adamw_step` = begin_update + adamw_range over the six segments, and
        ddp.FlatGradReducer(update_in_backward=True) applies the same record bucket by bucket while backward still runs."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.grad)
            self.exp_avg_sq = torch.zeros_like(self.grad)
        hp = self.hparams_fn(self.cfg)
        if group_lrs is None:
            if lr_factor is None:
                max_steps = max_steps or self.cfg["max_steps"]
                lr_factor = self.lr_factor(self.step_count, max_steps)
            group_lrs = [lr * lr_factor for lr, _ in hp]
        if group_wds is None:
            group_wds = [wd for _, wd in hp]
        self.step_count += 1
        return dict(lrs=[float(x) for x in group_lrs], wds=[float(x) for x in group_wds], b1=float(betas[0]), b2=float(betas[1]),
                    eps=float(eps), step=self.step_count, grad_scale=grad_scale)

    # A hipGraph-captured step replays with frozen kernel arguments: graph.GraphedStep makes the AdamW launches read this step's
    # learning rate and bias-corrected step size per group from `hyper_dev` ([6][2] fp32 on the device, uploaded before a replay).
    hyper_dev = None

    def _hyper_dev_ptr(self, gi):
        return None if self.hyper_dev is None else C.c_void_p(self.hyper_dev.data_ptr() + 8 * gi)

    @staticmethod
    def hyper_values(hyper):
        """[6][2] host floats {lr, lr * sqrt(1 - b2^t) / (1 - b1^t)} of a begin_update() record: what m3ae_adamw computes from
        (lr, step) on the host when hyper_dev is NULL."""
        t = hyper["step"]
        bc1, bc2 = 1.0 - hyper["b1"] ** t, 1.0 - hyper["b2"] ** t
        return [[lr, lr * math.sqrt(bc2) / bc1] for lr in hyper["lrs"]]

    @torch.no_grad()
    def adamw_range(self, a, b, gi, hyper):
        """The fused AdamW kernel over elements [a, b) of the flat buffers (inside optimizer group `gi`), on the current stream."""
        if b <= a:
            return
        from . import ops as _ops
        es = 4
        sh = C.c_void_p(self.shadow.data_ptr() + a * 2) if self.shadow is not None else None
        _lib.check(_lib.lib().m3ae_adamw(C.c_void_p(self.flat.data_ptr() + a * es), C.c_void_p(self.grad.data_ptr() + a * es),
                                         C.c_void_p(self.exp_avg.data_ptr() + a * es),
                                         C.c_void_p(self.exp_avg_sq.data_ptr() + a * es), sh, b - a, hyper["lrs"][gi],
                                         hyper["b1"], hyper["b2"], hyper["eps"], hyper["wds"][gi], hyper["step"],
                                         hyper["grad_scale"], self._hyper_dev_ptr(gi), _ops._stream()), "m3ae_adamw")

    @torch.no_grad()
    def adamw_step(self, max_steps=None, grad_scale=1.0, lr_factor=None, group_lrs=None, betas=(0.9, 0.98), eps=1e-8,
                   group_wds=None):
        """One AdamW step over the six group segments (m3ae_utils.py:206; transformers-4.6.0 AdamW semantics,
        betas (0.9, 0.98), eps 1e-8) + the polynomial-decay schedule, stepped per optimizer step.
        `group_lrs` (one learning rate per group, e.g. an Optimizer's param_groups after its scheduler stepped) overrides
        the built-in schedule."""
        self.adamw_apply(self.begin_update(max_steps, grad_scale, lr_factor, group_lrs, betas, eps, group_wds))

    @torch.no_grad()
    def adamw_apply(self, hyper):
        """The device part of an optimizer step for a begin_update() record: six fused launches + the transposed weight copies."""
        for gi in range(6):
            a, b = self.segments[gi]
            self.adamw_range(a, min(b, self.trainable_end), gi, hyper)
        self.sync_shadows(cast=False)

    def make_optimizer(self, max_steps=None):
        """([optimizer], [{"scheduler", "interval": "step"}]) in the shape m3ae_utils.set_schedule returns (:240-242)."""
        max_steps = max_steps or self.cfg["max_steps"]
        opt = FlatAdamW(self)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: self.lr_factor(step, max_steps))
        return [opt], [{"scheduler": sched, "interval": "step"}]

    def group_names(self):
        return [[n for n, _ in g] for g in self.groups[:6]]


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.Optimizer face of ParamStore's fused AdamW (the object m3ae_utils.set_schedule returns as
    `optimizer`, m3ae_utils.py:135-206): six param groups -- {base, head x lr_multiplier_head, multi_modal x
    lr_multiplier_multi_modal} x {weight decay, no decay} -- with the reference's lr / weight_decay / betas / eps per group,
    so a Lightning-style loop (`optimizer.step(); scheduler.step(); optimizer.zero_grad()`) and LambdaLR schedulers work.
    `step()` is one fused kernel launch per group over the flat buffers; the learning rates are read from
    `param_groups[i]["lr"]` (whatever a scheduler wrote there).  `state_dict()` / `load_state_dict()` round-trip the
    moments (per parameter, torch's layout) and the step count."""

    def __init__(self, store, grad_scale=1.0):
        self.store = store
        self.grad_scale = grad_scale
        groups = []
        for gi, (lr, wd) in enumerate(store.hparams_fn(store.cfg)):
            params = [p for _, p in store.groups[gi]]
            if not params:   # torch rejects empty groups: keep the slot with a placeholder so that indices stay group ids
                params = [torch.nn.Parameter(torch.zeros(0, device=store.device), requires_grad=False)]
            groups.append(dict(params=params, lr=lr, weight_decay=wd, m3ae_group=gi))
        super().__init__(groups, dict(lr=store.cfg["learning_rate"], betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0))

    def _bind_state(self):
        st = self.store
        if st.exp_avg is None:
            st.exp_avg = torch.zeros_like(st.grad)
            st.exp_avg_sq = torch.zeros_like(st.grad)
        for gi in range(6):
            for _, p in st.groups[gi]:
                o = st.offset[id(p)]
                self.state[p] = dict(step=torch.tensor(float(st.step_count)),
                                     exp_avg=st.exp_avg[o:o + p.numel()].view(p.shape),
                                     exp_avg_sq=st.exp_avg_sq[o:o + p.numel()].view(p.shape))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = {pg["m3ae_group"]: pg for pg in self.param_groups}
        self.store.adamw_step(group_lrs=[g[i]["lr"] for i in range(6)], group_wds=[g[i]["weight_decay"] for i in range(6)],
                              betas=g[0]["betas"], eps=g[0]["eps"], grad_scale=self.grad_scale)
        return loss

    def zero_grad(self, set_to_none=False):
        self.store.zero_grad()   # gradients are views of the flat buffer: never set to None

    def state_dict(self):
        self._bind_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        st = self.store
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        mine = [p for g in self.param_groups for p in g["params"]]
        if len(ids) != len(mine):
            raise ValueError("optimizer state_dict does not match this model's parameter groups")
        if st.exp_avg is None:
            st.exp_avg = torch.zeros_like(st.grad)
            st.exp_avg_sq = torch.zeros_like(st.grad)
        step = None
        for i, p in zip(ids, mine):
            s = state_dict["state"].get(i)
            if s is None or id(p) not in st.offset or p.numel() == 0:
                continue
            o = st.offset[id(p)]
            st.exp_avg[o:o + p.numel()].view(p.shape).copy_(s["exp_avg"])
            st.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(s["exp_avg_sq"])
            step = int(s["step"]) if step is None else step
        if step is not None:
            st.step_count = step
        for pg, sg in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "weight_decay", "betas", "eps", "initial_lr"):
                if k in sg:
                    pg[k] = sg[k]
        self._bind_state()
