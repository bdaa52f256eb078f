"""Trainer / entry-point shim: what `main.py` / `main_t5_m3ae.py` + `pl.Trainer` do for the hot path,
without Lightning or sacred.

    python main.py with data_root=synthetic num_gpus=1 num_nodes=1 task_finetune_vqa_vqa_rad \
        per_gpu_batchsize=64 clip16 text_roberta image_size=384 tokenizer=downloaded/roberta-base

accepts the argument grammar of run_scripts/*.sh (`config.parse_cli`), and reproduces the reference's training
arithmetic around `training_step`:

* gradient accumulation `grad_steps = max(batch_size // (per_gpu_batchsize * num_gpus * num_nodes), 1)` (main.py:50,70):
  the loss of every micro-batch is divided by `grad_steps` (Lightning's `accumulate_grad_batches`), gradients add up
  in place in the flat buffer, the gradient all-reduce runs on the LAST micro-batch only (ddp.FlatGradReducer);
* `max_steps` optimizer steps, or `max_epoch` epochs when `max_steps` is None / -1 (main.py:51-52);
* the 6-group AdamW + polynomial schedule stepped once per optimizer step (m3ae_utils.py:112-242, ParamStore);
* validation every `val_check_interval` of an epoch with the VQA score `val/the_metric` (my_metrics.py:58-83), best +
  last checkpoints (`ModelCheckpoint(save_top_k=1, monitor="val/the_metric", mode="max", save_last=True)`,
  main.py:36-43) written as `{"state_dict": ...}` with the REFERENCE's key names, so the files load in the upstream
  code and upstream checkpoints load here (`load_path`); `save_weights_only` when "finetune" is in `exp_name`;
* one process per GPU: launched under `python -m torch.distributed.run --nproc-per-node N main.py with ...`
  (RANK / LOCAL_RANK / WORLD_SIZE), RCCL through `torch.distributed`; `num_gpus` * `num_nodes` must equal the world.

Data: `data_root=<dir with vqa_vqa_rad_{train,val,test}.arrow>` runs the arrow input pipeline (m3ae_amd/data.py, SURVEY
8f-2: host decode + bicubic resize in a thread pool, pinned uint8 upload on a side stream, ToTensor + Normalize on
the GPU); `data_root=synthetic` (or empty) selects `SyntheticDataModule`, which serves deterministic batches with the
same collate schema (base_dataset.py:165-228).
"""
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

from . import config as config_mod
from . import ops, synth
from .ddp import FlatGradReducer


def log(msg, rank=0):
    if rank == 0:
        print(f"[m3ae] {msg}", file=sys.stderr, flush=True)


def grad_steps_of(cfg, world):
    """main.py:50 (num_gpus * num_nodes is the world size under Lightning ddp)."""
    return max(cfg["batch_size"] // (cfg["per_gpu_batchsize"] * world), 1)


def plan(cfg, world, train_samples):
    """Optimizer-step budget exactly as main.py:49-52 + Lightning derive it."""
    gs = grad_steps_of(cfg, world)
    # len(train_dataloader): DistributedSampler gives every rank ceil(n / world) samples, the loader keeps the ragged last
    # batch (drop_last=False) -> ceil; Lightning steps the optimizer on an epoch's last batch even when the accumulation
    # window is incomplete -> ceil again for the steps of an epoch
    micro_per_epoch = max(math.ceil(math.ceil(train_samples / world) / cfg["per_gpu_batchsize"]), 1)
    steps_per_epoch = max(math.ceil(micro_per_epoch / gs), 1)
    ms = cfg["max_steps"]
    if ms is None or ms < 0:
        # m3ae_utils.py:212-217: len(train_dataloader) * max_epochs // accumulate_grad_batches
        max_steps, max_epochs = max(micro_per_epoch * cfg["max_epoch"] // gs, 1), cfg["max_epoch"]
    else:
        max_steps, max_epochs = ms, 1000
    return dict(grad_steps=gs, micro_per_epoch=micro_per_epoch, steps_per_epoch=steps_per_epoch, max_steps=max_steps,
                max_epochs=max_epochs)


class SyntheticDataModule:
    """Stand-in for MTDataModule (multitask_datamodule.py:11-82): a fixed pool of deterministic batches with the
    reference collate schema, sharded by rank.  `train_samples` mimics the dataset length (VQA-RAD train: 3064)."""

    def __init__(self, cfg, rank=0, world=1, device="cuda", head="cls", pool=4, train_samples=None, val_samples=None):
        self.cfg, self.rank, self.world, self.device, self.head = cfg, rank, world, device, head
        self.B = cfg["per_gpu_batchsize"]
        self.train_samples = train_samples or cfg.get("synthetic_train_samples", 3064)
        self.val_samples = val_samples or cfg.get("synthetic_val_samples", 451)
        self.pretrain = any(cfg["loss_names"][k] > 0 for k in ("mlm", "mim", "itm"))
        self.pool = [self._make(i) for i in range(pool)]
        self.val_pool = [self._make(1000 + i) for i in range(2)]

    def _make(self, idx):
        c = self.cfg
        b = synth.synthetic_batch(self.B, text_len=c["max_text_len"], image_size=c["image_size"],
                                  vocab_size=c["vocab_size"], label_size=c["vqa_label_size"],
                                  rank=self.rank + self.world * idx, device=self.device, pretrain=self.pretrain)
        if self.head == "t5":
            lab = synth.det_randint("t5_labels", 2, 32128, (self.B, 6), salt=31 + self.rank + self.world * idx)
            lab[:, -1] = 1
            b["t5_labels"] = lab.to(self.device)
        if self.head == "decoder":  # [CLS] answer tokens [SEP] [PAD]... with BERT's special ids (m3ae_decoder.py:336-342)
            import torch
            T = 6
            ids = synth.det_randint("dec_tokens", 1000, 30522, (self.B, T), salt=37 + self.rank + self.world * idx)
            n = synth.det_randint("dec_len", 1, T - 1, (self.B,), salt=38 + self.rank + self.world * idx)
            pos = torch.arange(T).view(1, T)
            ids = torch.where(pos == 0, torch.full_like(ids, 101), ids)
            ids = torch.where(pos == (n.view(-1, 1) + 1), torch.full_like(ids, 102), ids)
            ids = torch.where(pos > (n.view(-1, 1) + 1), torch.zeros_like(ids), ids)
            b["decoder_tokens"] = ids.to(self.device)
        return b

    def train_batches(self, epoch):
        n = max(math.ceil(math.ceil(self.train_samples / self.world) / self.B), 1)   # == plan()'s micro_per_epoch
        for i in range(n):
            yield self.pool[(epoch * n + i) % len(self.pool)]

    def val_batches(self):
        n = max(self.val_samples // (self.B * self.world), 1)
        for i in range(n):
            yield self.val_pool[i % len(self.val_pool)]


def vqa_score(logits, targets):
    """VQAScore.update (my_metrics.py:66-79): one-hot of the arg-max logit against the soft targets."""
    idx = logits.float().argmax(dim=1)
    return targets.float().gather(1, idx.view(-1, 1)).sum(), logits.shape[0]


def state_dict_cpu(model):
    return {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items()}


class Trainer:
    def __init__(self, cfg, model, dm, rank=0, world=1, device="cuda", log_every=10):
        self.cfg, self.model, self.dm, self.rank, self.world, self.device = cfg, model, dm, rank, world, device
        self.store = model.store
        self.reducer = FlatGradReducer(self.store)
        self.plan = plan(cfg, world, dm.train_samples)
        if cfg["decay_power"] != "cosine" and not isinstance(cfg["decay_power"], (int, float)):
            raise ValueError(f"decay_power must be a number or 'cosine' (m3ae_utils.py:225), got {cfg['decay_power']!r}")
        # the reference's ([optimizer], [{"scheduler", "interval": "step"}]) (m3ae_utils.py:240-242), driven as Lightning drives it
        self.max_steps = self.plan["max_steps"]
        model.trainer_ref = self
        opts, scheds = model.configure_optimizers()
        self.optimizer, self.scheduler = opts[0], scheds[0]["scheduler"]
        self.global_step, self.epoch, self.best = 0, 0, -1.0
        self.log_every = log_every
        exp = cfg["exp_name"]
        run_name = f'{exp}-seed{cfg["seed"]}-from_{str(cfg["load_path"]).replace("/", "_")}'  # main.py:31
        self.ckpt_dir = os.path.join(cfg["log_dir"], run_name, "checkpoints")
        self.weights_only = "finetune" in exp  # main.py:42
        self.history = []

    # -- checkpoints (reference key names; loadable by m3ae_module.py:105-113 upstream) ------------------------
    def save(self, name, metric=None):
        if self.rank != 0:
            return None
        os.makedirs(self.ckpt_dir, exist_ok=True)
        ck = {"state_dict": state_dict_cpu(self.model), "global_step": self.global_step, "epoch": self.epoch,
              "hyper_parameters": {"config": {k: v for k, v in self.cfg.items()}}, "val/the_metric": metric}
        if not self.weights_only:   # Lightning's keys
            ck["optimizer_states"] = [self.optimizer.state_dict()]
            ck["lr_schedulers"] = [self.scheduler.state_dict()]
        path = os.path.join(self.ckpt_dir, name)
        torch.save(ck, path)
        return path

    def resume(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["state_dict"], strict=False)
        self.store.sync_shadows()
        self.global_step, self.epoch = ck.get("global_step", 0), ck.get("epoch", 0)
        of = ck.get("optimizer_flat")   # checkpoints written before the Optimizer face existed
        if ck.get("optimizer_states"):
            self.optimizer.load_state_dict(ck["optimizer_states"][0])
            self.scheduler.load_state_dict(ck["lr_schedulers"][0])
        elif of is not None and of["exp_avg"] is not None:
            self.store.exp_avg = of["exp_avg"].to(self.device)
            self.store.exp_avg_sq = of["exp_avg_sq"].to(self.device)
            self.store.step_count = of["step_count"]
        if not ck.get("optimizer_states"):
            if of is None:
                self.store.step_count = self.global_step
            self.scheduler.last_epoch = self.store.step_count   # LambdaLR: lr = base * lambda(last_epoch)
            for g, base in zip(self.optimizer.param_groups, self.scheduler.base_lrs):
                g["lr"] = base * self.store.lr_factor(self.store.step_count, self.max_steps)
        return ck

    # -- loops ------------------------------------------------------------------------------------------------
    def _loss(self, batch):
        out = self.model.training_step(batch)
        return out["loss"] if isinstance(out, dict) else out

    def validate(self):
        self.model.eval()
        tot, cnt = torch.zeros((), device=self.device), 0
        with torch.no_grad():
            for batch in self.dm.val_batches():
                n = batch["text_ids"].shape[0]
                ret = None
                if hasattr(self.model, "vqa_head_forward"):
                    self.model.set_task()
                    ret = self.model(batch, test=True)
                if ret is not None and "vqa_logits" in ret:
                    s, n = vqa_score(ret["vqa_logits"], ret["vqa_targets"])
                elif ret is not None:   # pre-training objectives: negative summed loss as the monitored quantity
                    s = -sum(v.float() for k, v in ret.items() if k.endswith("_loss")) * n
                else:                   # generator heads: negative teacher-forced loss
                    s = -self._loss(batch).float() * n
                tot += s
                cnt += n
        t = torch.stack([tot, torch.tensor(float(cnt), device=self.device)])
        if self.world > 1:
            dist.all_reduce(t)
        self.model.train()
        return (t[0] / t[1]).item()

    def fit(self):
        """The training loop, issued on the library's high-priority launch stream (ops.launch_stream: the image half's kernels are
        dispatched ahead of the text half's on the side stream); the caller's stream is current again when it returns."""
        if not torch.cuda.is_available():
            return self._fit()
        prev = ops.use_launch_stream()
        try:
            return self._fit()
        finally:
            torch.cuda.synchronize()
            torch.cuda.set_stream(prev)

    def _fit(self):
        P, cfg = self.plan, self.cfg
        gs = P["grad_steps"]
        self.model.train()
        ops.set_dropout_seed(cfg["seed"] * 1000003 + self.rank)
        vci = cfg["val_check_interval"]
        val_every = max(int(P["micro_per_epoch"] * vci), 1) if isinstance(vci, float) else int(vci)
        log(f"fit: world {self.world}, per-GPU batch {cfg['per_gpu_batchsize']}, grad_steps {gs}, "
            f"{P['steps_per_epoch']} optimizer steps/epoch, max_steps {P['max_steps']}", self.rank)
        t0, seen = time.perf_counter(), 0
        done = self.global_step >= P["max_steps"]
        while not done and self.epoch < P["max_epochs"]:
            micro, nbatch = 0, 0   # position inside the accumulation window / batches of this epoch
            it = iter(self.dm.train_batches(self.epoch))
            nxt = next(it, None)
            while nxt is not None:
                batch, nxt = nxt, next(it, None)
                # a window ends after grad_steps micro-batches -- or with the epoch (Lightning steps on the last batch)
                first, last = micro % gs == 0, (micro % gs == gs - 1 or nxt is None)
                if first:
                    self.optimizer.zero_grad()
                # exchange gradients only on the window's last micro-batch (Lightning skips the DDP sync otherwise)
                (self.reducer.attach() if last else self.reducer.detach())
                loss = self._loss(batch) / gs
                loss.backward()
                micro = 0 if last else micro + 1
                nbatch += 1
                seen += cfg["per_gpu_batchsize"] * self.world
                if last:
                    self.reducer.finish()
                    self.optimizer.grad_scale = self.reducer.grad_scale
                    self.optimizer.step()
                    self.scheduler.step()
                    self.global_step += 1
                    if self.global_step % self.log_every == 0 or self.global_step == 1:
                        lv = loss.item() * gs
                        dt = time.perf_counter() - t0
                        self.history.append((self.global_step, lv))
                        log(f"epoch {self.epoch} step {self.global_step}/{P['max_steps']} loss {lv:.4f} "
                            f"lr x{self.store.lr_factor(self.store.step_count, P['max_steps']):.4f} "
                            f"{seen / dt:.1f} pairs/s", self.rank)
                    if self.global_step >= P["max_steps"]:
                        done = True
                if nbatch % val_every == 0 or done:
                    m = self.validate()
                    log(f"val/the_metric {m:.4f} (best {max(self.best, m):.4f})", self.rank)
                    if m > self.best:
                        self.best = m
                        self.save("best.ckpt", m)
                    self.save("last.ckpt", m)
                if done:
                    break
            self.epoch += 1
        self.reducer.detach()
        return {"global_step": self.global_step, "best": self.best, "history": self.history}

    def test(self):
        m = self.validate()
        log(f"test/the_metric {m:.4f}", self.rank)
        return m


def init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from .ddp import rccl_group_options
        dist.init_process_group("nccl", device_id=dev, pg_options=rccl_group_options())  # nccl == RCCL on ROCm; its streams at the launch stream's priority
    return rank, world, dev


def build_model(cfg, head, device):
    from .modules import DecoderModel, M3AETransformerSS, T5VQA_MMEncoderInput
    if head == "t5":
        model = T5VQA_MMEncoderInput(cfg)
        model.unfreeze_top_layers(cfg["unfreeze_num_encoder_layers"], cfg["unfreeze_num_decoder_layers"])
    elif head == "decoder":
        model = DecoderModel(cfg)
    else:
        model = M3AETransformerSS(cfg)
    lp = cfg["load_path"]
    if lp and os.path.exists(lp):
        target = model.m3ae if head in ("t5", "decoder") else model
        target._load(lp)  # m3ae_module.py:104-113: ckpt["state_dict"], strict=False, pos-embed resize
    else:
        # no checkpoint on disk (offline box): deterministic random init of the named architecture
        synth.fill_deterministic(model)
    dt = torch.bfloat16 if cfg.get("compute_dtype", "bf16") == "bf16" else torch.float32
    model.finalize(device, dt)
    return model


def run(argv, head="cls", tokenizer=None):
    """Entry point shared by main.py / main_t5_m3ae.py / main_decoder_m3ae.py: `python main.py with k=v ...
    named_config ...`.  `tokenizer`: optional callable for the arrow pipeline (default: the `tokenizer=` directory)."""
    cfg = config_mod.parse_cli(argv)
    rank, world, dev = init_distributed()
    expect = (cfg["num_gpus"] if isinstance(cfg["num_gpus"], int) else len(cfg["num_gpus"])) * cfg["num_nodes"]
    if expect != world:
        raise SystemExit(f"num_gpus * num_nodes = {expect} but the launcher started {world} process(es): start one "
                         f"process per GPU with `python -m torch.distributed.run --nproc-per-node {expect} ...`")
    if cfg["per_gpu_batchsize"] <= 0:
        raise SystemExit("per_gpu_batchsize must be set (run_scripts/*.sh pass it explicitly)")
    torch.manual_seed(cfg["seed"])  # pl.seed_everything (main.py:19)
    root = cfg["data_root"]
    model = build_model(cfg, head, dev)
    if root in ("", "synthetic"):
        dm = SyntheticDataModule(cfg, rank, world, dev, head=head)
    else:
        from .data import ArrowDataModule  # arrow reader + CLIP transform + collate, prefetched
        if head != "cls" and getattr(model, "tokenizer", None) is None:
            raise SystemExit("the generator heads need their answer tokenizer (T5 / BERT vocabulary files) for real data: "
                             "construct the model with `tokenizer=` or use data_root=synthetic")
        dm = ArrowDataModule(cfg, rank, world, dev, tokenizer=tokenizer, head=head)
    tr = Trainer(cfg, model, dm, rank, world, dev)
    if cfg.get("resume_from"):
        tr.resume(cfg["resume_from"])
    out = {}
    if not cfg["test_only"]:
        out = tr.fit()
        if world > 1:
            dist.barrier()  # rank 0 has written best.ckpt
        if "finetune" in cfg["exp_name"] and os.path.exists(os.path.join(tr.ckpt_dir, "best.ckpt")):
            step = tr.global_step
            tr.resume(os.path.join(tr.ckpt_dir, "best.ckpt"))  # trainer.test(ckpt_path="best") (main.py:80)
            tr.global_step = out["global_step"] = step
    out["test"] = tr.test()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({k: v for k, v in out.items() if k != "history"}))
    return out
