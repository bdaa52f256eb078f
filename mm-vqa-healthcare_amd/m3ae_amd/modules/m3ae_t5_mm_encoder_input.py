"""T5VQA_MMEncoderInput on MI355X: frozen M3AE -> [prefix tokens | projected CLS] -> T5 encoder -> teacher-forced
decoder -> tied LM head -> cross-entropy (reference m3ae/modules/m3ae_t5_mm_encoder_input.py:12-295; the model
`main_t5_m3ae.py` trains, run_scripts/finetune_m3ae.sh).

Same class name / ctor (`T5VQA_MMEncoderInput(m3ae_config)`), same state_dict prefixes (`m3ae.*`, `t5.*`,
`feature_projection.*`), same `unfreeze_top_layers(num_encoder_layers, num_decoder_layers)` recipe, same
`training_step -> {'loss': ...}` contract.  Documented deviations:
  * the reference draws a FRESH random nn.Linear(1536 -> 512) for every sample of every call (:75-77,128-129) -- it is
    un-reproducible and untrained by construction.  Here it is ONE registered, frozen `cls_projection` (same shapes);
  * `prepare_inputs` is batched (no per-sample Python loop), same values: 2 prefix embeddings + projected CLS, zero
    padded to 512 rows, all-ones attention mask (:159-178) -- the padding rows are attended, as in the reference;
  * tokenisation happens outside: `batch["t5_labels"]` (int64 [B, T], pad 0, eos 1) or a `tokenizer` callable;
    "question:" is ids [822, 10] (the t5-small SentencePiece ids; not verifiable offline);
  * beam-search `generate` (t5.py) serves the inference / test branch (:207-218); the extra `generate` the reference runs
    inside every TRAINING step for its string metrics (:252-284) is not executed (metrics are out of scope).
"""
import torch
import torch.nn as nn

from .. import ops
from ..param_store import ParamStore, group_hparams_decoder, param_group_of_decoder
from .m3ae_module import M3AETransformerSS, _Base, _HParams, pl
from .t5 import T5ForConditionalGeneration

MAX_SEQ_LEN = 512  # m3ae_t5_mm_encoder_input.py:159


class T5VQA_MMEncoderInput(_Base):
    def __init__(self, m3ae_config, freeze_m3ae=True, freeze_t5_layers=True, tokenizer=None, t5_vocab=32128,
                 t5_dims=None):
        super().__init__()
        if pl is None:
            self.hparams = _HParams(m3ae_config=m3ae_config)
        else:
            self.save_hyperparameters(ignore=["tokenizer", "t5_dims"])
        self.m3ae = M3AETransformerSS(m3ae_config)
        if freeze_m3ae:
            for p in self.m3ae.parameters():
                p.requires_grad = False
        self.tokenizer = tokenizer
        self.t5 = T5ForConditionalGeneration(t5_dims or m3ae_config.get("t5_model_name", "t5-small"), t5_vocab)
        if freeze_t5_layers:
            for p in self.t5.parameters():
                p.requires_grad = False
        hs = self.m3ae.hparams.config["hidden_size"]
        d = self.t5.config.hidden_size
        self.feature_projection = nn.Linear(hs * 2, d)  # registered but unused, as in the reference (:40-43)
        self.cls_projection = nn.Linear(hs * 2, d)      # deviation: explicit stand-in for the per-call random Linear
        for p in self.cls_projection.parameters():
            p.requires_grad = False
        for p in self.feature_projection.parameters():
            p.requires_grad = False                       # never reached by a gradient in the reference either
        self.max_answer_length = m3ae_config.get("t5_max_length", 25)
        self.prefix_ids = [822, 10]                       # "question:"
        self.current_tasks = list()
        self.store = None

    def unfreeze_top_layers(self, num_encoder_layers=2, num_decoder_layers=2):
        """m3ae_t5_mm_encoder_input.py:79-96 (python negative-index semantics of range() included)."""
        for p in self.t5.parameters():
            p.requires_grad = False
        ne, nd = len(self.t5.encoder.block), len(self.t5.decoder.block)
        for i in range(ne - num_encoder_layers, ne):
            for p in self.t5.encoder.block[i].parameters():
                p.requires_grad = True
        for i in range(nd - num_decoder_layers, nd):
            for p in self.t5.decoder.block[i].layer[0].parameters():
                p.requires_grad = True
            for p in self.t5.decoder.block[i].layer[1].parameters():
                p.requires_grad = True

    def weight_units(self):
        return self.m3ae.weight_units() + self.t5.weight_units() + [self.cls_projection.weight]

    def finalize(self, device="cuda", compute_dtype=None):
        cfg = self.m3ae.hparams.config
        if compute_dtype is not None:
            self.m3ae._dtype = compute_dtype
        self.store = ParamStore(self, cfg, device, self.m3ae._dtype, self.weight_units,
                                group_fn=param_group_of_decoder, hparams_fn=group_hparams_decoder)
        self.m3ae.store = self.store
        return self

    def prepare_inputs(self, batch):
        """m3ae_t5_mm_encoder_input.py:100-190 with include_cls_feats=True, include_imagetext_feats=False (the
        run_scripts/finetune_m3ae.sh setting), batched."""
        with torch.no_grad():
            cls = self.m3ae.infer(batch)["multi_modal_cls_feats"]
        B = cls.shape[0]
        dev = cls.device
        dt = self.m3ae._dtype
        ids = torch.tensor(self.prefix_ids, device=dev).repeat(B)
        pre = self.t5.embed(ids.view(B, -1), dt)                                   # [B, P, d]
        proj = ops.linear(cls, self.cls_projection.weight, self.cls_projection.bias)  # [B, d]
        P = pre.shape[1]
        x = torch.zeros(B, MAX_SEQ_LEN, proj.shape[-1], dtype=dt, device=dev)
        x[:, :P] = pre
        x[:, P] = proj
        return {"inputs_embeds": x, "attention_mask": torch.ones(B, MAX_SEQ_LEN, dtype=torch.long, device=dev)}

    def labels_of(self, batch):
        if "t5_labels" in batch:
            return batch["t5_labels"]
        if self.tokenizer is None:
            raise ValueError("pass batch['t5_labels'] or construct with a tokenizer")
        flat = [a[0] for a in batch["vqa_answer"]]
        return self.tokenizer(flat, padding=True, truncation=True, return_tensors="pt").input_ids.to(
            batch["text_ids"].device)

    def forward(self, batch, test=False):
        """Training branch of m3ae_t5_mm_encoder_input.py:193-295."""
        if self.store is None:
            raise RuntimeError("call finalize(device) before the first forward")
        inputs = self.prepare_inputs(batch)
        if len(self.current_tasks) == 0 or test:  # inference / test: beam search (:207-218)
            with torch.no_grad():
                enc = self.t5.encoder(inputs["inputs_embeds"])
            return {"generated_ids": self.t5.generate(enc, num_beams=4, max_length=self.max_answer_length)}
        out = self.t5(inputs["inputs_embeds"], self.labels_of(batch))
        return {"vqa_loss": out.loss, "vqa_logits": out.logits}

    def training_step(self, batch, batch_idx=0):
        """m3ae_t5_mm_encoder_input.py:340-346."""
        self.current_tasks = [k for k, v in self.m3ae.hparams.config["loss_names"].items() if v > 0]
        output = self(batch)
        names = self.m3ae.hparams.config["loss_names"]
        total = sum(v * names[k.replace("_loss", "")] for k, v in output.items() if k.endswith("_loss"))
        return {"loss": total}

    def configure_optimizers(self):
        """m3ae_t5_utils.set_schedule_decoder (:290-375; `set_schedule` itself is commented out in the reference,
        SURVEY 9 #2): two groups, one lr, poly decay -- fused in ParamStore.adamw_step."""
        tr = getattr(self, "trainer_ref", None)
        return self.store.make_optimizer(getattr(tr, "max_steps", None) if tr is not None else None)
