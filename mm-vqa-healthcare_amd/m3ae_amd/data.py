"""Input pipeline for the hot path: arrow reader + CLIP transform + collate + tokenisation, producing
the batch dict of the drop-in boundary (8b) with the NEXT batch decoded, resized and uploaded while the current one
trains.

Reference pieces restated (host side, Python like the reference's):
  * `BaseDataset` (m3ae/datasets/base_dataset.py:12-228): `{data_dir}/{name}.arrow` written by prepro/make_arrow.py
    (:126-204: columns image, questions, answers, answer_labels, answer_scores, image_id, question_id, answer_type,
    split), one sample per (image row, question index) (`index_mapper`, :72-81);
  * `VQAVQARADDataset.__getitem__` (vqa_vqa_rad_dataset.py:24-43);
  * `clip_transform` (transforms/transform.py:60-67): PIL RGBA -> `Resize(size, BICUBIC)` (shorter side, torchvision's
    integer rounding) -> `CenterCrop(size)` -> RGB -> ToTensor -> Normalize(CLIP mean / std);
  * `collate` (base_dataset.py:165-228): fine-tuning keys, and with an `MLMCollator` the pre-training fields
    `text_ids_mlm` / `text_labels_mlm`;
  * the masked-language-model collators the datamodule picks (base_datamodule.py:62-69, third-party transformers==4.6.0
    `DataCollatorForWholeWordMask` / `DataCollatorForLanguageModeling`; the reference vendors the same file as
    m3ae/utils/data_collator.py:290-496) -> `MLMCollator`.

MI355X side: decode + bicubic resize stay on host cores (PIL releases the GIL; a thread pool of `num_workers`), the
crop is handed over as uint8 NHWC in PINNED memory (a quarter of the fp32 bytes over PCIe), copied on a side HIP
stream, and ToTensor + Normalize run in one kernel on the GPU (`m3ae_image_normalize_u8`, same IEEE arithmetic as
torch: bit-equal to the reference's tensor).  `ArrowDataModule.train_batches` keeps `prefetch` batches in flight.
"""
import ctypes as C
import io
import os
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from .synth import CLIP_MEAN, CLIP_STD


# ------------------------------------------------------------------------------------------------------------
# transform (host part)
# ------------------------------------------------------------------------------------------------------------
def clip_resize_crop(img, size):
    """PIL image -> uint8 [size, size, 3]: transform.py:60-64 (Resize BICUBIC on the RGBA image, CenterCrop, RGB)."""
    from PIL import Image
    img = img.convert("RGBA")                      # base_dataset.py:92-93
    w, h = img.size
    if w <= h:                                     # torchvision Resize(int): shorter side -> size, long side truncated
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    if (nw, nh) != (w, h):
        img = img.resize((nw, nh), Image.BICUBIC)
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    img = img.crop((left, top, left + size, top + size)).convert("RGB")
    return np.array(img, dtype=np.uint8)  # a writable copy (torch.from_numpy)


def normalize_on_device(u8_nhwc, stream=None):
    """uint8 [B, H, W, 3] (device) -> fp32 [B, 3, H, W]: ToTensor + Normalize in one kernel."""
    B, H, W, _ = u8_nhwc.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=u8_nhwc.device)
    mean = (C.c_float * 3)(*CLIP_MEAN)
    std = (C.c_float * 3)(*CLIP_STD)
    s = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    _lib.check(_lib.lib().m3ae_image_normalize_u8(C.c_void_p(u8_nhwc.data_ptr()), C.c_void_p(out.data_ptr()), B, H, W,
                                                  mean, std, s), "m3ae_image_normalize_u8")
    return out


# ------------------------------------------------------------------------------------------------------------
# tokenizer
# ------------------------------------------------------------------------------------------------------------
def load_tokenizer(cfg):
    """base_datamodule.py:13-26: `RobertaTokenizerFast.from_pretrained(path, local_files_only=True)`."""
    name = cfg["tokenizer"]
    from transformers import BertTokenizerFast, RobertaTokenizerFast
    if "roberta" in name:
        return RobertaTokenizerFast.from_pretrained(name, local_files_only=True)
    return BertTokenizerFast.from_pretrained(name, do_lower_case="uncased" in name, local_files_only=True)


# ------------------------------------------------------------------------------------------------------------
# masked-language-model collation (pre-training batches)
# ------------------------------------------------------------------------------------------------------------
class MLMCollator:
    """`mlm_collator` of base_datamodule.py:62-69, restated (host side; consumes Python's `random` and torch's CPU
    generator in the release's order, so a seeded run reproduces the release's draws -- tests/golden/mlm_collate.npz).

    whole_word=True (config.py:40): data_collator.py:381-496.  Per example, candidate words are runs of a token followed
    by its "##" continuations; "[CLS]" / "[SEP]" are skipped BY NAME, so with the RoBERTa vocabulary (no "##", specials
    called <s> </s> <pad>) every position -- specials and padding included -- is a one-token candidate.  The candidates are
    shuffled and taken until max(1, round(len * p)) positions are covered (len = the PADDED length the dataset
    produced, base_dataset.py:154-163), then specials and padding are struck from the selection: a short question
    ends up with fewer masked tokens than the count suggests.  Quirks kept as they are.
    whole_word=False: data_collator.py:290-378, Bernoulli(p) per non-special position.
    Both: selected positions keep their id as label (-100 elsewhere) and are rewritten 80 % -> <mask>, 10 % -> a uniform
    random id, 10 % unchanged."""

    def __init__(self, tokenizer, mlm_probability=0.15, whole_word=True, max_predictions=512):
        self.tok, self.p, self.whole_word, self.max_predictions = tokenizer, mlm_probability, whole_word, max_predictions
        if getattr(tokenizer, "mask_token_id", None) is None:
            raise ValueError("masked language modelling needs a tokenizer with a mask token")

    def _pad(self, rows):
        """_collate_batch (data_collator.py:253-287): right-pad with pad_token_id to the longest row."""
        n = max(len(r) for r in rows)
        out = torch.full((len(rows), n), int(self.tok.pad_token_id), dtype=torch.long)
        for i, r in enumerate(rows):
            out[i, : len(r)] = torch.as_tensor(r, dtype=torch.long)
        return out

    def _special(self, ids):
        return torch.tensor([self.tok.get_special_tokens_mask(r, already_has_special_tokens=True) for r in ids.tolist()],
                            dtype=torch.bool)

    def _word_selection(self, tokens):
        words = []
        for i, t in enumerate(tokens):
            if t in ("[CLS]", "[SEP]"):
                continue
            if words and t.startswith("##"):
                words[-1].append(i)
            else:
                words.append([i])
        random.shuffle(words)
        budget = min(self.max_predictions, max(1, int(round(len(tokens) * self.p))))
        chosen = []
        for w in words:
            if len(chosen) >= budget:
                break
            if len(chosen) + len(w) > budget or any(i in chosen for i in w):
                continue
            chosen.extend(w)
        sel = [0] * len(tokens)
        for i in chosen:
            sel[i] = 1
        return sel

    def __call__(self, encodings):
        rows = [list(e["input_ids"]) if isinstance(e, dict) else list(e) for e in encodings]
        ids = self._pad(rows)
        labels = ids.clone()
        if self.whole_word:
            # the selection rows are padded with pad_token_id too (any non-zero reads as "selected"); padding is struck below
            picked = self._pad([self._word_selection(self.tok.convert_ids_to_tokens(r)) for r in rows])
            picked.masked_fill_(self._special(labels), 0)
            picked.masked_fill_(labels.eq(int(self.tok.pad_token_id)), 0)
            picked = picked.bool()
        else:
            prob = torch.full(labels.shape, self.p)
            prob.masked_fill_(self._special(labels), 0.0)
            picked = torch.bernoulli(prob).bool()
        labels[~picked] = -100
        to_mask = torch.bernoulli(torch.full(labels.shape, 0.8)).bool() & picked
        ids[to_mask] = int(self.tok.mask_token_id)
        to_random = torch.bernoulli(torch.full(labels.shape, 0.5)).bool() & picked & ~to_mask
        ids[to_random] = torch.randint(len(self.tok), labels.shape, dtype=torch.long)[to_random]
        return {"input_ids": ids, "labels": labels}


# ------------------------------------------------------------------------------------------------------------
# dataset
# ------------------------------------------------------------------------------------------------------------
class ArrowVQADataset:
    """BaseDataset + VQAVQARADDataset for `{data_dir}/vqa_vqa_rad_{split}.arrow` (or any `names`)."""

    def __init__(self, data_dir, split, image_size, max_text_len, tokenizer, names=None):
        import pyarrow as pa
        self.names = names or [f"vqa_vqa_rad_{split}"]
        tables = []
        for name in self.names:
            path = os.path.join(data_dir, f"{name}.arrow")
            if os.path.isfile(path):
                tables.append(pa.ipc.RecordBatchFileReader(pa.memory_map(path, "r")).read_all())
        if not tables:
            raise FileNotFoundError(f"no arrow table for {self.names} under {data_dir!r}")
        self.table = pa.concat_tables(tables)
        self.image_size, self.max_text_len, self.tokenizer = image_size, max_text_len, tokenizer
        self.all_texts = self.table["questions"].to_pylist()
        self.index_mapper = [(i, j) for i, texts in enumerate(self.all_texts) for j in range(len(texts))]

    def __len__(self):
        return len(self.index_mapper)

    def image_u8(self, row):
        from PIL import Image
        return clip_resize_crop(Image.open(io.BytesIO(self.table["image"][row].as_py())), self.image_size)

    def __getitem__(self, index):
        row, qi = self.index_mapper[index]
        text = self.all_texts[row][qi]
        enc = self.tokenizer(text, padding="max_length", truncation=True, max_length=self.max_text_len)
        t = self.table
        return {
            "image_u8": self.image_u8(row),
            "text": text,
            "input_ids": list(enc["input_ids"]),
            "attention_mask": list(enc["attention_mask"]),
            "vqa_answer": t["answers"][row][qi].as_py(),
            "vqa_labels": t["answer_labels"][row][qi].as_py(),
            "vqa_scores": t["answer_scores"][row][qi].as_py(),
            "answer_types": t["answer_type"][row][qi].as_py(),
            "qid": t["question_id"][row][qi].as_py(),
        }


class ArrowCaptionDataset:
    """BaseDataset + ROCODataset / MedicatDataset (pretraining_roco_dataset.py:1-21, pretraining_medicat_dataset.py:1-21):
    `{data_dir}/{name}_{split}.arrow` written by prepro/make_arrow.py:40-63 (columns image, caption [list of str], image_id,
    split), one sample per (image row, caption index) (base_dataset.py:72-81), and `get_suite` (:141-163): the image, its
    caption, and `draw_false_image` negatives drawn as `random.randint(0, len - 1)` over THIS table's samples
    (`get_false_image`, :107-111) -- Python's `random` stream, as the reference consumes it, so a seeded run reproduces the
    reference's draws.  A sample that fails to decode is replaced by a random one (:158-160)."""

    def __init__(self, data_dir, name, split, image_size, max_text_len, tokenizer, draw_false_image=0):
        import pyarrow as pa
        assert split in ("train", "val", "test")
        self.names = [f"{name}_{split}"]
        path = os.path.join(data_dir, f"{self.names[0]}.arrow")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"no arrow table {path!r}")
        self.table = pa.ipc.RecordBatchFileReader(pa.memory_map(path, "r")).read_all()
        self.image_size, self.max_text_len, self.tokenizer = image_size, max_text_len, tokenizer
        self.draw_false_image = draw_false_image
        self.all_texts = self.table["caption"].to_pylist()
        assert isinstance(self.all_texts[0][0], str)
        self.index_mapper = [(i, j) for i, texts in enumerate(self.all_texts) for j in range(len(texts))]

    def __len__(self):
        return len(self.index_mapper)

    def image_u8(self, row):
        from PIL import Image
        return clip_resize_crop(Image.open(io.BytesIO(self.table["image"][row].as_py())), self.image_size)

    def get_suite(self, index):
        while True:
            try:
                row, ci = self.index_mapper[index]
                text = self.all_texts[row][ci]
                enc = self.tokenizer(text, padding="max_length", truncation=True, max_length=self.max_text_len)
                ret = {"image_u8": self.image_u8(row), "text": text, "input_ids": list(enc["input_ids"]),
                       "attention_mask": list(enc["attention_mask"]), "img_index": row, "cap_index": ci,
                       "raw_index": index, "replica": ci > 0}
                for rep in range(self.draw_false_image):
                    frow, _ = self.index_mapper[random.randint(0, len(self.index_mapper) - 1)]
                    ret[f"false_image_u8_{rep}"] = self.image_u8(frow)
                return ret
            except Exception as e:  # noqa: BLE001  (base_dataset.py:158-160)
                print(f"Error while read file idx {index} in {self.names[0]} -> {e}")
                index = random.randint(0, len(self.index_mapper) - 1)

    __getitem__ = get_suite


class ConcatDataset:
    """torch.utils.data.ConcatDataset as MTDataModule uses it (multitask_datamodule.py:36-40): datasets back to back."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.ends = np.cumsum([len(p) for p in self.parts]).tolist()

    def __len__(self):
        return self.ends[-1] if self.ends else 0

    def __getitem__(self, index):
        for p, end in zip(self.parts, self.ends):
            if index < end:
                return p[index - (end - len(p))]
        raise IndexError(index)


def collate_host(samples, pin=True, mlm_collator=None):
    """base_dataset.py:165-228: images stacked as uint8 NHWC, ids / masks as int64 tensors; with `mlm_collator` also
    `text_ids_mlm` / `text_labels_mlm` (:202-209; the reference always computes them, the fine-tuning step never reads
    them)."""
    B = len(samples)
    S = max(len(s["input_ids"]) for s in samples)
    img = torch.from_numpy(np.stack([s["image_u8"] for s in samples]))
    ids = torch.zeros((B, S), dtype=torch.long)
    mask = torch.zeros((B, S), dtype=torch.long)
    for i, s in enumerate(samples):
        ids[i, : len(s["input_ids"])] = torch.tensor(s["input_ids"])
        mask[i, : len(s["attention_mask"])] = torch.tensor(s["attention_mask"])
    extra = {}
    if mlm_collator is not None:
        m = mlm_collator([{"input_ids": s["input_ids"]} for s in samples])
        extra = {"text_ids_mlm": m["input_ids"], "text_labels_mlm": m["labels"]}
    for k in sorted(samples[0]):   # the negatives of the image-text matching objective (base_dataset.py:107-111, :173-195)
        if k.startswith("false_image_u8_"):
            extra[k] = torch.from_numpy(np.stack([s[k] for s in samples]))
    if pin and torch.cuda.is_available():
        img, ids, mask = img.pin_memory(), ids.pin_memory(), mask.pin_memory()
        extra = {k: v.pin_memory() for k, v in extra.items()}
    out = {"image_u8": img, "text_ids": ids, "text_masks": mask, **extra, "text": [s["text"] for s in samples]}
    for k in ("vqa_answer", "vqa_labels", "vqa_scores", "answer_types", "qid", "img_index", "cap_index", "raw_index",
              "replica"):
        if k in samples[0]:
            out[k] = [s[k] for s in samples]
    return out


def to_device_batch(hb, device, copy_stream=None):
    """Upload a host batch (pinned) and finish the transform on the GPU -> the 8b batch dict."""
    cur = torch.cuda.current_stream()
    cs = copy_stream or cur
    with torch.cuda.stream(cs):
        u8 = hb["image_u8"].to(device, non_blocking=True)
        ids = hb["text_ids"].to(device, non_blocking=True)
        mask = hb["text_masks"].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        mlm = {k: hb[k].to(device, non_blocking=True) for k in ("text_ids_mlm", "text_labels_mlm") if k in hb}
        fal = {"_" + k: hb[k].to(device, non_blocking=True) for k in hb if k.startswith("false_image_u8_")}
        ev.record(cs)
    out = {k: v for k, v in hb.items() if k not in ("image_u8", "text_ids", "text_masks", "text_ids_mlm", "text_labels_mlm")
           and not k.startswith("false_image_u8_")}
    out.update(_u8=u8, text_ids=ids, text_masks=mask, text_labels=None, _ready=ev, **mlm, **fal)
    return out


def finish_batch(db):
    """Called on the compute stream right before the step: wait for the upload, normalise, build labels."""
    torch.cuda.current_stream().wait_event(db.pop("_ready"))
    u8 = db.pop("_u8")
    db["image"] = [normalize_on_device(u8)]
    for k in [k for k in db if k.startswith("_false_image_u8_")]:
        f8 = db.pop(k)
        f8.record_stream(torch.cuda.current_stream())
        db["false_image_" + k[len("_false_image_u8_"):]] = [normalize_on_device(f8)]
    db["text_labels"] = torch.full_like(db["text_ids"], -100)
    cur = torch.cuda.current_stream()
    u8.record_stream(cur)
    for t in db.values():   # every tensor uploaded on the copy stream (text_ids_mlm / text_labels_mlm included) is now used here
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(cur)
    return db


# ------------------------------------------------------------------------------------------------------------
# datamodule
# ------------------------------------------------------------------------------------------------------------
class ArrowDataModule:
    """MTDataModule / BaseDataModule for the hot path: DistributedSampler-style sharding (seeded shuffle per epoch,
    multitask_datamodule.py:44-48), `per_gpu_batchsize` batches, background decode + upload."""

    def __init__(self, cfg, rank=0, world=1, device="cuda", tokenizer=None, prefetch=3, head="cls"):
        self.cfg, self.rank, self.world, self.device, self.head = cfg, rank, world, device, head
        self.B = cfg["per_gpu_batchsize"]
        self.tokenizer = tokenizer or load_tokenizer(cfg)
        root = cfg["data_root"]
        names = list(cfg.get("datasets") or ["vqa_vqa_rad"])
        if any(n in ("roco", "medicat") for n in names):   # the pre-training caption tables (config.py:22,31: draw_false_image = 1)
            mk = lambda split: ConcatDataset([ArrowCaptionDataset(root, n, split, cfg["image_size"], cfg["max_text_len"],
                                                                  self.tokenizer, cfg.get("draw_false_image", 0))
                                              for n in names])
        else:
            mk = lambda split: ArrowVQADataset(root, split, cfg["image_size"], cfg["max_text_len"], self.tokenizer)
        self.train_set = mk("train")
        self.val_set = self._try(mk, "val") or self.train_set
        self.test_set = self._try(mk, "test") or self.val_set
        self.train_samples, self.val_samples = len(self.train_set), len(self.val_set)
        self.workers = max(int(cfg.get("num_workers", 8)), 1)
        self.prefetch = prefetch
        self.copy_stream = torch.cuda.Stream(device=device) if torch.cuda.is_available() else None
        # base_datamodule.py:62-69; the reference builds it for every task, only the MLM objective reads its output
        self.mlm_collator = None
        if cfg.get("loss_names", {}).get("mlm", 0) > 0:
            self.mlm_collator = MLMCollator(self.tokenizer, cfg.get("mlm_prob", 0.15), cfg.get("whole_word_masking", True))

    @staticmethod
    def _try(mk, split):
        try:
            return mk(split)
        except FileNotFoundError:
            return None

    def _indices(self, ds, epoch, shuffle):
        idx = list(range(len(ds)))
        if shuffle:
            random.Random(self.cfg["seed"] * 1000 + epoch).shuffle(idx)
        total = (len(idx) + self.world - 1) // self.world * self.world  # DistributedSampler pads by wrapping around
        idx += idx[: total - len(idx)]
        return idx[self.rank::self.world]

    def _stream(self, ds, idx, drop_last):
        """Generator of device batches; host decode runs `prefetch` batches ahead in a thread pool."""
        chunks = [idx[i:i + self.B] for i in range(0, len(idx), self.B)]
        if drop_last:
            chunks = [c for c in chunks if len(c) == self.B]
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def producer():
            with ThreadPoolExecutor(self.workers) as pool:
                for c in chunks:
                    if stop.is_set():
                        break
                    q.put(collate_host(list(pool.map(ds.__getitem__, c)), mlm_collator=self.mlm_collator))
            q.put(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        pending = None
        try:
            while True:
                hb = q.get()
                nxt = None if hb is None else to_device_batch(hb, self.device, self.copy_stream)
                if pending is not None:
                    yield finish_batch(pending)   # its upload was issued one batch ago
                if nxt is None:
                    break
                pending = nxt
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)

    def train_batches(self, epoch):
        return self._stream(self.train_set, self._indices(self.train_set, epoch, True), drop_last=False)

    def val_batches(self):
        return self._stream(self.val_set, self._indices(self.val_set, 0, False), drop_last=False)
