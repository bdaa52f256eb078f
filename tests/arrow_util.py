"""Test helper: a tiny VQA-RAD-shaped arrow dataset (the on-disk schema of prepro/make_arrow.py:183-204) and a
vocabulary-free tokenizer with RoBERTa's special ids (<s> 0, <pad> 1, </s> 2)."""
import io
import os

import numpy as np


class HashTokenizer:
    vocab_size = 1000

    def __call__(self, text, padding="max_length", truncation=True, max_length=32, **kw):
        ids = [0] + [3 + (sum(ord(ch) * (i + 1) for i, ch in enumerate(w)) % 990) for w in text.lower().split()][: max_length - 2] + [2]
        mask = [1] * len(ids)
        if padding == "max_length":
            ids += [1] * (max_length - len(ids))
            mask += [0] * (max_length - len(mask))
        return {"input_ids": ids, "attention_mask": mask}


def make_image(kind, w, h, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    if kind == "split":  # left half red, right half blue
        a = np.zeros((h, w, 3), dtype=np.uint8)
        a[:, : w // 2, 0] = 255
        a[:, w // 2:, 2] = 255
        img = Image.fromarray(a, "RGB")
    elif kind == "gray":
        img = Image.fromarray(rng.randint(0, 256, (h, w), dtype=np.uint8), "L")
    elif kind == "const":
        img = Image.fromarray(np.full((h, w, 3), (10, 128, 250), dtype=np.uint8), "RGB")
    else:
        img = Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
    buf = io.BytesIO()
    img.save(buf, format="PNG")
    return buf.getvalue()


def write_split(root, split, n_images, seed=0):
    import pyarrow as pa
    kinds = ["split", "gray", "const", "noise"]
    rows = {k: [] for k in ("image", "questions", "answers", "answer_labels", "answer_scores", "image_id", "question_id",
                            "answer_type", "split")}
    qid = 0
    for i in range(n_images):
        kind = kinds[i % 4]
        w, h = [(500, 400), (300, 300), (200, 640), (96, 80)][i % 4]
        nq = 1 + i % 3
        rows["image"].append(make_image(kind, w, h, seed + i))
        rows["questions"].append([f"is there a {kind} finding number {j} in image {i} ?" for j in range(nq)])
        rows["answers"].append([[f"ans{(i + j) % 7}"] for j in range(nq)])
        rows["answer_labels"].append([[(i * 3 + j) % 498] for j in range(nq)])
        rows["answer_scores"].append([[1.0] for j in range(nq)])
        rows["image_id"].append(f"img{i}")
        rows["question_id"].append([qid + j for j in range(nq)])
        rows["answer_type"].append([(i + j) % 2 for j in range(nq)])
        rows["split"].append(split)
        qid += nq
    table = pa.table(rows)
    os.makedirs(root, exist_ok=True)
    with pa.OSFile(os.path.join(root, f"vqa_vqa_rad_{split}.arrow"), "wb") as sink:
        with pa.RecordBatchFileWriter(sink, table.schema) as writer:
            writer.write_table(table)
    return qid


class CollatorTokenizer:
    """Just enough tokenizer for the collators: RoBERTa special ids / names, or BERT-style names with "##" pieces."""

    def __init__(self, style, n=1000):
        self.style, self.n = style, n
        if style == "roberta":
            self.names = {0: "<s>", 1: "<pad>", 2: "</s>", 3: "<unk>", n - 1: "<mask>"}
            self.pad_token_id, self.mask_token_id = 1, n - 1
            self.special = {0, 1, 2}
        else:
            self.names = {0: "[PAD]", 100: "[UNK]", 101: "[CLS]", 102: "[SEP]", 103: "[MASK]"}
            self.pad_token_id, self.mask_token_id = 0, 103
            self.special = {0, 101, 102}
        self.mask_token, self._pad_token = self.names[self.mask_token_id], self.names[self.pad_token_id]
        self.padding_side = "right"

    def __len__(self):
        return self.n

    def _convert_id_to_token(self, i):
        if i in self.names:
            return self.names[i]
        return ("##p%d" if self.style == "bert" and i % 3 == 0 else "w%d") % i

    def convert_ids_to_tokens(self, ids):
        return [self._convert_id_to_token(int(i)) for i in ids]

    def convert_tokens_to_ids(self, t):
        return {v: k for k, v in self.names.items()}[t]

    def get_special_tokens_mask(self, ids, already_has_special_tokens=True):
        return [1 if int(i) in self.special else 0 for i in ids]


def collator_cases():
    rng = np.random.RandomState(7)
    cases = []
    for style in ("roberta", "bert"):
        tok = CollatorTokenizer(style)
        first, last = (0, 2) if style == "roberta" else (101, 102)
        for fixed in (True, False):     # padded to max_length by the dataset (the reference's case) / ragged
            rows = []
            for _ in range(6):
                n = int(rng.randint(4, 31))
                r = [first] + [int(v) for v in rng.randint(110, 990, size=n)] + [last]
                if fixed:
                    r += [tok.pad_token_id] * (32 - len(r))
                rows.append(r)
            cases.append((style, fixed, rows))
    return cases


def write_caption_split(root, name, split, n_images, seed=0):
    """A tiny pre-training caption table in the on-disk schema of prepro/make_arrow.py:40-63 (image, caption, image_id, split)."""
    import pyarrow as pa
    kinds = ["split", "gray", "const", "noise"]
    rows = {k: [] for k in ("image", "caption", "image_id", "split")}
    for i in range(n_images):
        w, h = [(500, 400), (300, 300), (200, 640), (96, 80)][i % 4]
        rows["image"].append(make_image(kinds[i % 4], w, h, seed + i))
        rows["caption"].append([f"{name} caption {j} of image {i} shows a {kinds[i % 4]} pattern" for j in range(1 + i % 2)])
        rows["image_id"].append(f"{name}{i}")
        rows["split"].append(split)
    table = pa.table(rows)
    os.makedirs(root, exist_ok=True)
    with pa.OSFile(os.path.join(root, f"{name}_{split}.arrow"), "wb") as sink:
        with pa.RecordBatchFileWriter(sink, table.schema) as writer:
            writer.write_table(table)
    return sum(len(c) for c in rows["caption"])
