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
