"""Batch provider with the contract of vqa/datasets/input_ops_vqa_tf_record_memft.py:6-82.

create(batch_size, data_dir, split, is_train, shuffle) returns an iterator of batch dicts
  id i64[B], image_id str[B], image_idx i64[B], q_intseq i32[B, Tmax_of_batch] (zero padded),
  q_intseq_len i32[B], answer_target f32[B, num_answers] (dense from sparse ids/scores)
The last batch of a pass may be short.  Behaviour kept from the reference pipeline:
  * train (is_train and shuffle): one shuffle, batches cached after batching (so the same
    batches repeat every epoch, input_ops...:61-78), repeated 1000 times;
  * otherwise a single ordered pass, then StopIteration (the reference's OutOfRangeError,
    vqa/evaler.py:119-123).
Storage: the reference's own layout -- `<data_dir>/<split>/<split>-*` TFRecord shards (tfrecord_io, no
TensorFlow) and `<data_dir>/data_info.hdf5` with /data_info/num_answers (hdf5_io, no h5py;
data/tools/vqa_v2/generator_tf_record_memft_genome.py:81-107) -- or the compact `<data_dir>/<split>.npz`
(ragged arrays + offsets) this package writes for synthetic splits; `data_info.json` {"num_answers": A} is
accepted where data_info.hdf5 is absent.
"""
from __future__ import annotations

import json
import os

import numpy as np


def read_num_answers(data_dir):
    """num_answers of a tf_record directory: /data_info/num_answers of data_info.hdf5
    (vqa/datasets/input_ops_vqa_tf_record_memft.py:13-15), else data_info.json."""
    h5 = os.path.join(data_dir, "data_info.hdf5")
    if os.path.exists(h5):
        from . import hdf5_io
        with hdf5_io.File(h5) as f:
            return int(f["data_info"]["num_answers"][()])
    js = os.path.join(data_dir, "data_info.json")
    if os.path.exists(js):
        with open(js) as f:
            return int(json.load(f)["num_answers"])
    raise FileNotFoundError("neither data_info.hdf5 nor data_info.json under %s" % data_dir)


def write_data_info(data_dir, num_answers, **extra):
    """data_info.hdf5 with the group the reference writes (num_answers as an int32 scalar; optional tables such as
    intseq_ans / intseq_ans_len / max_ans_len)."""
    from . import hdf5_io
    info = {"num_answers": np.array(int(num_answers), np.int32)}
    info.update(extra)
    hdf5_io.write(os.path.join(data_dir, "data_info.hdf5"), {"data_info": info})


class SplitData:
    """In-memory examples of one split (the fields of the reference's tf.Example schema,
    data/tools/vqa_v2/generator_tf_record_memft_genome.py:184-194)."""

    def __init__(self, qid, image_id, image_idx, q_flat, q_off, ans_ids, ans_scores, ans_off, num_answers):
        self.qid, self.image_id, self.image_idx = qid, image_id, image_idx
        self.q_flat, self.q_off = q_flat, q_off
        self.ans_ids, self.ans_scores, self.ans_off = ans_ids, ans_scores, ans_off
        self.num_answers = int(num_answers)

    def __len__(self):
        return len(self.qid)

    @staticmethod
    def load(data_dir, split):
        num_answers = read_num_answers(data_dir)
        z = np.load(os.path.join(data_dir, split + ".npz"), allow_pickle=False)
        return SplitData(z["qid"], z["image_id"], z["image_idx"], z["q_flat"], z["q_off"], z["ans_ids"],
                         z["ans_scores"], z["ans_off"], num_answers)

    def save(self, data_dir, split):
        os.makedirs(data_dir, exist_ok=True)
        write_data_info(data_dir, self.num_answers)
        np.savez(os.path.join(data_dir, split + ".npz"), qid=self.qid, image_id=self.image_id,
                 image_idx=self.image_idx, q_flat=self.q_flat, q_off=self.q_off, ans_ids=self.ans_ids,
                 ans_scores=self.ans_scores, ans_off=self.ans_off)

    def batch(self, rows):
        B = len(rows)
        lens = (self.q_off[rows + 1] - self.q_off[rows]).astype(np.int32)
        T = max(int(lens.max()) if B else 0, 1)
        q = np.zeros((B, T), np.int32)
        tgt = np.zeros((B, self.num_answers), np.float32)
        for i, r in enumerate(rows):
            q[i, :lens[i]] = self.q_flat[self.q_off[r]:self.q_off[r + 1]]
            a0, a1 = self.ans_off[r], self.ans_off[r + 1]
            tgt[i, self.ans_ids[a0:a1]] = self.ans_scores[a0:a1]      # tf.sparse_to_dense
        return {"id": self.qid[rows].astype(np.int64), "image_id": self.image_id[rows],
                "image_idx": self.image_idx[rows].astype(np.int64), "q_intseq": q, "q_intseq_len": lens,
                "answer_target": tgt}


def synthetic_split(num_examples, num_images, vocab_size, num_answers, max_len=14, min_len=3, seed=0):
    """Synthetic examples shaped like the VQA-v2 records: 1-3 answers per question
    with scores from {0.3, 0.6, 0.9, 1.0}, 3..14 tokens."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, max_len + 1, size=num_examples)
    q_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q_flat = rng.integers(1, max(vocab_size - 3, 2), size=int(q_off[-1])).astype(np.int32)
    nans = rng.integers(1, 4, size=num_examples)
    ans_off = np.concatenate([[0], np.cumsum(nans)]).astype(np.int64)
    ans_ids = np.concatenate([rng.choice(num_answers, size=k, replace=False) for k in nans]).astype(np.int32)
    ans_scores = rng.choice(np.array([0.3, 0.6, 0.9, 1.0], np.float32), size=int(ans_off[-1]))
    image_idx = rng.integers(0, num_images, size=num_examples).astype(np.int64)
    image_id = np.array(["synthetic-%08d.jpg" % i for i in image_idx])
    return SplitData(np.arange(num_examples, dtype=np.int64), image_id, image_idx, q_flat, q_off, ans_ids,
                     ans_scores, ans_off, num_answers)


def sort_by_length(batch):
    """The batch with its rows ordered by question length, longest first, plus `live_rows` (int32 [T]:
    live_rows[t] = number of questions longer than t) and `sort_order` (the permutation applied).  Every
    per-sample entry is permuted consistently, so losses, reports and results keyed by `id` are unchanged; the
    engine then runs each GRU step on the still-running prefix only (vqa_gru_seq_*_live)."""
    lens = np.asarray(batch["q_intseq_len"])
    B, T = np.asarray(batch["q_intseq"]).shape
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    out = {}
    for k, v in batch.items():
        if isinstance(v, np.ndarray) and v.shape[:1] == (B,):
            out[k] = np.ascontiguousarray(v[order])
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = [v[i] for i in order]
        else:
            out[k] = v
    sl = np.clip(lens[order], 0, T)
    out["live_rows"] = (sl[None, :] > np.arange(T)[:, None]).sum(1).astype(np.int32)
    out["sort_order"] = order
    return out


def create(batch_size, data_dir, split, is_train=True, scope="vqa", shuffle=True, seed=0, data=None,
           repeat=1000):
    """Iterator of batch dicts; `data` (a SplitData) overrides the files under data_dir."""
    if data is not None:
        d = data
    elif os.path.exists(os.path.join(data_dir, split + ".npz")):
        d = SplitData.load(data_dir, split)
    else:                                    # the reference's own tfrecord shards (no TensorFlow needed)
        from . import tfrecord_io
        d = tfrecord_io.load_vqa_split(data_dir, split, read_num_answers(data_dir))
    n = len(d)
    order = np.arange(n)
    if is_train and shuffle:
        np.random.default_rng(seed).shuffle(order)
    chunks = [order[i:i + batch_size] for i in range(0, n, batch_size)]

    def gen():
        cache = {}
        for _ in range(repeat if is_train else 1):
            for ci, rows in enumerate(chunks):
                if is_train:
                    if ci not in cache:
                        cache[ci] = d.batch(rows)          # dataset.cache() after padded_batch
                    yield cache[ci]
                else:
                    yield d.batch(rows)
    return gen()
