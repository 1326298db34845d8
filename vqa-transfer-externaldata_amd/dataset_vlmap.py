"""Visual-Genome pre-training dataset with the sampling contract of
vlmap_memft/datasets/dataset_vlmap.py:19-236, 308-353 (model_vlmap_bf_or_wordset_withatt_sp inputs).

Per image: up to 5 object and 5 attribute blank-fill entries are drawn at random (shuffle, truncate,
pad by repeating the last, `num` = number of valid ones), captions padded to the per-image max length
(batches to the per-batch max), one word set per entry taken round-robin from the answer's shuffled
word-set list (re-shuffled when exhausted).  The enwiki-context fields are not produced: the cfg-5
model does not read them.  Files: `<data_dir>/<split>_processed.pkl`, `<split>_image_info.pkl`,
`answer_dict.pkl`, `wordset_dict5.pkl` as in the reference, features from `<split>_vfeat.hdf5` (or `.npz`).
`synthetic_dataset` builds the same structures in memory.
"""
from __future__ import annotations

import logging
import os
import pickle
from collections import defaultdict

import numpy as np

log = logging.getLogger(__name__)

NUM_CONFIG = {"obj_blank_fill": 5, "attr_blank_fill": 5}      # dataset_vlmap.py (n_obj_bf, n_attr_bf)


def _load_pickle(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f, encoding="latin1")


class DataConfig:
    pass


class Dataset(object):
    def __init__(self, config=None, split="train", name="vlmap_memft", data=None, seed=0):
        """data: optional dict(ids, image_id2idx, processed, answer_dict, ws_dict, image_features,
        spatial_features, normal_boxes, num_boxes) replacing the files under config.data_dir."""
        self.name, self.split = name, split
        self.rng = np.random.RandomState(seed)
        if data is None:
            d = config.data_dir
            info = _load_pickle(os.path.join(d, "{}_image_info.pkl".format(split)))
            data = {"ids": info["image_ids"], "image_id2idx": info["image_id2idx"],
                    "processed": _load_pickle(os.path.join(d, "{}_processed.pkl".format(split))),
                    "answer_dict": _load_pickle(os.path.join(d, "answer_dict.pkl")),
                    "ws_dict": _load_pickle(os.path.join(d, "wordset_dict5.pkl"))}
            # '<split>_vfeat.hdf5' of the reference (vlmap_memft/datasets/dataset_vlmap.py:67-72), read without
            # h5py; '<split>_vfeat.npz' with the same keys as an alternative
            h5 = os.path.join(d, "{}_vfeat.hdf5".format(split))
            if os.path.exists(h5):
                from . import hdf5_io
                f = hdf5_io.File(h5)
                z = {k: f[k].read() for k in ("image_features", "spatial_features", "normal_boxes", "num_boxes")}
            else:
                z = np.load(os.path.join(d, "{}_vfeat.npz".format(split)))
            data.update(image_features=z["image_features"], spatial_features=z["spatial_features"],
                        normal_boxes=z["normal_boxes"], num_boxes=np.asarray(z["num_boxes"]))
        self._ids = list(data["ids"])
        self.image_id2idx = data["image_id2idx"]
        self.processed = data["processed"]
        self.answer_dict = data["answer_dict"]
        self.num_answers = len(self.answer_dict["vocab"])
        self.ws_dict = data["ws_dict"]
        # the reference reads 'ans2shuffled_wordset', which no script in its tree writes
        # (find_word_group.py writes 'ans2wordset'); accept either
        self._ans2ws = self.ws_dict.get("ans2shuffled_wordset", self.ws_dict.get("ans2wordset"))
        self.image_features, self.spatial_features = data["image_features"], data["spatial_features"]
        self.normal_boxes, self.num_boxes = data["normal_boxes"], data["num_boxes"]
        self.max_box_num, self.vfeat_dim = self.image_features.shape[1], self.image_features.shape[2]
        self.wordset_choice_idx = defaultdict(lambda: defaultdict(lambda: defaultdict(int)))

    def get_config(self):
        c = DataConfig()
        c.n_attr_bf, c.n_obj_bf = NUM_CONFIG["attr_blank_fill"], NUM_CONFIG["obj_blank_fill"]
        c.vfeat_dim, c.max_box_num = self.vfeat_dim, self.max_box_num
        return c

    def sample_wordset(self, e, category, task):
        label = e[task]
        wordsets = self._ans2ws[label]
        i = self.wordset_choice_idx[category][task][label]
        ws = wordsets[i]
        self.wordset_choice_idx[category][task][label] += 1
        if self.wordset_choice_idx[category][task][label] >= len(wordsets):
            self.rng.shuffle(wordsets)
            self.wordset_choice_idx[category][task][label] = 0
        return ws

    def get_data(self, image_id, with_features=True):
        """with_features=False (not in the reference): the row carries `image_idx` instead of its slices of the
        feature tables -- for trainers that keep the tables in HBM and gather there (create_ops(resident=True))"""
        idx = self.image_id2idx[image_id]
        if with_features:
            ret = {"image_id": np.array(image_id, np.int32), "image_ft": self.image_features[idx],
                   "spatial_ft": self.spatial_features[idx], "normal_boxes": self.normal_boxes[idx],
                   "num_boxes": self.num_boxes[idx]}
        else:
            ret = {"image_id": np.array(image_id, np.int32), "image_idx": np.array(idx, np.int64)}
        entry = self.processed[image_id]
        for cat, key in (("obj", "obj_blank_fill"), ("attr", "attr_blank_fill")):
            n = NUM_CONFIG[key]
            idx_list = list(range(len(entry[key])))
            self.rng.shuffle(idx_list)
            idx_list = idx_list[:n]
            num_valid = len(idx_list)
            while len(idx_list) < n:
                idx_list.append(idx_list[-1])
            maxlen = max(len(entry[key][i]["blank"]) for i in idx_list)
            blanks = np.zeros([n, maxlen], np.int32)
            weights, boxes, fills, blens, wsets = [], [], [], [], []
            for j, i in enumerate(idx_list):
                e = entry[key][i]
                w = np.zeros([self.max_box_num], np.float32)
                w[e["p_idx"]] = e["p_weight"]
                weights.append(w)
                boxes.append(e["normal_box"])
                blens.append(len(e["blank"]))
                blanks[j, :blens[j]] = e["blank"]
                fills.append(e["fill"])
                wsets.append(self.sample_wordset(e, cat, "fill"))
            ret.update({key + "/num": np.array(num_valid, np.int32), key + "/weights": np.array(weights, np.float32),
                        key + "/normal_boxes": np.array(boxes, np.float32), key + "/fills": np.array(fills, np.int32),
                        key + "/blanks": blanks, key + "/blanks_len": np.array(blens, np.int32),
                        key + "/wordsets": np.array(wsets, np.int32)})
        return ret

    @property
    def ids(self):
        return self._ids

    def __len__(self):
        return len(self._ids)


def _collate(rows):
    """padded_batch of the reference: captions zero-padded to the longest of the batch, everything else stacked"""
    out = {}
    for k in rows[0]:
        if k.endswith("/blanks"):
            L = max(r[k].shape[1] for r in rows)
            o = np.zeros((len(rows), rows[0][k].shape[0], L), rows[0][k].dtype)
            for i, r in enumerate(rows):
                o[i, :, :r[k].shape[1]] = r[k]
            out[k] = o
        else:
            out[k] = np.stack([np.asarray(r[k]) for r in rows])
    return out


def _batches(batch_size, dataset, is_train, shuffle, seed, repeat, resident, part=0, parts=1):
    """batches number part, part + parts, ... of the stream (every producer draws the same epoch orders)"""
    ids = list(dataset.ids)
    rng = np.random.RandomState(seed)
    g = 0
    for _ in range(repeat if is_train else 1):
        order = list(ids)
        if is_train and shuffle:
            rng.shuffle(order)
        for lo in range(0, len(order), batch_size):
            if g % parts == part:
                yield _collate([dataset.get_data(i, with_features=not resident) for i in order[lo:lo + batch_size]])
            g += 1


def _worker_main(q, batch_size, dataset, is_train, shuffle, seed, repeat, resident, part, parts):
    dataset.rng = np.random.RandomState((int(seed) + 1000003 * (part + 1)) % (2 ** 31 - 1))   # own sampling stream
    try:
        for b in _batches(batch_size, dataset, is_train, shuffle, seed, repeat, resident, part, parts):
            q.put(b)
    except BaseException:                 # hand the failure to the consumer instead of ending the stream silently
        import traceback
        q.put(("__producer_error__", traceback.format_exc()))
    finally:
        q.put(None)


def create_ops(batch_size, dataset, is_train=True, scope="vlmap_memft", shuffle=True, seed=0, repeat=1000,
               resident=False, workers=0, prefetch=0):
    """Iterator of padded batch dicts (dataset_vlmap.create_ops, :308-353): captions are padded to the
    longest of the batch; train repeats, eval is a single pass.

    Not in the reference (whose py_func map is sequential, :323-340, and was never the bottleneck of a 2018 GPU):
    * resident: rows carry `image_idx` instead of the 295 KB feature slices (the trainer gathers from tables in HBM);
    * workers K > 0: K forked producer processes, started HERE (call before anything initialises the GPU); producer w
      assembles batches w, w + K, ... of the same epoch orders with its own sampling stream, the parent hands them out
      in batch order, so a run is reproducible for a given K (the sampled entries differ from the K = 0 stream);
    * prefetch P > 0 (workers == 0): one producer thread runs P batches ahead of the consumer."""
    args = (batch_size, dataset, is_train, shuffle, seed, repeat, resident)
    if workers and workers > 0:
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            # a fork now would hand the children this process's GPU file descriptors: one producer thread instead
            log.warning("create_ops(workers=%d) after the GPU was initialised: using one producer thread", workers)
            workers, prefetch = 0, max(2, prefetch or 0)
    if workers and workers > 0:
        import multiprocessing as mp
        ctx = mp.get_context("fork")              # the dataset (pickles, feature tables) is shared copy-on-write
        qs = [ctx.Queue(maxsize=max(2, prefetch or 2)) for _ in range(workers)]
        procs = [ctx.Process(target=_worker_main, args=(qs[w],) + args + (w, workers), daemon=True) for w in range(workers)]
        for p in procs:
            p.start()

        def gen_mp():
            try:
                g = 0
                while True:
                    w = g % workers
                    while True:      # a producer killed from outside (OOM killer) never posts its sentinel: notice it
                        try:
                            b = qs[w].get(timeout=5.0)
                            break
                        except Exception:       # queue.Empty
                            if not procs[w].is_alive() and qs[w].empty():
                                raise RuntimeError("batch producer %d (pid %s) died with exit code %s without posting a "
                                                   "batch or an error" % (w, procs[w].pid, procs[w].exitcode))
                    if b is None:
                        return
                    if isinstance(b, tuple) and b and b[0] == "__producer_error__":
                        raise RuntimeError("batch producer %d failed:\n%s" % (g % workers, b[1]))
                    yield b
                    g += 1
            finally:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
        return gen_mp()
    if prefetch and prefetch > 0:
        import queue
        import threading
        q = queue.Queue(maxsize=prefetch)
        stop = threading.Event()

        def offer(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return
                except queue.Full:
                    continue

        def produce():
            try:
                for b in _batches(*args):
                    offer(b)
                    if stop.is_set():
                        return
            except BaseException:
                import traceback
                offer(("__producer_error__", traceback.format_exc()))
            finally:
                while not stop.is_set():
                    try:
                        q.put(None, timeout=0.2)
                        break
                    except queue.Full:
                        continue
        threading.Thread(target=produce, daemon=True).start()

        def gen_thread():
            try:
                while True:
                    b = q.get()
                    if b is None:
                        return
                    if isinstance(b, tuple) and b and b[0] == "__producer_error__":
                        raise RuntimeError("batch producer thread failed:\n%s" % b[1])
                    yield b
            finally:
                stop.set()
        return gen_thread()
    return _batches(*args)


def synthetic_dataset(num_images, Vq, n_ws, A, R=36, D=2048, max_len=10, seed=0):
    """In-memory dataset with the reference's structures (5-8 entries per image and category)."""
    rng = np.random.default_rng(seed)
    ids = list(range(1000, 1000 + num_images))
    ys, xs = np.sort(rng.random((num_images, R, 2)), -1), np.sort(rng.random((num_images, R, 2)), -1)
    nb = np.stack([ys[..., 0], xs[..., 0], ys[..., 1], xs[..., 1]], -1).astype(np.float32)
    spat = np.concatenate([nb, nb[..., 2:3] - nb[..., 0:1], nb[..., 3:4] - nb[..., 1:2]], -1).astype(np.float32)
    processed = {}
    for i in ids:
        ent = {}
        for key in NUM_CONFIG:
            lst = []
            for _ in range(int(rng.integers(3, 9))):
                ln = int(rng.integers(2, max_len + 1))
                y, x = np.sort(rng.random(2)), np.sort(rng.random(2))
                lst.append({"blank": rng.integers(1, Vq, size=ln).tolist(), "fill": int(rng.integers(0, A)),
                            "normal_box": np.array([y[0], x[0], y[1], x[1]], np.float32),
                            "p_idx": rng.choice(R, size=3, replace=False).tolist(),
                            "p_weight": (np.ones(3) / 3).tolist()})
            ent[key] = lst
        processed[i] = ent
    ws = {"vocab": ["ws%d" % i for i in range(n_ws)],
          "ans2shuffled_wordset": {a: rng.integers(0, n_ws, size=int(rng.integers(1, 4))).tolist() for a in range(A)}}
    adict = {"vocab": ["a%d" % i for i in range(A)], "dict": {"a%d" % i: i for i in range(A)}}
    return {"ids": ids, "image_id2idx": {i: j for j, i in enumerate(ids)}, "processed": processed,
            "answer_dict": adict, "ws_dict": ws,
            "image_features": np.maximum(rng.standard_normal((num_images, R, D)), 0).astype(np.float32),
            "spatial_features": spat, "normal_boxes": nb, "num_boxes": np.full(num_images, R, np.int32)}
