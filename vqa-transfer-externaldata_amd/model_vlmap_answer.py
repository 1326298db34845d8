"""MI355X counterpart of vqa/model_vlmap_answer.py (and, via model_standard.py, of
vqa/model_standard.py): same constructor, attributes and variable names; the graph
is replaced by libvqahot.so driven through FusionEngine.

Reference contract reproduced (vqa/model_vlmap_answer.py:17-100, 304-326):
  Model(batch, config, is_train=True, image_features=None) -> build() sets
  .loss, .losses, .report (13 keys), .output, .mid_result, .heavy_output, .vocab,
  .answer_dict; filter_train_vars / filter_transfer_vars select by top-level scope.
Differences forced by the platform: there is no deferred TF graph, so build() runs
the forward pass for the CURRENT self.batch (set_batch() swaps it); tensors are torch
CUDA tensors; the dropout that the reference draws inside tf.nn.dropout comes from
explicit reproducible keep-masks keyed by (config.seed, step).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from . import fusion as F
from . import hdf5_io
from .log import log

W_DIM = 300   # Word dimension           (vqa/model_vlmap_answer.py:10-12)
L_DIM = 1024  # Language dimension
V_DIM = 1024


def get_dummy_data():
    """util/__init__.py:5-17 (--debug 1): zero features for 500 images, one box each."""
    bn, bs, dim = 500, 36, 2048
    return (np.zeros([bn, bs, dim], np.float32), np.zeros([bn, bs, 6], np.float32),
            np.zeros([bn, bs, 4], np.float32), np.ones([bn], np.int32), bs, dim)


def _load_pickle(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:      # python-2 cPickle files of the reference
            f.seek(0)
            return pickle.load(f, encoding="latin1")


def learn_glove_init(vocab, glove=None, rng=None, oov_mean_initialize=False):
    """modules.LearnGloVe (vlmap/modules.py:415-448): rows of words found in GloVe get
    their vector, others zeros.  `glove` = {'dict': word->row, 'param': [n,300]} or None
    (GloVe files are download-only; then every row starts at zero like an OOV word, or
    small uniform noise when an rng is given for synthetic runs).
    oov_mean_initialize (the answer matrix of model_standard_word2vec): an entry that is not a GloVe word but a
    phrase of GloVe words gets the mean of their vectors; any other non-empty entry raises, as the reference does."""
    n = len(vocab["vocab"])
    w = np.zeros([n, W_DIM], np.float32)
    if glove is not None:
        for i, word in enumerate(vocab["vocab"]):
            j = glove["dict"].get(word)
            if j is not None and j < glove["param"].shape[0]:
                w[i] = glove["param"][j]
            elif oov_mean_initialize and word != "":
                words = word.split()
                if not all(x in glove["dict"] for x in words):
                    raise Exception("Unkown words {}".format(words))          # (sic) vlmap/modules.py:438
                w[i] = np.mean([glove["param"][glove["dict"][x]] for x in words], 0)
    elif rng is not None:
        w = rng.uniform(-0.01, 0.01, size=w.shape).astype(np.float32)
    return w


def answer_exist_mask(answer_dict, word_answer_dict=None):
    """modules.AnswerExistMask (vlmap/modules.py:575-586)."""
    mask = np.zeros([len(answer_dict["vocab"])], np.float32)
    if word_answer_dict is not None:
        for i, a in enumerate(answer_dict["vocab"]):
            if a in word_answer_dict["dict"]:
                mask[i] = 1.0
    else:
        mask += 1.0
    return mask


def word_weight_answer_init(answer_dict, input_dim, word_weights=None, default_bias=-100.0,
                            weight_name="class_weights", bias_name="class_biases"):
    """modules.WordWeightAnswer (vlmap/modules.py:589-627): head initialised by answer-string
    lookup into the exported class weights (`weight_name` / `bias_name` datasets of weights.hdf5); missing answers get
    weight 0 and bias -100."""
    dim = len(answer_dict["vocab"])
    weights = np.zeros([input_dim, dim], np.float32)
    biases = np.zeros([dim], np.float32) + default_bias
    if word_weights is not None:
        wd = word_weights["answer_dict"]["dict"]
        for i, a in enumerate(answer_dict["vocab"]):
            if a in wd:
                weights[:, i] = word_weights[weight_name][:, wd[a]]
                biases[i] = word_weights[bias_name][wd[a]]
    return weights, biases


def word_weight_embed_init(vocab, word_weights=None, weight_name="v_word"):
    """modules.WordWeightEmbed (vlmap/modules.py:393-412): rows of the question vocabulary found in the word-weight
    directory's vocab.pkl get that directory's `weight_name` row, every other row starts at zero."""
    w = np.zeros([len(vocab["vocab"]), W_DIM], np.float32)
    if word_weights is not None and weight_name in word_weights and word_weights.get("vocab") is not None:
        src, wd = word_weights[weight_name], word_weights["vocab"]["dict"]
        for i, word in enumerate(vocab["vocab"]):
            j = wd.get(word)
            if j is not None and j < src.shape[0]:
                w[i] = src[j]
    return w


def load_word_weight_dir(path):
    """word_weights_model-N/ written by export_word_weights: answer_dict.pkl + weights.hdf5 with class_weights /
    class_biases (vlmap/modules.py:598-601), read without h5py (hdf5_io); a weights.npz with the same keys is
    accepted as an alternative."""
    ad = _load_pickle(os.path.join(path, "answer_dict.pkl"))
    h5, npz = os.path.join(path, "weights.hdf5"), os.path.join(path, "weights.npz")
    # class_* of export_word_weights.py; v_class_* / l_class_* of export_noc_word_weights.py:72-75 (model_vlmap_answer_noc)
    # v_word + vocab.pkl: modules.WordWeightEmbed of the bi-directional models (vlmap/modules.py:397-406)
    wanted = ("class_weights", "class_biases", "v_class_weights", "v_class_biases", "l_class_weights", "l_class_biases", "v_word")
    if os.path.exists(h5):
        with hdf5_io.File(h5) as f:
            out = {k: np.array(f[k]) for k in wanted if k in f}
    elif os.path.exists(npz):
        z = np.load(npz)
        out = {k: z[k] for k in wanted if k in z.files}
    else:
        raise FileNotFoundError("neither weights.hdf5 nor weights.npz under %s" % path)
    out["answer_dict"] = ad
    vp = os.path.join(path, "vocab.pkl")
    out["vocab"] = _load_pickle(vp) if os.path.exists(vp) else None
    return out


def load_image_features(path):
    """(features, spatials, normal_boxes, num_boxes, max_box_num, vfeat_dim) of a region-feature file
    (vqa/model_vlmap_answer.py:59-70).  The format is taken from the file's magic, not its name: the reference's
    HDF5 (image_features, spatial_features, normal_boxes, num_boxes, data_info/{max_box_num,vfeat_dim}) is read
    through hdf5_io -- the big tables come back as np.memmap views, paged in while they are uploaded -- and an
    .npz with the same names is accepted as an alternative."""
    if hdf5_io.is_hdf5(path):
        f = hdf5_io.File(path)              # kept open: the returned arrays are views of the mapped file
        info = f["data_info"]
        return (f["image_features"].read(), f["spatial_features"].read(), f["normal_boxes"].read(),
                np.array(f["num_boxes"]), int(info["max_box_num"][()]), int(info["vfeat_dim"][()]))
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:2] != b"PK":
        raise ValueError("%s is neither an HDF5 file nor an .npz archive (magic %r)" % (path, magic))
    z = np.load(path)
    return (z["image_features"], z["spatial_features"], z["normal_boxes"], z["num_boxes"],
            int(z["max_box_num"]), int(z["vfeat_dim"]))


class Model(object):
    MODEL_TYPE = "vlmap_answer"

    def __init__(self, batch, config, is_train=True, image_features=None):
        """The five older ablations (model_vlmap_answer2 / _no_noise / _adapt / _full / _ent) predate `image_features` in
        the reference (their constructors are (batch, config, is_train)); the argument stays optional here for all."""
        self.batch = batch
        self.config = config
        self.image_dir = getattr(config, "image_dir", None)
        self.is_train = is_train
        self.device = torch.device(getattr(config, "device", "cuda:0"))

        self.word_weight_dir = getattr(config, "vlmap_word_weight_dir", None)
        if self.word_weight_dir is None and self.MODEL_TYPE in F.VLMAP_FAMILY + F.BI_FAMILY:
            log.warning("word_weight_dir is None")

        self.losses, self.report, self.mid_result = {}, {}, {}
        self.output, self.heavy_output, self.vis_image = {}, {}, {}

        # vocab / answer_dict: pickles at the reference's paths, or in-memory dicts on the config
        self.vocab = getattr(config, "vocab", None) or _load_pickle(config.vocab_path)
        self.answer_dict = getattr(config, "answer_dict", None) or _load_pickle(
            os.path.join(config.tf_record_dir, "answer_dict.pkl"))
        self.num_answer = len(self.answer_dict["vocab"])
        self.num_train_answer = self.answer_dict["num_train_answer"]
        A = self.num_answer
        self.train_answer_mask = (np.arange(A) < self.num_train_answer).astype(np.float32)
        self.test_answer_mask = 1.0 - self.train_answer_mask
        self.obj_answer_mask = np.asarray(self.answer_dict["is_object"], np.float32)
        self.attr_answer_mask = np.asarray(self.answer_dict["is_attribute"], np.float32)

        word_weights = None
        if self.word_weight_dir is not None:
            word_weights = getattr(config, "word_weights", None) or load_word_weight_dir(self.word_weight_dir)
        self.answer_exist_mask = answer_exist_mask(
            self.answer_dict, word_weights["answer_dict"] if word_weights else None)

        if getattr(config, "debug", 0):
            feats = get_dummy_data()
        elif image_features is None:
            log.infov("loading image features...")
            feats = load_image_features(config.vfeat_path)
            log.infov("done")
        else:
            feats = (image_features["features"], image_features["spatials"], image_features["normal_boxes"],
                     image_features["num_boxes"], image_features["max_box_num"], image_features["vfeat_dim"])
        (self.features, self.spatials, self.normal_boxes, self.num_boxes, self.max_box_num, self.vfeat_dim) = feats

        self._word_weights = word_weights
        self._step = 0
        self._engine = None
        self.build()

    # ---------------------------------------------------------------- variable filters
    def filter_train_vars(self, trainable_vars):
        return F.filter_train_vars(list(trainable_vars), self.MODEL_TYPE)

    def filter_transfer_vars(self, all_vars):
        return F.filter_transfer_vars(list(all_vars), self.MODEL_TYPE)

    # ---------------------------------------------------------------- engine plumbing
    def _initial_params(self, shapes):
        cfg = self.config
        seed = int(getattr(cfg, "seed", 123))
        g = torch.Generator().manual_seed(seed)
        rng = np.random.default_rng(seed)
        sc = F.scope_names(self.MODEL_TYPE)
        p = {}
        for n, s in shapes.items():
            if n == sc["embed"]:
                p[n] = learn_glove_init(self.vocab, getattr(cfg, "glove", None),
                                        rng if getattr(cfg, "debug", 0) or getattr(cfg, "synthetic", 0) else None)
            elif n == sc.get("embed2"):          # V_WordMap of the bi-directional models (:45-46)
                p[n] = word_weight_embed_init(self.vocab, self._word_weights)
            elif n.endswith("/weights") or n.endswith("/kernel"):
                lim = (6.0 / (s[0] + s[1])) ** 0.5            # layers.fully_connected: Xavier uniform
                p[n] = ((torch.rand(s, generator=g) * 2 - 1) * lim).numpy()
            elif n.endswith("gates/bias") or n.endswith("LayerNorm/gamma"):
                p[n] = np.ones(s, np.float32)                  # GRUCell gate bias 1.0, LN gamma 1
            else:
                p[n] = np.zeros(s, np.float32)
        if self.MODEL_TYPE in ("vlmap_answer",) + F.TWO_HEAD_FAMILY + F.ABLATION_FAMILY + F.BI_FAMILY:      # the other heads keep their Xavier / zero initialisation
            w, b = word_weight_answer_init(self.answer_dict, 2 * L_DIM, self._word_weights)
            p[sc["head"] + "/fc/weights"], p[sc["head"] + "/fc/biases"] = w, b
        elif self.MODEL_TYPE in F.NOC_FAMILY:      # WordWeightAnswerV / L from v_class_* / l_class_* (:190-202)
            for hd, pre in ((sc["head"], "v_"), (sc["head2"], "l_")):
                w, b = word_weight_answer_init(self.answer_dict, 2 * L_DIM, self._word_weights,
                                               weight_name=pre + "class_weights", bias_name=pre + "class_biases")
                p[hd + "/fc/weights"], p[hd + "/fc/biases"] = w, b
        return p

    def _engine_kwargs(self):
        """extra FusionEngine arguments of a model variant (none for the two base models)"""
        return {}

    _UPLOAD_CHUNK_BYTES = 256 << 20

    def _to_dev(self, a, dtype):
        if torch.is_tensor(a):
            return a.to(device=self.device, dtype=dtype).contiguous()
        a = np.asarray(a) if not isinstance(a, np.ndarray) else a
        if a.flags.writeable:
            return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype).contiguous()
        # read-only arrays (the np.memmap views of a feature file, load_image_features): torch must not alias them.
        # Uploaded in bounded slices -- each slice is copied on read, so a 36 GB table never sits in host RAM twice
        out = torch.empty(a.shape, dtype=dtype, device=self.device)
        if a.ndim == 0 or a.size == 0:
            if a.size:
                out.copy_(torch.from_numpy(np.array(a)))
            return out
        rows = max(1, int(self._UPLOAD_CHUNK_BYTES // max(a.nbytes // a.shape[0], 1)))
        for lo in range(0, a.shape[0], rows):
            out[lo:lo + rows].copy_(torch.from_numpy(np.array(a[lo:lo + rows])))
        return out

    def _make_engine(self, B, T):
        Vq = len(self.vocab["vocab"])
        shapes = F.variable_shapes(self.MODEL_TYPE, Vq, W_DIM, self.vfeat_dim, V_DIM, self.num_answer)
        eng = F.FusionEngine(model_type=self.MODEL_TYPE, B=B, R=self.max_box_num, D=self.vfeat_dim, H=V_DIM, T=T,
                             W=W_DIM, A=self.num_answer, Vq=Vq, N_img=len(self.features),
                             params=self._initial_params(shapes), device=self.device,
                             global_batch=getattr(self.config, "global_batch", None), **self._engine_kwargs())
        eng.bind_inputs(
            table=self._to_dev(self.features, torch.float32),        # the whole table lives in HBM (a1)
            nbox_table=self._to_dev(self.num_boxes, torch.int32),
            answer_masks={"train": self._to_dev(self.train_answer_mask, torch.float32),
                          "obj": self._to_dev(self.obj_answer_mask, torch.float32),
                          "attr": self._to_dev(self.attr_answer_mask, torch.float32),
                          "exist": self._to_dev(self.answer_exist_mask, torch.float32)})
        return eng

    @property
    def engine(self):
        return self._engine

    def variables(self):
        """name -> tensor for every variable (TF variable names)."""
        return dict(self._engine.params)

    def set_batch(self, batch):
        self.batch = batch

    _DEVICE_KEYS = (("image_idx", torch.int64), ("q_intseq", torch.int32), ("q_intseq_len", torch.int32),
                    ("answer_target", torch.float32))

    def to_device_batch(self, batch):
        """The batch dict with its four model inputs moved to the device (other entries -- ids, image_id -- kept)."""
        out = dict(batch)
        for k, dt in self._DEVICE_KEYS:
            out[k] = self._to_dev(batch[k], dt)
        return out

    def _device_batch(self):
        b = self.batch
        db = {k: self._to_dev(b[k], dt) for k, dt in self._DEVICE_KEYS}
        if b.get("live_rows") is not None:      # rows sorted by length (input_ops_vqa.sort_by_length)
            db["live_rows"] = b["live_rows"]
        return db

    # ---------------------------------------------------------------- build = forward
    REPORT_RENAME = None      # ((new key, key of the 13-scalar report), ...) for the variants with an older, shorter report

    def map_report(self, report, extra=None):
        """the step's 13 report scalars under this model's `report` keys (variants with an older report rename them),
        plus a variant's additional scalars (`extra`: FusionEngine.extra_report, e.g. latent_loss / entropy)"""
        out = dict(report) if self.REPORT_RENAME is None else {new: report[old] for new, old in self.REPORT_RENAME}
        if extra:
            out.update({k: v for k, v in extra.items() if k != "total_loss"})
        out.update(self._constant_report())
        return out

    def _constant_report(self):
        """report entries that are not computed from the batch (e.g. model_step of the oldest variants)"""
        return {}

    def _variant_inputs(self, eng, seed, row_offset, global_rows, dropout_off):
        """extra per-step inputs of a variant as keyword arguments of FusionEngine.forward (noise, keep_tile)"""
        return {}

    def build(self):
        """build network architecture and loss (here: run it on self.batch)"""
        db = self._device_batch()
        B, T = db["q_intseq"].shape
        gb = getattr(self.config, "global_batch", None)
        if self._engine is None:
            self._engine = self._make_engine(B, T)
        else:
            self._engine.resize(B, T, gb)
        eng = self._engine
        kj2 = None
        if getattr(self.config, "dropout_off", False):
            ka = kj = None
        else:   # tf.nn.dropout is applied unconditionally in the reference (also at eval time)
            # under data parallelism the stream is indexed by the global row (config.shard_row_offset, global_batch)
            ka, kj = eng.make_keep_masks(int(getattr(self.config, "seed", 123)), self._step,
                                         row_offset=int(getattr(self.config, "shard_row_offset", 0) or 0),
                                         global_rows=gb)
            if self.MODEL_TYPE in F.NOC_FAMILY:
                kj2 = eng.make_keep_mask_joint2(int(getattr(self.config, "seed", 123)), self._step,
                                                row_offset=int(getattr(self.config, "shard_row_offset", 0) or 0), global_rows=gb)
        extra_in = self._variant_inputs(eng, int(getattr(self.config, "seed", 123)),
                                        int(getattr(self.config, "shard_row_offset", 0) or 0), gb,
                                        bool(getattr(self.config, "dropout_off", False)))
        self._step += 1
        self._db, self._keep = db, (ka, kj)
        eng.forward(db, ka, kj, want_dz=self.is_train, keep_joint2=kj2, **extra_in)

        d = eng.dims
        A, R = d.A, d.R
        stats = eng.tensor("stats").view(B, 16)
        rep = eng.tensor("report")
        keys = [eng.lib.vqa_report_key(i).decode() for i in range(13)]
        xk = eng.EXTRA_REPORT_KEYS.get(self.MODEL_TYPE)
        self.report = self.map_report({k: rep[i] for i, k in enumerate(keys)},
                                      {xk[0]: rep[13], xk[1]: rep[14]} if xk else None)
        self.losses = {"answer": rep[0]}
        if xk:        # self.losses['latent'] / ['entropy'] = the weighted term; self.loss = their sum
            self.losses[{"vlmap_answer_full": "latent", "vlmap_answer_ent": "entropy"}[self.MODEL_TYPE]] = rep[14]
        self.loss = eng.loss()
        self.mid_result = {
            "num_V_ft": eng.tensor("num_V_ft"), "q_linear_v": eng.tensor("q_linear_v").view(B, -1),
            "att_score": eng.tensor("att_score").view(B, R), "pooled_V_ft": eng.tensor("pooled_V_ft").view(B, -1),
            "pooled_linear_l": eng.tensor("pooled_linear_l").view(B, -1),
            "l_linear_l": eng.tensor("l_linear_l").view(B, -1), "joint": eng.tensor("joint").view(B, -1),
            "logit": eng.tensor("logit").view(B, A), "pred": eng.tensor("pred"),
        }
        self.output = {
            "att_score": self.mid_result["att_score"], "logit": self.mid_result["logit"],
            "pred": self.mid_result["pred"],
            "test_obj_score": stats[:, 5], "test_obj_max_score": stats[:, 10],
            "test_attr_score": stats[:, 6], "test_attr_max_score": stats[:, 11],
            "all_score": stats[:, 2], "max_train_score": stats[:, 14],
        }
        self.heavy_output = {"condition": eng.tensor("condition").view(B, -1)}
        return self.loss

    # the two halves of optimize_loss that the Trainer drives (vqa/trainer.py:106-114)
    def backward(self, reducer=None):
        """reducer: optional dp.BucketedAllReduce -> gradient buckets are all-reduced while backward runs."""
        self._engine.backward(reducer=reducer)

    def apply_gradients(self, learning_rate, allreduce=None):
        if allreduce is not None:
            allreduce(self._engine.grad_flat)
        self._engine.optimizer_step(learning_rate)
