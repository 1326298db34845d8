"""MI355X counterpart of vqa/model_vqa.py -- the registry's oldest model and the default `--model_type` of
vqa/trainer.py:18,337.

One BasicLSTMCell(512) (modules.encode_L with its default cell_type, vlmap/modules.py:124-140) encodes the question AND every
candidate answer's token sequence (`data_info.hdf5`: intseq_ans / intseq_ans_len / num_answers, :35-45); `L2V` maps the question
code into the visual space, a dot-product `modules.attention` pools the 512-d region features of the model_vfeat extractor,
`V2L` maps the pooled feature back, and every (question, answer) pair is scored by
`classifier(tanh(answer_layer1(answer_ft) + pooled_layer1(pooled_map_L) + q_layer1(q_L_ft)))` (:232-257).  Embedding =
`modules.GloVe_vocab`: constant GloVe rows for all but the last three vocabulary entries, which are the trainable
`GloVe/learn` (vlmap/modules.py:451-467).  V2L / L2V train only with `config.ft_vlmap` (:63-74); transfer set V2L, L2V,
encode_L, GloVe (:76-88).  loss = mean_B sum_A sigmoid-CE (no train-answer mask); report = answer_loss, answer_accuracy;
mid_result = num_V_ft, normal_boxes, att_score, pred.  `model_type` 13 of the C step (csrc/legacy_vqa.inc, csrc/lstm_ops.hip).
The reference constructor is (batch, config, is_train); `image_features` stays an optional extra here."""
import os

import numpy as np
import torch

from . import fusion as F
from . import hdf5_io
from .log import log
from .model_vlmap_answer import Model as _Base, _load_pickle, get_dummy_data, load_image_features

W_DIM = 300   # vqa/model_vqa.py:10-13
L_DIM = 512
MAP_DIM = 512
V_DIM = 512


def glove_vocab_init(vocab, glove=None, rng=None):
    """modules.GloVe_vocab (vlmap/modules.py:451-467): (fixed [Vq-3, 300] = GloVe rows of vocab[:-3], learn [3, 300] ~
    U(-0.01, 0.01)).  Without GloVe files (download-only) the fixed rows are zero, or small noise for synthetic runs."""
    n = len(vocab["vocab"]) - 3
    fixed = np.zeros([n, W_DIM], np.float32)
    if glove is not None:
        for i, w in enumerate(vocab["vocab"][:-3]):
            fixed[i] = glove["param"][glove["dict"][w]]            # KeyError for an unknown word, as in the reference
    elif rng is not None:
        fixed = (0.3 * rng.standard_normal((n, W_DIM))).astype(np.float32)
    learn = (rng if rng is not None else np.random.default_rng(0)).uniform(-0.01, 0.01, size=(3, W_DIM)).astype(np.float32)
    return fixed, learn


class Model(_Base):
    MODEL_TYPE = "vqa"
    REPORT_RENAME = (("answer_loss", "answer_train_loss"), ("answer_accuracy", "answer_acc"))

    def __init__(self, batch, config, is_train=True, image_features=None):
        self.batch, self.config, self.is_train = batch, config, is_train
        self.image_dir = getattr(config, "image_dir", None)
        self.device = torch.device(getattr(config, "device", "cuda:0"))
        self.losses, self.report, self.mid_result = {}, {}, {}
        self.output, self.heavy_output, self.vis_image = {}, {}, {}
        self.vocab = getattr(config, "vocab", None) or _load_pickle(config.vocab_path)
        self.ft_vlmap = bool(getattr(config, "ft_vlmap", False))
        # answer candidates (:35-45): token sequences of every answer
        ans = getattr(config, "answer_intseq", None)
        if ans is not None:
            self.answer_intseq_value = np.asarray(ans, np.int32)
            self.answer_intseq_len_value = np.asarray(config.answer_intseq_len, np.int32)
            self.num_answer = int(self.answer_intseq_value.shape[0])
        else:
            log.infov("loading answer info..")
            with hdf5_io.File(os.path.join(config.tf_record_dir, "data_info.hdf5")) as f:
                info = f["data_info"]
                self.answer_intseq_value = np.array(info["intseq_ans"]).astype(np.int32)
                self.answer_intseq_len_value = np.array(info["intseq_ans_len"]).astype(np.int32)
                self.num_answer = int(np.asarray(info["num_answers"][()]))
        A = self.num_answer
        # the step's loss / score kernels take the answer masks of the newer models: nothing is masked here
        self.train_answer_mask, self.test_answer_mask = np.ones(A, np.float32), np.zeros(A, np.float32)
        self.obj_answer_mask = self.attr_answer_mask = np.zeros(A, np.float32)
        self.answer_exist_mask = np.ones(A, np.float32)
        if getattr(config, "debug", 0):
            feats = get_dummy_data()
        elif image_features is None:
            log.infov("loading image features...")
            feats = load_image_features(config.vfeat_path)
        else:
            feats = (image_features["features"], image_features["spatials"], image_features["normal_boxes"],
                     image_features["num_boxes"], image_features["max_box_num"], image_features["vfeat_dim"])
        (self.features, self.spatials, self.normal_boxes, self.num_boxes, self.max_box_num, self.vfeat_dim) = feats
        if int(self.vfeat_dim) != V_DIM:
            raise ValueError("model 'vqa' attends with a dot product: it needs %d-d region features (model_vfeat's), got %d"
                             % (V_DIM, int(self.vfeat_dim)))
        self._word_weights, self._step, self._engine = None, 0, None
        self.build()

    def filter_train_vars(self, trainable_vars):
        return F.filter_train_vars(list(trainable_vars), self.MODEL_TYPE, ft_vlmap=self.ft_vlmap)

    def _make_engine(self, B, T):
        cfg = self.config
        seed = int(getattr(cfg, "seed", 123))
        rng = np.random.default_rng(seed)
        g = torch.Generator().manual_seed(seed)
        Vq = len(self.vocab["vocab"])
        synthetic = getattr(cfg, "debug", 0) or getattr(cfg, "synthetic", 0)
        fixed, learn = glove_vocab_init(self.vocab, getattr(cfg, "glove", None), rng if synthetic else None)
        shapes = F.variable_shapes(self.MODEL_TYPE, Vq, W_DIM, V_DIM, L_DIM, self.num_answer, map_dim=MAP_DIM)
        p = {}
        for n, s in shapes.items():
            if n == "GloVe/learn":
                p[n] = learn
            elif n.endswith("/weights") or n.endswith("/kernel"):
                lim = (6.0 / (s[0] + s[1])) ** 0.5
                p[n] = ((torch.rand(s, generator=g) * 2 - 1) * lim).numpy()
            else:
                p[n] = np.zeros(s, np.float32)
        eng = F.FusionEngine(model_type=self.MODEL_TYPE, B=B, R=self.max_box_num, D=V_DIM, H=L_DIM, T=T, W=W_DIM,
                             A=self.num_answer, Vq=Vq, N_img=len(self.features), params=p, device=self.device,
                             global_batch=getattr(cfg, "global_batch", None), map_dim=MAP_DIM, ft_vlmap=self.ft_vlmap,
                             glove_fixed=fixed, answers={"intseq": self.answer_intseq_value, "len": self.answer_intseq_len_value})
        eng.bind_inputs(table=self._to_dev(self.features, torch.float32), nbox_table=self._to_dev(self.num_boxes, torch.int32),
                        answer_masks={"train": self._to_dev(self.train_answer_mask, torch.float32),
                                      "obj": self._to_dev(self.obj_answer_mask, torch.float32),
                                      "attr": self._to_dev(self.attr_answer_mask, torch.float32),
                                      "exist": self._to_dev(self.answer_exist_mask, torch.float32)})
        return eng

    def build(self):
        """build network architecture and loss (here: run it on self.batch); no dropout in this model"""
        db = self._device_batch()
        B, T = db["q_intseq"].shape
        gb = getattr(self.config, "global_batch", None)
        if self._engine is None:
            self._engine = self._make_engine(B, T)
        else:
            self._engine.resize(B, T, gb)
        eng = self._engine
        self._step += 1
        self._db = db
        eng.forward(db, None, None, want_dz=self.is_train)
        rep = eng.tensor("report")
        self.report = {"answer_loss": rep[0], "answer_accuracy": rep[2]}
        self.losses = {"answer": rep[0]}
        self.loss = rep[0]
        self.mid_result = {"num_V_ft": eng.tensor("num_V_ft"), "att_score": eng.tensor("att_score").view(B, eng.dims.R),
                           "pred": eng.tensor("pred"), "logit": eng.tensor("logit").view(B, eng.dims.A),
                           "q_L_ft": eng.tensor("q_L_ft").view(B, -1)}
        self.output = {"pred": self.mid_result["pred"], "att_score": self.mid_result["att_score"], "logit": self.mid_result["logit"]}
        return self.loss
