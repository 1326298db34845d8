"""Device-side engine of the fusion model: owns the flat parameter / gradient /
Adam-state buffers and the kernel workspace, and drives libvqahot.so.

Replaces what `session.run([loss, report, optimizer])` does in the reference
(vqa/trainer.py:275-287): forward + backward + clip_by_global_norm(20) + Adam of
vqa/model_vlmap_answer.py / vqa/model_standard.py.  Variable names are the
reference's TF variable names -- they are the checkpoint and
transfer contract.

Memory layout in HBM (all fp32):
  train_flat  = [LearnGloVe/embed_map | every other trainable var, name order]
  grad_flat   = same layout + 4 trailing floats; slot 0 of the tail carries the
                un-aggregated embedding-slice sum of squares so that ONE
                all-reduce moves gradients and that scalar under data parallel
  m_flat, v_flat = Adam moments, same layout as train_flat
  frozen_flat = variables excluded by filter_train_vars (vlmap_answer only)
  workspace   = activations + backward scratch, carved by the library
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8     # tf.train.AdamOptimizer defaults
CLIP_NORM = 20.0                                   # vqa/trainer.py:111

FROZEN_TOP_SCOPES_VLMAP = ("q_linear_l", "pooled_linear_l", "joint_fc", "WordWeightAnswer")
TRANSFER_TOP_SCOPES_VLMAP = ("q_linear_l", "pooled_linear_l", "joint_fc")

_INT_TENSORS = {"num_V_ft", "pred"}


# model_standard and its variants: one architecture, every variable trainable (vqa/model_standard.py:80-84,
# vqa/model_standard_word2vec.py, vqa/model_standard_testmask.py:64-68)
STANDARD_FAMILY = ("standard", "standard_word2vec", "standard_testmask")
# model_vlmap_answer and its variant with a second, trainable head on the fixed joint (vqa/model_vlmap_answer_vqa_all2.py)
# "no composition" (vqa/model_vlmap_answer_noc.py = model_vlmap_answer_nocarch.py): joint_v / joint_l instead of joint_fc,
# WordWeightAnswerV / WordWeightAnswerL instead of WordWeightAnswer
NOC_FAMILY = ("vlmap_answer_noc", "vlmap_answer_nocarch")
FROZEN_TOP_SCOPES_NOC = ("q_linear_l", "pooled_linear_l", "joint_v", "joint_l", "WordWeightAnswerV", "WordWeightAnswerL")
TRANSFER_TOP_SCOPES_NOC = ("q_linear_l", "pooled_linear_l", "joint_v", "joint_l")
TWO_HEAD_FAMILY = ("vlmap_answer_vqa_all", "vlmap_answer_vqa_all2")       # fixed head + trainable TunedWordWeightAnswer
# the five older ablations of model_vlmap_answer, each the base model with ONE change (include/vqa_hot.h: VQA_MODEL_*):
# vqa/model_vlmap_answer2.py, _no_noise.py, _adapt.py, _full.py, _ent.py
ABLATION_FAMILY = ("vlmap_answer2", "vlmap_answer_no_noise", "vlmap_answer_adapt", "vlmap_answer_full", "vlmap_answer_ent")
VLMAP_FAMILY = ("vlmap_answer",) + TWO_HEAD_FAMILY + NOC_FAMILY + ABLATION_FAMILY
# the bi-directional-GRU generation: vqa/model_vlmap_finetune.py and model_vlmap_only.py (same graph, different train set)
BI_FAMILY = ("vlmap_finetune", "vlmap_only")
FROZEN_TOP_SCOPES_ONLY = ("V_WordMap", "v_word_fc", "q_linear_v", "v_linear_v", "hadamard_attention", "q_linear_l",
                          "pooled_linear_l", "joint_fc", "WordWeightAnswer")                     # model_vlmap_only.py:64-76
TRANSFER_TOP_SCOPES_BI = ("v_word_fc", "q_linear_v", "v_linear_v", "hadamard_attention", "q_linear_l", "pooled_linear_l",
                          "joint_fc")                                                            # model_vlmap_finetune.py:70-87
# the registry's oldest model and the default of vqa/trainer.py: vqa/model_vqa.py (LSTM over questions and answers, L2V / V2L,
# dot-product attention over model_vfeat's 512-d features, a broadcast tanh scoring layer)
LEGACY_FAMILY = ("vqa",)
LSTM_SCOPE = "encode_L/rnn/basic_lstm_cell"
NUM_MARGINAL = 200            # vqa/model_vlmap_answer_ent.py:16
W_ENTROPY = 0.1               # vqa/model_vlmap_answer_ent.py:14
LATENT_LOSS_WEIGHT = 0.1      # vqa/model_vlmap_answer_full.py:33


def scope_names(model_type):
    """logical layer -> TF variable scope (vqa/model_vlmap_answer.py:126-185,
    vqa/model_standard.py:223-275)."""
    if model_type in NOC_FAMILY:
        return {"embed": "LearnGloVe/embed_map", "v_linear_v": "v_linear_v",
                "gru_gates": "encode_L/rnn/gru_cell/gates", "gru_cand": "encode_L/rnn/gru_cell/candidate",
                "q_linear_v": "q_linear_v", "score": "hadamard_attention/compute/score",
                "pooled_linear_l": "pooled_linear_l", "q_linear_l": "q_linear_l",
                "joint_fc": "joint_v", "joint2": "joint_l", "head": "WordWeightAnswerV", "head2": "WordWeightAnswerL"}
    if model_type in LEGACY_FAMILY:
        return {"embed": "GloVe/learn"}
    if model_type in BI_FAMILY:
        g = "encode_L_bi/bidirectional_rnn/%s/gru_cell/%s"
        return {"embed": "LearnGloVe/embed_map", "embed2": "V_WordMap/embed_map", "v_linear_v": "v_linear_v",
                "gru_gates": g % ("fw", "gates"), "gru_cand": g % ("fw", "candidate"),
                "gru_bw_gates": g % ("bw", "gates"), "gru_bw_cand": g % ("bw", "candidate"),
                "q_att_key": "q_att_key", "q_att_query": "q_att_query", "word_score": "word_attention/compute/score",
                "v_word_fc": "v_word_fc", "q_linear_v": "q_linear_v", "score": "hadamard_attention/compute/score",
                "pooled_linear_l": "pooled_linear_l", "q_linear_l": "q_linear_l", "joint_fc": "joint_fc",
                "head": "WordWeightAnswer"}
    if model_type in VLMAP_FAMILY:
        pre, head = "", "WordWeightAnswer"
    elif model_type in STANDARD_FAMILY:
        pre, head = "reasoning/", "reasoning/classifier"
    else:
        raise ValueError("unknown model_type %r" % (model_type,))
    return {"embed": "LearnGloVe/embed_map", "v_linear_v": "v_linear_v",
            "gru_gates": "encode_L/rnn/gru_cell/gates", "gru_cand": "encode_L/rnn/gru_cell/candidate",
            "q_linear_v": "q_linear_v", "score": "hadamard_attention/compute/score",
            "pooled_linear_l": pre + "pooled_linear_l", "q_linear_l": pre + "q_linear_l",
            "joint_fc": pre + "joint_fc", "head": head,
            # vlmap_answer_vqa_all2 only (:202-220): the tuned head and the two tuned layers that feed nothing
            "head2": "TunedWordWeightAnswer", "tuned_q_linear_l": "tuned_q_linear_l", "tuned_joint_fc": "tuned_joint_fc",
            # the ablations: answer2 (:127-130) | no_noise + full (:122-125) | full (:124-131) | adapt (:132-135)
            "q_L_ft2": "q_L_ft2", "q_L_mean": "q_L_mean", "q_L_log_sigma_sq": "q_L_log_sigma_sq", "v_adapt": "v_adapt"}


def variable_shapes(model_type, Vq, W, D, H, A, map_dim=None):
    """name -> shape for every variable of the model."""
    sc = scope_names(model_type)
    if model_type in LEGACY_FAMILY:      # vqa/model_vqa.py: H = L_DIM, D = V_DIM = vfeat_dim, map_dim = MAP_DIM (:10-13)
        M = int(map_dim or H)
        s = {"GloVe/learn": (3, W), LSTM_SCOPE + "/kernel": (W + H, 4 * H), LSTM_SCOPE + "/bias": (4 * H,)}
        for scope, fin, fout, bias in (("L2V/fc_1", H, M, True), ("L2V/fc_2", M, M, True), ("L2V/Linear", M, D, True),
                                       ("V2L/fc_1", D, M, True), ("V2L/fc_2", M, M, True), ("V2L/Linear", M, H, True),
                                       ("reasoning/answer_layer1", H, H, False), ("reasoning/pooled_layer1", H, H, False),
                                       ("reasoning/q_layer1", H, H, True), ("reasoning/classifier", H, 1, True)):
            s[scope + "/fc/weights"] = (fin, fout)
            if bias:
                s[scope + "/fc/biases"] = (fout,)
        return s
    s = {sc["embed"]: (Vq, W)}

    def fc(scope, fin, fout, ln):
        s[scope + "/fc/weights"] = (fin, fout)
        s[scope + "/fc/biases"] = (fout,)
        if ln:
            s[scope + "/LayerNorm/beta"] = (fout,)
            s[scope + "/LayerNorm/gamma"] = (fout,)

    if model_type in BI_FAMILY:          # encode_L_bidirection: two cells of H / 2 units each (vlmap/modules.py:103-104)
        if H % 2:
            raise ValueError("the bi-directional encoder is built for an even language dimension")
        h = H // 2
        s[sc["embed2"]] = (Vq, W)
        fc(sc["v_linear_v"], D, H, True)
        for gates, cand in ((sc["gru_gates"], sc["gru_cand"]), (sc["gru_bw_gates"], sc["gru_bw_cand"])):
            s[gates + "/kernel"], s[gates + "/bias"] = (W + h, 2 * h), (2 * h,)
            s[cand + "/kernel"], s[cand + "/bias"] = (W + h, h), (h,)
        fc(sc["q_att_key"], H, H, True)
        fc(sc["q_att_query"], H, H, True)
        fc(sc["word_score"], H, 1, False)
        fc(sc["v_word_fc"], W, H, True)
        fc(sc["q_linear_v"], H, H, True)
        fc(sc["score"], H, 1, False)
        fc(sc["pooled_linear_l"], D, H, True)
        fc(sc["q_linear_l"], H, H, True)
        fc(sc["joint_fc"], H, 2 * H, True)
        fc(sc["head"], 2 * H, A, False)
        return s
    fc(sc["v_linear_v"], D, H, True)
    s[sc["gru_gates"] + "/kernel"] = (W + H, 2 * H)
    s[sc["gru_gates"] + "/bias"] = (2 * H,)
    s[sc["gru_cand"] + "/kernel"] = (W + H, H)
    s[sc["gru_cand"] + "/bias"] = (H,)
    fc(sc["q_linear_v"], H, H, True)
    fc(sc["score"], H, 1, False)
    fc(sc["pooled_linear_l"], H if model_type == "vlmap_answer_adapt" else D, H, True)   # adapt pools the H-wide v_adapt
    fc(sc["q_linear_l"], H, H, True)
    if model_type == "vlmap_answer2":
        fc(sc["q_L_ft2"], H, H, True)
    if model_type in ("vlmap_answer_no_noise", "vlmap_answer_full"):
        fc(sc["q_L_mean"], H, H, False)
    if model_type == "vlmap_answer_full":
        fc(sc["q_L_log_sigma_sq"], H, H, False)
    if model_type == "vlmap_answer_adapt":
        fc(sc["v_adapt"], D, H, True)
    fc(sc["joint_fc"], H, 2 * H, True)
    # standard_word2vec: the classifier maps into the 300-d word space (vqa/model_standard_word2vec.py:180-183)
    fc(sc["head"], 2 * H, W if model_type == "standard_word2vec" else A, False)
    if model_type in TWO_HEAD_FAMILY:
        fc(sc["tuned_q_linear_l"], H, H, True)
        fc(sc["tuned_joint_fc"], H, 2 * H, True)
        fc(sc["head2"], 2 * H, A, False)
    if model_type in NOC_FAMILY:
        fc(sc["joint2"], H, 2 * H, True)
        fc(sc["head2"], 2 * H, A, False)
    return s


def filter_train_vars(names, model_type, ft_vlmap=False):
    """vqa/model_vlmap_answer.py:81-89 / vqa/model_standard.py:80-84 on variable names (ft_vlmap: vqa/model_vqa.py:63-74)."""
    if model_type in LEGACY_FAMILY:
        return [n for n in names if ft_vlmap or n.split("/")[0] not in ("V2L", "L2V")]
    if model_type in STANDARD_FAMILY or model_type == "vlmap_finetune":      # model_vlmap_finetune.py:64-68: everything
        return list(names)
    if model_type == "vlmap_only":
        return [n for n in names if n.split("/")[0] not in FROZEN_TOP_SCOPES_ONLY]
    frozen = FROZEN_TOP_SCOPES_NOC if model_type in NOC_FAMILY else FROZEN_TOP_SCOPES_VLMAP
    return [n for n in names if n.split("/")[0] not in frozen]


def filter_transfer_vars(names, model_type):
    """vqa/model_vlmap_answer.py:91-100 / vqa/model_standard.py:86-93."""
    if model_type in STANDARD_FAMILY:
        return [n for n in names if n.split("/")[0] in ("encode_L", "GloVe")]
    if model_type in BI_FAMILY:
        return [n for n in names if n.split("/")[0] in TRANSFER_TOP_SCOPES_BI]
    if model_type in LEGACY_FAMILY:                                   # vqa/model_vqa.py:76-88
        return [n for n in names if n.split("/")[0] in ("V2L", "L2V", "encode_L", "GloVe")]
    keep = TRANSFER_TOP_SCOPES_NOC if model_type in NOC_FAMILY else TRANSFER_TOP_SCOPES_VLMAP
    return [n for n in names if n.split("/")[0] in keep]


def _pad4(n):
    return (n + 3) // 4 * 4


def flat_layout(model_type, shapes, ft_vlmap=False):
    """Where every variable sits in the flat buffers (pure host arithmetic, no GPU): train_flat / grad_flat order =
    the order backward completes the gradients (vqa_fusion_backward_phases):
    [embedding | GRU candidate/* | GRU gates/* | everything else by name] (+ 4 tail floats in grad_flat), every
    variable padded to a multiple of 4 floats.  `buckets` are the slices of grad_flat FusionEngine.backward hands to a
    bucketed reducer, in the order it starts them."""
    sc = scope_names(model_type)
    names = sorted(shapes)
    train = filter_train_vars(names, model_type, ft_vlmap)
    # scatter-added tables first (the bi-directional models have a second one, V_WordMap, trainable in vlmap_finetune):
    # they are zeroed before every backward and their norm is taken over the un-aggregated slices (tail slot)
    embeds = [sc["embed"]] + ([sc["embed2"]] if sc.get("embed2") in train else [])
    is_gru = lambda n: n.startswith("encode_L/") or n.startswith("encode_L_bi/")
    gru_first = sorted((n for n in train if is_gru(n)), key=lambda n: ("/candidate/" not in n, n))
    train_names = embeds + gru_first + [n for n in train if n not in embeds and not is_gru(n)]
    frozen_names = [n for n in names if n not in train]

    def carve(name_list):
        off, table = 0, {}
        for n in name_list:
            cnt = int(np.prod(shapes[n]))
            table[n] = (off, cnt)
            off += _pad4(cnt)
        return table, off

    train_tab, n_train = carve(train_names)
    frozen_tab, n_frozen = carve(frozen_names)
    embed_floats = sum(_pad4(int(np.prod(shapes[n]))) for n in embeds)
    # the GRU tensors must sit right after the tables, and inside them [candidate/* | gates/*]: the gate half is reduced
    # while the candidate half is computed
    gru = [n for n in train_names if is_gru(n)]
    assert train_names[len(embeds):len(embeds) + len(gru)] == gru, "flat layout: the GRU kernels must follow the embeddings"
    gru_end = embed_floats + sum(_pad4(int(np.prod(shapes[n]))) for n in gru)
    cand = [n for n in gru if "/candidate/" in n]
    assert gru[:len(cand)] == cand, "flat layout: candidate/* precede gates/*"
    gru_mid = embed_floats + sum(_pad4(int(np.prod(shapes[n]))) for n in cand)
    e, m, g, n = embed_floats, gru_mid, gru_end, n_train
    return dict(train_names=train_names, frozen_names=frozen_names, train_tab=train_tab, n_train=n_train,
                frozen_tab=frozen_tab, n_frozen=n_frozen, embed_floats=e, gru_mid=m, gru_end=g,
                buckets=[(g, n), (0, e), (n, n + 4), (m, g), (e, m)])


class FusionEngine:
    MODEL_TYPE_ID = {"vlmap_answer": 0, "standard": 1, "standard_word2vec": 2, "standard_testmask": 3,
                     "vlmap_answer_vqa_all2": 4, "vlmap_answer_noc": 5, "vlmap_answer_nocarch": 5, "vlmap_answer_vqa_all": 6,
                     "vlmap_answer2": 7, "vlmap_answer_no_noise": 8, "vlmap_answer_adapt": 9, "vlmap_answer_full": 10,
                     "vlmap_answer_ent": 11, "vlmap_finetune": 12, "vlmap_only": 12, "vqa": 13}

    def __init__(self, *, model_type, B, R, D, H, T, W, A, Vq, N_img, params, device="cuda:0",
                 keep_att=0.8, keep_joint=0.5, global_batch=None, deterministic=None, answer_glove=None,
                 fused_gather=False, num_marginal=NUM_MARGINAL, ent_cols=None, map_dim=None, ft_vlmap=False,
                 glove_fixed=None, answers=None):
        """deterministic=True: run-to-run bitwise reproducible steps (the embedding-gradient scatter-add switches
        from float atomics to an atomic-free kernel, ~30 us slower at bs 512).  Per engine: the choice travels in
        vqa_dims_t.flags with every call, no process-wide library state is touched.
        fused_gather=True: no feature-gather pass; v_linear_v's GEMM reads the table rows through image_idx
        (vqa_gemm_f32_gather).  Same step time as the default at bs 512, 151 MB less HBM traffic.
        num_marginal / ent_cols (vlmap_answer_ent): pairings per question, and how many leading head columns the
        regulariser computes (None: from the answer masks at bind_inputs -- 1 + the last known training answer).
        map_dim / ft_vlmap / glove_fixed / answers (model_type 'vqa', vqa/model_vqa.py): MAP_DIM of L2V / V2L; whether those
        two train (config.ft_vlmap); the constant GloVe rows [Vq-3, W] of modules.GloVe_vocab; the candidate answers' token
        sequences {'intseq': i32 [A, La], 'len': i32 [A]} (data_info.hdf5)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.VqaHotError("FusionEngine needs a GPU (no CPU fallback)")
        self.device = torch.device(device)
        self.model_type = model_type
        self.sc = scope_names(model_type)
        self.dims = _lib.Dims(B=B, R=R, D=D, H=H, T=T, W=W, A=A, Vq=Vq, N_img=N_img,
                              model_type=self.MODEL_TYPE_ID[model_type],
                              keep_att=keep_att, keep_joint=keep_joint,
                              inv_global_batch=1.0 / float(global_batch or B),
                              flags=(_lib.FLAG_DETERMINISTIC if deterministic else 0) |
                                    (_lib.FLAG_FUSED_GATHER if fused_gather else 0),
                              num_marginal=int(num_marginal) if model_type == "vlmap_answer_ent" else 0,
                              ent_cols=int(ent_cols or A) if model_type == "vlmap_answer_ent" else 0,
                              extra_weight={"vlmap_answer_ent": W_ENTROPY, "vlmap_answer_full": LATENT_LOSS_WEIGHT}.get(model_type, 0.0))
        self._ent_cols_given = ent_cols is not None
        self.clip_norm = CLIP_NORM       # tf.contrib.layers.optimize_loss(clip_gradients=...): 20.0 (vqa/trainer.py:111), 0.25 in
                                         # vqa/trainer_standard.py:95 -- the caller's Trainer sets it
        self.glove_fixed, self._answers = None, None
        if model_type in LEGACY_FAMILY:
            if glove_fixed is None or answers is None:
                raise ValueError("model_type 'vqa' needs glove_fixed [Vq-3, W] and answers {'intseq' [A, La], 'len' [A]}")
            to_t = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(device=self.device, dtype=dt).contiguous()
            self.glove_fixed = to_t(glove_fixed, torch.float32)
            self._answers = (to_t(answers["intseq"], torch.int32), to_t(answers["len"], torch.int32))
            if tuple(self.glove_fixed.shape) != (Vq - 3, W) or self._answers[0].shape[0] != A or self._answers[1].numel() != A:
                raise ValueError("glove_fixed must be [Vq-3, W], answers['intseq'] [A, La], answers['len'] [A]")
            self.dims.map_dim, self.dims.La = int(map_dim or H), int(self._answers[0].shape[1])
        self.shapes = variable_shapes(model_type, Vq, W, D, H, A, map_dim=map_dim)
        lay = flat_layout(model_type, self.shapes, ft_vlmap=ft_vlmap)
        self.train_names, self.frozen_names = lay["train_names"], lay["frozen_names"]
        self._train_tab, self.n_train = lay["train_tab"], lay["n_train"]
        self._frozen_tab, self.n_frozen = lay["frozen_tab"], lay["n_frozen"]
        self.embed_floats, self.gru_mid, self.gru_end = lay["embed_floats"], lay["gru_mid"], lay["gru_end"]
        self._buckets = lay["buckets"]
        f32 = dict(dtype=torch.float32, device=self.device)
        self.train_flat = torch.zeros(self.n_train, **f32)
        self.frozen_flat = torch.zeros(max(self.n_frozen, 4), **f32)
        self.grad_flat = torch.zeros(self.n_train + 4, **f32)
        self.m_flat = torch.zeros(self.n_train, **f32)
        self.v_flat = torch.zeros(self.n_train, **f32)
        self.norm_sq = torch.zeros(4, **f32)
        self.sumsq_ws = torch.zeros(int(self.lib.vqa_sumsq_workspace_floats(self.n_train)) + 4, **f32)
        self.step_count = 0

        self.params, self.grads = {}, {}
        for n, (off, cnt) in self._train_tab.items():
            self.params[n] = self.train_flat[off:off + cnt].view(self.shapes[n])
            self.grads[n] = self.grad_flat[off:off + cnt].view(self.shapes[n])
        for n, (off, cnt) in self._frozen_tab.items():
            self.params[n] = self.frozen_flat[off:off + cnt].view(self.shapes[n])
        # standard_word2vec: the constant [W, A] GloVe matrix of the answers -- a tf.constant in the reference, so it
        # is neither a variable, nor trained, nor in checkpoints
        self.answer_glove = None
        if model_type == "standard_word2vec":
            if answer_glove is None:
                raise ValueError("standard_word2vec needs answer_glove [W, A]")
            g = torch.as_tensor(np.asarray(answer_glove) if not torch.is_tensor(answer_glove) else answer_glove)
            if tuple(g.shape) != (W, A):
                raise ValueError("answer_glove has shape %s, expected %s" % (tuple(g.shape), (W, A)))
            self.answer_glove = g.to(device=self.device, dtype=torch.float32).contiguous()
        self.load_params(params)

        ws_bytes = int(self.lib.vqa_fusion_workspace_bytes(C.byref(self.dims)))
        if ws_bytes <= 0:
            raise _lib.VqaHotError("vqa_fusion_workspace_bytes rejected the dims")
        self.workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device=self.device)
        self._p_struct = self._make_struct(self.params, all_required=True)
        self._g_struct = self._make_struct(self.grads, all_required=False)
        self._tensor_cache = {}
        self._reset_recurrence_word()
        self._batch_keepalive = None

    # ------------------------------------------------------------------ parameters
    def load_params(self, params, strict=True):
        for n in self.shapes:
            if n not in params:
                if strict:
                    raise KeyError("missing variable %s" % n)
                continue
            src = params[n]
            t = torch.as_tensor(np.asarray(src) if not torch.is_tensor(src) else src)
            if tuple(t.shape) != tuple(self.shapes[n]):
                raise ValueError("variable %s has shape %s, expected %s" % (n, tuple(t.shape), self.shapes[n]))
            self.params[n].copy_(t.to(torch.float32))

    def state_dict(self):
        """Flat name -> CPU tensor archive with the reference's variable names
        (+ Adam slots and global_step), the analogue of tf.train.Saver's checkpoint."""
        out = {n: p.detach().cpu().clone() for n, p in self.params.items()}
        for n, (off, cnt) in self._train_tab.items():
            out[n + "/Adam"] = self.m_flat[off:off + cnt].view(self.shapes[n]).cpu().clone()
            out[n + "/Adam_1"] = self.v_flat[off:off + cnt].view(self.shapes[n]).cpu().clone()
        out["global_step"] = torch.tensor(self.step_count, dtype=torch.int64)
        return out

    def load_state_dict(self, sd, var_names=None):
        names = var_names if var_names is not None else list(self.shapes)
        self.load_params({n: sd[n] for n in names}, strict=False)
        if var_names is None:
            for n, (off, cnt) in self._train_tab.items():
                if n + "/Adam" in sd:
                    self.m_flat[off:off + cnt].copy_(sd[n + "/Adam"].reshape(-1))
                    self.v_flat[off:off + cnt].copy_(sd[n + "/Adam_1"].reshape(-1))
            if "global_step" in sd:
                self.step_count = int(sd["global_step"])

    def _make_struct(self, table, all_required):
        sc = self.sc

        def ptr(name):
            t = table.get(name)
            if t is None:
                if all_required:
                    raise KeyError(name)
                return None
            return t.data_ptr()

        def fc(scope, ln):
            return _lib.Fc(w=ptr(scope + "/fc/weights"), b=ptr(scope + "/fc/biases"),
                           beta=ptr(scope + "/LayerNorm/beta") if ln else None,
                           gamma=ptr(scope + "/LayerNorm/gamma") if ln else None)

        if self.model_type in LEGACY_FAMILY:
            def plain(scope):
                return _lib.Fc(w=ptr(scope + "/fc/weights"), b=ptr(scope + "/fc/biases") if scope + "/fc/biases" in self.shapes else None)
            return _lib.Params(
                glove_fixed=self.glove_fixed.data_ptr() if all_required else None, glove_learn=ptr("GloVe/learn"),
                lstm_k=ptr(LSTM_SCOPE + "/kernel"), lstm_b=ptr(LSTM_SCOPE + "/bias"),
                l2v=(_lib.Fc * 3)(plain("L2V/fc_1"), plain("L2V/fc_2"), plain("L2V/Linear")),
                v2l=(_lib.Fc * 3)(plain("V2L/fc_1"), plain("V2L/fc_2"), plain("V2L/Linear")),
                answer_layer1=plain("reasoning/answer_layer1"), pooled_layer1=plain("reasoning/pooled_layer1"),
                q_layer1=plain("reasoning/q_layer1"), classifier=plain("reasoning/classifier"))
        return _lib.Params(
            embed=ptr(sc["embed"]), v_linear_v=fc(sc["v_linear_v"], True),
            gru_wg=ptr(sc["gru_gates"] + "/kernel"), gru_bg=ptr(sc["gru_gates"] + "/bias"),
            gru_wc=ptr(sc["gru_cand"] + "/kernel"), gru_bc=ptr(sc["gru_cand"] + "/bias"),
            q_linear_v=fc(sc["q_linear_v"], True), score=fc(sc["score"], False),
            pooled_linear_l=fc(sc["pooled_linear_l"], True), q_linear_l=fc(sc["q_linear_l"], True),
            joint_fc=fc(sc["joint_fc"], True), head=fc(sc["head"], False),
            answer_glove=self.answer_glove.data_ptr() if self.answer_glove is not None else None,
            head2=fc(sc["head2"], False) if self.model_type in TWO_HEAD_FAMILY + NOC_FAMILY else _lib.Fc(),
            joint2=fc(sc["joint2"], True) if self.model_type in NOC_FAMILY else _lib.Fc(),
            q_L_ft2=fc(sc["q_L_ft2"], True) if self.model_type == "vlmap_answer2" else _lib.Fc(),
            q_L_mean=fc(sc["q_L_mean"], False) if self.model_type in ("vlmap_answer_no_noise", "vlmap_answer_full") else _lib.Fc(),
            q_L_log_sigma_sq=fc(sc["q_L_log_sigma_sq"], False) if self.model_type == "vlmap_answer_full" else _lib.Fc(),
            v_adapt=fc(sc["v_adapt"], True) if self.model_type == "vlmap_answer_adapt" else _lib.Fc(),
            **({} if self.model_type not in BI_FAMILY else dict(
                embed2=ptr(sc["embed2"]),
                gru_bw_wg=ptr(sc["gru_bw_gates"] + "/kernel"), gru_bw_bg=ptr(sc["gru_bw_gates"] + "/bias"),
                gru_bw_wc=ptr(sc["gru_bw_cand"] + "/kernel"), gru_bw_bc=ptr(sc["gru_bw_cand"] + "/bias"),
                q_att_key=fc(sc["q_att_key"], True), q_att_query=fc(sc["q_att_query"], True),
                word_score=fc(sc["word_score"], False), v_word_fc=fc(sc["v_word_fc"], True))))

    def resize(self, B, T, global_batch=None):
        """Re-target the engine to another batch size / padded question length (the reference pads
        each batch to its own max length, input_ops_vqa_tf_record_memft.py:62-71, and the last batch
        of an epoch is short).  Parameters and optimiser state are untouched; the workspace only
        grows."""
        d = self.dims
        if B == d.B and T == d.T and global_batch is None:
            return
        self.drop_graphs()            # captured step graphs hold this shape's buffers
        d.B, d.T = B, T
        d.inv_global_batch = 1.0 / float(global_batch or B)
        need = int(self.lib.vqa_fusion_workspace_bytes(C.byref(d)))
        if need <= 0:
            raise _lib.VqaHotError("vqa_fusion_workspace_bytes rejected the dims")
        if need > self.workspace.numel():
            self.drop_graphs()
            self.workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)
        self._tensor_cache = {}
        self._reset_recurrence_word()
        for a in ("_keep_att", "_keep_joint", "_keep_joint2", "_keep_tile", "_noise", "_keep_word"):
            if hasattr(self, a):
                delattr(self, a)

    def _grow_workspace(self):
        need = int(self.lib.vqa_fusion_workspace_bytes(C.byref(self.dims)))
        if need <= 0:
            raise _lib.VqaHotError("vqa_fusion_workspace_bytes rejected the dims")
        if need > self.workspace.numel():
            self.drop_graphs()
            self.workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)
        self._tensor_cache = {}
        self._reset_recurrence_word()

    # ------------------------------------------------------------------ workspace views
    def tensor(self, name):
        """Named intermediate (reference mid_result / output key) as a torch view."""
        if name in self._tensor_cache:
            return self._tensor_cache[name]
        off, n = C.c_int64(), C.c_int64()
        _lib.check(self.lib.vqa_fusion_tensor(C.byref(self.dims), name.encode(), C.byref(off), C.byref(n)),
                   "vqa_fusion_tensor(%s)" % name)
        raw = self.workspace[off.value:off.value + 4 * n.value]
        t = raw.view(torch.int32 if name in _INT_TENSORS else torch.float32)
        self._tensor_cache[name] = t
        return t

    # ------------------------------------------------------------------ step pieces
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def bind_inputs(self, *, table, nbox_table, answer_masks):
        """Device-resident feature table [N,R,D] f32, num_boxes i32 [N], float [A] masks."""
        self._table, self._nbox = table, nbox_table
        self._amask = answer_masks
        if self.model_type == "vlmap_answer_ent" and not self._ent_cols_given:
            # the regulariser only looks at answers with train * exist > 0.5 (vqa/model_vlmap_answer_ent.py:63-65) and the
            # train mask is a prefix (:44-46): the head GEMM of the pairings stops after the last such column, rounded up
            # to the GEMM's 32-column granularity
            sel = torch.nonzero((answer_masks["train"] * answer_masks["exist"]) > 0.5)
            last = int(sel.max()) + 1 if sel.numel() else 1
            cols = min(self.dims.A, (last + 31) // 32 * 32)
            if cols != self.dims.ent_cols:
                self.dims.ent_cols = cols
                self._grow_workspace()

    def _batch_struct(self, batch, keep_att, keep_joint, keep_joint2=None, noise=None, keep_tile=None, keep_word=None):
        d = self.dims
        if keep_word is not None:
            assert keep_word.dtype == torch.uint8 and keep_word.numel() == d.B * d.T * d.H
        if self.model_type == "vlmap_answer_full":
            if noise is None:
                raise ValueError("vlmap_answer_full needs the reparameterisation noise [B, H] (make_noise)")
            assert noise.dtype == torch.float32 and noise.numel() == d.B * d.H and noise.is_contiguous()
        if keep_tile is not None:
            assert keep_tile.dtype == torch.uint8 and keep_tile.numel() == d.B * d.num_marginal * 2 * d.H
        assert batch["image_idx"].dtype == torch.int64 and batch["image_idx"].numel() == d.B
        assert batch["q_intseq"].dtype == torch.int32 and tuple(batch["q_intseq"].shape) == (d.B, d.T)
        assert batch["q_intseq_len"].dtype == torch.int32
        assert tuple(batch["answer_target"].shape) == (d.B, d.A)
        if keep_att is not None:
            assert keep_att.dtype == torch.uint8 and keep_att.numel() == d.B * d.R * d.H
        if keep_joint is not None:
            assert keep_joint.dtype == torch.uint8 and keep_joint.numel() == d.B * 2 * d.H
        live = batch.get("live_rows")
        if live is not None:      # host int32[T]: rows sorted by length, longest first (input_ops_vqa.sort_by_length)
            live = np.ascontiguousarray(live, dtype=np.int32)
            assert live.shape == (d.T,) and (np.diff(live) <= 0).all() and 0 <= live[-1] and live[0] <= d.B
        if keep_joint2 is not None:
            assert keep_joint2.dtype == torch.uint8 and keep_joint2.numel() == d.B * 2 * d.H
        self._batch_keepalive = (batch, keep_att, keep_joint, keep_joint2, live, noise, keep_tile, keep_word)
        am = self._amask
        return _lib.Batch(
            table=self._table.data_ptr(), nbox_table=self._nbox.data_ptr(),
            image_idx=batch["image_idx"].data_ptr(), q_intseq=batch["q_intseq"].data_ptr(),
            q_intseq_len=batch["q_intseq_len"].data_ptr(), answer_target=batch["answer_target"].data_ptr(),
            train_mask=am["train"].data_ptr(), obj_mask=am["obj"].data_ptr(), attr_mask=am["attr"].data_ptr(),
            exist_mask=am["exist"].data_ptr(),
            keep_att=keep_att.data_ptr() if keep_att is not None else None,
            keep_joint=keep_joint.data_ptr() if keep_joint is not None else None,
            keep_joint2=keep_joint2.data_ptr() if keep_joint2 is not None else None,
            live_rows=live.ctypes.data if live is not None else None,
            noise=noise.data_ptr() if noise is not None else None,
            keep_tile=keep_tile.data_ptr() if keep_tile is not None else None,
            keep_word=keep_word.data_ptr() if keep_word is not None else None,
            answer_intseq=self._answers[0].data_ptr() if self._answers is not None else None,
            answer_intseq_len=self._answers[1].data_ptr() if self._answers is not None else None)

    def forward(self, batch, keep_att=None, keep_joint=None, want_dz=True, keep_joint2=None, noise=None, keep_tile=None,
                keep_word=None):
        """keep_joint2: vlmap_answer_noc only -- the keep-mask of l_joint (keep_joint is v_joint's);
        noise: vlmap_answer_full only -- standard-normal draws [B, H] of the reparameterisation (make_noise);
        keep_tile: vlmap_answer_ent only -- keep-mask [B, num_marginal, 2H] of the pairings' dropout (make_keep_mask_tile);
        keep_word: vlmap_finetune / vlmap_only -- keep-mask [B, T, H] of the word attention's dropout (make_keep_mask_word)"""
        self._bs = self._batch_struct(batch, keep_att, keep_joint, keep_joint2, noise, keep_tile, keep_word)
        _lib.check(self.lib.vqa_fusion_forward(C.byref(self.dims), C.byref(self._p_struct), C.byref(self._bs),
                                               C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(),
                                               1 if want_dz else 0, self._stream()), "vqa_fusion_forward")

    def _backward_phases(self, phases):
        tail = self.grad_flat[self.n_train:]
        _lib.check(self.lib.vqa_fusion_backward_phases(
            C.byref(self.dims), C.byref(self._p_struct), C.byref(self._g_struct), C.byref(self._bs),
            C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(), C.c_void_p(tail.data_ptr()), phases,
            self._stream()), "vqa_fusion_backward_phases")

    def backward(self, reducer=None):
        """All gradients into grad_flat.  With a bucketed `reducer` (dp.BucketedAllReduce) the three
        dependency-ordered phases are launched one by one and each finished bucket's all-reduce is started
        right away, so it overlaps the next phase: [everything but GRU/embedding] during BPTT, the
        embedding bucket (+ slice sum of squares) and the GRU gate bucket during the GRU weight-gradient GEMMs;
        only the GRU candidate bucket's reduction (5.4 MB) is exposed."""
        # only the embedding gradient is scatter-added; everything else is overwritten
        self.grad_flat[:self.embed_floats].zero_()
        if reducer is None:
            self._backward_phases(15)
            return
        gf = self.grad_flat
        rest, emb, tail, gates, cand = (gf[lo:hi] for lo, hi in self._buckets)       # flat_layout: disjoint, cover grad_flat
        self._backward_phases(1)
        reducer.start(rest)
        self._backward_phases(2)
        reducer.start(emb)
        reducer.start(tail)
        self._backward_phases(4)
        reducer.start(gates)                      # GRU gates (10.8 MB) reduce under the candidate GEMMs
        self._backward_phases(8)
        reducer.start(cand)                       # GRU candidate (5.4 MB): the only exposed reduction
        reducer.finish()

    def optimizer_step(self, lr):
        """clip_by_global_norm(20) + Adam on the flat buffers.  The norm uses the
        dense gradients of every non-embedding train var plus the un-aggregated
        embedding slices (tail slot), as tf.clip_by_global_norm does for IndexedSlices."""
        e = self.embed_floats
        dense = self.grad_flat[e:self.n_train]
        tail = self.grad_flat[self.n_train:]
        _lib.check(self.lib.vqa_sumsq(C.c_void_p(dense.data_ptr()), dense.numel(), C.c_void_p(tail.data_ptr()),
                                      C.c_void_p(self.norm_sq.data_ptr()), C.c_void_p(self.sumsq_ws.data_ptr()),
                                      self.sumsq_ws.numel(), self._stream()), "vqa_sumsq")
        self.step_count += 1
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        _lib.check(self.lib.vqa_clip_adam(C.c_void_p(self.train_flat.data_ptr()), C.c_void_p(self.grad_flat.data_ptr()),
                                          C.c_void_p(self.m_flat.data_ptr()), C.c_void_p(self.v_flat.data_ptr()),
                                          self.n_train, C.c_void_p(self.norm_sq.data_ptr()), float(self.clip_norm), lr_t,
                                          ADAM_B1, ADAM_B2, ADAM_EPS, self._stream()), "vqa_clip_adam")

    def train_step(self, batch, keep_att, keep_joint, lr, allreduce=None, keep_joint2=None, noise=None, keep_tile=None,
                   keep_word=None):
        self.forward(batch, keep_att, keep_joint, want_dz=True, keep_joint2=keep_joint2, noise=noise, keep_tile=keep_tile,
                     keep_word=keep_word)
        if allreduce is not None and hasattr(allreduce, "start"):
            self.backward(reducer=allreduce)          # bucketed, overlapped with the backward phases
        else:
            self.backward()
            if allreduce is not None:
                allreduce(self.grad_flat)
        self.optimizer_step(lr)

    # ------------------------------------------------------------------ whole step as one hipGraph replay
    def train_step_graph(self, batch, lr, seed, step):
        """One train step = refresh the step's inputs in place + ONE vqa_graph_launch.
        The first call for a shape (B, T, live_rows) captures forward -> backward -> device-side Adam rate -> norm ->
        clip + Adam on a private stream into an executable graph (csrc/graph.hip); later calls copy the batch into the
        graph's static input buffers, regenerate the dropout masks (and a variant's noise / pairing mask) for
        (seed, step) in place, and replay.  Same kernels, same arithmetic as train_step (bitwise, with
        deterministic=True); single process only (a data-parallel step keeps the eager path and its bucketed reducer).
        The Adam step count lives on the device (vqa_adam_lr_step) and is kept equal to self.step_count.
        EXPERIMENTAL and not the default anywhere: on this ROCm release replay is SLOWER than eager launches (3.96 against
        3.54 ms per bs-512 step, profiles/r4_graph_bench.txt), and mixing eager train_step calls with replays on ONE engine
        gave run-to-run differences in two of five model types (tools/dbg/graph_dbg.py) -- use an engine either way."""
        d = self.dims
        live = batch.get("live_rows")
        live_key = None if live is None else tuple(int(x) for x in np.asarray(live))
        key = (d.B, d.T, live_key)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
            self._g_stream = torch.cuda.Stream(device=self.device)
            self._g_step = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._g_lr = torch.zeros(1, dtype=torch.float64, device=self.device)
            self._g_lr_t = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._g_step_value, self._g_lr_value = None, None
        g = self._graphs.get(key)
        cur = torch.cuda.current_stream(self.device)
        if g is None:
            static = {k: torch.empty_like(batch[k]) for k in ("image_idx", "q_intseq", "q_intseq_len", "answer_target")}
            if live is not None:
                static["live_rows"] = np.ascontiguousarray(live, dtype=np.int32)
            g = {"static": static, "exec": None}
        st = g["static"]
        self._g_stream.wait_stream(cur)
        with torch.cuda.stream(self._g_stream):
            for k in ("image_idx", "q_intseq", "q_intseq_len", "answer_target"):
                st[k].copy_(batch[k], non_blocking=True)
            if self._g_step_value != self.step_count:
                self._g_step.fill_(self.step_count)
            if self._g_lr_value != lr:
                self._g_lr.fill_(lr)
                self._g_lr_value = lr
            ka, kj = self.make_keep_masks(seed, step)
            extra = {}
            if self.model_type in NOC_FAMILY:
                extra["keep_joint2"] = self.make_keep_mask_joint2(seed, step)
            if self.model_type == "vlmap_answer_full":
                extra["noise"] = self.make_noise(seed, step)
            if self.model_type == "vlmap_answer_ent":
                extra["keep_tile"] = self.make_keep_mask_tile(seed, step)
            if self.model_type in BI_FAMILY:
                extra["keep_word"] = self.make_keep_mask_word(seed, step)
            if g["exec"] is None:
                sp = C.c_void_p(self._g_stream.cuda_stream)
                _lib.check(self.lib.vqa_graph_capture_begin(sp), "vqa_graph_capture_begin")
                try:
                    self.forward(st, ka, kj, want_dz=True, **extra)
                    self.backward()
                    e = self.embed_floats
                    dense, tail = self.grad_flat[e:self.n_train], self.grad_flat[self.n_train:]
                    _lib.check(self.lib.vqa_sumsq(C.c_void_p(dense.data_ptr()), dense.numel(), C.c_void_p(tail.data_ptr()),
                                                  C.c_void_p(self.norm_sq.data_ptr()), C.c_void_p(self.sumsq_ws.data_ptr()),
                                                  self.sumsq_ws.numel(), sp), "vqa_sumsq")
                    _lib.check(self.lib.vqa_adam_lr_step(C.c_void_p(self._g_step.data_ptr()), C.c_void_p(self._g_lr.data_ptr()),
                                                         ADAM_B1, ADAM_B2, C.c_void_p(self._g_lr_t.data_ptr()), sp),
                               "vqa_adam_lr_step")
                    _lib.check(self.lib.vqa_clip_adam_dev(
                        C.c_void_p(self.train_flat.data_ptr()), C.c_void_p(self.grad_flat.data_ptr()),
                        C.c_void_p(self.m_flat.data_ptr()), C.c_void_p(self.v_flat.data_ptr()), self.n_train,
                        C.c_void_p(self.norm_sq.data_ptr()), float(self.clip_norm), C.c_void_p(self._g_lr_t.data_ptr()), ADAM_B1, ADAM_B2,
                        ADAM_EPS, sp), "vqa_clip_adam_dev")
                except Exception:
                    self.lib.vqa_graph_capture_abort(sp)
                    raise
                ex, n_nodes = C.c_void_p(), C.c_int()
                _lib.check(self.lib.vqa_graph_capture_end(sp, C.byref(ex), C.byref(n_nodes)), "vqa_graph_capture_end")
                g["exec"], g["nodes"], g["keep"] = ex, int(n_nodes.value), (ka, kj, extra, self._bs, self._batch_keepalive)
                self._graphs[key] = g
            _lib.check(self.lib.vqa_graph_launch(g["exec"], C.c_void_p(self._g_stream.cuda_stream)), "vqa_graph_launch")
        cur.wait_stream(self._g_stream)
        self.step_count += 1
        self._g_step_value = self.step_count
        return g["nodes"]

    def drop_graphs(self):
        """destroys the captured step graphs (they hold the workspace's addresses: call before the workspace moves)"""
        for g in getattr(self, "_graphs", {}).values():
            if g.get("exec") is not None:
                self.lib.vqa_graph_destroy(g["exec"])
        self._graphs = {}

    # ------------------------------------------------------------------ results
    def _reset_recurrence_word(self):
        """(re)locates the sticky error word of the weight-stationary GRU launches in the CURRENT workspace layout and
        zeroes it: a resize moves the buffer over bytes that held something else"""
        self._ws_err_word = False
        if getattr(self, "workspace", None) is None or not hasattr(self, "dims"):
            self._ws_err_word = None
            return
        off, n = C.c_int64(), C.c_int64()
        if self.lib.vqa_fusion_tensor(C.byref(self.dims), b"gru_ws", C.byref(off), C.byref(n)) == 0:
            self._ws_err_word = self.workspace[off.value + 4 * 1023: off.value + 4 * 1024].view(torch.int32)
            self._ws_err_word.zero_()

    def check_recurrence(self):
        """The weight-stationary GRU launches (csrc/gru_ws.hip) need all their 256 workgroups resident at once; if something
        else keeps CUs from them for seconds (another process computing on the same GPU), their bounded waits give up, set an
        error word in the workspace and the step's numbers are garbage.  Called wherever results are fetched: raises instead
        of returning them."""
        if getattr(self, "_ws_err_word", None) is None:
            self._reset_recurrence_word()
        if self._ws_err_word is None:
            return
        if self._ws_err_word is not False and int(self._ws_err_word.item()) != 0:
            raise _lib.VqaHotError("the persistent GRU recurrence timed out waiting for its workgroups (is another process "
                                   "computing on this GPU?): results of this step are invalid; VQA_HOT_GRU_WS=0 selects the "
                                   "per-step kernels")

    def report(self, global_rows=None, group=None):
        """The 13 report scalars of the reference (vqa/model_vlmap_answer.py:275-288) for the batch this engine
        ran.  Under data parallelism pass global_rows (= sum of the shard sizes): the per-sample statistics are
        summed over the shard, SUM-all-reduced, and the means / guarded ratios are taken over the GLOBAL batch by
        the same kernel -- every rank then reports what one process on the whole batch would."""
        import torch.distributed as dist
        self.check_recurrence()
        if global_rows is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
            r = self.tensor("report")[:13].cpu().numpy()
        else:
            d = self.dims
            s = self.tensor("stats").view(d.B, 16).sum(0)
            if dist.get_backend(group) == "gloo":
                h = s.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                s = h.to(self.device)
            else:
                dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
            mean = (s / float(global_rows)).contiguous()
            out = torch.zeros(16, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.vqa_report_reduce(C.c_void_p(mean.data_ptr()), 1, C.c_void_p(out.data_ptr()),
                                                  self._stream()), "vqa_report_reduce")
            r = out[:13].cpu().numpy()
        return {self.lib.vqa_report_key(i).decode(): float(r[i]) for i in range(13)}

    EXTRA_REPORT_KEYS = {"vlmap_answer_full": ("latent_loss", "train_latent_loss"),       # vqa/model_vlmap_answer_full.py:222-223
                         "vlmap_answer_ent": ("entropy", "weighted_entropy")}               # vqa/model_vlmap_answer_ent.py:293-294

    def extra_report(self, global_rows=None, group=None):
        """vlmap_answer_full / _ent: the two report scalars behind the 13 (report[13], report[14]) and the model's total
        loss (report[15] = answer_train_loss + the weighted term); {} for every other model.  Data parallel: the
        per-sample terms travel in stats[:, 15] and are reduced like the other statistics."""
        keys = self.EXTRA_REPORT_KEYS.get(self.model_type)
        if keys is None:
            return {}
        import torch.distributed as dist
        if global_rows is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
            r = self.tensor("report")[13:16].cpu().numpy()
            return {keys[0]: float(r[0]), keys[1]: float(r[1]), "total_loss": float(r[2])}
        d = self.dims
        s = self.tensor("stats").view(d.B, 16)[:, [0, 15]].sum(0)
        if dist.get_backend(group) == "gloo":
            h = s.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            s = h
        else:
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        a, e = (float(x) / float(global_rows) for x in s.cpu())
        return {keys[0]: e, keys[1]: d.extra_weight * e, "total_loss": a + d.extra_weight * e}

    def loss(self):
        """the scalar optimize_loss minimises: sum of the model's `losses` (vqa/model_vlmap_answer.py:304-306)"""
        return self.tensor("report")[15 if self.model_type in self.EXTRA_REPORT_KEYS else 0]

    def make_keep_masks(self, seed, step, row_offset=0, global_rows=None):
        """Reproducible dropout keep-masks for (seed, step) -- the explicit stand-in for
        tf.nn.dropout's internal RNG (vlmap/modules.py:82, model_vlmap_answer.py:180).
        The mask stream is indexed by the GLOBAL batch row: a data-parallel shard passes its first global row
        (row_offset) and the global batch size, and draws exactly the bits one process running the whole batch
        would draw for those rows -- ranks never reuse each other's bits."""
        d = self.dims
        Bg = int(global_rows) if global_rows is not None else d.B
        n_att, n_j = d.B * d.R * d.H, d.B * 2 * d.H
        if not hasattr(self, "_keep_att"):
            self._keep_att = torch.empty(n_att, dtype=torch.uint8, device=self.device)
            self._keep_joint = torch.empty(n_j, dtype=torch.uint8, device=self.device)
        off = step * (Bg * d.R * d.H + Bg * 2 * d.H)
        _lib.check(self.lib.vqa_dropout_mask(C.c_void_p(self._keep_att.data_ptr()), n_att, seed,
                                             off + row_offset * d.R * d.H, d.keep_att, self._stream()),
                   "vqa_dropout_mask")
        _lib.check(self.lib.vqa_dropout_mask(C.c_void_p(self._keep_joint.data_ptr()), n_j, seed,
                                             off + Bg * d.R * d.H + row_offset * 2 * d.H, d.keep_joint,
                                             self._stream()), "vqa_dropout_mask")
        return self._keep_att, self._keep_joint

    def make_keep_mask_word(self, seed, step, row_offset=0, global_rows=None):
        """vlmap_finetune / vlmap_only: keep-mask [B, T, H] of the question self-attention's tf.nn.dropout(., 0.8)
        (modules.hadamard_attention under scope word_attention), its own region of the (seed, step) stream"""
        d = self.dims
        Bg = int(global_rows) if global_rows is not None else d.B
        per_row = d.T * d.H
        n = d.B * per_row
        if not hasattr(self, "_keep_word"):
            self._keep_word = torch.empty(n, dtype=torch.uint8, device=self.device)
        off = (3 << 40) + step * (Bg * per_row) + row_offset * per_row
        _lib.check(self.lib.vqa_dropout_mask(C.c_void_p(self._keep_word.data_ptr()), n, seed, off, d.keep_att,
                                             self._stream()), "vqa_dropout_mask")
        return self._keep_word

    def make_keep_mask_tile(self, seed, step, row_offset=0, global_rows=None):
        """vlmap_answer_ent: keep-mask [B, num_marginal, 2H] of tf.nn.dropout(tile_joint, 0.5)
        (vqa/model_vlmap_answer_ent.py:205), its own region of the (seed, step) stream, indexed by the global batch row"""
        d = self.dims
        Bg = int(global_rows) if global_rows is not None else d.B
        per_row = d.num_marginal * 2 * d.H
        n = d.B * per_row
        if not hasattr(self, "_keep_tile"):
            self._keep_tile = torch.empty(n, dtype=torch.uint8, device=self.device)
        off = (2 << 40) + step * (Bg * per_row) + row_offset * per_row
        _lib.check(self.lib.vqa_dropout_mask(C.c_void_p(self._keep_tile.data_ptr()), n, seed, off, d.keep_joint,
                                             self._stream()), "vqa_dropout_mask")
        return self._keep_tile

    def make_noise(self, seed, step, row_offset=0, global_rows=None):
        """vlmap_answer_full: the standard-normal draws [B, H] that tf.random_normal(seed=123) produces inside the
        reference graph (vqa/model_vlmap_answer_full.py:133), as an explicit reproducible input keyed by (seed, step) and
        the global batch row"""
        d = self.dims
        Bg = int(global_rows) if global_rows is not None else d.B
        n = d.B * d.H
        if not hasattr(self, "_noise"):
            self._noise = torch.empty(n, dtype=torch.float32, device=self.device)
        off = step * (Bg * d.H) + row_offset * d.H
        _lib.check(self.lib.vqa_normal_noise(C.c_void_p(self._noise.data_ptr()), n, seed, off, self._stream()),
                   "vqa_normal_noise")
        return self._noise

    def make_keep_mask_joint2(self, seed, step, row_offset=0, global_rows=None):
        """vlmap_answer_noc: the second dropout site's keep-mask (l_joint), from its own region of the same stream"""
        d = self.dims
        Bg = int(global_rows) if global_rows is not None else d.B
        n_j = d.B * 2 * d.H
        if not hasattr(self, "_keep_joint2"):
            self._keep_joint2 = torch.empty(n_j, dtype=torch.uint8, device=self.device)
        off = (1 << 40) + step * (Bg * 2 * d.H) + row_offset * 2 * d.H      # far beyond the two masks of make_keep_masks
        _lib.check(self.lib.vqa_dropout_mask(C.c_void_p(self._keep_joint2.data_ptr()), n_j, seed, off, d.keep_joint,
                                             self._stream()), "vqa_dropout_mask")
        return self._keep_joint2
