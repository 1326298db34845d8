"""CPU oracle for the VQA fusion-model hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for
this path and its arithmetic lives in
tensorflow-gpu==1.6.0 (requirements.txt:1), which is not installable here.
This file is therefore a NumPy restatement of the reference's graph code plus
the published TF-1.6 op semantics it calls; it is pinned only by (i) the
known-answer tests in tests/test_oracle_known_answers.py, (ii) an independent
torch-autograd restatement (oracle/torch_ref.py) and (iii) central finite
differences in float64.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (vqa-transfer-externaldata_amd/) never does.

Reference lines restated (paths relative to /root/reference):
  * feature gather            vqa/model_vlmap_answer.py:110-123
  * fc_layer (FC -> LN -> act) vlmap/modules.py:630-650
  * embedding lookup          vqa/model_vlmap_answer.py:134, vlmap/modules.py:415-448
  * encode_L (GRU)            vlmap/modules.py:124-140
  * hadamard_attention        vlmap/modules.py:67-97
  * attention_pooling         vlmap/modules.py:23-39
  * fusion MLP + head         vqa/model_vlmap_answer.py:163-187 (WordWeightAnswer
                              vlmap/modules.py:589-627); vqa/model_standard.py:251-275
  * loss + report             vqa/model_vlmap_answer.py:192-288; vqa/model_standard.py:281-374
  * word2vec head variant     vqa/model_standard_word2vec.py:180-202 (classifier FC 2048 -> 300, logits =
                              joint2 x a fixed [300, A] GloVe matrix of the answers; train loss masked by the
                              train-answer mask like vlmap_answer)
  * optimiser                 vqa/trainer.py:87-114 (optimize_loss: global-norm
                              clip 20.0 then Adam)
TF semantics followed: SURVEY.md section 5.2 items 1-9.
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-12          # tf.contrib.layers.layer_norm -> nn.batch_normalization(variance_epsilon=1e-12)
KEEP_ATT = 0.8          # vlmap/modules.py:82
KEEP_JOINT = 0.5        # vqa/model_vlmap_answer.py:180
CLIP_NORM = 20.0        # vqa/trainer.py:111
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults

# Variable names = the checkpoint / transfer contract.
FC_LN_SCOPES_VLMAP = ["v_linear_v", "q_linear_v", "pooled_linear_l", "q_linear_l", "joint_fc"]
FROZEN_TOP_SCOPES_VLMAP = ("q_linear_l", "pooled_linear_l", "joint_fc", "WordWeightAnswer")
TRANSFER_TOP_SCOPES_VLMAP = ("q_linear_l", "pooled_linear_l", "joint_fc")

OUTPUT_GLOVE = "reasoning/output_glove:const"
# model_standard and its variants share one architecture (every variable trainable, fusion MLP under 'reasoning/',
# plain 'classifier' head); standard_testmask (vqa/model_standard_testmask.py) is model_standard with the training
# loss masked by the train-answer mask (:266-268) and an older, shorter report (:295-304)
STANDARD_FAMILY = ("standard", "standard_word2vec", "standard_testmask")
# The five older ablations of model_vlmap_answer (constructors without `image_features`, so only
# vqa/trainer.py:84 builds them).  Each is model_vlmap_answer with ONE change:
#   vlmap_answer2        vqa/model_vlmap_answer2.py:127-131,164  q_L_ft2 = fc_layer(q_L_ft, LN, tanh) feeds q_linear_l
#                        (heavy_output['condition'] = q_L_ft2); q_linear_v still reads q_L_ft
#   vlmap_answer_no_noise  vqa/model_vlmap_answer_no_noise.py:122-125,157  q_L_mean = linear fc_layer(q_L_ft) feeds q_linear_l
#   vlmap_answer_full    vqa/model_vlmap_answer_full.py:124-134,166,217-223,272-276  VAE reparameterisation:
#                        q_linear_l reads q_L_mean + noise * sqrt(exp(q_L_log_sigma_sq)), noise = random_normal(seed=123)
#                        (an explicit input here, like the dropout masks); loss += 0.1 * KL latent loss
#   vlmap_answer_adapt   vqa/model_vlmap_answer_adapt.py:132-142  v_adapt = fc_layer(V_ft, LN over the [R, H] block, ReLU)
#                        is what the attention pools (pooled_V_ft is [B, H], pooled_linear_l maps H -> H)
#   vlmap_answer_ent     vqa/model_vlmap_answer_ent.py:16,191-211,281-292  marginal-entropy regulariser: joint_fc + head
#                        on NUM_MARGINAL pairings of every question with (stop-gradient) pooled features of the batch
ABLATION_FAMILY = ("vlmap_answer2", "vlmap_answer_no_noise", "vlmap_answer_full", "vlmap_answer_adapt", "vlmap_answer_ent")
LATENT_LOSS_WEIGHT = 0.1   # vqa/model_vlmap_answer_full.py:33
W_ENTROPY = 0.1            # vqa/model_vlmap_answer_ent.py:14
NUM_MARGINAL = 200         # vqa/model_vlmap_answer_ent.py:16
TRAIN_MASKED_LOSS = ("vlmap_answer", "standard_word2vec", "standard_testmask", "vlmap_answer_noc",
                     "vlmap_answer_nocarch") + ABLATION_FAMILY
# the three oldest of them (_no_noise, _adapt, _full) carry the 9-key report of model_standard_testmask
# (vqa/model_vlmap_answer_no_noise.py:210-219, _adapt.py:211-220, _full.py:220-232); _full adds its latent terms
OLD_REPORT_TYPES = ("vlmap_answer_no_noise", "vlmap_answer_adapt", "vlmap_answer_full")
# vqa/model_vlmap_answer_vqa_all2.py: model_vlmap_answer (fixed transferred fusion MLP + WordWeightAnswer head, :128-196)
# plus a TRAINABLE second head `TunedWordWeightAnswer` on the same `joint` (:216-220); logits are summed (:226-227), the
# loss is untuned * train_mask + tuned (:240-241) and the prediction takes the tuned logit on training answers and the
# fixed one on test answers (:243-244).  `tuned_q_linear_l` / `tuned_joint_fc` (:202-214) are built but feed nothing:
# the tuned head reads `joint`, not `tuned_joint` (:216-217) -- reproduced: variables that never receive a gradient.
# vqa/model_vlmap_answer_noc.py (= model_vlmap_answer_nocarch.py, the two files are identical): "no composition" --
# instead of joint_fc(pooled_linear_l * l_linear_l) two separate branches, joint_v(pooled_linear_l) and joint_l(l_linear_l),
# each FC + LN + ReLU + dropout 0.5 (:177-188) with its own transferred head WordWeightAnswerV / WordWeightAnswerL
# (weights.hdf5 datasets v_class_* / l_class_*, :190-202); logit = v_logit + l_logit (:204); loss / report as in
# model_vlmap_answer; frozen: q_linear_l, pooled_linear_l, joint_v, joint_l and both heads (:80-90), transferred: the four
# layers (:92-103)
NOC_FAMILY = ("vlmap_answer_noc", "vlmap_answer_nocarch")
FROZEN_TOP_SCOPES_NOC = ("q_linear_l", "pooled_linear_l", "joint_v", "joint_l", "WordWeightAnswerV", "WordWeightAnswerL")
TRANSFER_TOP_SCOPES_NOC = ("q_linear_l", "pooled_linear_l", "joint_v", "joint_l")
# vqa/model_vlmap_answer_vqa_all.py = _vqa_all2 except (diff of the two files): the fixed logits of answers the word-weight
# directory does not know are replaced by the row minimum of the fixed logits (:192-194); the tuned loss is taken on the
# SUM, ce(logit + tuned_logit) (:236-237); the training loss masks both terms, sum_a (untuned + tuned) * train_mask
# (:241-242); pred = argmax(logit + tuned_logit) (:244)
VLMAP_FAMILY = ("vlmap_answer", "vlmap_answer_vqa_all", "vlmap_answer_vqa_all2") + NOC_FAMILY + ABLATION_FAMILY
TWO_HEAD_FAMILY = ("vlmap_answer_vqa_all", "vlmap_answer_vqa_all2")
# report keys of vqa/model_standard_testmask.py:295-304 in terms of the 13 keys of the current models
TESTMASK_REPORT = {"answer_train_loss": "answer_train_loss", "answer_report_loss": "answer_report_loss",
                   "answer_accuracy": "answer_acc", "exist_answer_accuracy": "exist_acc",
                   "test_answer_accuracy": "test_acc", "normal_test_answer_accuracy": "normal_test_acc",
                   "max_exist_answer_accuracy": "max_exist_acc", "test_max_answer_accuracy": "test_max_acc",
                   "test_max_exist_answer_accuracy": "test_max_exist_acc"}


def testmask_report(report):
    """the 9 report scalars of vqa/model_standard_testmask.py:295-304 from the 13-key report"""
    return {k: report[v] for k, v in TESTMASK_REPORT.items()}



def is_const(name):
    return name.endswith(":const")


REPORT_KEYS = [
    "answer_train_loss", "answer_report_loss", "answer_acc", "exist_acc", "test_acc",
    "normal_test_acc", "normal_test_object_acc", "normal_test_attribute_acc",
    "normal_exist_acc", "normal_train_exist_acc", "max_exist_acc", "test_max_acc",
    "test_max_exist_acc",
]


def scope_names(model_type: str) -> dict:
    """Maps logical layer -> TF variable scope for the two models.

    vqa/model_vlmap_answer.py:126-185 vs vqa/model_standard.py:223-275 (the
    latter nests the fusion MLP under 'reasoning/' and uses a plain
    'classifier' fc_layer as head).
    """
    if model_type in VLMAP_FAMILY:
        pre = ""
        head = "WordWeightAnswer"
    elif model_type in STANDARD_FAMILY:
        pre = "reasoning/"
        head = "reasoning/classifier"
    else:
        raise ValueError("unknown model_type %r" % (model_type,))
    return {
        # standard_word2vec only: the constant [W, A] answer-GloVe matrix (a tf.constant, not a variable:
        # vqa/model_standard_word2vec.py:185-188); kept beside the variables under a name no filter selects
        "glove": OUTPUT_GLOVE,
        "embed": "LearnGloVe/embed_map",
        "v_linear_v": "v_linear_v",
        "gru_gates": "encode_L/rnn/gru_cell/gates",
        "gru_cand": "encode_L/rnn/gru_cell/candidate",
        "q_linear_v": "q_linear_v",
        "score": "hadamard_attention/compute/score",
        "pooled_linear_l": pre + "pooled_linear_l",
        "q_linear_l": pre + "q_linear_l",
        "joint_fc": pre + "joint_fc",
        "head": head,
        # vlmap_answer_vqa_all2 only
        "head2": "TunedWordWeightAnswer", "tuned_q_linear_l": "tuned_q_linear_l", "tuned_joint_fc": "tuned_joint_fc",
        # vlmap_answer_noc / nocarch only
        "joint_v": "joint_v", "joint_l": "joint_l", "headV": "WordWeightAnswerV", "headL": "WordWeightAnswerL",
        # the ablations (ABLATION_FAMILY): answer2 / no_noise + full / full / adapt
        "q_L_ft2": "q_L_ft2", "q_L_mean": "q_L_mean", "q_L_log_sigma_sq": "q_L_log_sigma_sq", "v_adapt": "v_adapt",
    }


def train_var_names(params: dict, model_type: str) -> list:
    """filter_train_vars: vqa/model_vlmap_answer.py:81-89, vqa/model_standard.py:80-84."""
    names = sorted(n for n in params.keys() if not is_const(n))
    if model_type in STANDARD_FAMILY:
        return names
    frozen = FROZEN_TOP_SCOPES_NOC if model_type in NOC_FAMILY else FROZEN_TOP_SCOPES_VLMAP
    return [n for n in names if n.split("/")[0] not in frozen]


def transfer_var_names(params: dict, model_type: str) -> list:
    """filter_transfer_vars: vqa/model_vlmap_answer.py:91-100 (standard: :86-93)."""
    names = sorted(n for n in params.keys() if not is_const(n))
    if model_type in STANDARD_FAMILY:
        return [n for n in names if n.split("/")[0] in ("encode_L", "GloVe")]
    keep = TRANSFER_TOP_SCOPES_NOC if model_type in NOC_FAMILY else TRANSFER_TOP_SCOPES_VLMAP
    return [n for n in names if n.split("/")[0] in keep]


# ----------------------------------------------------------------------------
# parameter initialisation
# ----------------------------------------------------------------------------
def init_params(rng, model_type="vlmap_answer", Vq=64, W=300, D=2048, H=1024, A=3000,
                dtype=np.float32, head="random"):
    """Random-init weights of the reference architecture.

    layers.fully_connected: Xavier-uniform weights, zero bias; layer_norm: beta 0,
    gamma 1; GRUCell: gates bias 1.0, candidate bias 0 (tf.contrib.rnn.GRUCell);
    embeddings small uniform (GloVe files are not available offline).
    head='random' Xavier head; head='untrained' reproduces WordWeightAnswer with
    word_weight_dir=None (weights 0, bias -100: vlmap/modules.py:601-602).
    """
    sc = scope_names(model_type)
    p = {}

    def xavier(fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)

    def fc(scope, fin, fout, ln):
        p[scope + "/fc/weights"] = xavier(fin, fout)
        p[scope + "/fc/biases"] = np.zeros(fout, dtype)
        if ln:
            p[scope + "/LayerNorm/beta"] = np.zeros(fout, dtype)
            p[scope + "/LayerNorm/gamma"] = np.ones(fout, dtype)

    p[sc["embed"]] = rng.uniform(-0.01, 0.01, size=(Vq, W)).astype(dtype)
    fc(sc["v_linear_v"], D, H, True)
    p[sc["gru_gates"] + "/kernel"] = xavier(W + H, 2 * H)
    p[sc["gru_gates"] + "/bias"] = np.ones(2 * H, dtype)
    p[sc["gru_cand"] + "/kernel"] = xavier(W + H, H)
    p[sc["gru_cand"] + "/bias"] = np.zeros(H, dtype)
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
    if model_type in NOC_FAMILY:
        fc(sc["joint_v"], H, 2 * H, True)
        fc(sc["joint_l"], H, 2 * H, True)
        for hd in (sc["headV"], sc["headL"]):
            if head == "untrained":
                p[hd + "/fc/weights"] = np.zeros((2 * H, A), dtype)
                p[hd + "/fc/biases"] = np.full(A, -100.0, dtype)
            else:
                fc(hd, 2 * H, A, False)
        return p
    fc(sc["joint_fc"], H, 2 * H, True)
    if model_type in TWO_HEAD_FAMILY:
        fc(sc["tuned_q_linear_l"], H, H, True)
        fc(sc["tuned_joint_fc"], H, 2 * H, True)
        fc(sc["head2"], 2 * H, A, False)                    # fc_layer(joint, num_answer, use_bias=True), Xavier / zero
    if model_type == "standard_word2vec":
        fc(sc["head"], 2 * H, W, False)                     # 'classifier' FC to the 300-d word space
        p[sc["glove"]] = (0.3 * rng.standard_normal((W, A))).astype(dtype)   # stands in for the answers' GloVe vectors
    elif head == "untrained":
        p[sc["head"] + "/fc/weights"] = np.zeros((2 * H, A), dtype)
        p[sc["head"] + "/fc/biases"] = np.full(A, -100.0, dtype)
    else:
        fc(sc["head"], 2 * H, A, False)
    return p


def perturb_ln_params(params, rng, scale=0.1):
    """Make LN beta/gamma and biases non-trivial so parity tests exercise them."""
    for k in params:
        if k.endswith("LayerNorm/beta") or k.endswith("/biases"):
            params[k] = (params[k] + scale * rng.standard_normal(params[k].shape)).astype(params[k].dtype)
        elif k.endswith("LayerNorm/gamma"):
            params[k] = (params[k] + scale * rng.standard_normal(params[k].shape)).astype(params[k].dtype)
    return params


# ----------------------------------------------------------------------------
# primitive forward ops
# ----------------------------------------------------------------------------
def sigmoid(x):
    # numerically stable logistic, same value as tf.sigmoid to rounding
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def fc_forward(x, W, b):
    """layers.fully_connected on the last axis."""
    return x @ W + b


def layer_norm_forward(x, gamma, beta):
    """tf.contrib.layers.layer_norm defaults: statistics over ALL
    axes except 0, gamma/beta on the last axis, biased variance, eps 1e-12."""
    axes = tuple(range(1, x.ndim))
    mu = x.mean(axis=axes, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=axes, keepdims=True)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(LN_EPS))
    xhat = (x - mu) * rstd
    return xhat * gamma + beta, xhat, rstd


def fc_ln_relu_forward(x, params, scope):
    """modules.fc_layer(use_bias, use_ln, relu): vlmap/modules.py:630-650."""
    pre = fc_forward(x, params[scope + "/fc/weights"], params[scope + "/fc/biases"])
    ln, xhat, rstd = layer_norm_forward(pre, params[scope + "/LayerNorm/gamma"],
                                        params[scope + "/LayerNorm/beta"])
    y = np.maximum(ln, 0)
    return y, (x, pre, xhat, rstd, ln)


def fc_ln_tanh_forward(x, params, scope):
    """modules.fc_layer(use_bias, use_ln, activation_fn=tf.tanh): vqa/model_vlmap_answer2.py:127-130."""
    pre = fc_forward(x, params[scope + "/fc/weights"], params[scope + "/fc/biases"])
    ln, xhat, rstd = layer_norm_forward(pre, params[scope + "/LayerNorm/gamma"],
                                        params[scope + "/LayerNorm/beta"])
    return np.tanh(ln), (x, pre, xhat, rstd, ln)


def old_report(report):
    """the 9 report scalars of the three oldest ablations (same keys as model_standard_testmask) from the 13-key report"""
    return {k: report[v] for k, v in TESTMASK_REPORT.items()}


def marginal_index(B, M):
    """vqa/model_vlmap_answer_ent.py:196-198: reshape(tile(pooled_linear_l, [M, 1]), [-1, M, L]) -- row (i, m) of the
    tiled tensor is flat row i * M + m of the [M * B, L] tile, i.e. pooled_linear_l[(i * M + m) % B]."""
    return (np.arange(B)[:, None] * M + np.arange(M)[None, :]) % B


def gru_forward(x, lens, Wg, bg, Wc, bc):
    """tf.contrib.rnn.GRUCell under tf.nn.dynamic_rnn(sequence_length).

    x [B,T,W]; returns final state [B,H] and the per-step tape for backward.
    r,u = split(sigmoid([x,h]Wg+bg)) with r FIRST; c = tanh([x, r*h]Wc+bc);
    h' = u*h + (1-u)*c; for t >= len the state is copied through.
    """
    B, T, Wd = x.shape
    H = Wc.shape[1]
    h = np.zeros((B, H), x.dtype)
    tape = []
    for t in range(T):
        xt = x[:, t, :]
        g = sigmoid(np.concatenate([xt, h], axis=1) @ Wg + bg)
        r, u = g[:, :H], g[:, H:]
        rh = r * h
        c = np.tanh(np.concatenate([xt, rh], axis=1) @ Wc + bc)
        hn = u * h + (1 - u) * c
        live = (t < lens)[:, None]
        hn = np.where(live, hn, h)
        tape.append((xt, h, r, u, c, live))
        h = hn
    return h, tape


def hadamard_attention_forward(v, nbox, qv, w, b, mask_att):
    """vlmap/modules.py:67-97.  mask_att is the explicit Bernoulli(keep=.8)
    0/1 mask replacing tf.nn.dropout's internal RNG."""
    dt = v.dtype.type
    feat = v * qv[:, None, :]
    feat = feat * mask_att * dt(1.0 / KEEP_ATT)
    s = feat @ w[:, 0] + b[0]                      # [B,R]
    R = v.shape[1]
    valid = np.arange(R)[None, :] < nbox[:, None]   # tf.sequence_mask
    s = np.where(valid, s, dt(-np.inf))
    with np.errstate(invalid="ignore"):
        m = s.max(axis=1, keepdims=True)
        e = np.exp(s - m)
        att = e / e.sum(axis=1, keepdims=True)      # nbox == 0 -> NaN row, as TF
    return att, feat


def sigmoid_ce(z, t):
    """tf.nn.sigmoid_cross_entropy_with_logits."""
    return np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))


def _guarded_ratio(num, den):
    # tf.where(tf.equal(den, 0), den, num / den)
    return den if den == 0 else num / den


def loss_and_report(z, tgt, answer_masks, model_type):
    """vqa/model_vlmap_answer.py:192-288 / vqa/model_standard.py:281-374.

    answer_masks: dict of float [A] arrays: train, obj, attr, exist.
    """
    dt = z.dtype.type
    train = answer_masks["train"]
    test = dt(1) - train
    obj, attr, exist = answer_masks["obj"], answer_masks["attr"], answer_masks["exist"]
    ell = sigmoid_ce(z, tgt)
    report_loss = ell.sum(axis=1).mean()
    if model_type in TRAIN_MASKED_LOSS:     # model_standard_word2vec.py:199-201 and model_standard_testmask.py:266-268 mask too
        train_loss = (ell * train).sum(axis=1).mean()
    else:
        train_loss = report_loss
    pred = np.argmax(z, axis=1).astype(np.int32)        # first max
    B = z.shape[0]
    tp = tgt[np.arange(B), pred]                         # one_hot(pred) * target, summed

    def at_pred(mask):
        return tp * mask[pred]

    out = {
        "pred": pred,
        "all_score": tp,
        "max_train_score": (tgt * train).max(axis=1),
        "test_obj_score": at_pred(test * obj),
        "test_obj_max_score": (tgt * test * obj).max(axis=1),
        "test_attr_score": at_pred(test * attr),
        "test_attr_max_score": (tgt * test * attr).max(axis=1),
    }
    acc = tp.mean()
    exist_acc = at_pred(exist).mean()
    test_acc = at_pred(test).mean()
    test_obj_acc = out["test_obj_score"].mean()
    test_attr_acc = out["test_attr_score"].mean()
    train_exist_acc = at_pred(exist * train).mean()
    max_exist = (tgt * exist).max(axis=1).mean()
    max_train_exist = (tgt * exist * train).max(axis=1).mean()
    test_obj_max = out["test_obj_max_score"].mean()
    test_attr_max = out["test_attr_max_score"].mean()
    test_max = (tgt * test).max(axis=1).mean()
    test_max_exist = (tgt * exist * test).max(axis=1).mean()
    report = {
        "answer_train_loss": train_loss,
        "answer_report_loss": report_loss,
        "answer_acc": acc,
        "exist_acc": exist_acc,
        "test_acc": test_acc,
        "normal_test_acc": _guarded_ratio(test_acc, test_max),
        "normal_test_object_acc": _guarded_ratio(test_obj_acc, test_obj_max),
        "normal_test_attribute_acc": _guarded_ratio(test_attr_acc, test_attr_max),
        "normal_exist_acc": _guarded_ratio(exist_acc, max_exist),
        "normal_train_exist_acc": _guarded_ratio(train_exist_acc, max_train_exist),
        "max_exist_acc": max_exist,
        "test_max_acc": test_max,
        "test_max_exist_acc": test_max_exist,
    }
    return train_loss, report, out, ell


def loss_and_report_all1(z1m, z2, tgt, answer_masks):
    """vqa/model_vlmap_answer_vqa_all.py:228-339: z1m = fixed logits with unknown answers at the row minimum, z2 = tuned
    logits; loss terms ce(z1m) and ce(z1m + z2), BOTH masked by the train-answer mask in the training loss; pred from the sum."""
    train = answer_masks["train"]
    ell = sigmoid_ce(z1m, tgt) + sigmoid_ce(z1m + z2, tgt)
    _, report, out, _ = loss_and_report(z1m + z2, tgt, answer_masks, "standard")
    train_loss = (ell * train).sum(axis=1).mean()
    report["answer_train_loss"] = train_loss
    report["answer_report_loss"] = ell.sum(axis=1).mean()
    return train_loss, report, out


def loss_and_report_all2(z1, z2, tgt, answer_masks):
    """vqa/model_vlmap_answer_vqa_all2.py:232-339: z1 = fixed WordWeightAnswer logits, z2 = TunedWordWeightAnswer logits.
    train loss = mean_B sum_A (ce(z1) * train_mask + ce(z2)); report loss = mean_B sum_A (ce(z1) + ce(z2));
    pred = argmax(z1 * test_mask + z2 * train_mask); everything downstream of pred as in model_vlmap_answer."""
    dt = z1.dtype.type
    train = answer_masks["train"]
    test = dt(1) - train
    ell1, ell2 = sigmoid_ce(z1, tgt), sigmoid_ce(z2, tgt)
    mixed = z1 * test + z2 * train
    _, report, out, _ = loss_and_report(mixed, tgt, answer_masks, "standard")     # pred and the 11 accuracy scalars
    train_loss = (ell1 * train + ell2).sum(axis=1).mean()
    report["answer_train_loss"] = train_loss
    report["answer_report_loss"] = (ell1 + ell2).sum(axis=1).mean()
    return train_loss, report, out


# ----------------------------------------------------------------------------
# full forward
# ----------------------------------------------------------------------------
def forward(params, batch, table, nbox_table, answer_masks, masks, model_type="vlmap_answer", stop_grad_values=None):
    """Forward of model_vlmap_answer / model_standard.

    stop_grad_values (tests only): {'pooled_linear_l': array} holds the value behind vlmap_answer_ent's tf.stop_gradient
    fixed while parameters are perturbed, so that finite differences see the gradient TF propagates.

    batch: dict image_idx i64[B], q_intseq i32[B,T], q_intseq_len i32[B],
           answer_target f[B,A]  (vqa/datasets/input_ops_vqa_tf_record_memft.py:47-71)
    table f[N,R,D], nbox_table i[N]; masks: {'att': 0/1 [B,R,H], 'joint': 0/1 [B,2H]}.
    Returns (loss, report, output, mid, tape).
    """
    sc = scope_names(model_type)
    dt = table.dtype.type
    idx = batch["image_idx"]
    V = np.take(table, idx, axis=0)                                   # a1
    nb = np.take(nbox_table, idx, axis=0)
    v, t_v = fc_ln_relu_forward(V, params, sc["v_linear_v"])           # a2
    e = params[sc["embed"]][batch["q_intseq"]]                          # a3
    h, t_gru = gru_forward(e, batch["q_intseq_len"],                    # a4
                           params[sc["gru_gates"] + "/kernel"], params[sc["gru_gates"] + "/bias"],
                           params[sc["gru_cand"] + "/kernel"], params[sc["gru_cand"] + "/bias"])
    qv, t_qv = fc_ln_relu_forward(h, params, sc["q_linear_v"])          # a5
    att, feat = hadamard_attention_forward(                              # a6
        v, nb, qv, params[sc["score"] + "/fc/weights"], params[sc["score"] + "/fc/biases"],
        masks["att"])
    va = t_va = None
    if model_type == "vlmap_answer_adapt":                               # :132-142: the attention pools v_adapt, not V_ft
        va, t_va = fc_ln_relu_forward(V, params, sc["v_adapt"])
        p = np.einsum("br,brd->bd", att, va)
    else:
        p = np.einsum("br,brd->bd", att, V)                              # a7
    pl, t_pl = fc_ln_relu_forward(p, params, sc["pooled_linear_l"])      # a8
    # what q_linear_l reads: the GRU state, or one of the ablations' layers on top of it
    lin_in, t_lin, extra_losses, extra_report = h, None, {}, {}
    if model_type == "vlmap_answer2":
        lin_in, t_lin = fc_ln_tanh_forward(h, params, sc["q_L_ft2"])
    elif model_type in ("vlmap_answer_no_noise", "vlmap_answer_full"):
        qm = fc_forward(h, params[sc["q_L_mean"] + "/fc/weights"], params[sc["q_L_mean"] + "/fc/biases"])
        lin_in, t_lin = qm, dict(qm=qm)
        if model_type == "vlmap_answer_full":
            qs = fc_forward(h, params[sc["q_L_log_sigma_sq"] + "/fc/weights"], params[sc["q_L_log_sigma_sq"] + "/fc/biases"])
            sigma = np.sqrt(np.exp(qs))
            lin_in = qm + masks["noise"] * sigma
            latent = dt(-0.5) * (dt(1) + qs - qm * qm - np.exp(qs)).sum(axis=-1).mean()      # :272-276
            t_lin = dict(qm=qm, qs=qs, sigma=sigma)
            extra_losses["latent"] = dt(LATENT_LOSS_WEIGHT) * latent
            extra_report.update(latent_loss=latent, train_latent_loss=extra_losses["latent"])
    ll, t_ll = fc_ln_relu_forward(lin_in, params, sc["q_linear_l"])
    if model_type in NOC_FAMILY:
        vj0, t_vj = fc_ln_relu_forward(pl, params, sc["joint_v"])        # :177-181
        lj0, t_lj = fc_ln_relu_forward(ll, params, sc["joint_l"])        # :183-188
        vj = vj0 * masks["joint"] * dt(1.0 / KEEP_JOINT)
        lj = lj0 * masks["joint_l"] * dt(1.0 / KEEP_JOINT)
        zv = fc_forward(vj, params[sc["headV"] + "/fc/weights"], params[sc["headV"] + "/fc/biases"])
        zl = fc_forward(lj, params[sc["headL"] + "/fc/weights"], params[sc["headL"] + "/fc/biases"])
        z = zv + zl                                                      # :204
        loss, report, out, ell = loss_and_report(z, batch["answer_target"], answer_masks, model_type)
        out["att_score"], out["logit"] = att, z
        mid = {"num_V_ft": nb, "q_linear_v": qv, "att_score": att, "pooled_V_ft": p, "pooled_linear_l": pl,
               "l_linear_l": ll, "v_joint": vj, "l_joint": lj, "logit": z, "pred": out["pred"], "v_linear_v": v,
               "condition": h, "V_ft": V}
        tape = dict(V=V, nb=nb, v=v, t_v=t_v, e=e, h=h, t_gru=t_gru, qv=qv, t_qv=t_qv, att=att, feat=feat, p=p, pl=pl,
                    t_pl=t_pl, ll=ll, t_ll=t_ll, vj=vj, lj=lj, t_vj=t_vj, t_lj=t_lj, z=z)
        return loss, report, out, mid, tape
    jin = pl * ll
    j0, t_j = fc_ln_relu_forward(jin, params, sc["joint_fc"])            # a9
    j = j0 * masks["joint"] * dt(1.0 / KEEP_JOINT)
    z = fc_forward(j, params[sc["head"] + "/fc/weights"], params[sc["head"] + "/fc/biases"])  # a10
    j2 = None
    if model_type == "standard_word2vec":                   # logit = joint2 x output_glove
        j2 = z
        z = j2 @ params[sc["glove"]]
    z1 = z2 = None
    extra_mid = {}
    z1_raw = None
    if model_type in TWO_HEAD_FAMILY:
        z1 = z
        z2 = fc_forward(j, params[sc["head2"] + "/fc/weights"], params[sc["head2"] + "/fc/biases"])   # reads `joint` (:216-217)
        if model_type == "vlmap_answer_vqa_all":
            z1_raw = z1
            ex = answer_masks["exist"]
            z1 = z1 * ex + z1.min(axis=1, keepdims=True) * (dt(1) - ex)                # :192-194
            loss, report, out = loss_and_report_all1(z1, z2, batch["answer_target"], answer_masks)
        else:
            loss, report, out = loss_and_report_all2(z1, z2, batch["answer_target"], answer_masks)
        z = z1 + z2                                                                    # output['logit'] (:226-227)
        # the dead branch (:202-214), for mid_result only: needs its own dropout mask when asked for
        tll, _ = fc_ln_relu_forward(h, params, sc["tuned_q_linear_l"])
        extra_mid.update(tuned_l_linear_l=tll, logit_fixed=z1, logit_tuned=z2)
        if "tuned_joint" in masks:
            tj0, _ = fc_ln_relu_forward(pl * tll, params, sc["tuned_joint_fc"])
            extra_mid["tuned_joint"] = tj0 * masks["tuned_joint"] * dt(1.0 / KEEP_JOINT)
    else:
        loss, report, out, ell = loss_and_report(z, batch["answer_target"], answer_masks, model_type)  # a11
    t_ent = None
    if model_type == "vlmap_answer_ent":
        # Maximum entropy regularisation (vqa/model_vlmap_answer_ent.py:191-211, 281-292): every question is paired with
        # M pooled visual features of the batch (stop-gradient), pushed through joint_fc (+ its own dropout) and the
        # head; the softmax over the known training answers is averaged over the M pairings and the NEGATIVE entropy of
        # that marginal joins the loss with weight 0.1.  layer_norm of the [B, M, 2H] tensor normalises over (M, 2H).
        Bn = pl.shape[0]
        M = masks["tile_joint"].shape[1]
        pl_const = pl if stop_grad_values is None else stop_grad_values["pooled_linear_l"]
        tp = pl_const[marginal_index(Bn, M)]                                               # [B, M, H], no gradient
        tin = tp * ll[:, None, :]
        tj0, t_tj = fc_ln_relu_forward(tin, params, sc["joint_fc"])
        tj = tj0 * masks["tile_joint"] * dt(1.0 / KEEP_JOINT)
        tz = fc_forward(tj, params[sc["head"] + "/fc/weights"], params[sc["head"] + "/fc/biases"])
        sel = (answer_masks["exist"] * answer_masks["train"]) > 0.5                      # train_exist_answer_mask_bool (:63-65)
        mz = tz[:, :, sel]
        ez = np.exp(mz - mz.max(axis=-1, keepdims=True))
        prob = ez / ez.sum(axis=-1, keepdims=True)
        mprob = prob.mean(axis=1)                                                          # [B, #train-exist answers]
        neg_ent = (mprob * np.log(mprob + dt(1e-8))).sum(axis=-1).mean()
        extra_losses["entropy"] = dt(W_ENTROPY) * neg_ent
        extra_report.update(entropy=neg_ent, weighted_entropy=extra_losses["entropy"])
        extra_mid["marginal_prob"] = mprob
        t_ent = dict(tp=tp, tj=tj, t_tj=t_tj, prob=prob, mprob=mprob, sel=sel, M=M)
    if model_type in OLD_REPORT_TYPES:
        report = old_report(report)
    report.update(extra_report)
    for v_ in extra_losses.values():         # self.loss = sum of self.losses (:250-252)
        loss = loss + v_
    out["att_score"] = att
    out["logit"] = z
    mid = {"num_V_ft": nb, "q_linear_v": qv, "att_score": att, "pooled_V_ft": p,
           "pooled_linear_l": pl, "l_linear_l": ll, "joint": j, "logit": z,
           "pred": out["pred"], "v_linear_v": v, "condition": lin_in if model_type == "vlmap_answer2" else h, "V_ft": V}
    if va is not None:
        mid["v_adapt"] = va
    if model_type in ("vlmap_answer_no_noise", "vlmap_answer_full"):
        mid["q_L_mean"] = t_lin["qm"]
    if model_type == "vlmap_answer_full":
        mid["q_L_log_sigma_sq"], mid["q_L_mean_noise"] = t_lin["qs"], lin_in
    mid.update(extra_mid)
    tape = dict(V=V, nb=nb, v=v, t_v=t_v, e=e, h=h, t_gru=t_gru, qv=qv, t_qv=t_qv, att=att,
                feat=feat, p=p, pl=pl, t_pl=t_pl, ll=ll, t_ll=t_ll, jin=jin, j0=j0, t_j=t_j,
                j=j, z=z, j2=j2, z1=z1, z2=z2, z1_raw=z1_raw, va=va, t_va=t_va, t_lin=t_lin, t_ent=t_ent,
                model_type=model_type)
    return loss, report, out, mid, tape


# ----------------------------------------------------------------------------
# analytic backward
# ----------------------------------------------------------------------------
def _fc_ln_relu_backward(dy, tape, params, scope, grads, need_dx=True, act="relu"):
    x, pre, xhat, rstd, ln = tape
    gamma = params[scope + "/LayerNorm/gamma"]
    dln = dy * (ln > 0) if act == "relu" else dy * (1 - np.tanh(ln) ** 2)
    red = tuple(range(dln.ndim - 1))
    grads[scope + "/LayerNorm/beta"] = dln.sum(axis=red)
    grads[scope + "/LayerNorm/gamma"] = (dln * xhat).sum(axis=red)
    dxhat = dln * gamma
    axes = tuple(range(1, dxhat.ndim))
    m1 = dxhat.mean(axis=axes, keepdims=True)
    m2 = (dxhat * xhat).mean(axis=axes, keepdims=True)
    dpre = rstd * (dxhat - m1 - xhat * m2)
    x2 = x.reshape(-1, x.shape[-1])
    d2 = dpre.reshape(-1, dpre.shape[-1])
    grads[scope + "/fc/weights"] = x2.T @ d2
    grads[scope + "/fc/biases"] = d2.sum(axis=0)
    if need_dx:
        return dpre @ params[scope + "/fc/weights"].T
    return None


def backward(params, batch, answer_masks, masks, tape, model_type="vlmap_answer"):
    """Gradients of the training loss wrt EVERY variable (frozen ones included;
    the optimiser picks train vars).  Returns (grads, dx_embed_slices) where
    dx_embed_slices [B,T,W] are the un-aggregated IndexedSlices values of the
    embedding gradient."""
    sc = scope_names(model_type)
    dt = tape["z"].dtype.type
    g = {}
    z = tape["z"]
    B = z.shape[0]
    tgt = batch["answer_target"]
    if model_type in TWO_HEAD_FAMILY:
        z = tape["z1"]
    dz = (sigmoid(z) - tgt) / dt(B)
    if model_type in TRAIN_MASKED_LOSS or model_type in TWO_HEAD_FAMILY:
        dz = dz * answer_masks["train"]
    dz2 = None
    if model_type == "vlmap_answer_vqa_all":
        dz2 = (sigmoid(tape["z1"] + tape["z2"]) - tgt) / dt(B) * answer_masks["train"]      # the tuned term sees the SUM, masked
        dz = dz + dz2                                                                       # d loss / d (masked fixed logits)
        # back through logit * exist + min(logit) * (1 - exist): tf.reduce_min hands its gradient to the minimum, split
        # evenly over ties
        ex = answer_masks["exist"]
        raw = tape["z1_raw"]
        is_min = (raw == raw.min(axis=1, keepdims=True)).astype(raw.dtype)
        to_min = (dz * (dt(1) - ex)).sum(axis=1, keepdims=True)
        dz = dz * ex + is_min * (to_min / is_min.sum(axis=1, keepdims=True))
    if model_type in NOC_FAMILY:
        for hd, jt in ((sc["headV"], "vj"), (sc["headL"], "lj")):
            g[hd + "/fc/weights"] = tape[jt].T @ dz
            g[hd + "/fc/biases"] = dz.sum(axis=0)
        sc_keep = dt(1.0 / KEEP_JOINT)
        dpl = _fc_ln_relu_backward(dz @ params[sc["headV"] + "/fc/weights"].T * masks["joint"] * sc_keep,
                                   tape["t_vj"], params, sc["joint_v"], g)
        dll = _fc_ln_relu_backward(dz @ params[sc["headL"] + "/fc/weights"].T * masks["joint_l"] * sc_keep,
                                   tape["t_lj"], params, sc["joint_l"], g)
        return _backward_below_joint(params, batch, masks, tape, sc, g, dpl, dll, dt)
    Wh = params[sc["head"] + "/fc/weights"]
    if model_type == "standard_word2vec":
        dz = dz @ params[sc["glove"]].T                     # gradient wrt joint2; the GloVe matrix is a constant
    g[sc["head"] + "/fc/weights"] = tape["j"].T @ dz
    g[sc["head"] + "/fc/biases"] = dz.sum(axis=0)
    dj = dz @ Wh.T
    if model_type in TWO_HEAD_FAMILY:
        if dz2 is None:
            dz2 = (sigmoid(tape["z2"]) - tgt) / dt(B)           # _all2: the tuned term of the loss is not masked (:240-241)
        g[sc["head2"] + "/fc/weights"] = tape["j"].T @ dz2
        g[sc["head2"] + "/fc/biases"] = dz2.sum(axis=0)
        dj = dj + dz2 @ params[sc["head2"] + "/fc/weights"].T
        for scope in (sc["tuned_q_linear_l"], sc["tuned_joint_fc"]):   # no path to the loss: TF hands back None
            for v in ("/fc/weights", "/fc/biases", "/LayerNorm/beta", "/LayerNorm/gamma"):
                g[scope + v] = np.zeros_like(params[scope + v])
    dj0 = dj * masks["joint"] * dt(1.0 / KEEP_JOINT)
    djin = _fc_ln_relu_backward(dj0, tape["t_j"], params, sc["joint_fc"], g)
    dpl = djin * tape["ll"]
    dll = djin * tape["pl"]
    if model_type == "vlmap_answer_ent":
        # the regulariser's path: marginal -> softmax of every pairing -> head -> dropout -> LN(M x 2H) + ReLU -> joint_fc
        # -> l_linear_l only (pooled_linear_l is behind tf.stop_gradient, :197)
        te = tape["t_ent"]
        prob, mprob, M = te["prob"], te["mprob"], te["M"]
        eps = dt(1e-8)
        dmp = dt(W_ENTROPY) / dt(B) * (np.log(mprob + eps) + mprob / (mprob + eps))
        dprob = np.broadcast_to(dmp[:, None, :] / dt(M), prob.shape)
        dmz = prob * (dprob - (prob * dprob).sum(axis=-1, keepdims=True))
        dtz = np.zeros(prob.shape[:2] + (z.shape[1],), z.dtype)
        dtz[:, :, te["sel"]] = dmz
        g[sc["head"] + "/fc/weights"] = g[sc["head"] + "/fc/weights"] + te["tj"].reshape(-1, te["tj"].shape[-1]).T @ dtz.reshape(-1, z.shape[1])
        g[sc["head"] + "/fc/biases"] = g[sc["head"] + "/fc/biases"] + dtz.sum(axis=(0, 1))
        dtj0 = (dtz @ Wh.T) * masks["tile_joint"] * dt(1.0 / KEEP_JOINT)
        g2 = {}
        dtin = _fc_ln_relu_backward(dtj0, te["t_tj"], params, sc["joint_fc"], g2)
        for k_, v_ in g2.items():                    # joint_fc is one set of variables used at both call sites
            g[k_] = g[k_] + v_
        dll = dll + (dtin * te["tp"]).sum(axis=1)
    return _backward_below_joint(params, batch, masks, tape, sc, g, dpl, dll, dt)


def _backward_below_joint(params, batch, masks, tape, sc, g, dpl, dll, dt):
    """everything upstream of pooled_linear_l / l_linear_l (shared by all model types)"""
    z = tape["z"]
    model_type = tape.get("model_type")
    B = z.shape[0]
    dp = _fc_ln_relu_backward(dpl, tape["t_pl"], params, sc["pooled_linear_l"], g)
    dh = _fc_ln_relu_backward(dll, tape["t_ll"], params, sc["q_linear_l"], g)      # gradient wrt what q_linear_l read
    if model_type == "vlmap_answer2":
        dh = _fc_ln_relu_backward(dh, tape["t_lin"], params, sc["q_L_ft2"], g, act="tanh")
    elif model_type in ("vlmap_answer_no_noise", "vlmap_answer_full"):
        tl, h_ = tape["t_lin"], tape["h"]
        dqm = dh
        if model_type == "vlmap_answer_full":
            # x = mean + noise * exp(ls / 2); latent = -0.5 mean_B sum (1 + ls - mean^2 - exp(ls)), weight 0.1
            lw = dt(LATENT_LOSS_WEIGHT) / dt(B)
            dqs = dh * masks["noise"] * dt(0.5) * tl["sigma"] + lw * dt(0.5) * (np.exp(tl["qs"]) - dt(1))
            dqm = dh + lw * tl["qm"]
            g[sc["q_L_log_sigma_sq"] + "/fc/weights"] = h_.T @ dqs
            g[sc["q_L_log_sigma_sq"] + "/fc/biases"] = dqs.sum(axis=0)
        g[sc["q_L_mean"] + "/fc/weights"] = h_.T @ dqm
        g[sc["q_L_mean"] + "/fc/biases"] = dqm.sum(axis=0)
        dh = dqm @ params[sc["q_L_mean"] + "/fc/weights"].T
        if model_type == "vlmap_answer_full":
            dh = dh + dqs @ params[sc["q_L_log_sigma_sq"] + "/fc/weights"].T
    # attention pooling: p = sum_r att * V   (V is an input: no dV; adapt pools the trainable v_adapt instead)
    V, att = tape["V"], tape["att"]
    if model_type == "vlmap_answer_adapt":
        datt = np.einsum("bd,brd->br", dp, tape["va"])
        _fc_ln_relu_backward(att[:, :, None] * dp[:, None, :], tape["t_va"], params, sc["v_adapt"], g, need_dx=False)
    else:
        datt = np.einsum("bd,brd->br", dp, V)
    ds = att * (datt - (att * datt).sum(axis=1, keepdims=True))      # softmax backward (masked rows: att=0)
    w = params[sc["score"] + "/fc/weights"]
    g[sc["score"] + "/fc/weights"] = np.einsum("br,brh->h", ds, tape["feat"])[:, None]
    g[sc["score"] + "/fc/biases"] = np.array([ds.sum()], dtype=z.dtype)
    dfeat = ds[:, :, None] * w[None, None, :, 0]
    dfeat = dfeat * masks["att"] * dt(1.0 / KEEP_ATT)
    dv = dfeat * tape["qv"][:, None, :]
    dqv = (dfeat * tape["v"]).sum(axis=1)
    _fc_ln_relu_backward(dv, tape["t_v"], params, sc["v_linear_v"], g, need_dx=False)
    dh = dh + _fc_ln_relu_backward(dqv, tape["t_qv"], params, sc["q_linear_v"], g)
    # GRU BPTT
    Wg = params[sc["gru_gates"] + "/kernel"]
    Wc = params[sc["gru_cand"] + "/kernel"]
    Wd = tape["e"].shape[2]
    H = Wc.shape[1]
    dWg = np.zeros_like(Wg); dbg = np.zeros(2 * H, z.dtype)
    dWc = np.zeros_like(Wc); dbc = np.zeros(H, z.dtype)
    T = len(tape["t_gru"])
    dx = np.zeros_like(tape["e"])
    for t in range(T - 1, -1, -1):
        xt, hp, r, u, c, live = tape["t_gru"][t]
        dh_live = np.where(live, dh, 0)
        dh_pass = np.where(live, 0, dh)
        du = dh_live * (hp - c)
        dc = dh_live * (1 - u)
        dhp = dh_live * u
        dc_pre = dc * (1 - c * c)
        xin_c = np.concatenate([xt, r * hp], axis=1)
        dWc += xin_c.T @ dc_pre
        dbc += dc_pre.sum(axis=0)
        dxin_c = dc_pre @ Wc.T
        drh = dxin_c[:, Wd:]
        dr = drh * hp
        dhp = dhp + drh * r
        dg_pre = np.concatenate([dr * r * (1 - r), du * u * (1 - u)], axis=1)
        xin_g = np.concatenate([xt, hp], axis=1)
        dWg += xin_g.T @ dg_pre
        dbg += dg_pre.sum(axis=0)
        dxin_g = dg_pre @ Wg.T
        dhp = dhp + dxin_g[:, Wd:]
        dx[:, t, :] = dxin_c[:, :Wd] + dxin_g[:, :Wd]
        dh = dhp + dh_pass
    g[sc["gru_gates"] + "/kernel"] = dWg
    g[sc["gru_gates"] + "/bias"] = dbg
    g[sc["gru_cand"] + "/kernel"] = dWc
    g[sc["gru_cand"] + "/bias"] = dbc
    dE = np.zeros_like(params[sc["embed"]])
    np.add.at(dE, batch["q_intseq"].reshape(-1), dx.reshape(-1, Wd))
    g[sc["embed"]] = dE
    return g, dx


# ----------------------------------------------------------------------------
# optimiser: tf.contrib.layers.optimize_loss(Adam, clip_gradients=20.0)
# ----------------------------------------------------------------------------
def global_norm(grads, train_names, dx_embed, embed_name):
    """clip_ops.global_norm over the train-var gradients; the embedding
    gradient is an IndexedSlices whose norm is taken over the UN-AGGREGATED
    slice values."""
    acc = 0.0
    for n in train_names:
        if n == embed_name:
            acc += float((dx_embed.astype(np.float64) ** 2).sum())
        else:
            acc += float((grads[n].astype(np.float64) ** 2).sum())
    return np.sqrt(acc)


def clip_adam_step(params, grads, train_names, state, lr, dx_embed, embed_name,
                   clip=CLIP_NORM):
    """One optimiser step in place.  state = {'step': int, 'm': {}, 'v': {}}.

    clip_by_global_norm: g * clip / max(norm, clip).  Adam (TF1):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t*m/(sqrt(v)+eps).  Sparse Adam
    after de-duplication == dense Adam with zero rows.
    """
    norm = global_norm(grads, train_names, dx_embed, embed_name)
    scale = clip / max(norm, clip)
    state["step"] += 1
    t = state["step"]
    for n in train_names:
        dt = params[n].dtype.type
        gsc = grads[n] * dt(scale)
        m = state["m"].setdefault(n, np.zeros_like(params[n]))
        v = state["v"].setdefault(n, np.zeros_like(params[n]))
        m[...] = dt(ADAM_B1) * m + dt(1 - ADAM_B1) * gsc
        v[...] = dt(ADAM_B2) * v + dt(1 - ADAM_B2) * gsc * gsc
        lr_t = lr * np.sqrt(1 - ADAM_B2 ** t) / (1 - ADAM_B1 ** t)
        params[n] -= dt(lr_t) * m / (np.sqrt(v) + dt(ADAM_EPS))
    return norm


def new_opt_state():
    return {"step": 0, "m": {}, "v": {}}


def train_step(params, batch, table, nbox_table, answer_masks, masks, state, lr=1e-3,
               model_type="vlmap_answer"):
    """Trainer.run_train_step restated: forward, backward, clip, Adam (vqa/trainer.py:275-287)."""
    loss, report, out, mid, tape = forward(params, batch, table, nbox_table, answer_masks, masks, model_type)
    grads, dx = backward(params, batch, answer_masks, masks, tape, model_type)
    names = train_var_names(params, model_type)
    norm = clip_adam_step(params, grads, names, state, lr, dx, scope_names(model_type)["embed"])
    return loss, report, out, mid, grads, norm


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------
def make_answer_masks(rng, A, num_train_answer, dtype=np.float32, exist_all=True):
    train = (np.arange(A) < num_train_answer).astype(dtype)
    is_obj = (rng.random(A) < 0.5)
    obj = is_obj.astype(dtype)
    attr = (~is_obj).astype(dtype)
    exist = np.ones(A, dtype) if exist_all else (rng.random(A) < 0.7).astype(dtype)
    return {"train": train, "obj": obj, "attr": attr, "exist": exist}


def make_batch(rng, B, T, Vq, A, N, dtype=np.float32, ragged=True, min_len=3):
    idx = rng.integers(0, N, size=B).astype(np.int64)
    lens = (rng.integers(min_len, T + 1, size=B) if ragged else np.full(B, T)).astype(np.int32)
    q = rng.integers(0, max(Vq - 3, 1), size=(B, T)).astype(np.int32)
    q[np.arange(T)[None, :] >= lens[:, None]] = 0          # zero padding
    tgt = np.zeros((B, A), dtype)
    for b in range(B):
        k = int(rng.integers(1, 4))
        ids = rng.choice(A, size=k, replace=False)
        tgt[b, ids] = rng.choice(np.array([0.3, 0.6, 0.9, 1.0], dtype), size=k)
    return {"image_idx": idx, "q_intseq": q, "q_intseq_len": lens, "answer_target": tgt}


def make_table(rng, N, R, D, dtype=np.float32, full_boxes=True):
    table = np.maximum(rng.standard_normal((N, R, D)), 0).astype(dtype)
    nbox = np.full(N, R, np.int32) if full_boxes else rng.integers(1, R + 1, size=N).astype(np.int32)
    return table, nbox


def make_dropout_masks(rng, B, R, H, dtype=np.float32, model_type=None, num_marginal=NUM_MARGINAL):
    m = {"att": (rng.random((B, R, H)) < KEEP_ATT).astype(dtype),
         "joint": (rng.random((B, 2 * H)) < KEEP_JOINT).astype(dtype)}
    if model_type in NOC_FAMILY:          # second dropout site: l_joint (`joint` is v_joint's mask)
        m["joint_l"] = (rng.random((B, 2 * H)) < KEEP_JOINT).astype(dtype)
    if model_type == "vlmap_answer_full":  # tf.random_normal(seed=123) of the reparameterisation, an explicit input here
        m["noise"] = rng.standard_normal((B, H)).astype(dtype)
    if model_type == "vlmap_answer_ent":   # tf.nn.dropout(tile_joint, 0.5): its own mask, [B, M, 2H]
        m["tile_joint"] = (rng.random((B, num_marginal, 2 * H)) < KEEP_JOINT).astype(dtype)
    return m
