"""CPU oracle for the oldest model of the registry, vqa/model_vqa.py (`--model_type vqa`, the default of
vqa/trainer.py:18,337).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (no reference fixtures; TF 1.6 absent -- see oracle/vqa_oracle.py).  NumPy forward restating the
reference graph, plus an independently composed torch forward whose autograd supplies the reference gradients.

Reference lines (relative to /root/reference):
  * Model.__init__ / filters     vqa/model_vqa.py:19-88 (answers' token sequences from data_info.hdf5, :35-45;
                                 V2L / L2V train only with config.ft_vlmap, :63-74)
  * build                        vqa/model_vqa.py:185-277
  * GloVe_vocab                  vlmap/modules.py:451-467   (constant GloVe rows + a trainable [3, 300] `GloVe/learn` for the
                                 last three vocabulary entries)
  * encode_L (cell_type='LSTM' is the default the model uses, :212, :229)   vlmap/modules.py:124-140:
                                 tf.contrib.rnn.BasicLSTMCell(512) under tf.nn.dynamic_rnn(sequence_length): gates
                                 i, j, f, o = split([x, h] W + b); c' = c sigmoid(f + 1) + sigmoid(i) tanh(j);
                                 h' = tanh(c') sigmoid(o); state carried through past the length; output = final h.
                                 ONE set of LSTM variables encodes the questions and all candidate answers (AUTO_REUSE)
  * L2V / V2L                    vlmap/modules.py:533-549, 508-524 (three fc_layers, ReLU / tanh, no LayerNorm)
  * attention (dot product)      vlmap/modules.py:42-64
  * attention_pooling            vlmap/modules.py:23-39
The features this model reads are model_vfeat's 512-d region features (V_DIM = 512 = vfeat_dim; the dot-product attention
needs equal widths).  logit[b, a] = w . tanh(A1 answer_ft[a] + P1 pooled_map_L[b] + Q1 q_L_ft[b] + bq) + bc  (:232-257),
loss = mean_B sum_A sigmoid-CE with NO train-answer mask (:263-266), report = answer_loss, answer_accuracy (:274-275).
"""
from __future__ import annotations

import numpy as np

from . import vqa_oracle as O

REPORT_KEYS = ["answer_loss", "answer_accuracy"]
FIXED = "GloVe/fixed:const"          # the constant part of GloVe_vocab's embed_map: not a variable
LSTM = "encode_L/rnn/basic_lstm_cell"


def variable_shapes(Vq, W=300, D=512, L=512, M=512, A=3000):
    s = {"GloVe/learn": (3, W), LSTM + "/kernel": (W + L, 4 * L), LSTM + "/bias": (4 * L,)}

    def fc(scope, fin, fout, bias=True):
        s[scope + "/fc/weights"] = (fin, fout)
        if bias:
            s[scope + "/fc/biases"] = (fout,)

    fc("L2V/fc_1", L, M); fc("L2V/fc_2", M, M); fc("L2V/Linear", M, D)
    fc("V2L/fc_1", D, M); fc("V2L/fc_2", M, M); fc("V2L/Linear", M, L)
    fc("reasoning/answer_layer1", L, L, bias=False)
    fc("reasoning/pooled_layer1", L, L, bias=False)
    fc("reasoning/q_layer1", L, L)
    fc("reasoning/classifier", L, 1)
    return s


def init_params(rng, Vq=64, W=300, D=512, L=512, M=512, A=3000, dtype=np.float32):
    p = {FIXED: (0.3 * rng.standard_normal((Vq - 3, W))).astype(dtype)}
    for n, sh in variable_shapes(Vq, W, D, L, M, A).items():
        if n == "GloVe/learn":
            p[n] = rng.uniform(-0.01, 0.01, size=sh).astype(dtype)
        elif n.endswith("/weights") or n.endswith("/kernel"):
            lim = np.sqrt(6.0 / (sh[0] + sh[1]))
            p[n] = rng.uniform(-lim, lim, size=sh).astype(dtype)
        else:
            p[n] = np.zeros(sh, dtype)                      # BasicLSTMCell bias 0 (forget_bias 1.0 is added in the cell)
    return p


def train_var_names(params, ft_vlmap):
    names = sorted(n for n in params if not O.is_const(n))
    if ft_vlmap:
        return names
    return [n for n in names if n.split("/")[0] not in ("V2L", "L2V")]


def transfer_var_names(params):
    return [n for n in sorted(params) if not O.is_const(n) and n.split("/")[0] in ("V2L", "L2V", "encode_L", "GloVe")]


def lstm_final_h(x, lens, K, b):
    """dynamic_rnn(BasicLSTMCell, sequence_length): x [N, T, W] -> final h [N, L]."""
    N, T, _ = x.shape
    L = K.shape[1] // 4
    c = np.zeros((N, L), x.dtype)
    h = np.zeros((N, L), x.dtype)
    for t in range(T):
        g = np.concatenate([x[:, t], h], 1) @ K + b
        i, j, f, o = g[:, :L], g[:, L:2 * L], g[:, 2 * L:3 * L], g[:, 3 * L:]
        cn = c * O.sigmoid(f + x.dtype.type(1.0)) + O.sigmoid(i) * np.tanh(j)
        hn = np.tanh(cn) * O.sigmoid(o)
        live = (t < lens)[:, None]
        c, h = np.where(live, cn, c), np.where(live, hn, h)
    return h


def _fc(x, p, scope, act=None):
    y = x @ p[scope + "/fc/weights"]
    if scope + "/fc/biases" in p:
        y = y + p[scope + "/fc/biases"]
    return np.maximum(y, 0) if act == "relu" else (np.tanh(y) if act == "tanh" else y)


def forward(p, batch, table, nbox_table, answers):
    """answers = {'intseq': i32 [A, La], 'len': i32 [A]} (data_info.hdf5: intseq_ans / intseq_ans_len).
    Returns (loss, report, out, mid)."""
    glove = np.concatenate([p[FIXED], p["GloVe/learn"]], 0)
    idx = batch["image_idx"]
    V = np.take(table, idx, axis=0)
    nb = np.take(nbox_table, idx, axis=0)
    K, b = p[LSTM + "/kernel"], p[LSTM + "/bias"]
    q_ft = lstm_final_h(glove[batch["q_intseq"]], batch["q_intseq_len"], K, b)
    q_map_V = _fc(_fc(_fc(q_ft, p, "L2V/fc_1", "relu"), p, "L2V/fc_2", "relu"), p, "L2V/Linear")
    s = np.einsum("brd,bd->br", V, q_map_V)
    valid = np.arange(V.shape[1])[None, :] < nb[:, None]
    s = np.where(valid, s, -np.inf)
    e = np.exp(s - s.max(1, keepdims=True))
    att = e / e.sum(1, keepdims=True)
    pooled = np.einsum("br,brd->bd", att, V)
    pooled_map_L = _fc(_fc(_fc(pooled, p, "V2L/fc_1", "tanh"), p, "V2L/fc_2", "tanh"), p, "V2L/Linear")
    a_ft = lstm_final_h(glove[answers["intseq"]], answers["len"], K, b)
    al = _fc(a_ft, p, "reasoning/answer_layer1")
    pl = _fc(pooled_map_L, p, "reasoning/pooled_layer1")
    ql = _fc(q_ft, p, "reasoning/q_layer1")
    layer1 = np.tanh(al[None, :, :] + pl[:, None, :] + ql[:, None, :])
    z = layer1 @ p["reasoning/classifier/fc/weights"][:, 0] + p["reasoning/classifier/fc/biases"][0]
    tgt = batch["answer_target"]
    loss = O.sigmoid_ce(z, tgt).sum(1).mean()
    pred = np.argmax(z, 1).astype(np.int32)
    acc = tgt[np.arange(len(pred)), pred].mean()
    mid = {"num_V_ft": nb, "q_L_ft": q_ft, "q_map_V": q_map_V, "att_score": att, "pooled_V_ft": pooled,
           "pooled_map_L": pooled_map_L, "answer_ft": a_ft, "logit": z, "pred": pred}
    return loss, {"answer_loss": loss, "answer_accuracy": acc}, {"pred": pred, "att_score": att}, mid


def torch_loss_and_grads(p, batch, table, nbox_table, answers):
    """float64 torch restatement (F.linear, split-based cell, masked_fill softmax).  Returns (loss, mid, grads, slice_sq):
    grads for every variable; slice_sq = the sum of squares of the un-aggregated IndexedSlices that reach GloVe/learn
    (question and answer lookups of the last three vocabulary rows)."""
    import torch
    import torch.nn.functional as F
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    P = {k: t64(v).requires_grad_(not O.is_const(k)) for k, v in p.items()}
    glove = torch.cat([P[FIXED], P["GloVe/learn"]], 0)
    K, b = P[LSTM + "/kernel"], P[LSTM + "/bias"]
    L = K.shape[1] // 4

    def lstm(ids, lens):
        x = F.embedding(torch.from_numpy(ids).long(), glove)
        x.retain_grad()
        lens = torch.from_numpy(lens).long()
        c = x.new_zeros(x.shape[0], L)
        h = x.new_zeros(x.shape[0], L)
        for t in range(x.shape[1]):
            i, j, f, o = (torch.cat([x[:, t], h], 1) @ K + b).split(L, dim=1)
            cn = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            hn = torch.tanh(cn) * torch.sigmoid(o)
            live = (lens > t)[:, None]
            c, h = torch.where(live, cn, c), torch.where(live, hn, h)
        return h, x

    def fc(x, scope, act=None):
        y = F.linear(x, P[scope + "/fc/weights"].t(), P.get(scope + "/fc/biases"))
        return torch.relu(y) if act == "relu" else (torch.tanh(y) if act == "tanh" else y)

    idx = torch.from_numpy(batch["image_idx"])
    V = t64(table).index_select(0, idx)
    nb = torch.from_numpy(nbox_table).long().index_select(0, idx)
    q_ft, xq = lstm(batch["q_intseq"], batch["q_intseq_len"])
    q_map_V = fc(fc(fc(q_ft, "L2V/fc_1", "relu"), "L2V/fc_2", "relu"), "L2V/Linear")
    s = torch.bmm(V, q_map_V.unsqueeze(-1)).squeeze(-1)
    valid = torch.arange(V.shape[1])[None, :] < nb[:, None]
    att = torch.softmax(s.masked_fill(~valid, float("-inf")), -1)
    pooled = torch.bmm(att.unsqueeze(1), V).squeeze(1)
    pooled_map_L = fc(fc(fc(pooled, "V2L/fc_1", "tanh"), "V2L/fc_2", "tanh"), "V2L/Linear")
    a_ft, xa = lstm(answers["intseq"], answers["len"])
    layer1 = torch.tanh(fc(a_ft, "reasoning/answer_layer1").unsqueeze(0) + fc(pooled_map_L, "reasoning/pooled_layer1").unsqueeze(1)
                        + fc(q_ft, "reasoning/q_layer1").unsqueeze(1))
    z = fc(layer1, "reasoning/classifier").squeeze(-1)
    loss = F.binary_cross_entropy_with_logits(z, t64(batch["answer_target"]), reduction="none").sum(-1).mean()
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else np.zeros(t.shape)) for k, t in P.items() if not O.is_const(k)}
    Vq = glove.shape[0]
    sq = 0.0
    for ids, x in ((batch["q_intseq"], xq), (answers["intseq"], xa)):
        sel = torch.from_numpy(ids).long() >= Vq - 3
        sq += float((x.grad[sel] ** 2).sum())
    mid = {"q_L_ft": q_ft, "q_map_V": q_map_V, "att_score": att, "pooled_V_ft": pooled, "pooled_map_L": pooled_map_L,
           "answer_ft": a_ft, "logit": z}
    return float(loss.detach()), {k: t.detach().numpy() for k, t in mid.items()}, grads, sq


def train_step(p, batch, table, nbox_table, answers, state, lr, ft_vlmap):
    """forward, backward, clip_by_global_norm(20), Adam on the train variables, in place (vqa/trainer.py:87-114)."""
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    loss, mid, grads, sq = torch_loss_and_grads(to64(p), batch, table.astype(np.float64), nbox_table, answers)
    names = train_var_names(p, ft_vlmap)
    norm = np.sqrt(sum((sq if n == "GloVe/learn" else float((grads[n].astype(np.float64) ** 2).sum())) for n in names))
    scale = O.CLIP_NORM / max(norm, O.CLIP_NORM)
    state["step"] += 1
    t = state["step"]
    lr_t = lr * np.sqrt(1 - O.ADAM_B2 ** t) / (1 - O.ADAM_B1 ** t)
    for n in names:
        dt = p[n].dtype.type
        g = (grads[n] * scale).astype(p[n].dtype)
        m = state["m"].setdefault(n, np.zeros_like(p[n]))
        v = state["v"].setdefault(n, np.zeros_like(p[n]))
        m[...] = dt(O.ADAM_B1) * m + dt(1 - O.ADAM_B1) * g
        v[...] = dt(O.ADAM_B2) * v + dt(1 - O.ADAM_B2) * g * g
        p[n] -= dt(lr_t) * m / (np.sqrt(v) + dt(O.ADAM_EPS))
    return loss, grads, norm


def make_answers(rng, A, Vq, La=4):
    lens = rng.integers(1, La + 1, size=A).astype(np.int32)
    seq = rng.integers(0, Vq, size=(A, La)).astype(np.int32)
    seq[np.arange(La)[None, :] >= lens[:, None]] = 0
    return {"intseq": seq, "len": lens}
