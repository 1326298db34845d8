"""CPU oracle for the bi-directional-GRU generation of the VQA model: vqa/model_vlmap_finetune.py and
vqa/model_vlmap_only.py (the two files differ in filter_train_vars only).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (no reference fixtures; TF 1.6 absent -- see oracle/vqa_oracle.py).  NumPy forward restating the
reference graph, plus an independently composed torch forward whose autograd supplies the reference gradients (the layer
math shared with model_vlmap_answer -- fc_layer, GRUCell, hadamard_attention, the loss -- is pinned by the hand-derived
backward of oracle/vqa_oracle.py).

Reference lines (relative to /root/reference):
  * Model.__init__ / filters          vqa/model_vlmap_finetune.py:17-87 (vlmap_only: :64-86)
  * build                             vqa/model_vlmap_finetune.py:89-211
  * encode_L_bidirection              vlmap/modules.py:100-122 (GRUCell(512) forward and backward,
                                      tf.nn.bidirectional_dynamic_rnn(sequence_length): the backward cell runs on
                                      reverse_sequence(inputs, len) and its outputs are reversed back the same way;
                                      outputs past the length are zero, states are carried through)
  * WordWeightEmbed                   vlmap/modules.py:393-412
  * fc_layer / hadamard_attention / attention_pooling / WordWeightAnswer   vlmap/modules.py:630-650, 67-97, 23-39, 589-627
What differs from model_vlmap_answer: the question code is concat(final fw state, final bw state) (2 x 512); the
attention's query is NOT a layer on that code but q_linear_v(pooled_q_v) where pooled_q_v pools a second word embedding
(V_WordMap -> v_word_fc) with a question SELF-attention (word_attention: keys fc_layer(bi-GRU outputs), query
fc_layer(question code)); fc_layer on a [B, T, .] tensor normalises over the whole [T, 1024] block of a sample, padded
positions included (T = padded width of the batch); the report has three scalars (:207-211).
"""
from __future__ import annotations

import numpy as np

from . import vqa_oracle as O

KEEP_ATT = O.KEEP_ATT
REPORT_KEYS = ["answer_train_loss", "answer_report_loss", "answer_accuracy"]
MODEL_TYPES = ("vlmap_finetune", "vlmap_only")
# vqa/model_vlmap_only.py:64-76: everything that the pre-training produced stays fixed
FROZEN_TOP_SCOPES_ONLY = ("V_WordMap", "v_word_fc", "q_linear_v", "v_linear_v", "hadamard_attention", "q_linear_l",
                          "pooled_linear_l", "joint_fc", "WordWeightAnswer")
# vqa/model_vlmap_finetune.py:70-87 (same in model_vlmap_only.py)
TRANSFER_TOP_SCOPES = ("v_word_fc", "q_linear_v", "v_linear_v", "hadamard_attention", "q_linear_l", "pooled_linear_l",
                       "joint_fc")
GRU = "encode_L_bi/bidirectional_rnn/%s/gru_cell/%s"


def variable_shapes(Vq, W=300, D=2048, H=1024, A=3000):
    h1, h2 = (H + 1) // 2, H // 2                        # dim1 = ceil(dim / 2), dim2 = floor(dim / 2)  (modules.py:103-104)
    s = {"LearnGloVe/embed_map": (Vq, W), "V_WordMap/embed_map": (Vq, W)}

    def fc(scope, fin, fout, ln):
        s[scope + "/fc/weights"] = (fin, fout)
        s[scope + "/fc/biases"] = (fout,)
        if ln:
            s[scope + "/LayerNorm/beta"] = (fout,)
            s[scope + "/LayerNorm/gamma"] = (fout,)

    fc("v_linear_v", D, H, True)
    for d, h in (("fw", h1), ("bw", h2)):
        s[GRU % (d, "gates") + "/kernel"] = (W + h, 2 * h)
        s[GRU % (d, "gates") + "/bias"] = (2 * h,)
        s[GRU % (d, "candidate") + "/kernel"] = (W + h, h)
        s[GRU % (d, "candidate") + "/bias"] = (h,)
    fc("q_att_key", H, H, True)
    fc("q_att_query", H, H, True)
    fc("word_attention/compute/score", H, 1, False)
    fc("v_word_fc", W, H, True)
    fc("q_linear_v", H, H, True)
    fc("hadamard_attention/compute/score", H, 1, False)
    fc("pooled_linear_l", D, H, True)
    fc("q_linear_l", H, H, True)
    fc("joint_fc", H, 2 * H, True)
    fc("WordWeightAnswer", 2 * H, A, False)
    return s


def init_params(rng, Vq=64, W=300, D=2048, H=1024, A=3000, dtype=np.float32):
    p = {}
    for n, sh in variable_shapes(Vq, W, D, H, A).items():
        if n.endswith("embed_map"):
            p[n] = rng.uniform(-0.05, 0.05, size=sh).astype(dtype)
        elif n.endswith("/weights") or n.endswith("/kernel"):
            lim = np.sqrt(6.0 / (sh[0] + sh[1]))
            p[n] = rng.uniform(-lim, lim, size=sh).astype(dtype)
        elif n.endswith("gates/bias") or n.endswith("LayerNorm/gamma"):
            p[n] = np.ones(sh, dtype)
        else:
            p[n] = np.zeros(sh, dtype)
    return p


def train_var_names(params, model_type):
    names = sorted(params)
    if model_type == "vlmap_finetune":                   # :64-68: every trainable variable
        return names
    return [n for n in names if n.split("/")[0] not in FROZEN_TOP_SCOPES_ONLY]


def transfer_var_names(params, model_type):
    return [n for n in sorted(params) if n.split("/")[0] in TRANSFER_TOP_SCOPES]


def reverse_sequence(x, lens):
    """tf.reverse_sequence(x, seq_lengths=lens, seq_axis=1, batch_axis=0): the first lens[b] steps of row b reversed,
    the rest left where they are."""
    out = x.copy()
    for b, n in enumerate(lens):
        out[b, :n] = x[b, :n][::-1]
    return out


def gru_outputs(x, lens, Wg, bg, Wc, bc):
    """dynamic_rnn(GRUCell, sequence_length): per-step outputs [B, T, h] (zero past the length) and the final state."""
    B, T, _ = x.shape
    h = Wc.shape[1]
    state = np.zeros((B, h), x.dtype)
    outs = np.zeros((B, T, h), x.dtype)
    for t in range(T):
        g = O.sigmoid(np.concatenate([x[:, t], state], 1) @ Wg + bg)
        r, u = g[:, :h], g[:, h:]
        c = np.tanh(np.concatenate([x[:, t], r * state], 1) @ Wc + bc)
        new = u * state + (1 - u) * c
        live = (t < lens)[:, None]
        outs[:, t] = np.where(live, new, 0)
        state = np.where(live, new, state)
    return outs, state


def encode_L_bidirection(x, lens, p):
    fw = [p[GRU % ("fw", k) + s] for k, s in (("gates", "/kernel"), ("gates", "/bias"), ("candidate", "/kernel"), ("candidate", "/bias"))]
    bw = [p[GRU % ("bw", k) + s] for k, s in (("gates", "/kernel"), ("gates", "/bias"), ("candidate", "/kernel"), ("candidate", "/bias"))]
    of, sf = gru_outputs(x, lens, *fw)
    ob, sb = gru_outputs(reverse_sequence(x, lens), lens, *bw)
    ob = reverse_sequence(ob, lens)
    return np.concatenate([of, ob], -1), np.concatenate([sf, sb], -1)


def forward(p, batch, table, nbox_table, answer_masks, masks):
    """masks: {'att': [B,R,H], 'word': [B,T,H], 'joint': [B,2H]} 0/1 keep masks of the three tf.nn.dropout sites
    (image attention, word attention, joint).  Returns (loss, report, out, mid)."""
    dt = table.dtype.type
    idx = batch["image_idx"]
    lens = batch["q_intseq_len"]
    q = batch["q_intseq"]
    V = np.take(table, idx, axis=0)
    nb = np.take(nbox_table, idx, axis=0)
    v, _ = O.fc_ln_relu_forward(V, p, "v_linear_v")
    e = p["LearnGloVe/embed_map"][q]
    q_map, q_ft = encode_L_bidirection(e, lens, p)                                       # [B,T,H], [B,H]
    key, _ = O.fc_ln_relu_forward(q_map, p, "q_att_key")                                  # LN over [T, H]
    query, _ = O.fc_ln_relu_forward(q_ft, p, "q_att_query")
    w_att, _ = O.hadamard_attention_forward(key, lens, query, p["word_attention/compute/score/fc/weights"],
                                            p["word_attention/compute/score/fc/biases"], masks["word"])
    qv_embed = p["V_WordMap/embed_map"][q]
    q_v_ft, _ = O.fc_ln_relu_forward(qv_embed, p, "v_word_fc")                            # LN over [T, H]
    pooled_q_v = np.einsum("bt,bth->bh", w_att, q_v_ft)
    qv, _ = O.fc_ln_relu_forward(pooled_q_v, p, "q_linear_v")
    att, _ = O.hadamard_attention_forward(v, nb, qv, p["hadamard_attention/compute/score/fc/weights"],
                                          p["hadamard_attention/compute/score/fc/biases"], masks["att"])
    pooled = np.einsum("br,brd->bd", att, V)
    pl, _ = O.fc_ln_relu_forward(pooled, p, "pooled_linear_l")
    ll, _ = O.fc_ln_relu_forward(q_ft, p, "q_linear_l")
    j0, _ = O.fc_ln_relu_forward(pl * ll, p, "joint_fc")
    j = j0 * masks["joint"] * dt(1.0 / O.KEEP_JOINT)
    z = O.fc_forward(j, p["WordWeightAnswer/fc/weights"], p["WordWeightAnswer/fc/biases"])
    tgt = batch["answer_target"]
    ell = O.sigmoid_ce(z, tgt)
    train_loss = (ell * answer_masks["train"]).sum(1).mean()
    report_loss = ell.sum(1).mean()
    pred = np.argmax(z, axis=1).astype(np.int32)
    acc = tgt[np.arange(len(pred)), pred].mean()
    report = {"answer_train_loss": train_loss, "answer_report_loss": report_loss, "answer_accuracy": acc}
    mid = {"num_V_ft": nb, "v_linear_v": v, "q_L_map": q_map, "q_L_ft": q_ft, "q_att_key": key, "q_att_query": query,
           "w_att_score": w_att, "q_v_ft": q_v_ft, "pooled_q_v": pooled_q_v, "q_linear_v": qv, "att_score": att,
           "pooled_V_ft": pooled, "pooled_linear_l": pl, "l_linear_l": ll, "joint": j, "logit": z, "pred": pred}
    return train_loss, report, {"pred": pred, "att_score": att}, mid


# ------------------------------------------------------------------------------------------------------------------
# independently composed torch forward; its autograd is the gradient oracle
# ------------------------------------------------------------------------------------------------------------------
def torch_loss_and_grads(p, batch, table, nbox_table, answer_masks, masks, gates=None):
    """float64 torch restatement (nn.functional ops, packed reversal by index arithmetic instead of per-row slicing).
    Returns (loss, mid, grads, slices): grads for EVERY variable (the optimiser picks the train set); `slices` =
    {'LearnGloVe/embed_map': dE [B,T,W], 'V_WordMap/embed_map': dE2 [B,T,W]} the un-aggregated IndexedSlices values of
    the two embedding gradients (their sum of squares enters clip_by_global_norm)."""
    import torch
    import torch.nn.functional as F
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    P = {k: t64(v).requires_grad_(True) for k, v in p.items()}
    idx = torch.from_numpy(batch["image_idx"])
    q = torch.from_numpy(batch["q_intseq"]).long()
    lens = torch.from_numpy(batch["q_intseq_len"]).long()
    tgt = t64(batch["answer_target"])
    B, T = q.shape

    def fc_ln_relu(x, scope):
        pre = F.linear(x, P[scope + "/fc/weights"].t(), P[scope + "/fc/biases"])
        ln = F.layer_norm(pre, pre.shape[1:], eps=O.LN_EPS)              # statistics over every non-batch axis
        return torch.relu(ln * P[scope + "/LayerNorm/gamma"] + P[scope + "/LayerNorm/beta"])

    def gru(x, d):
        Wg, bg = P[GRU % (d, "gates") + "/kernel"], P[GRU % (d, "gates") + "/bias"]
        Wc, bc = P[GRU % (d, "candidate") + "/kernel"], P[GRU % (d, "candidate") + "/bias"]
        h = Wc.shape[1]
        s = x.new_zeros(B, h)
        outs = []
        for t in range(T):
            r, u = torch.sigmoid(torch.cat([x[:, t], s], 1) @ Wg + bg).split(h, dim=1)
            c = torch.tanh(torch.cat([x[:, t], r * s], 1) @ Wc + bc)
            new = u * s + (1 - u) * c
            live = (lens > t)[:, None]
            outs.append(torch.where(live, new, torch.zeros_like(new)))
            s = torch.where(live, new, s)
        return torch.stack(outs, 1), s

    # reverse_sequence as a gather: position t of row b reads len_b - 1 - t when t < len_b, itself otherwise
    ar = torch.arange(T)[None, :].expand(B, T)
    rev = torch.where(ar < lens[:, None], lens[:, None] - 1 - ar, ar)

    def hadamard(mem, mem_len, qry, scope, keep):
        feat = mem * qry[:, None, :] * t64(keep) / KEEP_ATT
        s = F.linear(feat, P[scope + "/fc/weights"].t(), P[scope + "/fc/biases"]).squeeze(-1)
        valid = torch.arange(mem.shape[1])[None, :] < mem_len[:, None]
        return torch.softmax(s.masked_fill(~valid, float("-inf")), dim=-1)

    V = t64(table).index_select(0, idx)
    nb = torch.from_numpy(nbox_table).long().index_select(0, idx)
    v = fc_ln_relu(V, "v_linear_v")
    e = F.embedding(q, P["LearnGloVe/embed_map"])
    e.retain_grad()
    of, sf = gru(e, "fw")
    ob, sb = gru(torch.gather(e, 1, rev[:, :, None].expand_as(e)), "bw")
    ob = torch.gather(ob, 1, rev[:, :, None].expand_as(ob))
    q_map, q_ft = torch.cat([of, ob], -1), torch.cat([sf, sb], -1)
    key = fc_ln_relu(q_map, "q_att_key")
    query = fc_ln_relu(q_ft, "q_att_query")
    w_att = hadamard(key, lens, query, "word_attention/compute/score", masks["word"])
    e2 = F.embedding(q, P["V_WordMap/embed_map"])
    e2.retain_grad()
    q_v_ft = fc_ln_relu(e2, "v_word_fc")
    pooled_q_v = torch.bmm(w_att.unsqueeze(1), q_v_ft).squeeze(1)
    qv = fc_ln_relu(pooled_q_v, "q_linear_v")
    att = hadamard(v, nb, qv, "hadamard_attention/compute/score", masks["att"])
    pooled = torch.bmm(att.unsqueeze(1), V).squeeze(1)
    pl = fc_ln_relu(pooled, "pooled_linear_l")
    ll = fc_ln_relu(q_ft, "q_linear_l")
    j = fc_ln_relu(pl * ll, "joint_fc") * t64(masks["joint"]) / O.KEEP_JOINT
    z = torch.addmm(P["WordWeightAnswer/fc/biases"], j, P["WordWeightAnswer/fc/weights"])
    loss = (F.binary_cross_entropy_with_logits(z, tgt, reduction="none") * t64(answer_masks["train"])).sum(-1).mean()
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else np.zeros(t.shape)) for k, t in P.items()}
    mid = {"q_L_map": q_map, "q_L_ft": q_ft, "q_att_key": key, "w_att_score": w_att, "q_v_ft": q_v_ft,
           "pooled_q_v": pooled_q_v, "q_linear_v": qv, "att_score": att, "pooled_V_ft": pooled, "joint": j, "logit": z}
    slices = {"LearnGloVe/embed_map": e.grad.numpy(), "V_WordMap/embed_map": e2.grad.numpy()}
    return float(loss.detach()), {k: t.detach().numpy() for k, t in mid.items()}, grads, slices


def global_norm(grads, train_names, slices):
    """clip_ops.global_norm over the train-variable gradients; embedding gradients are IndexedSlices whose norm is
    taken over the un-aggregated slice values (as in oracle/vqa_oracle.global_norm)."""
    acc = 0.0
    for n in train_names:
        g = slices[n] if n in slices else grads[n]
        acc += float((np.asarray(g, np.float64) ** 2).sum())
    return np.sqrt(acc)


def train_step(p, batch, table, nbox_table, answer_masks, masks, state, lr, model_type):
    """forward, backward, clip_by_global_norm(20), Adam on the model's train variables (vqa/trainer.py:87-114) in place."""
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    loss, mid, grads, slices = torch_loss_and_grads(to64(p), batch, table.astype(np.float64), nbox_table, to64(answer_masks),
                                                    to64(masks))
    names = train_var_names(p, model_type)
    norm = global_norm(grads, names, slices)
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


def make_masks(rng, B, R, T, H, dtype=np.float32):
    return {"att": (rng.random((B, R, H)) < KEEP_ATT).astype(dtype), "word": (rng.random((B, T, H)) < KEEP_ATT).astype(dtype),
            "joint": (rng.random((B, 2 * H)) < O.KEEP_JOINT).astype(dtype)}
