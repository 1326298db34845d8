"""CPU oracle for the cfg-5 pre-training model.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (no reference fixtures; TF 1.6 absent -- see oracle/vqa_oracle.py).  NumPy forward
restating vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py, and an independently composed torch
forward whose autograd supplies the reference gradients (the hand-derived backward of the fusion model in
oracle/vqa_oracle.py already pins the shared layer math).

Reference lines (relative to /root/reference):
  * Model.__init__ / build                vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:15-94
  * build_object_V_ft / build_attribute_V_ft   :323-364, :414-455   (spatial attention over 36 regions,
                                                                     V_ft / spatial_ft tiled x n entries)
  * build_object_wordset / attribute_wordset   :366-412, :457-503
  * build_object_blank_fill / attribute        :505-556, :558-609
  * n_way_classification_loss                  :675-706
  * modules: fc_layer vlmap/modules.py:630-650, hadamard_attention :67-97, attention_pooling :23-39,
    encode_L :124-140, learn_embedding_map :351-358, LearnGloVe :415-448
LayerNorm variables of an fc_layer scope that several call sites enter (modules.py:630-650, `layers.layer_norm(out)`
with no scope inside `with tf.variable_scope(scope, reuse=tf.AUTO_REUSE)`) -- the TF 1.x rule, the SAME one
oracle/conv_oracle.py applies to I2V's BatchNorm (DESIGN.md section 2 spells it out):
  * `variable_scope(None, default_name='LayerNorm')` uniquifies against VariableStore.variable_scopes_count of
    '<parent>/LayerNorm' (`_get_unique_variable_scope`);
  * leaving a STRING-named scope runs `close_variable_subscopes(name)`, which zeroes the counts of everything below it
    (`_pure_variable_scope.__exit__`), so the next call site that re-enters 'pooled_linear_l' finds the count of
    'pooled_linear_l/LayerNorm' at 0, gets the un-suffixed name again, and AUTO_REUSE hands it the existing beta / gamma.
  => ONE LayerNorm per shared FC, trained by every call site (`ln_shared=True`, the default here).
The per-call-site reading of SURVEY 5.1 (`LayerNorm`, `LayerNorm_1`, ... in graph build order: object_V_ft,
attribute_V_ft, object_blank_fill, attribute_blank_fill, object_wordset, attribute_wordset) is kept as the second mode:
the set of variable names in `p` decides (`.../LayerNorm_1/...` present = per call site), exactly as a reference
checkpoint would.
"""
from __future__ import annotations

import numpy as np

from . import vqa_oracle as O

TOP_K = 5
KINDS = ("obj", "attr")


def ln_name(scope, idx):
    return scope + ("/LayerNorm" if idx == 0 else "/LayerNorm_%d" % idx)


def ln_shared_in(names):
    """True when the variable names are those of the shared-LayerNorm graph (no `<scope>/LayerNorm_<k>/`)."""
    return not any("/LayerNorm_" in k for k in names)


def variable_shapes(Vq, n_ws, A, W=300, D=2048, H=1024, ln_shared=True):
    s = {"wordset_map/learn": (n_ws, W), "V_GloVe/embed_map": (Vq, W), "L_GloVe/embed_map": (Vq, W),
         "LearnAnswerGloVe/embed_map": (A, W)}

    def fc(scope, fin, fout, n_ln):
        s[scope + "/fc/weights"] = (fin, fout)
        s[scope + "/fc/biases"] = (fout,)
        for i in range(min(n_ln, 1) if ln_shared else n_ln):
            s[ln_name(scope, i) + "/beta"] = (fout,)
            s[ln_name(scope, i) + "/gamma"] = (fout,)

    fc("spat_v_linear_v", 6, H, 2)
    fc("spat_q_linear_v", 6, H, 2)
    fc("spat_att/compute/score", H, 1, 0)
    s["encode_L_blank/rnn/gru_cell/gates/kernel"] = (W + H, 2 * H)
    s["encode_L_blank/rnn/gru_cell/gates/bias"] = (2 * H,)
    s["encode_L_blank/rnn/gru_cell/candidate/kernel"] = (W + H, H)
    s["encode_L_blank/rnn/gru_cell/candidate/bias"] = (H,)
    fc("pooled_linear_l", D, H, 4)
    fc("q_linear_l", H, H, 4)
    fc("joint_fc", H, 2 * H, 4)
    fc("wordset_ft", W, H, 2)
    fc("classifier", 2 * H, A, 0)
    return s


# variables that exist in the graph but receive no gradient in this model
NO_GRAD_VARS = ("V_GloVe/embed_map", "LearnAnswerGloVe/embed_map")


def init_params(rng, Vq, n_ws, A, W=300, D=2048, H=1024, dtype=np.float32, perturb=True, ln_shared=True):
    p = {}
    for n, shp in variable_shapes(Vq, n_ws, A, W, D, H, ln_shared).items():
        if n.endswith("/weights") or n.endswith("/kernel"):
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            p[n] = rng.uniform(-lim, lim, size=shp)
        elif n.endswith("gates/bias") or n.endswith("/gamma"):
            p[n] = np.ones(shp)
        elif n.endswith("embed_map") or n.endswith("/learn"):
            p[n] = rng.uniform(-0.01, 0.01, size=shp)
        else:
            p[n] = np.zeros(shp)
        if perturb and (n.endswith("/beta") or n.endswith("/gamma") or n.endswith("/biases")):
            p[n] = p[n] + 0.1 * rng.standard_normal(shp)
        p[n] = p[n].astype(dtype)
    return p


def make_batch(rng, B, n, R, D, L, Vq, n_ws, A, dtype=np.float32):
    """Batch dict of vlmap_memft/datasets/dataset_vlmap.py:128-236 (the fields this model reads)."""
    b = {"image_ft": np.maximum(rng.standard_normal((B, R, D)), 0).astype(dtype),
         "num_boxes": rng.integers(max(R // 2, 1), R + 1, size=B).astype(np.int32)}
    ys = np.sort(rng.random((B, R, 2)), -1)
    xs = np.sort(rng.random((B, R, 2)), -1)
    nb = np.stack([ys[..., 0], xs[..., 0], ys[..., 1], xs[..., 1]], -1)
    b["spatial_ft"] = np.concatenate([nb, nb[..., 2:3] - nb[..., 0:1], nb[..., 3:4] - nb[..., 1:2]], -1).astype(dtype)
    for k in KINDS:
        ys = np.sort(rng.random((B, n, 2)), -1)
        xs = np.sort(rng.random((B, n, 2)), -1)
        b[k + "_blank_fill/normal_boxes"] = np.stack([ys[..., 0], xs[..., 0], ys[..., 1], xs[..., 1]], -1).astype(dtype)
        b[k + "_blank_fill/fills"] = rng.integers(0, A, size=(B, n)).astype(np.int32)
        lens = rng.integers(1, L + 1, size=(B, n)).astype(np.int32)
        blanks = rng.integers(1, Vq, size=(B, n, L)).astype(np.int32)
        blanks[np.arange(L)[None, None, :] >= lens[..., None]] = 0
        b[k + "_blank_fill/blanks"], b[k + "_blank_fill/blanks_len"] = blanks, lens
        b[k + "_blank_fill/wordsets"] = rng.integers(0, n_ws, size=(B, n)).astype(np.int32)
        b[k + "_blank_fill/num"] = rng.integers(1, n + 1, size=B).astype(np.int32)
    return b


def make_masks(rng, B, n, R, H, dtype=np.float32):
    m = {}
    for k in KINDS:
        m[k + "/att"] = (rng.random((B * n, R, H)) < O.KEEP_ATT).astype(dtype)
        m[k + "/bf_joint"] = (rng.random((B, n, 2 * H)) < O.KEEP_JOINT).astype(dtype)
        m[k + "/ws_joint"] = (rng.random((B, n, 2 * H)) < O.KEEP_JOINT).astype(dtype)
    return m


def _fc_ln(x, p, scope, ln_idx, act):
    """modules.fc_layer: FC on the last axis, layer_norm over ALL non-batch axes, activation."""
    pre = x @ p[scope + "/fc/weights"] + p[scope + "/fc/biases"]
    if ln_shared_in(p):
        ln_idx = 0
    ln, _, _ = O.layer_norm_forward(pre, p[ln_name(scope, ln_idx) + "/gamma"], p[ln_name(scope, ln_idx) + "/beta"])
    return np.maximum(ln, 0) if act == "relu" else np.tanh(ln)


def n_way_classification_loss(logits, labels, valid):
    """:675-706 with a mask.  logits [B,n,A], labels int [B,n], valid float [B,n]."""
    z = logits - logits.max(-1, keepdims=True)
    lse = np.log(np.exp(z).sum(-1))
    ce = lse - np.take_along_axis(z, labels[..., None], -1)[..., 0]
    den = valid.sum()
    loss = (ce * valid).sum() / den
    pred = logits.argmax(-1)
    acc = ((pred == labels) * valid).sum() / den
    zl = np.take_along_axis(logits, labels[..., None], -1)
    idx = np.arange(logits.shape[-1])
    before = ((logits > zl) | ((logits == zl) & (idx < labels[..., None]))).sum(-1)   # tf.nn.top_k tie order
    topk = ((before < TOP_K) * valid).sum() / den
    return loss, acc, topk


def forward(p, batch, masks, n):
    dt = batch["image_ft"].dtype.type
    B, R, D = batch["image_ft"].shape
    report, losses, mid = {}, {}, {}
    for ki, k in enumerate(KINDS):
        # build_{object,attribute}_V_ft
        key = batch[k + "_blank_fill/normal_boxes"]
        key6 = np.concatenate([key, key[..., 2:3] - key[..., 0:1], key[..., 3:4] - key[..., 1:2]], -1)
        v = _fc_ln(batch["spatial_ft"], p, "spat_v_linear_v", ki, "relu")          # [B,R,H]; identical for the n tiles
        qv = _fc_ln(key6, p, "spat_q_linear_v", ki, "relu")                        # [B,n,H], LN over (n,H)
        vt = np.repeat(v, n, axis=0)                                               # [B*n,R,H]
        att, _ = O.hadamard_attention_forward(vt, np.repeat(batch["num_boxes"], n), qv.reshape(B * n, -1),
                                              p["spat_att/compute/score/fc/weights"],
                                              p["spat_att/compute/score/fc/biases"], masks[k + "/att"])
        pooled = np.einsum("qr,qrd->qd", att, np.repeat(batch["image_ft"], n, axis=0)).reshape(B, n, D)
        mid[k + "/att"], mid[k + "/pooled_V_ft"] = att, pooled
        valid = (np.arange(n)[None, :] < batch[k + "_blank_fill/num"][:, None]).astype(pooled.dtype)
        fills = batch[k + "_blank_fill/fills"].astype(np.int64)

        def head(l_ft, ln_idx, jmask):
            vl = _fc_ln(pooled, p, "pooled_linear_l", ln_idx, "relu")
            ll = _fc_ln(l_ft, p, "q_linear_l", ln_idx, "relu")
            j = _fc_ln(vl * ll, p, "joint_fc", ln_idx, "relu") * jmask * dt(1.0 / O.KEEP_JOINT)
            return j @ p["classifier/fc/weights"] + p["classifier/fc/biases"]

        # build_{object,attribute}_blank_fill
        blanks = batch[k + "_blank_fill/blanks"]
        L = blanks.shape[-1]
        e = p["L_GloVe/embed_map"][blanks.reshape(B * n, L)]
        bf, _ = O.gru_forward(e, batch[k + "_blank_fill/blanks_len"].reshape(-1),
                              p["encode_L_blank/rnn/gru_cell/gates/kernel"], p["encode_L_blank/rnn/gru_cell/gates/bias"],
                              p["encode_L_blank/rnn/gru_cell/candidate/kernel"],
                              p["encode_L_blank/rnn/gru_cell/candidate/bias"])
        logit = head(bf.reshape(B, n, -1), ki, masks[k + "/bf_joint"])
        loss, acc, topk = n_way_classification_loss(logit, fills, valid)
        losses[k + "_blank_fill"] = loss
        report[k + "_blank_fill_loss"], report[k + "_blank_fill_acc"] = loss, acc
        report[k + "_blank_fill_top_%d_acc" % TOP_K] = topk
        mid[k + "/bf_logit"] = logit
        # build_{object,attribute}_wordset
        ws = np.tanh(p["wordset_map/learn"][batch[k + "_blank_fill/wordsets"]])
        wf = _fc_ln(ws, p, "wordset_ft", ki, "tanh")
        logit = head(wf, 2 + ki, masks[k + "/ws_joint"])
        loss, acc, topk = n_way_classification_loss(logit, fills, valid)
        losses[k + "_wordset"] = loss
        report[k + "_wordset_loss"], report[k + "_wordset_acc"] = loss, acc
        report[k + "_wordset_top_%d_acc" % TOP_K] = topk
        mid[k + "/ws_logit"] = logit
    total = sum(losses.values())
    report["total_loss"] = total
    return total, report, mid


# ----------------------------------------------------------------------------- torch restatement
RELU_SITES = tuple("%s/%s" % (k, s) for k in KINDS for s in ("v", "qv", "bf/vl", "bf/ll", "bf/j", "ws/vl", "ws/ll", "ws/j"))


def torch_loss_and_grads(p, batch, masks, n, dtype=None, gates=None, capture=None):
    """Independent torch composition (F.linear / manual LN / torch GRU loop / F.cross_entropy) + autograd.
    Returns (total_loss, report-losses, grads dict, embedding slice grads dict).

    gates: optional {site: bool array} for the ReLU sites in RELU_SITES ("<kind>/v", "<kind>/qv",
    "<kind>/<bf|ws>/{vl,ll,j}").  relu(x) is then evaluated as x * gate with the GIVEN sign pattern -- the
    gate-conditioned gradient: with the gates of a float32 forward pass fixed, the loss is a smooth function of the
    parameters and a float32 backward must agree with this float64 one to rounding, whereas the unconditioned
    comparison also sees every gate on which float32 and float64 disagree (pre-activations within ~1e-6 of 0).
    capture: optional dict that receives this run's own sign pattern per site."""
    import torch
    import torch.nn.functional as F
    dtype = dtype or torch.float64
    P = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in p.items()}
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    B, R, D = batch["image_ft"].shape
    img, spat = t(batch["image_ft"]), t(batch["spatial_ft"])

    shared = ln_shared_in(p)

    def fc_ln(x, scope, i, act, site=None):
        i = 0 if shared else i
        pre = F.linear(x, P[scope + "/fc/weights"].t(), P[scope + "/fc/biases"])
        dims = tuple(range(1, pre.dim()))
        mu = pre.mean(dims, keepdim=True)
        var = pre.var(dims, unbiased=False, keepdim=True)
        ln = (pre - mu) * torch.rsqrt(var + O.LN_EPS) * P[ln_name(scope, i) + "/gamma"] + P[ln_name(scope, i) + "/beta"]
        if act == "relu" and capture is not None:
            capture[site] = (ln.detach() > 0).numpy()
        if act == "relu" and gates is not None:
            return ln * torch.as_tensor(np.asarray(gates[site]).reshape(tuple(ln.shape))).to(dtype)
        return torch.relu(ln) if act == "relu" else torch.tanh(ln)

    def gru(x, lens):
        Wg, bg = P["encode_L_blank/rnn/gru_cell/gates/kernel"], P["encode_L_blank/rnn/gru_cell/gates/bias"]
        Wc, bc = P["encode_L_blank/rnn/gru_cell/candidate/kernel"], P["encode_L_blank/rnn/gru_cell/candidate/bias"]
        H = Wc.shape[1]
        h = x.new_zeros(x.shape[0], H)
        for s in range(x.shape[1]):
            g = torch.sigmoid(torch.cat([x[:, s], h], 1) @ Wg + bg)
            r, u = g.split(H, 1)
            c = torch.tanh(torch.cat([x[:, s], r * h], 1) @ Wc + bc)
            h = torch.where((lens > s)[:, None], u * h + (1 - u) * c, h)
        return h

    losses, slices = {}, {}
    total = 0
    for ki, k in enumerate(KINDS):
        key = t(batch[k + "_blank_fill/normal_boxes"])
        key6 = torch.cat([key, key[..., 2:3] - key[..., 0:1], key[..., 3:4] - key[..., 1:2]], -1)
        v = fc_ln(spat, "spat_v_linear_v", ki, "relu", k + "/v")
        qv = fc_ln(key6, "spat_q_linear_v", ki, "relu", k + "/qv").reshape(B * n, -1)
        vt = v.repeat_interleave(n, 0)
        feat = vt * qv[:, None, :] * t(masks[k + "/att"]) / O.KEEP_ATT
        s = F.linear(feat, P["spat_att/compute/score/fc/weights"].t(), P["spat_att/compute/score/fc/biases"])[..., 0]
        nbv = torch.tensor(np.repeat(batch["num_boxes"], n))
        s = torch.where(torch.arange(R)[None, :] < nbv[:, None], s, torch.full_like(s, float("-inf")))
        att = torch.softmax(s, -1)
        pooled = torch.bmm(att[:, None, :], img.repeat_interleave(n, 0))[:, 0].reshape(B, n, D)
        valid = t((np.arange(n)[None, :] < batch[k + "_blank_fill/num"][:, None]).astype(np.float64))
        fills = torch.tensor(batch[k + "_blank_fill/fills"].astype(np.int64))

        def head(l_ft, i, jm):
            site = "%s/%s/" % (k, "bf" if i < 2 else "ws")
            vl = fc_ln(pooled, "pooled_linear_l", i, "relu", site + "vl")
            ll = fc_ln(l_ft, "q_linear_l", i, "relu", site + "ll")
            j = fc_ln(vl * ll, "joint_fc", i, "relu", site + "j") * t(jm) / O.KEEP_JOINT
            z = F.linear(j, P["classifier/fc/weights"].t(), P["classifier/fc/biases"])
            ce = F.cross_entropy(z.reshape(B * n, -1), fills.reshape(-1), reduction="none").reshape(B, n)
            return (ce * valid).sum() / valid.sum()

        blanks = torch.tensor(batch[k + "_blank_fill/blanks"].astype(np.int64)).reshape(B * n, -1)
        e = F.embedding(blanks, P["L_GloVe/embed_map"])
        e.retain_grad()
        slices[k + "/blank_embed"] = e
        bf = gru(e, torch.tensor(batch[k + "_blank_fill/blanks_len"].reshape(-1).astype(np.int64)))
        losses[k + "_blank_fill"] = head(bf.reshape(B, n, -1), ki, masks[k + "/bf_joint"])
        wse = F.embedding(torch.tensor(batch[k + "_blank_fill/wordsets"].astype(np.int64)), P["wordset_map/learn"])
        wse.retain_grad()
        slices[k + "/wordset_embed"] = wse
        wf = fc_ln(torch.tanh(wse), "wordset_ft", ki, "tanh")
        losses[k + "_wordset"] = head(wf, 2 + ki, masks[k + "/ws_joint"])
    for vloss in losses.values():
        total = total + vloss
    total.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in P.items()}
    return float(total.detach()), {k: float(v.detach()) for k, v in losses.items()}, grads, \
        {k: v.grad.numpy() for k, v in slices.items()}
