"""Second, independently composed CPU restatement (torch ops + autograd).

TEST INFRASTRUCTURE ONLY (same rules as oracle/vqa_oracle.py: importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
product path).  PARITY UNPINNED by the reference (no fixtures exist);
this file exists so that the NumPy oracle's hand-written backward is checked
against autograd of a differently-composed forward (F.linear, manual LN over
dims (1,2), a python GRU loop with torch.where masking).

Reference lines: vqa/model_vlmap_answer.py:102-288, vlmap/modules.py:23-39,
67-97,124-140,630-650; TF semantics per SURVEY.md 5.2.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import vqa_oracle as O


def _t(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def params_to_torch(params, dtype=torch.float32, requires_grad=True):
    out = {}
    for k, v in params.items():
        t = _t(v, dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def _fc_ln_relu(x, P, scope):
    pre = F.linear(x, P[scope + "/fc/weights"].t(), P[scope + "/fc/biases"])
    dims = tuple(range(1, pre.dim()))
    mu = pre.mean(dim=dims, keepdim=True)
    var = pre.var(dim=dims, unbiased=False, keepdim=True)
    ln = (pre - mu) * torch.rsqrt(var + O.LN_EPS) * P[scope + "/LayerNorm/gamma"] + P[scope + "/LayerNorm/beta"]
    return torch.relu(ln)


def _gru(x, lens, Wg, bg, Wc, bc):
    B, T, _ = x.shape
    H = Wc.shape[1]
    h = x.new_zeros(B, H)
    for t in range(T):
        xt = x[:, t]
        g = torch.sigmoid(torch.cat([xt, h], 1) @ Wg + bg)
        r, u = g.split(H, dim=1)
        c = torch.tanh(torch.cat([xt, r * h], 1) @ Wc + bc)
        hn = u * h + (1 - u) * c
        h = torch.where((lens > t)[:, None], hn, h)
    return h


def forward(P, batch, table, nbox_table, answer_masks, masks, model_type="vlmap_answer",
            dtype=torch.float32):
    sc = O.scope_names(model_type)
    idx = torch.from_numpy(batch["image_idx"])
    q = torch.from_numpy(batch["q_intseq"]).long()
    lens = torch.from_numpy(batch["q_intseq_len"]).long()
    tgt = _t(batch["answer_target"], dtype)
    tab = table if torch.is_tensor(table) else _t(table, dtype)
    nbt = nbox_table if torch.is_tensor(nbox_table) else torch.from_numpy(nbox_table).long()
    m_att = masks["att"] if torch.is_tensor(masks["att"]) else _t(masks["att"], dtype)
    m_j = masks["joint"] if torch.is_tensor(masks["joint"]) else _t(masks["joint"], dtype)
    V = tab.index_select(0, idx)
    nb = nbt.index_select(0, idx)
    v = _fc_ln_relu(V, P, sc["v_linear_v"])
    e = F.embedding(q, P[sc["embed"]])
    h = _gru(e, lens, P[sc["gru_gates"] + "/kernel"], P[sc["gru_gates"] + "/bias"],
             P[sc["gru_cand"] + "/kernel"], P[sc["gru_cand"] + "/bias"])
    qv = _fc_ln_relu(h, P, sc["q_linear_v"])
    feat = v * qv.unsqueeze(1) * m_att / O.KEEP_ATT
    s = F.linear(feat, P[sc["score"] + "/fc/weights"].t(), P[sc["score"] + "/fc/biases"]).squeeze(-1)
    R = V.shape[1]
    valid = torch.arange(R)[None, :] < nb[:, None]
    s = torch.where(valid, s, torch.full_like(s, float("-inf")))
    att = torch.softmax(s, dim=-1)
    extra_loss, extra_mid = 0.0, {}
    if model_type == "vlmap_answer_adapt":          # vqa/model_vlmap_answer_adapt.py:132-142
        va = _fc_ln_relu(V, P, sc["v_adapt"])
        p = (att.unsqueeze(-1) * va).sum(1)
        extra_mid["v_adapt"] = va
    else:
        p = torch.bmm(att.unsqueeze(1), V).squeeze(1)
    pl = _fc_ln_relu(p, P, sc["pooled_linear_l"])
    lin_in = h
    if model_type == "vlmap_answer2":               # vqa/model_vlmap_answer2.py:127-131: FC + LN + tanh
        pre = torch.addmm(P[sc["q_L_ft2"] + "/fc/biases"], h, P[sc["q_L_ft2"] + "/fc/weights"])
        lin_in = torch.tanh(F.layer_norm(pre, pre.shape[1:], eps=O.LN_EPS) * P[sc["q_L_ft2"] + "/LayerNorm/gamma"]
                            + P[sc["q_L_ft2"] + "/LayerNorm/beta"])
    elif model_type in ("vlmap_answer_no_noise", "vlmap_answer_full"):
        qm = torch.addmm(P[sc["q_L_mean"] + "/fc/biases"], h, P[sc["q_L_mean"] + "/fc/weights"])
        lin_in = qm
        extra_mid["q_L_mean"] = qm
        if model_type == "vlmap_answer_full":       # vqa/model_vlmap_answer_full.py:128-134, 272-276
            qs = torch.addmm(P[sc["q_L_log_sigma_sq"] + "/fc/biases"], h, P[sc["q_L_log_sigma_sq"] + "/fc/weights"])
            noise = masks["noise"] if torch.is_tensor(masks["noise"]) else _t(masks["noise"], dtype)
            lin_in = qm + noise * torch.sqrt(torch.exp(qs))
            kl = -0.5 * torch.sum(1 + qs - qm.pow(2) - qs.exp(), dim=-1).mean()
            extra_loss = extra_loss + O.LATENT_LOSS_WEIGHT * kl
            extra_mid.update(q_L_log_sigma_sq=qs, q_L_mean_noise=lin_in, latent_loss=kl)
    ll = _fc_ln_relu(lin_in, P, sc["q_linear_l"])
    if model_type in O.NOC_FAMILY:        # vqa/model_vlmap_answer_noc.py:177-204, composed independently
        m_jl = masks["joint_l"] if torch.is_tensor(masks["joint_l"]) else _t(masks["joint_l"], dtype)
        vj = _fc_ln_relu(pl, P, sc["joint_v"]) * m_j / O.KEEP_JOINT
        lj = _fc_ln_relu(ll, P, sc["joint_l"]) * m_jl / O.KEEP_JOINT
        z = torch.addmm(P[sc["headV"] + "/fc/biases"], vj, P[sc["headV"] + "/fc/weights"]) + \
            torch.addmm(P[sc["headL"] + "/fc/biases"], lj, P[sc["headL"] + "/fc/weights"])
        loss = (F.binary_cross_entropy_with_logits(z, tgt, reduction="none") * _t(answer_masks["train"], dtype)).sum(-1).mean()
        return loss, {"v_linear_v": v, "condition": h, "q_linear_v": qv, "att_score": att, "pooled_V_ft": p,
                      "pooled_linear_l": pl, "l_linear_l": ll, "v_joint": vj, "l_joint": lj, "logit": z, "embed": e}
    j = _fc_ln_relu(pl * ll, P, sc["joint_fc"]) * m_j / O.KEEP_JOINT
    z = F.linear(j, P[sc["head"] + "/fc/weights"].t(), P[sc["head"] + "/fc/biases"])
    if model_type == "standard_word2vec":
        z = z @ P[sc["glove"]].detach()
    if model_type == "vlmap_answer_vqa_all":        # vqa/model_vlmap_answer_vqa_all.py:188-244, composed independently
        z2 = F.linear(j, P[sc["head2"] + "/fc/weights"].t(), P[sc["head2"] + "/fc/biases"])
        train, exist = _t(answer_masks["train"], dtype), _t(answer_masks["exist"], dtype)
        z1m = torch.where(exist > 0, z, z.amin(dim=1, keepdim=True).expand_as(z))
        per_answer = (F.binary_cross_entropy_with_logits(z1m, tgt, reduction="none")
                      + F.binary_cross_entropy_with_logits(z1m + z2, tgt, reduction="none")) * train
        loss = per_answer.sum(-1).mean()
        mid = {"v_linear_v": v, "condition": h, "q_linear_v": qv, "att_score": att, "pooled_V_ft": p,
               "pooled_linear_l": pl, "l_linear_l": ll, "joint": j, "logit": z1m + z2, "embed": e,
               "pred": torch.argmax(z1m + z2, dim=-1)}
        return loss, mid
    if model_type == "vlmap_answer_vqa_all2":       # vqa/model_vlmap_answer_vqa_all2.py:216-241, composed independently
        z2 = F.linear(j, P[sc["head2"] + "/fc/weights"].t(), P[sc["head2"] + "/fc/biases"])
        train = _t(answer_masks["train"], dtype)
        per_answer = (F.binary_cross_entropy_with_logits(z, tgt, reduction="none") * train
                      + F.binary_cross_entropy_with_logits(z2, tgt, reduction="none"))
        loss = per_answer.sum(-1).mean()
        mid = {"v_linear_v": v, "condition": h, "q_linear_v": qv, "att_score": att, "pooled_V_ft": p,
               "pooled_linear_l": pl, "l_linear_l": ll, "joint": j, "logit": z + z2, "embed": e,
               "pred": torch.argmax(z * (1 - train) + z2 * train, dim=-1),
               "tuned_l_linear_l": _fc_ln_relu(h, P, sc["tuned_q_linear_l"])}
        return loss, mid
    ell = F.binary_cross_entropy_with_logits(z, tgt, reduction="none")
    if model_type in O.TRAIN_MASKED_LOSS:
        loss = (ell * _t(answer_masks["train"], dtype)).sum(-1).mean()
    else:
        loss = ell.sum(-1).mean()
    if model_type == "vlmap_answer_ent":            # vqa/model_vlmap_answer_ent.py:191-211, 281-292, with the TF ops' torch twins
        m_t = masks["tile_joint"] if torch.is_tensor(masks["tile_joint"]) else _t(masks["tile_joint"], dtype)
        B_, M = m_t.shape[0], m_t.shape[1]
        tile = pl.detach().repeat(M, 1).reshape(B_, M, -1)                   # tf.tile([M, 1]) then reshape [-1, M, L]
        tj = _fc_ln_relu(tile * ll.unsqueeze(1), P, sc["joint_fc"]) * m_t / O.KEEP_JOINT
        tz = F.linear(tj, P[sc["head"] + "/fc/weights"].t(), P[sc["head"] + "/fc/biases"])
        keep = torch.from_numpy((np.asarray(answer_masks["exist"]) * np.asarray(answer_masks["train"])) > 0.5)
        prob = torch.softmax(tz[:, :, keep], dim=-1)
        marg = prob.mean(dim=1)
        neg_ent = (marg * torch.log(marg + 1e-8)).sum(-1).mean()
        loss = loss + O.W_ENTROPY * neg_ent
        extra_mid.update(marginal_prob=marg, entropy=neg_ent)
    loss = loss + extra_loss
    mid = {"v_linear_v": v, "condition": lin_in if model_type == "vlmap_answer2" else h, "q_linear_v": qv, "att_score": att,
           "pooled_V_ft": p, "pooled_linear_l": pl, "l_linear_l": ll, "joint": j, "logit": z, "embed": e}
    mid.update(extra_mid)
    return loss, mid


def loss_and_grads(params, batch, table, nbox_table, answer_masks, masks, model_type="vlmap_answer",
                   dtype=torch.float64):
    """Returns (loss, mid, grads-as-numpy) via autograd; also the un-aggregated
    embedding slice gradient dx [B,T,W]."""
    P = params_to_torch(params, dtype)
    loss, mid = forward(P, batch, table, nbox_table, answer_masks, masks, model_type, dtype)
    mid["embed"].retain_grad()
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in P.items()}
    dx = mid["embed"].grad.numpy()
    return float(loss.detach()), {k: v.detach().numpy() for k, v in mid.items()}, grads, dx


@torch.no_grad()
def _adam_(p, g, m, v, lr_t):
    m.mul_(O.ADAM_B1).add_(g, alpha=1 - O.ADAM_B1)
    v.mul_(O.ADAM_B2).addcmul_(g, g, value=1 - O.ADAM_B2)
    p.sub_(lr_t * m / (v.sqrt() + O.ADAM_EPS))


class CpuTrainStep:
    """fp32 torch-CPU train step used as bench.py's cpu_baseline ('port'): the
    same math as the HIP path (forward, autograd backward, global-norm clip with
    un-aggregated embedding slices, Adam) on the host cores."""

    def __init__(self, params, table, nbox_table, answer_masks, model_type="vlmap_answer", lr=1e-3):
        self.model_type = model_type
        self.P = params_to_torch(params, torch.float32)
        self.table = _t(table, torch.float32)
        self.nbox = torch.from_numpy(np.asarray(nbox_table)).long()
        self.answer_masks = answer_masks
        self.train_names = O.train_var_names(params, model_type)
        for k, v in self.P.items():
            v.requires_grad_(k in self.train_names)
        self.m = {k: torch.zeros_like(self.P[k]) for k in self.train_names}
        self.v = {k: torch.zeros_like(self.P[k]) for k in self.train_names}
        self.step = 0
        self.lr = lr
        self.embed_name = O.scope_names(model_type)["embed"]

    def __call__(self, batch, masks):
        for k in self.train_names:
            self.P[k].grad = None
        loss, mid = forward(self.P, batch, self.table, self.nbox, self.answer_masks, masks,
                            self.model_type, torch.float32)
        mid["embed"].retain_grad()
        loss.backward()
        sq = 0.0
        for k in self.train_names:
            g = mid["embed"].grad if k == self.embed_name else self.P[k].grad
            sq += float((g.double() ** 2).sum())
        norm = sq ** 0.5
        scale = O.CLIP_NORM / max(norm, O.CLIP_NORM)
        self.step += 1
        lr_t = self.lr * (1 - O.ADAM_B2 ** self.step) ** 0.5 / (1 - O.ADAM_B1 ** self.step)
        for k in self.train_names:
            _adam_(self.P[k].data, self.P[k].grad * scale, self.m[k], self.v[k], lr_t)
        return float(loss.detach()), norm
