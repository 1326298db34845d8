"""cfg-5 pre-training ("task discovery") model on MI355X: the counterpart of
vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:54-94, 323-609, 675-706.

Spatial attention over the 36 regions for n = 5 annotated boxes per image, blank-fill (GRU over the
caption with a blank) and word-set conditioned heads for objects and attributes, all four heads
sharing pooled_linear_l / q_linear_l / joint_fc / classifier with ONE LayerNorm variable set per call
site (TF's un-scoped layer_norm: LayerNorm, LayerNorm_1, ... in graph build order), softmax-CE over
the obj3000+attr1000 answers with top-1 / top-5 accuracy.  Forward and the hand-derived backward are
composed here from the C-ABI ops (libvqahot.so); the effective batch is B*n rows and the x n tile of
V_ft / spatial_ft that the reference materialises (:324-333) never exists: the attention kernels take
`rep = n` queries per memory and v_linear_v of the (identical) tiles is computed once per image.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib, ops

TOP_K = 5
KINDS = ("obj", "attr")
KEEP_ATT, KEEP_JOINT = 0.8, 0.5
ADAM_B1, ADAM_B2, ADAM_EPS, CLIP_NORM = 0.9, 0.999, 1e-8, 20.0
NO_GRAD_VARS = ("V_GloVe/embed_map", "LearnAnswerGloVe/embed_map")      # created for export only
SPARSE_VARS = ("L_GloVe/embed_map", "wordset_map/learn")                # IndexedSlices gradients


def ln_name(scope, idx):
    return scope + ("/LayerNorm" if idx == 0 else "/LayerNorm_%d" % idx)


def variable_shapes(Vq, n_ws, A, W=300, D=2048, H=1024):
    s = {"wordset_map/learn": (n_ws, W), "V_GloVe/embed_map": (Vq, W), "L_GloVe/embed_map": (Vq, W),
         "LearnAnswerGloVe/embed_map": (A, W)}

    def fc(scope, fin, fout, n_ln):
        s[scope + "/fc/weights"] = (fin, fout)
        s[scope + "/fc/biases"] = (fout,)
        for i in range(n_ln):
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


def init_random_params(rng, Vq, n_ws, A, W=300, D=2048, H=1024):
    """Random-init weights of the architecture (Xavier-uniform FCs, GRU gate bias 1, LN gamma 1,
    embeddings U(-0.01, 0.01); GloVe vectors are download-only)."""
    p = {}
    for n, shp in variable_shapes(Vq, n_ws, A, W, D, H).items():
        if n.endswith("/weights") or n.endswith("/kernel"):
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            p[n] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
        elif n.endswith("gates/bias") or n.endswith("/gamma"):
            p[n] = np.ones(shp, np.float32)
        elif n.endswith("embed_map") or n.endswith("/learn"):
            p[n] = rng.uniform(-0.01, 0.01, size=shp).astype(np.float32)
        else:
            p[n] = np.zeros(shp, np.float32)
    return p


def add_length_sort(batch):
    """Host-side: for each blank-fill category the permutation that orders the B*n captions by length (longest
    first), its inverse and live_rows[t] = #captions longer than t.  The engine then embeds / encodes the
    captions in that order, runs every GRU step on the live prefix only, and un-permutes the final states."""
    for k in ("obj", "attr"):
        kl, kb = k + "_blank_fill/blanks_len", k + "_blank_fill/blanks"
        if kl not in batch or torch.is_tensor(batch[kl]):
            continue
        lens = np.asarray(batch[kl]).reshape(-1).astype(np.int64)
        L = int(np.asarray(batch[kb]).shape[-1])
        perm = np.argsort(-lens, kind="stable")
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        sl = np.clip(lens[perm], 0, L)
        batch[k + "_blank_fill/sort"] = {"perm": perm, "inv": inv,
                                         "live_rows": (sl[None, :] > np.arange(L)[:, None]).sum(1).astype(np.int32)}
    return batch


def _pad4(n):
    return (n + 3) // 4 * 4


class PretrainEngine:
    def __init__(self, *, n, R, D, H, W, A, Vq, n_ws, params, device="cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.VqaHotError("PretrainEngine needs a GPU (no CPU fallback)")
        self.device = torch.device(device)
        self.n, self.R, self.D, self.H, self.W, self.A = n, R, D, H, W, A
        self.shapes = variable_shapes(Vq, n_ws, A, W, D, H)
        dense = sorted(k for k in self.shapes if k not in NO_GRAD_VARS and k not in SPARSE_VARS)
        self.train_names = list(SPARSE_VARS) + dense
        off, self._tab = 0, {}
        for k in self.train_names:
            cnt = int(np.prod(self.shapes[k]))
            self._tab[k] = (off, cnt)
            off += _pad4(cnt)
        self.n_train = off
        self.sparse_floats = sum(_pad4(int(np.prod(self.shapes[k]))) for k in SPARSE_VARS)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.train_flat = torch.zeros(self.n_train, **f32)
        self.grad_flat = torch.zeros(self.n_train + 4, **f32)      # tail slot 0: un-aggregated slice sum of squares
        self.m_flat, self.v_flat = torch.zeros(self.n_train, **f32), torch.zeros(self.n_train, **f32)
        self.norm_sq = torch.zeros(4, **f32)
        self.sumsq_ws = torch.zeros(int(self.lib.vqa_sumsq_workspace_floats(self.n_train)) + 4, **f32)
        self.params, self.grads = {}, {}
        for k, (o, cnt) in self._tab.items():
            self.params[k] = self.train_flat[o:o + cnt].view(self.shapes[k])
            self.grads[k] = self.grad_flat[o:o + cnt].view(self.shapes[k])
        for k in NO_GRAD_VARS:
            self.params[k] = torch.zeros(self.shapes[k], **f32)
        for k in self.shapes:
            self.params[k].copy_(torch.as_tensor(np.asarray(params[k])).to(torch.float32))
        self.step_count = 0
        self.report = {}

    # ------------------------------------------------------------------ layer helpers
    def _fc_ln(self, x2d, scope, ln_idx, rows, act, keep=None, keep_prob=1.0):
        p = self.params
        pre = ops.gemm(x2d, p[scope + "/fc/weights"], bias=p[scope + "/fc/biases"])
        y, mean, rstd = ops.ln_act_fwd(pre, p[ln_name(scope, ln_idx) + "/gamma"], p[ln_name(scope, ln_idx) + "/beta"],
                                       rows, act, keep, keep_prob)
        return y, (x2d, pre, mean, rstd, scope, ln_idx, rows, act, keep, keep_prob)

    def _acc(self, name, value):
        """first contribution overwrites (grad buffers are not cleared), later ones add"""
        g = self.grads[name]
        if name in self._touched:
            ops.add_inplace(g.view(-1), value.reshape(-1).contiguous())
        else:
            g.copy_(value.view(g.shape))
            self._touched.add(name)

    def _fc_ln_bwd(self, dy, tape, need_dx=True):
        x2d, pre, mean, rstd, scope, ln_idx, rows, act, keep, keep_prob = tape
        p = self.params
        ln = ln_name(scope, ln_idx)
        dpre, dgamma, dbeta, dbias = ops.ln_act_bwd(dy, pre, mean, rstd, p[ln + "/gamma"], p[ln + "/beta"], rows, act,
                                                    keep, keep_prob)
        self._acc(ln + "/gamma", dgamma)
        self._acc(ln + "/beta", dbeta)
        self._acc(scope + "/fc/biases", dbias)
        wname = scope + "/fc/weights"
        gw = self.grads[wname]
        if wname in self._touched:          # shared weights: dW += x^T dpre (GEMM with C as its own addend)
            ops.gemm(x2d, dpre, transA=True, addend=gw, out=gw)
        else:
            ops.gemm(x2d, dpre, transA=True, out=gw)
            self._touched.add(wname)
        return ops.gemm(dpre, p[wname], transB=True) if need_dx else None

    def make_keep_masks(self, B, seed, step):
        """reproducible dropout keep-masks for (seed, step): {kind/att, kind/bf_joint, kind/ws_joint}"""
        n, R, H = self.n, self.R, self.H
        out, off = {}, step * (2 * (B * n * R * H + 2 * B * n * 2 * H))
        for k in KINDS:
            for name, cnt, keep in ((k + "/att", B * n * R * H, KEEP_ATT), (k + "/bf_joint", B * n * 2 * H, KEEP_JOINT),
                                    (k + "/ws_joint", B * n * 2 * H, KEEP_JOINT)):
                out[name] = ops.dropout_mask(cnt, seed, off, keep, self.device)
                off += cnt
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, batch, masks):
        """batch: device tensors with the keys of vlmap_memft/datasets/dataset_vlmap.py:128-236 that the
        model reads; masks: uint8 keep-masks (or None = no dropout) keyed '<kind>/att|bf_joint|ws_joint'."""
        p, n, R, D, H, W = self.params, self.n, self.R, self.D, self.H, self.W
        img = batch["image_ft"].contiguous()
        self._img = img
        self._mask_att = {k: masks[k + "/att"] for k in KINDS} if masks is not None else {}
        B = img.shape[0]
        Bn = B * n
        spat = batch["spatial_ft"].reshape(B * R, 6).contiguous()
        nb = batch["num_boxes"].to(torch.int32).contiguous()
        t = {"B": B, "kinds": {}}
        stats_all = {}
        mk = (lambda key: masks[key]) if masks is not None else (lambda key: None)
        for ki, k in enumerate(KINDS):
            kt = {}
            key = batch[k + "_blank_fill/normal_boxes"].reshape(Bn, 4)
            key6 = torch.cat([key, key[:, 2:3] - key[:, 0:1], key[:, 3:4] - key[:, 1:2]], 1).contiguous()
            v, kt["v_t"] = self._fc_ln(spat, "spat_v_linear_v", ki, R, "relu")            # once per image
            qv, kt["qv_t"] = self._fc_ln(key6, "spat_q_linear_v", ki, n, "relu")          # LN over (n, H)
            att, pooled = ops.attn_pool_fwd_rep(v.view(B, R, H), qv, img, nb, p["spat_att/compute/score/fc/weights"],
                                                p["spat_att/compute/score/fc/biases"], n, mk(k + "/att"), KEEP_ATT)
            kt.update(v=v, qv=qv, att=att, pooled=pooled)
            valid = (torch.arange(n, device=self.device)[None, :] < batch[k + "_blank_fill/num"][:, None]) \
                .float().reshape(Bn).contiguous()
            inv_valid = (1.0 / valid.sum()).reshape(1).contiguous()
            fills = batch[k + "_blank_fill/fills"].reshape(Bn).to(torch.int32).contiguous()
            kt.update(valid=valid)

            def head(l_ft, ln_idx, jmask, tag):
                vl, t_vl = self._fc_ln(pooled, "pooled_linear_l", ln_idx, n, "relu")
                ll, t_ll = self._fc_ln(l_ft, "q_linear_l", ln_idx, n, "relu")
                jin = ops.mul(vl, ll)
                j, t_j = self._fc_ln(jin, "joint_fc", ln_idx, n, "relu", jmask, KEEP_JOINT)
                z = ops.gemm(j, p["classifier/fc/weights"], bias=p["classifier/fc/biases"])
                stats, dz = ops.softmax_ce(z, fills, valid, inv_valid, TOP_K, want_dz=True)
                kt[tag] = dict(vl=vl, ll=ll, t_vl=t_vl, t_ll=t_ll, t_j=t_j, j=j, z=z, dz=dz)
                stats_all[k + "_" + tag] = (stats, inv_valid)

            # blank fill: GRU over the caption with a blank (L_GloVe embedding)
            blanks = batch[k + "_blank_fill/blanks"].reshape(Bn, -1).to(torch.int32).contiguous()
            L = blanks.shape[1]
            lens = batch[k + "_blank_fill/blanks_len"].reshape(Bn).to(torch.int32).contiguous()
            srt = batch.get(k + "_blank_fill/sort")
            live = None
            if srt is not None:        # captions in length order: the recurrence skips finished ones (add_length_sort)
                perm = torch.as_tensor(srt["perm"], device=self.device)
                kt["inv"], kt["perm"], live = torch.as_tensor(srt["inv"], device=self.device), perm, srt["live_rows"]
                blanks, lens = blanks.index_select(0, perm).contiguous(), lens.index_select(0, perm).contiguous()
            x_tm = ops.embed_fwd(p["L_GloVe/embed_map"], blanks)                           # [L, Bn, W]
            Wg, Wc = p["encode_L_blank/rnn/gru_cell/gates/kernel"], p["encode_L_blank/rnn/gru_cell/candidate/kernel"]
            xp = torch.empty(L * Bn, 3 * H, dtype=torch.float32, device=self.device)
            x2 = x_tm.view(L * Bn, W)
            ops.gemm(x2, Wg[:W], bias=p["encode_L_blank/rnn/gru_cell/gates/bias"], out=xp[:, :2 * H])
            ops.gemm(x2, Wc[:W], bias=p["encode_L_blank/rnn/gru_cell/candidate/bias"], out=xp[:, 2 * H:])
            hs, gtape = ops.gru_seq_fwd(xp, Wg[W:], Wc[W:], lens, L, Bn, H, live_rows=live)
            kt.update(blanks=blanks, lens=lens, x_tm=x_tm, hs=hs, gtape=gtape, L=L, live=live)
            head(hs[L] if live is None else hs[L].index_select(0, kt["inv"]), ki, mk(k + "/bf_joint"), "blank_fill")
            # word set: tanh(embedding) -> FC + LN + tanh
            wsid = batch[k + "_blank_fill/wordsets"].reshape(Bn, 1).to(torch.int32).contiguous()
            wse = ops.embed_fwd(p["wordset_map/learn"], wsid).view(Bn, W)
            ws = ops.tanh_fwd(wse)
            wf, kt["wf_t"] = self._fc_ln(ws, "wordset_ft", ki, n, "tanh")
            kt.update(wsid=wsid, ws=ws)
            head(wf, 2 + ki, mk(k + "/ws_joint"), "wordset")
            t["kinds"][k] = kt
        self._tape, self._stats = t, stats_all
        return stats_all

    def fetch_report(self):
        """report dict of the reference (13 scalars): <kind>_<task>_{loss,acc,top_5_acc}, total_loss"""
        rep, total = {}, 0.0
        for key, (stats, inv) in self._stats.items():
            s = (stats.sum(0) * inv).cpu().numpy()
            rep[key + "_loss"], rep[key + "_acc"] = float(s[0]), float(s[1])
            rep[key + "_top_%d_acc" % TOP_K] = float(s[2])
            total += float(s[0])
        rep["total_loss"] = total
        self.report = rep
        return rep

    # ------------------------------------------------------------------ backward
    def backward(self):
        p, n, R, D, H, W = self.params, self.n, self.R, self.D, self.H, self.W
        t = self._tape
        B = t["B"]
        Bn = B * n
        self._touched = set()
        for k in SPARSE_VARS:
            self.grads[k].zero_()
        slice_sq = []
        Wc_cls = p["classifier/fc/weights"]
        for ki, k in enumerate(KINDS):
            kt = t["kinds"][k]
            dpooled = None

            def head_bwd(tag):
                nonlocal dpooled
                h = kt[tag]
                gw = self.grads["classifier/fc/weights"]
                if "classifier/fc/weights" in self._touched:
                    ops.gemm(h["j"], h["dz"], transA=True, addend=gw, out=gw)
                else:
                    ops.gemm(h["j"], h["dz"], transA=True, out=gw)
                    self._touched.add("classifier/fc/weights")
                self._acc("classifier/fc/biases", ops.colsum(h["dz"]))
                dj = ops.gemm(h["dz"], Wc_cls, transB=True)
                djin = self._fc_ln_bwd(dj, h["t_j"])
                dvl, dll = ops.mul_bwd(djin, h["vl"], h["ll"])
                dpl = self._fc_ln_bwd(dvl, h["t_vl"])
                dpooled = dpl if dpooled is None else ops.add_inplace(dpooled, dpl)
                return self._fc_ln_bwd(dll, h["t_ll"])

            # blank fill -> GRU -> L_GloVe
            dbf = head_bwd("blank_fill")
            L = kt["L"]
            Wg, Wc = p["encode_L_blank/rnn/gru_cell/gates/kernel"], p["encode_L_blank/rnn/gru_cell/candidate/kernel"]
            if kt.get("live") is not None:
                dbf = dbf.index_select(0, kt["perm"]).contiguous()      # into the length-sorted caption order
            dxp = ops.gru_seq_bwd(dbf, Wg[W:], Wc[W:], kt["lens"], kt["hs"], kt["gtape"], L, Bn, H,
                                  live_rows=kt.get("live")).view(L * Bn, 3 * H)
            x2 = kt["x_tm"].view(L * Bn, W)
            hs_prev = kt["hs"][:L].reshape(L * Bn, H)
            rh = kt["gtape"][3].view(L * Bn, H)
            gWg, gWc = self.grads["encode_L_blank/rnn/gru_cell/gates/kernel"], \
                self.grads["encode_L_blank/rnn/gru_cell/candidate/kernel"]
            first = "gru" not in self._touched
            for A_, B_, out in ((x2, dxp[:, :2 * H], gWg[:W]), (hs_prev, dxp[:, :2 * H], gWg[W:]),
                                (x2, dxp[:, 2 * H:], gWc[:W]), (rh, dxp[:, 2 * H:], gWc[W:])):
                ops.gemm(A_, B_, transA=True, out=out, addend=None if first else out)
            bsum = ops.colsum(dxp)
            self._acc("encode_L_blank/rnn/gru_cell/gates/bias", bsum[:2 * H])
            self._acc("encode_L_blank/rnn/gru_cell/candidate/bias", bsum[2 * H:])
            self._touched.add("gru")
            dx = ops.gemm(dxp[:, :2 * H], Wg[:W], transB=True)
            ops.gemm(dxp[:, 2 * H:], Wc[:W], transB=True, addend=dx, out=dx)
            ops.embed_bwd_into(dx.view(L, Bn, W), kt["blanks"], self.grads["L_GloVe/embed_map"], lens=kt["lens"])
            slice_sq.append(ops.sumsq(dx.view(-1)))
            # word set -> wordset_ft -> tanh -> wordset_map
            dwf = head_bwd("wordset")
            dws = self._fc_ln_bwd(dwf, kt["wf_t"])
            dwse = ops.tanh_bwd(dws, kt["ws"])
            ops.embed_bwd_into(dwse.view(1, Bn, W), kt["wsid"], self.grads["wordset_map/learn"])
            slice_sq.append(ops.sumsq(dwse.view(-1)))
            # spatial attention
            dv, dqv, dw, db = ops.attn_pool_bwd_rep(dpooled, kt["v"].view(B, R, H), kt["qv"], self._img, kt["att"],
                                                    p["spat_att/compute/score/fc/weights"], n, self._mask_att.get(k),
                                                    KEEP_ATT)
            self._acc("spat_att/compute/score/fc/weights", dw)
            self._acc("spat_att/compute/score/fc/biases", db)
            self._fc_ln_bwd(dv.view(B * R, H), kt["v_t"], need_dx=False)
            self._fc_ln_bwd(dqv, kt["qv_t"], need_dx=False)
        self.grad_flat[self.n_train] = torch.stack(slice_sq).sum()

    def optimizer_step(self, lr):
        """clip_by_global_norm(20) + Adam; the two embedding tables contribute their UN-AGGREGATED slice
        gradients to the norm (tf.clip_by_global_norm on IndexedSlices), see fusion.FusionEngine."""
        dense = self.grad_flat[self.sparse_floats:self.n_train]
        tail = self.grad_flat[self.n_train:]
        P = lambda x: C.c_void_p(x.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.vqa_sumsq(P(dense), dense.numel(), P(tail), P(self.norm_sq), P(self.sumsq_ws),
                                      self.sumsq_ws.numel(), st), "vqa_sumsq")
        self.step_count += 1
        s = self.step_count
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** s) / (1.0 - ADAM_B1 ** s)
        _lib.check(self.lib.vqa_clip_adam(P(self.train_flat), P(self.grad_flat), P(self.m_flat), P(self.v_flat),
                                          self.n_train, P(self.norm_sq), CLIP_NORM, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS,
                                          st), "vqa_clip_adam")

    def train_step(self, batch, masks, lr):
        self.forward(batch, masks)
        self.backward()
        self.optimizer_step(lr)

    def state_dict(self):
        """name -> CPU tensor with the reference's variable names, the Adam slots of every trained variable
        (`<var>/Adam`, `<var>/Adam_1`) and global_step -- what tf.train.Saver keeps (vlmap_memft/trainer.py);
        same layout as fusion.FusionEngine.state_dict."""
        out = {k: v.detach().cpu().clone() for k, v in self.params.items()}
        for k, (o, cnt) in self._tab.items():
            out[k + "/Adam"] = self.m_flat[o:o + cnt].view(self.shapes[k]).cpu().clone()
            out[k + "/Adam_1"] = self.v_flat[o:o + cnt].view(self.shapes[k]).cpu().clone()
        out["global_step"] = torch.tensor(self.step_count, dtype=torch.int64)
        return out

    def load_state_dict(self, sd):
        """Restores parameters, Adam moments and the step count (beta powers), so a resumed run continues the
        optimiser trajectory of an uninterrupted one."""
        for k in self.shapes:
            if k in sd:
                self.params[k].copy_(torch.as_tensor(sd[k]).to(torch.float32))
        for k, (o, cnt) in self._tab.items():
            if k + "/Adam" in sd:
                self.m_flat[o:o + cnt].copy_(torch.as_tensor(sd[k + "/Adam"]).reshape(-1))
                self.v_flat[o:o + cnt].copy_(torch.as_tensor(sd[k + "/Adam_1"]).reshape(-1))
        if "global_step" in sd:
            self.step_count = int(sd["global_step"])


def export_word_weights(state_dict, vocab, answer_dict, save_dir):
    """vlmap_memft/export_word_weights.py:35-82: the bridge from pre-training to the VQA model
    (modules.WordWeightAnswer reads class_weights / class_biases by answer string).  Writes weights.hdf5 with the
    reference's five datasets (hdf5_io, no h5py) and vocab.pkl / answer_dict.pkl like the reference."""
    import os
    import pickle
    from . import hdf5_io
    if os.path.exists(save_dir):
        raise ValueError("Do not overwrite: {}".format(save_dir))
    os.makedirs(save_dir)
    g = lambda k: np.asarray(state_dict[k].cpu() if torch.is_tensor(state_dict[k]) else state_dict[k])
    hdf5_io.write(os.path.join(save_dir, "weights.hdf5"),
                  {"v_word": g("V_GloVe/embed_map"), "l_word": g("L_GloVe/embed_map"),
                   "l_answer_word": g("LearnAnswerGloVe/embed_map"), "class_weights": g("classifier/fc/weights"),
                   "class_biases": g("classifier/fc/biases")})
    with open(os.path.join(save_dir, "vocab.pkl"), "wb") as f:
        pickle.dump(vocab, f)
    with open(os.path.join(save_dir, "answer_dict.pkl"), "wb") as f:
        pickle.dump(answer_dict, f)
    return save_dir
