"""cfg-5 pre-training ("task discovery") model on MI355X: the counterpart of
vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:54-94, 323-609, 675-706.

Spatial attention over the 36 regions for n = 5 annotated boxes per image, blank-fill (GRU over the
caption with a blank) and word-set conditioned heads for objects and attributes, all four heads
sharing pooled_linear_l / q_linear_l / joint_fc / classifier, softmax-CE over the obj3000+attr1000 answers with
top-1 / top-5 accuracy.  LayerNorm variables of those shared fc_layer scopes: ONE per scope, trained by every call
site (`ln_shared=True`, the default: TF 1.x zeroes a string-named scope's sub-scope counts when it exits, so the
un-scoped layer_norm is `LayerNorm` again at the next call site and AUTO_REUSE shares it -- DESIGN.md section 2), or one per
call site (`LayerNorm`, `LayerNorm_1`, ... in graph build order) when a checkpoint carries those names:
`ln_shared_in(names)` decides and `load_state_dict` switches the engine over.  Forward and the hand-derived backward are
ONE C call each (vqa_pretrain_forward / vqa_pretrain_backward, csrc/pretrain_model.hip: every kernel of the
pass is enqueued from C++, the workspace is carved from the dims, no torch op runs inside the step); the
effective batch is B*n rows and the x n tile of V_ft / spatial_ft that the reference materialises (:324-333)
never exists: the attention kernels take `rep = n` queries per memory and v_linear_v of the (identical) tiles
is computed once per image.
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
SPARSE_VARS = ("wordset_map/learn", "L_GloVe/embed_map")                # IndexedSlices gradients
# Order of the dense variables in the flat buffers = the order in which the phases of vqa_pretrain_backward_phases
# complete their gradients, so every data-parallel bucket is one contiguous range:
#   [wordset_map | L_GloVe | GRU (phase 2) | stacked heads (phase 1) | spatial attention, wordset_ft (phase 8) | tail]
PHASE_SCOPES = (("encode_L_blank/",), ("classifier/", "joint_fc/", "pooled_linear_l/", "q_linear_l/"),
                ("spat_att/", "spat_q_linear_v/", "spat_v_linear_v/", "wordset_ft/"))


def ln_name(scope, idx):
    return scope + ("/LayerNorm" if idx == 0 else "/LayerNorm_%d" % idx)


def ln_shared_in(names):
    """True when the variable names are those of the shared-LayerNorm graph (no `<scope>/LayerNorm_<k>/...`)."""
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


def init_random_params(rng, Vq, n_ws, A, W=300, D=2048, H=1024, ln_shared=True):
    """Random-init weights of the architecture (Xavier-uniform FCs, GRU gate bias 1, LN gamma 1,
    embeddings U(-0.01, 0.01); GloVe vectors are download-only)."""
    p = {}
    for n, shp in variable_shapes(Vq, n_ws, A, W, D, H, ln_shared).items():
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
    """Host-side: the blank-fill captions of both categories are encoded as ONE batch of 2*B*n rows (object rows first).
    This adds 'blank_fill/sort' = the permutation that orders those rows by length (longest first), its inverse and
    live_rows[t] = #captions longer than t.  The engine then embeds / encodes the captions in that order, runs every GRU
    step on the live prefix only, and un-permutes the final states."""
    keys = [k + "_blank_fill/blanks_len" for k in KINDS]
    if any(k not in batch or torch.is_tensor(batch[k]) for k in keys):
        return batch
    lens = np.concatenate([np.asarray(batch[k]).reshape(-1) for k in keys]).astype(np.int64)
    L = int(np.asarray(batch[KINDS[0] + "_blank_fill/blanks"]).shape[-1])
    perm = np.argsort(-lens, kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    sl = np.clip(lens[perm], 0, L)
    batch["blank_fill/sort"] = {"perm": perm, "inv": inv,
                                "live_rows": (sl[None, :] > np.arange(L)[:, None]).sum(1).astype(np.int32)}
    return batch


def _pad4(n):
    return (n + 3) // 4 * 4


class PretrainEngine:
    def __init__(self, *, n, R, D, H, W, A, Vq, n_ws, params, device="cuda:0", deterministic=False, ln_shared=None):
        """ln_shared: one LayerNorm per shared fc_layer scope (True) or one per call site (False); None = whatever the
        variable names in `params` say (`.../LayerNorm_1/...` present -> per call site), as for a checkpoint."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.VqaHotError("PretrainEngine needs a GPU (no CPU fallback)")
        self.device = torch.device(device)
        self.n, self.R, self.D, self.H, self.W, self.A = n, R, D, H, W, A
        self.Vq, self.n_ws, self.deterministic = Vq, n_ws, bool(deterministic)
        self.step_count = 0
        self.report = {}
        self.workspace, self.dims = None, None
        self._layout(ln_shared_in(params) if ln_shared is None else bool(ln_shared))
        for k in self.shapes:
            self.params[k].copy_(torch.as_tensor(np.asarray(params[k])).to(torch.float32))

    def _layout(self, ln_shared):
        """Flat parameter / gradient / Adam buffers and the C structs for one of the two LayerNorm variable sets."""
        self.ln_shared = bool(ln_shared)
        self.shapes = variable_shapes(self.Vq, self.n_ws, self.A, self.W, self.D, self.H, self.ln_shared)
        dense = sorted(k for k in self.shapes if k not in NO_GRAD_VARS and k not in SPARSE_VARS)
        groups = [[k for k in dense if k.startswith(sc)] for sc in PHASE_SCOPES]
        assert sorted(sum(groups, [])) == dense, "a variable outside the phase scopes"
        self.train_names = list(SPARSE_VARS) + sum(groups, [])
        off, self._tab = 0, {}
        for k in self.train_names:
            cnt = int(np.prod(self.shapes[k]))
            self._tab[k] = (off, cnt)
            off += _pad4(cnt)
        self.n_train = off
        # bucket bounds (floats): wordset_map [0, b0), L_GloVe [b0, b1), GRU [b1, b2), heads [b2, b3), rest [b3, n_train)
        ends, o = [], 0
        for names in ([SPARSE_VARS[0]], [SPARSE_VARS[1]], groups[0], groups[1], groups[2]):
            o += sum(_pad4(int(np.prod(self.shapes[k]))) for k in names)
            ends.append(o)
        self._bounds = tuple(ends)
        assert ends[-1] == self.n_train
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
        self._p_struct = self._param_struct(self.params)
        self._g_struct = self._param_struct(self.grads)

    def _flags(self):
        return (_lib.FLAG_DETERMINISTIC if self.deterministic else 0) | (_lib.FLAG_SHARED_LN if self.ln_shared else 0)

    # ------------------------------------------------------------------ C-ABI plumbing
    def make_keep_masks(self, B, seed, step, row_offset=0, global_rows=None):
        """reproducible dropout keep-masks for (seed, step): {kind/att, kind/bf_joint, kind/ws_joint}.  The stream is
        indexed by the GLOBAL image row: a data-parallel shard of B images passes its first global row (row_offset) and
        the global batch size and draws exactly the bits one process on the whole batch would draw for its rows."""
        n, R, H = self.n, self.R, self.H
        Bg = int(global_rows) if global_rows is not None else B
        out, off = {}, step * (2 * (Bg * n * R * H + 2 * Bg * n * 2 * H))
        for k in KINDS:
            for name, per_image, keep in ((k + "/att", n * R * H, KEEP_ATT), (k + "/bf_joint", n * 2 * H, KEEP_JOINT),
                                          (k + "/ws_joint", n * 2 * H, KEEP_JOINT)):
                out[name] = ops.dropout_mask(B * per_image, seed, off + row_offset * per_image, keep, self.device)
                off += Bg * per_image
        return out

    def _param_struct(self, table):
        def fc(scope, n_ln):
            f = _lib.PtFc(w=table[scope + "/fc/weights"].data_ptr(), b=table[scope + "/fc/biases"].data_ptr())
            for i in range(min(n_ln, 1) if self.ln_shared else n_ln):
                f.beta[i] = table[ln_name(scope, i) + "/beta"].data_ptr()
                f.gamma[i] = table[ln_name(scope, i) + "/gamma"].data_ptr()
            return f
        g = "encode_L_blank/rnn/gru_cell/"
        return _lib.PtParams(
            wordset_map=table["wordset_map/learn"].data_ptr(), l_glove=table["L_GloVe/embed_map"].data_ptr(),
            spat_v_linear_v=fc("spat_v_linear_v", 2), spat_q_linear_v=fc("spat_q_linear_v", 2),
            spat_att_score=fc("spat_att/compute/score", 0), gru_wg=table[g + "gates/kernel"].data_ptr(),
            gru_bg=table[g + "gates/bias"].data_ptr(), gru_wc=table[g + "candidate/kernel"].data_ptr(),
            gru_bc=table[g + "candidate/bias"].data_ptr(), pooled_linear_l=fc("pooled_linear_l", 4),
            q_linear_l=fc("q_linear_l", 4), joint_fc=fc("joint_fc", 4), wordset_ft=fc("wordset_ft", 2),
            classifier=fc("classifier", 0))

    def _dev(self, v, dtype):
        t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def bind_tables(self, image_features, spatial_features, num_boxes):
        """Keep the feature tables of the dataset ([N,R,D] f32, [N,R,6] f32, [N] i32) in HBM; batches may then carry
        `image_idx` (i64 [B]) instead of image_ft / spatial_ft / num_boxes and the rows are gathered on the device
        (`vqa_gather_features`), the cfg-5 counterpart of SURVEY row a1: no per-step np.take of 151 MB on the host and
        no H2D copy of it.  The reference holds the whole '<split>_vfeat.hdf5' in host RAM (dataset_vlmap.py:67-72)."""
        f = self._dev(image_features, torch.float32)
        assert f.dim() == 3 and f.shape[1] == self.R and f.shape[2] == self.D, f.shape
        self._tables = (f, self._dev(spatial_features, torch.float32), self._dev(np.asarray(num_boxes), torch.int32))
        self._gathered = {}

    def _gather(self, idx):
        """image_ft / spatial_ft / num_boxes of a batch from the bound tables, into buffers owned by the engine"""
        from . import ops
        table, spat, nb = self._tables
        B = int(idx.numel())
        buf = self._gathered.get(B)
        if buf is None:
            buf = self._gathered[B] = {"spatial_ft": torch.empty(B, self.R, spat.shape[2], device=self.device)}
        V, n = ops.gather_features(table, nb, idx)
        torch.index_select(spat, 0, idx, out=buf["spatial_ft"])
        return V, buf["spatial_ft"], n

    def _batch_struct(self, batch, masks):
        """C view of one batch (device tensors with the keys of vlmap_memft/datasets/dataset_vlmap.py:128-236 that
        the model reads).  Converted tensors are cached on the batch dict, so a batch that is fed again (the input
        pipeline caches its batches) costs no conversion."""
        cache = batch.setdefault("_pt_dev", {}) if isinstance(batch, dict) else {}
        def get(key, dtype):
            if key not in cache:
                cache[key] = self._dev(batch[key], dtype)
            return cache[key]
        keep = [cache, masks]
        if "image_idx" in batch and "image_ft" not in batch:
            if getattr(self, "_tables", None) is None:
                raise ValueError("batch carries image_idx but no feature tables are bound (PretrainEngine.bind_tables)")
            V, sp, nb = self._gather(get("image_idx", torch.int64))
            keep.append((V, sp, nb))
            bs = _lib.PtBatch(image_ft=V.data_ptr(), spatial_ft=sp.data_ptr(), num_boxes=nb.data_ptr())
            B = V.shape[0]
        else:
            bs = _lib.PtBatch(image_ft=get("image_ft", torch.float32).data_ptr(),
                              spatial_ft=get("spatial_ft", torch.float32).data_ptr(),
                              num_boxes=get("num_boxes", torch.int32).data_ptr())
            B = cache["image_ft"].shape[0]
        L = None
        for ki, k in enumerate(KINDS):
            pre = k + "_blank_fill/"
            kb = bs.kind[ki]
            kb.normal_boxes = get(pre + "normal_boxes", torch.float32).data_ptr()
            for f in ("fills", "blanks", "blanks_len", "wordsets", "num"):
                setattr(kb, f, get(pre + f, torch.int32).data_ptr())
            Lk = int(cache[pre + "blanks"].shape[-1])
            assert L is None or L == Lk, "object / attribute captions must be padded to one length"
            L = Lk
            if masks is not None:
                kb.keep_att = masks[k + "/att"].data_ptr()
                kb.keep_bf_joint = masks[k + "/bf_joint"].data_ptr()
                kb.keep_ws_joint = masks[k + "/ws_joint"].data_ptr()
        srt = batch.get("blank_fill/sort")
        if srt is not None:       # captions in length order: the recurrence skips finished ones (add_length_sort)
            if "_dev" not in srt:
                live = np.ascontiguousarray(srt["live_rows"], dtype=np.int32)
                assert live.shape == (L,) and len(srt["perm"]) == 2 * int(cache[KINDS[0] + "_blank_fill/blanks_len"].numel())
                srt["_dev"] = (self._dev(np.asarray(srt["perm"]), torch.int32),
                               self._dev(np.asarray(srt["inv"]), torch.int32), live)
            perm, inv, live = srt["_dev"]
            bs.perm, bs.inv, bs.live_rows = perm.data_ptr(), inv.data_ptr(), live.ctypes.data
            keep.append(srt["_dev"])
        return bs, B, L, keep

    def tensor(self, name, dtype=torch.float32):
        """Named intermediate of the last forward as a torch view of the workspace (vqa_pretrain_tensor)."""
        off, n = C.c_int64(), C.c_int64()
        _lib.check(self.lib.vqa_pretrain_tensor(C.byref(self.dims), name.encode(), C.byref(off), C.byref(n)),
                   "vqa_pretrain_tensor(%s)" % name)
        return self.workspace[off.value:off.value + 4 * n.value].view(dtype)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ forward / backward
    def global_valid_counts(self, batch, group=None):
        """(#valid object entries, #valid attribute entries) of the GLOBAL batch = the denominators of the masked mean
        losses (n_way_classification_loss, :675-706): this shard's counts, SUM-all-reduced over the ranks.  A trainer
        whose ranks slice one global batch can compute the same numbers from the host arrays without a collective."""
        import torch.distributed as dist
        cnt = torch.stack([torch.as_tensor(batch[k + "_blank_fill/num"]).to(torch.int64).clamp(0, self.n).sum()
                           for k in KINDS]).to(torch.float64)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "gloo":
                cnt = cnt.cpu()
            else:
                cnt = cnt.to(self.device)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        return tuple(float(v) for v in cnt.cpu())

    def forward(self, batch, masks, want_dz=True, global_valid=None):
        """batch: dict of arrays / tensors (keys of dataset_vlmap's batches; optional 'blank_fill/sort' from
        add_length_sort); masks: uint8 keep-masks keyed '<kind>/att|bf_joint|ws_joint' or None (no dropout);
        global_valid: data parallel only -- (object, attribute) valid-entry counts of the global batch."""
        bs, B, L, keep = self._batch_struct(batch, masks)
        d = _lib.PtDims(B=B, n=self.n, R=self.R, D=self.D, H=self.H, W=self.W, A=self.A, Vq=self.Vq, n_ws=self.n_ws, L=L,
                        flags=self._flags(), keep_att=KEEP_ATT,
                        keep_joint=KEEP_JOINT)
        if global_valid is not None:
            d.global_valid[0], d.global_valid[1] = float(global_valid[0]), float(global_valid[1])
        need = int(self.lib.vqa_pretrain_workspace_bytes(C.byref(d)))
        if need <= 0:
            raise _lib.VqaHotError("vqa_pretrain_workspace_bytes rejected the dims")
        if self.workspace is None or need > self.workspace.numel():
            self.workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)
        self.dims, self._bs, self._keepalive = d, bs, keep
        _lib.check(self.lib.vqa_pretrain_forward(C.byref(d), C.byref(self._p_struct), C.byref(bs),
                                                 C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(),
                                                 1 if want_dz else 0, self._stream()), "vqa_pretrain_forward")
        Bn = B * self.n
        self._tape = {"B": B, "kinds": {
            k: {"att": self.tensor(k + "/att").view(Bn, self.R), "pooled": self.tensor(k + "/pooled").view(Bn, self.D),
                "blank_fill": {"z": self.tensor(k + "/bf/z").view(Bn, self.A)},
                "wordset": {"z": self.tensor(k + "/ws/z").view(Bn, self.A)}} for k in KINDS}}

    def fetch_report(self, reduce=False, group=None):
        """report dict of the reference (13 scalars): <kind>_<task>_{loss,acc,top_5_acc}, total_loss.  reduce: data
        parallel -- every scalar is a sum over the shard's rows already divided by the GLOBAL valid count
        (forward(global_valid=...)), so a SUM all-reduce gives what one process on the whole batch reports."""
        import torch.distributed as dist
        r = self.tensor("report")[:13]
        if reduce and dist.is_initialized() and dist.get_world_size(group) > 1:
            r = r.cpu() if dist.get_backend(group) == "gloo" else r.clone()
            dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
        r = r.cpu().numpy()
        self.report = {self.lib.vqa_pretrain_report_key(i).decode(): float(r[i]) for i in range(13)}
        return self.report

    def _backward_phases(self, phases):
        tail = self.grad_flat[self.n_train:]
        _lib.check(self.lib.vqa_pretrain_backward_phases(
            C.byref(self.dims), C.byref(self._p_struct), C.byref(self._g_struct), C.byref(self._bs),
            C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(), C.c_void_p(tail.data_ptr()), phases,
            self._stream()), "vqa_pretrain_backward_phases")

    def backward(self, reducer=None):
        """All gradients into grad_flat (vlmap_memft/trainer.py:129-137: optimize_loss over every variable).  With a
        bucketed `reducer` (dp.BucketedAllReduce) the dependency-ordered phases are enqueued one by one and each
        finished bucket's all-reduce starts right away: the stacked heads' 50+ MB reduce under the back-propagation
        through time, the GRU kernels under the dx GEMM + embedding scatter-add, L_GloVe under the spatial-attention
        backward; only the last, small bucket (word sets, spatial FCs, slice sum of squares) is exposed."""
        if reducer is None:
            self._backward_phases(15)
            return
        b0, b1, b2, b3 = self._bounds[:4]
        n = self.n_train
        self._backward_phases(1)
        reducer.start(self.grad_flat[b2:b3])
        self._backward_phases(2)
        reducer.start(self.grad_flat[b1:b2])
        self._backward_phases(4)
        reducer.start(self.grad_flat[b0:b1])
        self._backward_phases(8)
        reducer.start(self.grad_flat[:b0])
        reducer.start(self.grad_flat[b3:])          # spatial attention / wordset_ft gradients + the tail (slice sum of squares)
        reducer.finish()

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

    def train_step(self, batch, masks, lr, allreduce=None, global_valid=None):
        """forward -> backward -> (gradient all-reduce) -> clip + Adam.  Data parallel: `allreduce` is a
        dp.BucketedAllReduce (overlapped with the backward phases) or any callable on grad_flat; pass the global
        valid counts so that every shard divides by the global denominators and a SUM reduce gives the gradient of
        the global-batch losses; all ranks then apply the identical update."""
        self.forward(batch, masks, global_valid=global_valid)
        if allreduce is not None and hasattr(allreduce, "start"):
            self.backward(reducer=allreduce)
        else:
            self.backward()
            if allreduce is not None:
                allreduce(self.grad_flat)
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

    def load_state_dict(self, sd, strict=True):
        """Restores parameters, Adam moments and the step count (beta powers), so a resumed run continues the
        optimiser trajectory of an uninterrupted one.  The LayerNorm variable set follows the NAMES in the checkpoint:
        `<scope>/LayerNorm_1/...` present -> one LayerNorm per call site, absent -> one per shared scope; an engine
        built for the other set is laid out again before loading.  strict: a model variable missing from the
        checkpoint raises (a silently skipped name would resume from the initial weights)."""
        model_keys = [k for k in sd if k != "global_step" and not k.endswith(("/Adam", "/Adam_1"))]
        shared = ln_shared_in(model_keys)
        if shared != self.ln_shared:
            self._layout(shared)
            self.workspace, self.dims = None, None
        missing = [k for k in self.shapes if k not in sd]
        if missing and strict:
            raise KeyError("checkpoint lacks %d model variables, e.g. %s" % (len(missing), ", ".join(sorted(missing)[:4])))
        for k in self.shapes:
            if k in sd:
                self.params[k].copy_(torch.as_tensor(sd[k]).to(torch.float32))
        for k, (o, cnt) in self._tab.items():
            if k + "/Adam" in sd:
                self.m_flat[o:o + cnt].copy_(torch.as_tensor(sd[k + "/Adam"]).reshape(-1))
                self.v_flat[o:o + cnt].copy_(torch.as_tensor(sd[k + "/Adam_1"]).reshape(-1))
        if "global_step" in sd:
            self.step_count = int(sd["global_step"])
        return missing


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
