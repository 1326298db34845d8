"""Op-level host wrappers over the C ABI (torch tensors in, torch tensors out).

Mirrors vlmap/modules.py's free functions at op granularity so the parity tests
read like tests of the reference's modules.  Every function enqueues on the
current torch stream and raises VqaHotError on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def gemm(A, B, transA=False, transB=False, bias=None, addend=None, split_k=0, out=None):
    """C = op(A) @ op(B) (+bias) (+addend)  on the f32 MFMA (layers.fully_connected and its grads)."""
    lib = _lib.load()
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(-1) == 1 and B.stride(-1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[0] if transB else B.shape[1]
    assert (B.shape[1] if transB else B.shape[0]) == K
    Cm = out if out is not None else _f32(M, N, like=A)
    nws = int(lib.vqa_gemm_workspace_floats(int(transA), int(transB), M, N, K, split_k))
    ws = _f32(max(nws, 4), like=A)
    _lib.check(lib.vqa_gemm_f32(int(transA), int(transB), M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(Cm),
                                Cm.stride(0), _p(bias), _p(addend), addend.stride(0) if addend is not None else 0,
                                split_k, _p(ws), ws.numel(), _st(A)), "vqa_gemm_f32")
    return Cm


def gather_features(table, nbox_table, idx):
    lib = _lib.load()
    N, R, D = table.shape
    B = idx.numel()
    V = _f32(B, R, D, like=table)
    nb = torch.empty(B, dtype=torch.int32, device=table.device)
    _lib.check(lib.vqa_gather_features(_p(table), _p(nbox_table), _p(idx), _p(V), _p(nb), B, R, D, N, _st(table)),
               "vqa_gather_features")
    return V, nb


def embed_fwd(E, q):
    lib = _lib.load()
    B, T = q.shape
    Vq, W = E.shape
    x = _f32(T, B, W, like=E)
    _lib.check(lib.vqa_embed_fwd(_p(E), _p(q), _p(x), B, T, W, Vq, _st(E)), "vqa_embed_fwd")
    return x


def embed_bwd(dx_tm, q, Vq, lens=None):
    """lens (i32 [B]): skip the zero-padded positions t >= lens[b] (their dx is exactly zero)."""
    lib = _lib.load()
    T, B, W = dx_tm.shape
    dE = torch.zeros(Vq, W, dtype=torch.float32, device=dx_tm.device)
    _lib.check(lib.vqa_embed_bwd_len(_p(dx_tm), _p(q), _p(lens), _p(dE), B, T, W, Vq, _st(dx_tm)), "vqa_embed_bwd_len")
    return dE


def ln_relu_fwd(pre, gamma, beta, rows=1, keepmask=None, keep_prob=1.0):
    lib = _lib.load()
    M, N = pre.shape
    G = M // rows
    y = torch.empty_like(pre)
    mean, rstd = _f32(G, like=pre), _f32(G, like=pre)
    _lib.check(lib.vqa_ln_relu_fwd(_p(pre), _p(gamma), _p(beta), _p(keepmask), keep_prob, _p(y), _p(mean), _p(rstd),
                                   G, rows, N, _st(pre)), "vqa_ln_relu_fwd")
    return y, mean, rstd


def colsum(X):
    lib = _lib.load()
    M, N = X.shape
    out = _f32(N, like=X)
    ws = _f32(max(int(lib.vqa_colsum_workspace_floats(M, N)), 4), like=X)
    _lib.check(lib.vqa_colsum(_p(X), M, N, X.stride(0), _p(out), _p(ws), ws.numel(), _st(X)), "vqa_colsum")
    return out


def colsum3(X0, X1, X2):
    """Column sums of three equally shaped matrices in one pair of launches."""
    lib = _lib.load()
    M, N = X0.shape
    assert X1.shape == X0.shape and X2.shape == X0.shape and X0.stride(0) == X1.stride(0) == X2.stride(0)
    outs = [_f32(N, like=X0) for _ in range(3)]
    ws = _f32(max(3 * int(lib.vqa_colsum_workspace_floats(M, N)), 4), like=X0)
    _lib.check(lib.vqa_colsum3(_p(X0), _p(X1), _p(X2), M, N, X0.stride(0), _p(outs[0]), _p(outs[1]), _p(outs[2]),
                               _p(ws), ws.numel(), _st(X0)), "vqa_colsum3")
    return outs


def ln_relu_bwd(dy, pre, mean, rstd, gamma, beta, rows=1, keepmask=None, keep_prob=1.0, want_params=True):
    lib = _lib.load()
    M, N = pre.shape
    G = M // rows
    dpre = torch.empty_like(pre)
    pg = _f32(G, N, like=pre) if want_params else None
    pb = _f32(G, N, like=pre) if want_params else None
    pbias = _f32(G, N, like=pre) if want_params else None
    _lib.check(lib.vqa_ln_relu_bwd(_p(dy), _p(pre), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(keepmask), keep_prob,
                                   _p(dpre), _p(pg), _p(pb), _p(pbias), G, rows, N, _st(pre)), "vqa_ln_relu_bwd")
    if not want_params:
        return dpre, None, None, None
    dgamma, dbeta, dbias = colsum3(pg, pb, pbias)
    return dpre, dgamma, dbeta, dbias


def attn_pool_fwd(v, qv, V, nb, w, bias, keepmask=None, keep_prob=1.0):
    """modules.hadamard_attention + modules.attention_pooling."""
    lib = _lib.load()
    B, R, H = v.shape
    D = V.shape[2]
    att, pooled = _f32(B, R, like=v), _f32(B, D, like=v)
    _lib.check(lib.vqa_attn_pool_fwd(_p(v), _p(qv), _p(V), _p(nb), _p(w), _p(bias), _p(keepmask), keep_prob, _p(att),
                                     _p(pooled), B, R, H, D, _st(v)), "vqa_attn_pool_fwd")
    return att, pooled


def attn_pool_bwd(dpooled, v, qv, V, att, w, keepmask=None, keep_prob=1.0):
    lib = _lib.load()
    B, R, H = v.shape
    D = V.shape[2]
    dv, dqv = torch.empty_like(v), torch.empty_like(qv)
    pdw, pdb = _f32(B, H, like=v), _f32(B, 1, like=v)
    _lib.check(lib.vqa_attn_pool_bwd(_p(dpooled), _p(v), _p(qv), _p(V), _p(att), _p(w), _p(keepmask), keep_prob,
                                     _p(dv), _p(dqv), _p(pdw), _p(pdb), B, R, H, D, _st(v)), "vqa_attn_pool_bwd")
    return dv, dqv, colsum(pdw), colsum(pdb)


def loss_fwd(z, target, masks, use_train_mask=True, inv_batch=None, want_dz=True):
    lib = _lib.load()
    B, A = z.shape
    stats = _f32(B, 16, like=z)
    pred = torch.empty(B, dtype=torch.int32, device=z.device)
    dz = torch.empty_like(z) if want_dz else None
    _lib.check(lib.vqa_loss_fwd(_p(z), _p(target), _p(masks["train"]), _p(masks["obj"]), _p(masks["attr"]),
                                _p(masks["exist"]), int(use_train_mask), inv_batch if inv_batch else 1.0 / B,
                                _p(stats), _p(pred), _p(dz), B, A, _st(z)), "vqa_loss_fwd")
    report = _f32(16, like=z)
    _lib.check(lib.vqa_report_reduce(_p(stats), B, _p(report), _st(z)), "vqa_report_reduce")
    keys = [lib.vqa_report_key(i).decode() for i in range(13)]
    return stats, pred, dz, dict(zip(keys, report[:13].cpu().tolist()))


def gemm_shortk(A, B, bias=None, scale=None, residual=None, relu=False, out=None):
    """C = [relu]((A @ B) * scale + bias + residual) for K <= 304 with A stationary in registers (csrc/gemm_shortk.hip).
    gemm() and the extractor's convolutions route qualifying shapes here themselves; this is the direct entry."""
    lib = _lib.load()
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(-1) == 1 and B.stride(-1) == 1
    M, K = A.shape
    N = B.shape[1]
    assert B.shape[0] == K
    Cm = out if out is not None else _f32(M, N, like=A)
    _lib.check(lib.vqa_gemm_shortk_nn(M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(Cm), Cm.stride(0), _p(bias),
                                      _p(scale), _p(residual), residual.stride(0) if residual is not None else 0,
                                      int(bool(relu)), _st(A)), "vqa_gemm_shortk_nn")
    return Cm


def gemm_bf16x3_ex(A, B, transA=False, bias=None, split_k=1, out=None):
    """EXPERIMENT: op(A) @ B (+ bias) on the bf16 matrix pipe through three-way splits; transA: A is [K, M]; split k with
    a deterministic slab sum (csrc/gemm_bf16x3.hip)."""
    lib = _lib.load()
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(-1) == 1 and B.stride(-1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[1]
    assert B.shape[0] == K
    Cm = out if out is not None else _f32(M, N, like=A)
    nws = int(lib.vqa_gemm_bf16x3_workspace_floats(M, N, K, split_k))
    ws = _f32(max(nws, 4), like=A)
    _lib.check(lib.vqa_gemm_bf16x3(int(transA), M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(Cm), Cm.stride(0),
                                   _p(bias), split_k, _p(ws), ws.numel(), _st(A)), "vqa_gemm_bf16x3")
    return Cm


def gemm_bf16x3(A, B, bias=None, out=None):
    """EXPERIMENT: C = A @ B (+ bias) through three-way bf16 splits and six bf16 MFMA products per a*b (f32-equivalent
    products; csrc/gemm_bf16x3.hip).  Whole 128 x 128 x 32 tiles only; the product path uses gemm() (exact f32 MFMA)."""
    lib = _lib.load()
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(-1) == 1 and B.stride(-1) == 1
    M, K = A.shape
    N = B.shape[1]
    assert B.shape[0] == K
    Cm = out if out is not None else _f32(M, N, like=A)
    _lib.check(lib.vqa_gemm_bf16x3_nn(M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(Cm), Cm.stride(0), _p(bias),
                                      _st(A)), "vqa_gemm_bf16x3_nn")
    return Cm


def dropout_mask(n, seed, offset, keep_prob, device):
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.check(lib.vqa_dropout_mask(_p(out), n, seed, offset, keep_prob, _st(out)), "vqa_dropout_mask")
    return out


def sumsq(g, extra=None):
    lib = _lib.load()
    out = _f32(4, like=g)
    ws = _f32(int(lib.vqa_sumsq_workspace_floats(g.numel())) + 4, like=g)
    _lib.check(lib.vqa_sumsq(_p(g), g.numel(), _p(extra), _p(out), _p(ws), ws.numel(), _st(g)), "vqa_sumsq")
    return out[0]


# ---------------------------------------------------------------- ops used by the pre-training model
def ln_act_fwd(pre, gamma, beta, rows=1, act="relu", keepmask=None, keep_prob=1.0):
    """modules.fc_layer's layer_norm + activation ('relu' | 'tanh') (+ dropout)."""
    lib = _lib.load()
    M, N = pre.shape
    G = M // rows
    y = torch.empty_like(pre)
    mean, rstd = _f32(G, like=pre), _f32(G, like=pre)
    _lib.check(lib.vqa_ln_act_fwd(_p(pre), _p(gamma), _p(beta), _p(keepmask), keep_prob, _p(y), _p(mean), _p(rstd), G,
                                  rows, N, 0 if act == "relu" else 1, _st(pre)), "vqa_ln_act_fwd")
    return y, mean, rstd


def ln_act_bwd(dy, pre, mean, rstd, gamma, beta, rows=1, act="relu", keepmask=None, keep_prob=1.0):
    lib = _lib.load()
    M, N = pre.shape
    G = M // rows
    dpre = torch.empty_like(pre)
    pg, pb, pbias = _f32(G, N, like=pre), _f32(G, N, like=pre), _f32(G, N, like=pre)
    _lib.check(lib.vqa_ln_act_bwd(_p(dy), _p(pre), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(keepmask), keep_prob,
                                  _p(dpre), _p(pg), _p(pb), _p(pbias), G, rows, N, 0 if act == "relu" else 1,
                                  _st(pre)), "vqa_ln_act_bwd")
    dgamma, dbeta, dbias = colsum3(pg, pb, pbias)        # one pair of launches instead of three
    return dpre, dgamma, dbeta, dbias


def attn_pool_fwd_rep(v, qv, V, nb, w, bias, rep, keepmask=None, keep_prob=1.0):
    """`rep` queries per memory: v [B,R,H], V [B,R,D], nb [B]; qv [B*rep,H] -> att [B*rep,R], pooled [B*rep,D]."""
    lib = _lib.load()
    B, R, H = v.shape
    D = V.shape[2]
    att, pooled = _f32(B * rep, R, like=v), _f32(B * rep, D, like=v)
    _lib.check(lib.vqa_attn_pool_fwd_rep(_p(v), _p(qv), _p(V), _p(nb), _p(w), _p(bias), _p(keepmask), keep_prob,
                                         _p(att), _p(pooled), B, rep, R, H, D, _st(v)), "vqa_attn_pool_fwd_rep")
    return att, pooled


def attn_pool_bwd_rep(dpooled, v, qv, V, att, w, rep, keepmask=None, keep_prob=1.0):
    lib = _lib.load()
    B, R, H = v.shape
    D = V.shape[2]
    dv, dqv = torch.empty_like(v), torch.empty_like(qv)
    pdw, pdb = _f32(B * rep, H, like=v), _f32(B * rep, 1, like=v)
    _lib.check(lib.vqa_attn_pool_bwd_rep(_p(dpooled), _p(v), _p(qv), _p(V), _p(att), _p(w), _p(keepmask), keep_prob,
                                         _p(dv), _p(dqv), _p(pdw), _p(pdb), B, rep, R, H, D, _st(v)),
               "vqa_attn_pool_bwd_rep")
    return dv, dqv, colsum(pdw), colsum(pdb)


def tanh_fwd(x):
    lib = _lib.load()
    y = torch.empty_like(x)
    _lib.check(lib.vqa_tanh_fwd(_p(x), _p(y), x.numel(), _st(x)), "vqa_tanh_fwd")
    return y


def tanh_bwd(dy, y):
    lib = _lib.load()
    dx = torch.empty_like(y)
    _lib.check(lib.vqa_tanh_bwd(_p(dy), _p(y), _p(dx), y.numel(), _st(y)), "vqa_tanh_bwd")
    return dx


def mul(a, b):
    lib = _lib.load()
    z = torch.empty_like(a)
    _lib.check(lib.vqa_mul(_p(a), _p(b), _p(z), a.numel(), _st(a)), "vqa_mul")
    return z


def mul_bwd(dz, a, b):
    lib = _lib.load()
    da, db = torch.empty_like(a), torch.empty_like(b)
    _lib.check(lib.vqa_mul_bwd(_p(dz), _p(a), _p(b), _p(da), _p(db), a.numel(), _st(a)), "vqa_mul_bwd")
    return da, db


def add_inplace(acc, x):
    lib = _lib.load()
    _lib.check(lib.vqa_add_inplace(_p(acc), _p(x), acc.numel(), _st(acc)), "vqa_add_inplace")
    return acc


def embed_bwd_into(dx_tm, q, dE, lens=None):
    """dE[q[b,t],:] += dx_tm[t,b,:] (scatter-add into an existing gradient buffer); lens as in embed_bwd."""
    lib = _lib.load()
    T, B, W = dx_tm.shape
    _lib.check(lib.vqa_embed_bwd_len(_p(dx_tm), _p(q), _p(lens), _p(dE), B, T, W, dE.shape[0], _st(dx_tm)),
               "vqa_embed_bwd_len")


def _live_ptr(live_rows, T):
    live = np.ascontiguousarray(live_rows, dtype=np.int32)
    assert live.shape == (T,)
    return live


def gru_seq_fwd(xp, Wg_h, Wc_h, lens, T, B, H, live_rows=None):
    """Fused recurrence.  xp [T,B,3H] (x-projections + biases); returns hs [T+1,B,H] and the tape.
    live_rows (host int32 [T], rows sorted by length, longest first): run each step on the live prefix only."""
    lib = _lib.load()
    hs = torch.zeros(T + 1, B, H, dtype=torch.float32, device=xp.device)
    r, u, c, rh = (_f32(T, B, H, like=xp) for _ in range(4))
    if live_rows is None:
        _lib.check(lib.vqa_gru_seq_fwd(_p(xp), _p(Wg_h), _p(Wc_h), _p(lens), _p(hs), _p(r), _p(u), _p(c), _p(rh), T, B,
                                       H, _st(xp)), "vqa_gru_seq_fwd")
    else:
        live = _live_ptr(live_rows, T)
        _lib.check(lib.vqa_gru_seq_fwd_live(_p(xp), _p(Wg_h), _p(Wc_h), _p(lens), live.ctypes.data, _p(hs), _p(r), _p(u),
                                            _p(c), _p(rh), T, B, H, _st(xp)), "vqa_gru_seq_fwd_live")
    return hs, (r, u, c, rh)


def gru_seq_bwd(dh_T, Wg_h, Wc_h, lens, hs, tape, T, B, H, live_rows=None):
    """Returns dxp [T,B,3H] = (dr_pre | du_pre | dc_pre); dh_T is consumed.  live_rows as in gru_seq_fwd."""
    lib = _lib.load()
    r, u, c, rh = tape
    dxp = _f32(T, B, 3 * H, like=hs)
    scratch = _f32(B, H, like=hs)
    if live_rows is None:
        _lib.check(lib.vqa_gru_seq_bwd(_p(dh_T), _p(Wg_h), _p(Wc_h), _p(lens), _p(hs), _p(r), _p(u), _p(c), _p(dxp),
                                       _p(scratch), T, B, H, _st(hs)), "vqa_gru_seq_bwd")
    else:
        live = _live_ptr(live_rows, T)
        _lib.check(lib.vqa_gru_seq_bwd_live(_p(dh_T), _p(Wg_h), _p(Wc_h), _p(lens), live.ctypes.data, _p(hs), _p(r),
                                            _p(u), _p(c), _p(dxp), _p(scratch), T, B, H, _st(hs)), "vqa_gru_seq_bwd_live")
    return dxp


def softmax_ce(z, label, valid, inv_valid_sum, topk=5, want_dz=True):
    """n_way_classification_loss rows: stats [rows,4] = {ce, top1, topk, valid} (x valid); dz for backward."""
    lib = _lib.load()
    rows, A = z.shape
    stats = _f32(rows, 4, like=z)
    dz = torch.empty_like(z) if want_dz else None
    _lib.check(lib.vqa_softmax_ce_fwd(_p(z), _p(label), _p(valid), topk, _p(inv_valid_sum), _p(stats), _p(dz), rows, A,
                                      _st(z)), "vqa_softmax_ce_fwd")
    return stats, dz
