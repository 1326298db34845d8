"""GPU parity of each C-ABI op against the CPU oracle (bit-exact for integer
outputs, stated float tolerances otherwise).  Runs only on the MI355X box."""
import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O

pytestmark = pytest.mark.gpu

ops = None


@pytest.fixture(scope="module", autouse=True)
def _load():
    global ops
    import vqa_transfer_externaldata_amd  # noqa: F401
    from vqa_transfer_externaldata_amd import ops as _ops
    ops = _ops
    assert torch.cuda.is_available()


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def close(got, want, rtol, atol, msg=""):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=msg)


GEMM_SHAPES = [
    # (M, N, K) incl. ragged / unaligned / tiny / split-k / big-tile shapes
    (1, 1, 1), (5, 21, 12), (7, 20, 300), (64, 64, 16), (130, 70, 33), (512, 1024, 1024), (96, 3000, 2048),
    (300, 2048, 1000), (18432 // 8, 1024, 2048), (2048, 1024, 4608), (37, 41, 53),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_gemm_matches_f64(M, N, K, layout):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    D = rng.standard_normal((M, N)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64) + bias + D
    if layout == "nn":
        got = ops.gemm(dev(A), dev(B), bias=dev(bias), addend=dev(D))
    elif layout == "nt":
        got = ops.gemm(dev(A), dev(B.T), transB=True, bias=dev(bias), addend=dev(D))
    else:
        got = ops.gemm(dev(A.T), dev(B), transA=True, bias=dev(bias), addend=dev(D))
    scale = np.sqrt(K) + 1
    close(got, want, rtol=1e-5, atol=2e-6 * scale * 4, msg=layout)


@pytest.mark.parametrize("cfg", list(range(24)))
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_gemm_every_tile_config(cfg, layout):
    """Each tile configuration forced on one shape that is ragged in M (rows past M are clamped in the
    steady-state loader), has a partial last k tile (masked loader) and needs several full tiles first."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    M, N, K = 200, 264, 1000
    rng = np.random.default_rng(cfg * 3 + len(layout))
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    D = rng.standard_normal((M, N)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64) + bias + D
    try:
        assert lib.vqa_gemm_set_config(cfg) == 0
        if layout == "nn":
            got = ops.gemm(dev(A), dev(B), bias=dev(bias), addend=dev(D), split_k=1)
        elif layout == "nt":
            got = ops.gemm(dev(A), dev(B.T), transB=True, bias=dev(bias), addend=dev(D), split_k=1)
        else:
            got = ops.gemm(dev(A.T), dev(B), transA=True, bias=dev(bias), addend=dev(D), split_k=1)
        got = got.cpu().numpy()
    finally:
        lib.vqa_gemm_set_config(-1)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=3e-4, err_msg="cfg %d %s" % (cfg, layout))


@pytest.mark.parametrize("K", [508, 512, 516, 544, 548, 576, 1028])
@pytest.mark.parametrize("cfg,layout", [(20, "nn"), (20, "nt"), (21, "nn"), (19, "tn"), (20, "tn")])
def test_gemm_staggered_loop_edges(cfg, layout, K):
    """the SIMD-partner stagger of the 8-wave loop (k loops of >= 16 full tiles: one entry fetch, pairs of intervals,
    one transition interval, then the generic remainder): k lengths around the threshold, odd and even tile counts,
    with and without a partial last tile, ragged M and N"""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    M, N = 300, 200
    rng = np.random.default_rng(K + cfg)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64)
    try:
        assert lib.vqa_gemm_set_config(cfg) == 0
        if layout == "nn":
            got = ops.gemm(dev(A), dev(B), split_k=1)
        elif layout == "nt":
            got = ops.gemm(dev(A), dev(B.T), transB=True, split_k=1)
        else:
            got = ops.gemm(dev(A.T), dev(B), transA=True, split_k=1)
        got = got.cpu().numpy()
    finally:
        lib.vqa_gemm_set_config(-1)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=3e-4)


def test_gemm_identity_asymmetric_and_inplace_addend():
    # A = I with an asymmetric B catches a transposed C write; addend aliasing C (GRU in-place form)
    n = 96
    B = np.arange(n * n, dtype=np.float32).reshape(n, n) / 7.0
    got = ops.gemm(dev(np.eye(n, dtype=np.float32)), dev(B))
    np.testing.assert_array_equal(got.cpu().numpy(), B)
    Cbuf = dev(np.ones((n, n), np.float32))
    ops.gemm(dev(np.eye(n, dtype=np.float32)), dev(B), addend=Cbuf, out=Cbuf)
    np.testing.assert_array_equal(Cbuf.cpu().numpy(), B + 1)


def test_gemm_strided_views_like_gru():
    # C and addend are column slices of a wider buffer (xp[:, :2H] with ld = 3H)
    rng = np.random.default_rng(0)
    M, H = 40, 32
    xp = rng.standard_normal((M, 3 * H)).astype(np.float32)
    h = rng.standard_normal((M, H)).astype(np.float32)
    W = rng.standard_normal((H, 2 * H)).astype(np.float32)
    xpd = dev(xp)
    ops.gemm(dev(h), dev(W), addend=xpd[:, :2 * H], out=xpd[:, :2 * H])
    want = xp.copy()
    want[:, :2 * H] += h @ W
    close(xpd, want, 1e-5, 1e-5)


def test_gemm_split_k_deterministic():
    rng = np.random.default_rng(1)
    A = dev(rng.standard_normal((4096, 300)).astype(np.float32))
    B = dev(rng.standard_normal((4096, 256)).astype(np.float32))
    c1 = ops.gemm(A, B, transA=True, split_k=8)
    c2 = ops.gemm(A, B, transA=True, split_k=8)
    assert torch.equal(c1, c2)
    close(c1, A.cpu().numpy().astype(np.float64).T @ B.cpu().numpy().astype(np.float64), 1e-5, 2e-3)


def test_gather_features_bit_exact():
    rng = np.random.default_rng(2)
    table, nbox = O.make_table(rng, 50, 36, 64, full_boxes=False)
    idx = rng.integers(0, 50, size=17).astype(np.int64)
    V, nb = ops.gather_features(dev(table), dev(nbox), dev(idx))
    np.testing.assert_array_equal(V.cpu().numpy(), table[idx])
    np.testing.assert_array_equal(nb.cpu().numpy(), nbox[idx])


def test_embedding_gather_and_scatter():
    rng = np.random.default_rng(3)
    Vq, W, B, T = 40, 300, 9, 14
    E = rng.standard_normal((Vq, W)).astype(np.float32)
    q = rng.integers(0, Vq, size=(B, T)).astype(np.int32)
    x = ops.embed_fwd(dev(E), dev(q))
    np.testing.assert_array_equal(x.cpu().numpy(), E[q].transpose(1, 0, 2))
    dx = rng.standard_normal((T, B, W)).astype(np.float32)
    dE = ops.embed_bwd(dev(dx), dev(q), Vq)
    want = np.zeros((Vq, W), np.float64)
    np.add.at(want, q.T.reshape(-1), dx.reshape(-1, W).astype(np.float64))
    close(dE, want, 1e-5, 1e-5)
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    dxm = dx.copy()
    for b in range(B):
        dxm[lens[b]:, b] = 0.0
    want_len = np.zeros((Vq, W), np.float64)
    np.add.at(want_len, q.T.reshape(-1), dxm.reshape(-1, W).astype(np.float64))
    qh = np.full((B, T), 7, np.int32)                       # a hot row: every token the same id
    try:
        for det in (0, 1):                                  # float atomics (default) / atomic-free owner waves
            assert lib.vqa_set_deterministic(det) == 0
            close(ops.embed_bwd(dev(dx), dev(q), Vq), want, 1e-5, 1e-5)
            # positions past the sequence length are skipped when lens is given (their dx is zero in the model)
            close(ops.embed_bwd(dev(dx), dev(q), Vq, lens=dev(lens)), want_len, 1e-5, 1e-5)
            d1, d2 = ops.embed_bwd(dev(dx), dev(qh), Vq), ops.embed_bwd(dev(dx), dev(qh), Vq)
            close(d1[7], dx.reshape(-1, W).astype(np.float64).sum(0), 1e-5, 1e-4)
            rest = d1.clone()
            rest[7] = 0
            assert float(rest.abs().max()) == 0.0
            if det:
                assert torch.equal(d1, d2)                  # bitwise reproducible
    finally:
        lib.vqa_set_deterministic(0)


@pytest.mark.parametrize("G,rows,N", [(7, 1, 1024), (5, 36, 1024), (3, 1, 2048), (4, 6, 16), (3, 5, 21), (2, 36, 128)])
@pytest.mark.parametrize("drop", [False, True])
def test_ln_relu_fwd_bwd(G, rows, N, drop):
    rng = np.random.default_rng(G * 100 + rows * 10 + N)
    pre = (rng.standard_normal((G, rows, N)) * 2 + 0.5)
    gamma = 1 + 0.2 * rng.standard_normal(N); beta = 0.2 * rng.standard_normal(N)
    keep = (rng.random((G, rows, N)) < 0.5) if drop else None
    y64, xhat, rstd = O.layer_norm_forward(pre, gamma, beta)
    ln = y64
    y64 = np.maximum(y64, 0) * (keep / 0.5 if drop else 1.0)
    f = lambda a: dev(a.astype(np.float32))
    km = dev(keep.astype(np.uint8).reshape(G * rows, N)) if drop else None
    y, mean, rs = ops.ln_relu_fwd(f(pre.reshape(G * rows, N)), f(gamma), f(beta), rows, km, 0.5)
    close(y, y64.reshape(G * rows, N), 1e-4, 2e-5)
    close(rs, rstd.reshape(G), 1e-5, 0)
    dy = rng.standard_normal((G, rows, N))
    dln = dy * (keep / 0.5 if drop else 1.0) * (ln > 0)
    dxhat = dln * gamma
    m1 = dxhat.mean(axis=(1, 2), keepdims=True); m2 = (dxhat * xhat).mean(axis=(1, 2), keepdims=True)
    dpre64 = rstd * (dxhat - m1 - xhat * m2)
    dpre, dg, db, dbias = ops.ln_relu_bwd(f(dy.reshape(G * rows, N)), f(pre.reshape(G * rows, N)), mean, rs, f(gamma),
                                          f(beta), rows, km, 0.5)
    sc = np.abs(dpre64).max()
    close(dpre, dpre64.reshape(G * rows, N), 1e-3, 2e-4 * sc)
    close(dg, (dln * xhat).sum(axis=(0, 1)), 1e-3, 1e-3)
    close(db, dln.sum(axis=(0, 1)), 1e-3, 1e-3)
    close(dbias, dpre64.sum(axis=(0, 1)), 1e-3, 2e-3 * max(sc, 1))


@pytest.mark.parametrize("B,R,H,D", [(3, 36, 1024, 2048), (4, 5, 16, 24), (2, 100, 64, 32)])
@pytest.mark.parametrize("drop", [False, True])
def test_attention_pool_fwd_bwd(B, R, H, D, drop):
    rng = np.random.default_rng(B + R + H)
    v = np.maximum(rng.standard_normal((B, R, H)), 0); qv = np.maximum(rng.standard_normal((B, H)), 0)
    V = np.maximum(rng.standard_normal((B, R, D)), 0)
    w = rng.standard_normal((H, 1)) * 0.1; bias = np.array([0.2])
    nb = rng.integers(1, R + 1, size=B).astype(np.int32); nb[0] = R
    keep = (rng.random((B, R, H)) < 0.8).astype(np.float64) if drop else np.ones((B, R, H))
    att64, feat = O.hadamard_attention_forward(v, nb, qv, w, bias, keep if drop else np.ones_like(keep) * 0.8)
    p64 = np.einsum("br,brd->bd", att64, V)
    f = lambda a: dev(a.astype(np.float32))
    km = dev(keep.astype(np.uint8)) if drop else None
    # without dropout the oracle call above used mask = keep_prob so that mask/keep == 1
    att, pooled = ops.attn_pool_fwd(f(v), f(qv), f(V), dev(nb), f(w[:, 0]), f(bias), km, 0.8)
    close(att, att64, 1e-4, 1e-6)
    close(pooled, p64, 1e-4, 1e-5)
    assert np.all(att.cpu().numpy()[np.arange(R)[None, :] >= nb[:, None]] == 0)     # masked regions exactly 0
    dp = rng.standard_normal((B, D))
    datt = np.einsum("bd,brd->br", dp, V)
    ds = att64 * (datt - (att64 * datt).sum(1, keepdims=True))
    m = keep / 0.8 if drop else np.ones_like(keep)
    dfeat = ds[:, :, None] * w[None, None, :, 0] * m
    dv64 = dfeat * qv[:, None, :]; dqv64 = (dfeat * v).sum(1)
    dw64 = np.einsum("br,brh->h", ds, v * qv[:, None, :] * m); db64 = ds.sum()
    dv, dqv, dw, db = ops.attn_pool_bwd(f(dp), f(v), f(qv), f(V), att, f(w[:, 0]), km, 0.8)
    close(dv, dv64, 1e-3, 1e-4 * np.abs(dv64).max())
    close(dqv, dqv64, 1e-3, 1e-4 * np.abs(dqv64).max())
    close(dw, dw64, 1e-3, 1e-4 * np.abs(dw64).max())
    assert abs(float(db[0]) - db64) < 1e-4


def test_attention_num_box_one_and_zero():
    rng = np.random.default_rng(9)
    B, R, H, D = 2, 36, 64, 32
    f = lambda a: dev(a.astype(np.float32))
    v = rng.standard_normal((B, R, H)); qv = rng.standard_normal((B, H)); V = rng.standard_normal((B, R, D))
    att, pooled = ops.attn_pool_fwd(f(v), f(qv), f(V), dev(np.array([1, 0], np.int32)), f(np.ones(H)), f(np.zeros(1)))
    a = att.cpu().numpy()
    np.testing.assert_array_equal(a[0], np.eye(R, dtype=np.float32)[0])          # num_box = 1 -> [1,0,...]
    np.testing.assert_array_equal(pooled.cpu().numpy()[0], V[0, 0].astype(np.float32))
    assert np.all(np.isnan(a[1]))                                                  # num_box = 0 -> NaN row, as TF


@pytest.mark.parametrize("B,A", [(6, 3000), (5, 21), (300, 257)])
@pytest.mark.parametrize("model_type", ["vlmap_answer", "standard"])
def test_loss_report_pred_bit_exact_and_dz(B, A, model_type):
    rng = np.random.default_rng(A)
    z = rng.standard_normal((B, A)).astype(np.float32) * 3
    z[0, :] = -100.0                               # all-tie row -> argmax 0
    if A > 10:
        z[1, 5] = z[1, 9] = 50.0                   # tie -> first index
    batch = O.make_batch(rng, B, 4, 10, A, 5)
    am = O.make_answer_masks(rng, A, int(A * 0.75), exist_all=False)
    loss, report, out, ell = O.loss_and_report(z.astype(np.float64), batch["answer_target"].astype(np.float64),
                                               {k: v.astype(np.float64) for k, v in am.items()}, model_type)
    dm = {k: dev(v) for k, v in am.items()}
    stats, pred, dz, rep = ops.loss_fwd(dev(z), dev(batch["answer_target"]), dm, model_type == "vlmap_answer")
    np.testing.assert_array_equal(pred.cpu().numpy(), out["pred"])               # integer: bit exact
    assert pred.cpu().numpy()[0] == 0
    for k in O.REPORT_KEYS:
        assert abs(rep[k] - report[k]) <= 2e-5 * max(1.0, abs(report[k])), (k, rep[k], report[k])
    dz64 = (O.sigmoid(z.astype(np.float64)) - batch["answer_target"]) / B
    if model_type == "vlmap_answer":
        dz64 = dz64 * am["train"]
    close(dz, dz64, 1e-5, 1e-8)
    st = stats.cpu().numpy()
    np.testing.assert_allclose(st[:, 2], out["all_score"], rtol=0, atol=0)
    np.testing.assert_allclose(st[:, 14], out["max_train_score"], rtol=0, atol=0)


def test_colsum_and_sumsq_and_mask():
    rng = np.random.default_rng(4)
    X = rng.standard_normal((7168, 96)).astype(np.float32)
    close(ops.colsum(dev(X)), X.astype(np.float64).sum(0), 1e-4, 1e-3)
    for M, N in ((512, 1024), (37, 70), (1, 4)):      # two-stage, ragged and single-stage shapes
        Xs = [rng.standard_normal((M, N)).astype(np.float32) for _ in range(3)]
        outs = ops.colsum3(*[dev(x) for x in Xs])
        for x, o in zip(Xs, outs):
            close(o, x.astype(np.float64).sum(0), 1e-4, 1e-3)
            np.testing.assert_array_equal(o.cpu().numpy(), ops.colsum(dev(x)).cpu().numpy())   # same summation order
    g = rng.standard_normal(1000003).astype(np.float32)
    buf = torch.zeros(1000004, device="cuda")[:1000003]
    buf.copy_(dev(g))
    assert abs(float(ops.sumsq(buf)) / float((g.astype(np.float64) ** 2).sum()) - 1) < 1e-5
    m1 = ops.dropout_mask(1 << 20, 123, 0, 0.8, "cuda")
    m2 = ops.dropout_mask(1 << 20, 123, 0, 0.8, "cuda")
    m3 = ops.dropout_mask(1 << 19, 123, 1 << 19, 0.8, "cuda")
    assert torch.equal(m1, m2) and torch.equal(m1[1 << 19:], m3)                  # counter based
    for off in (1, 2, 3, 5, 4096 + 7):                                            # ... at any offset, not only aligned ones
        assert torch.equal(ops.dropout_mask(1000, 123, off, 0.8, "cuda"), m1[off:off + 1000]), off
    assert float(ops.dropout_mask(1 << 16, 9, 0, 1.0, "cuda").float().min()) == 1.0
    assert float(ops.dropout_mask(1 << 16, 9, 0, 0.5, "cuda").float().mean()) == pytest.approx(0.5, abs=1e-2)
    assert abs(float(m1.float().mean()) - 0.8) < 5e-3
    # ragged length and an unaligned destination give the same stream of bits as the 16-byte-store form
    big = torch.zeros((1 << 12) + 64, dtype=torch.uint8, device="cuda")
    from vqa_transfer_externaldata_amd import _lib
    import ctypes as C
    lib = _lib.load()
    for off, n in ((0, 4099), (3, 4099), (16, 33)):
        view = big[off:off + n]
        _lib.check(lib.vqa_dropout_mask(C.c_void_p(view.data_ptr()), n, 123, 0, 0.8, None), "vqa_dropout_mask")
        torch.cuda.synchronize()
        assert torch.equal(view, m1[:n])


@pytest.mark.parametrize("M,N", [(5, 8), (700, 1024), (25600, 3072), (129, 260)])
def test_colsum_16_byte_form_equals_the_scalar_form(M, N):
    """N, ldx multiples of 4 take the 16-byte kernels: bit for bit the scalar kernels' result (same per-column order),
    here against the scalar path taken by an N - 1 wide view of the same matrix"""
    rng = np.random.default_rng(M + N)
    X = dev(rng.standard_normal((M, N)).astype(np.float32))
    v4 = ops.colsum(X)                                   # 16-byte form
    sc = ops.colsum(X[:, :N - 1])                        # N - 1 columns, row stride N: scalar form
    assert torch.equal(v4[:N - 1], sc)
    close(v4, X.double().sum(0).cpu().numpy(), 1e-4, 1e-3 * max(1.0, np.sqrt(M) / 10))


@pytest.mark.parametrize("M,N", [(7, 40), (512, 1024), (25600, 300), (3, 5)])
def test_colsum_accumulating_forms(M, N):
    """vqa_colsum_acc / vqa_colsum3_acc: out (+)= column sums, chosen per output; one- and two-pass reductions"""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M + N)
    Xs = [dev(rng.standard_normal((M, N)).astype(np.float32)) for _ in range(3)]
    base = [dev(rng.standard_normal(N).astype(np.float32)) for _ in range(3)]
    P = lambda t: C.c_void_p(t.data_ptr())
    wsn = 3 * int(lib.vqa_colsum_workspace_floats(M, N)) + 4
    ws = torch.empty(wsn, device="cuda")
    plain = [ops.colsum(x) for x in Xs]
    for mask in (0, 5, 7, 2):
        outs = [b.clone() for b in base]
        _lib.check(lib.vqa_colsum3_acc(P(Xs[0]), P(Xs[1]), P(Xs[2]), M, N, N, P(outs[0]), P(outs[1]), P(outs[2]), mask,
                                       P(ws), wsn, None), "vqa_colsum3_acc")
        for i in range(3):
            want = plain[i] + base[i] if (mask >> i) & 1 else plain[i]
            assert torch.equal(outs[i], want), (mask, i)
    for acc in (0, 1):
        out = base[0].clone()
        _lib.check(lib.vqa_colsum_acc(P(Xs[0]), M, N, N, P(out), acc, P(ws), wsn, None), "vqa_colsum_acc")
        assert torch.equal(out, plain[0] + base[0] if acc else plain[0])


def test_gru_pack_and_unpack_of_the_input_rows():
    """vqa_gru_pack_wx / vqa_gru_unpack_dwx: x rows of the gate [W+H, 2H] and candidate [W+H, H] kernels side by side
    as [W, 3H] (+ biases as [3H]); the packed gradient goes back into the x rows only"""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    W, H = 12, 8
    g = torch.Generator(device="cuda").manual_seed(2)
    wg = torch.randn(W + H, 2 * H, device="cuda", generator=g)
    wc = torch.randn(W + H, H, device="cuda", generator=g)
    bg, bc = torch.randn(2 * H, device="cuda", generator=g), torch.randn(H, device="cuda", generator=g)
    wx, bx = torch.empty(W, 3 * H, device="cuda"), torch.empty(3 * H, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.vqa_gru_pack_wx(P(wg), P(wc), P(bg), P(bc), P(wx), P(bx), W, H, None), "pack")
    assert torch.equal(wx, torch.cat([wg[:W], wc[:W]], 1)) and torch.equal(bx, torch.cat([bg, bc]))
    dwx = torch.randn(W, 3 * H, device="cuda", generator=g)
    gwg, gwc = torch.full_like(wg, 7.0), torch.full_like(wc, 9.0)
    _lib.check(lib.vqa_gru_unpack_dwx(P(dwx), P(gwg), P(gwc), W, H, None), "unpack")
    assert torch.equal(gwg[:W], dwx[:, :2 * H]) and torch.equal(gwc[:W], dwx[:, 2 * H:])
    assert bool((gwg[W:] == 7.0).all()) and bool((gwc[W:] == 9.0).all())               # the h rows are not touched


def test_constant_column_of_the_inputs_yields_the_bias_gradient():
    """vqa_embed_fwd_ld / vqa_gru_unpack_dwx_bias: the time-major inputs carry a constant 1 in column W (zeros up to the
    row stride), so X^T dXP holds the x rows of both kernels' gradients in rows 0..W-1 and the two bias gradients
    (column sums of dXP) in row W -- rnn GRUCell's bias gradients of vqa/model_vlmap_answer.py:97-105"""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    B, T, W, H, Vq = 5, 3, 12, 8, 20
    Wp = ((W + 1 + 3) // 4) * 4
    g = torch.Generator(device="cuda").manual_seed(3)
    E = torch.randn(Vq, W, device="cuda", generator=g)
    q = torch.randint(0, Vq, (B, T), device="cuda", generator=g, dtype=torch.int32)
    x = torch.full((T, B, Wp), 5.0, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.vqa_embed_fwd_ld(P(E), P(q), P(x), B, T, W, Vq, Wp, None), "embed_ld")
    assert torch.equal(x[:, :, :W], ops.embed_fwd(E, q))
    assert bool((x[:, :, W] == 1).all()) and bool((x[:, :, W + 1:] == 0).all())
    dxp = torch.randn(T * B, 3 * H, device="cuda", generator=g)
    dwx = ops.gemm(x.view(T * B, Wp), dxp, transA=True)
    gwg, gwc = torch.full((W + H, 2 * H), 7.0, device="cuda"), torch.full((W + H, H), 9.0, device="cuda")
    gbg, gbc = torch.empty(2 * H, device="cuda"), torch.empty(H, device="cuda")
    _lib.check(lib.vqa_gru_unpack_dwx_bias(P(dwx), P(gwg), P(gwc), P(gbg), P(gbc), W, H, None), "unpack_bias")
    ref = x.view(T * B, Wp)[:, :W].double().t() @ dxp.double()
    assert (gwg[:W].double() - ref[:, :2 * H]).abs().max() < 1e-4 and (gwc[:W].double() - ref[:, 2 * H:]).abs().max() < 1e-4
    cs = dxp.double().sum(0)
    assert (gbg.double() - cs[:2 * H]).abs().max() < 1e-4 and (gbc.double() - cs[2 * H:]).abs().max() < 1e-4
    assert bool((gwg[W:] == 7.0).all()) and bool((gwc[W:] == 9.0).all())
    with pytest.raises(_lib.VqaHotError if hasattr(_lib, "VqaHotError") else Exception):
        _lib.check(lib.vqa_gru_unpack_dwx_bias(P(dwx), P(gwg), P(gwc), P(gbg), None, W, H, None), "unpack_bias")
    with pytest.raises(Exception):
        _lib.check(lib.vqa_embed_fwd_ld(P(E), P(q), P(x), B, T, W, Vq, W - 1, None), "embed_ld")


@pytest.mark.parametrize("rows,A", [(7, 12), (33, 4000), (5, 4096), (9, 1000), (4, 1028), (6, 3001), (3, 4100), (2, 8192)])
def test_softmax_ce_rows_match_f64_and_both_kernels_agree(rows, A):
    """vqa_softmax_ce_fwd (n_way_classification_loss, vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:675-706):
    {ce, top-1, top-k, valid} per row and dz against float64; the register-resident kernel (A <= 4096, A % 4 == 0) and
    the three-pass kernel agree exactly on the integer outputs and to rounding on the rest.  Ties: tf.nn.top_k /
    argmax order equal logits by index."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(rows * 7919 + A)
    z = (rng.standard_normal((rows, A)) * 3).astype(np.float32)
    label = rng.integers(0, A, size=rows).astype(np.int32)
    z[0, :] = 1.5                                          # all equal: argmax 0, the label's rank is its index
    label[0] = 3
    if rows > 2:
        z[1, label[1]] = z[1].max() + 1                    # a clear top-1 hit
        z[2, [5, A - 1]] = z[2].max() + 2                  # tie for the maximum: index 5 wins
        label[2] = A - 1
    valid = (rng.random(rows) < 0.7).astype(np.float32)
    valid[0] = 1
    inv = np.array([1.0 / valid.sum()], np.float32)
    topk = 5
    outs = []
    try:
        for fast in (1, 0):
            lib.vqa_softmax_set_fast(fast)
            st, dz = ops.softmax_ce(dev(z), dev(label), dev(valid), dev(inv), topk=topk)
            outs.append((st.cpu().numpy(), dz.cpu().numpy()))
    finally:
        lib.vqa_softmax_set_fast(1)
    z64 = z.astype(np.float64)
    mx = z64.max(1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(z64 - mx).sum(1))
    ar = np.arange(rows)
    zl = z64[ar, label]
    ce = (lse - zl) * valid
    top1 = (z64.argmax(1) == label) * valid
    rank = ((z64 > zl[:, None]) | ((z64 == zl[:, None]) & (np.arange(A)[None, :] < label[:, None]))).sum(1)
    tk = (rank < topk) * valid
    onehot = np.zeros((rows, A)); onehot[ar, label] = 1
    dz_ref = (np.exp(z64 - lse[:, None]) - onehot) * (valid * inv[0])[:, None]
    for st, dz in outs:
        assert np.abs(st[:, 0] - ce).max() < 2e-5 * max(1, np.abs(ce).max())
        assert np.array_equal(st[:, 1], top1) and np.array_equal(st[:, 2], tk) and np.array_equal(st[:, 3], valid)
        assert np.abs(dz - dz_ref).max() < 3e-6
    assert np.array_equal(outs[0][0][:, 1:], outs[1][0][:, 1:])
    assert np.abs(outs[0][1] - outs[1][1]).max() < 1e-6 and np.abs(outs[0][0][:, 0] - outs[1][0][:, 0]).max() < 1e-5


def test_errors_are_reported_not_swallowed():
    from vqa_transfer_externaldata_amd import VqaHotError
    a = torch.zeros(4, 4, device="cuda")
    with pytest.raises(VqaHotError):
        ops.gemm(a.t().contiguous(), a, transA=True, transB=True)                 # unsupported layout
    with pytest.raises(VqaHotError):
        ops.attn_pool_fwd(torch.zeros(1, 2, 6, device="cuda"), torch.zeros(1, 6, device="cuda"),
                          torch.zeros(1, 2, 8, device="cuda"), torch.ones(1, dtype=torch.int32, device="cuda"),
                          torch.zeros(6, device="cuda"), torch.zeros(1, device="cuda"))   # H % 4 != 0


@pytest.mark.parametrize("B,R,N,K,Nt", [(7, 36, 1024, 2048, 20), (64, 36, 1024, 2048, 200), (3, 5, 64, 32, 4), (9, 36, 128, 256, 6)])
def test_gather_fused_gemm_equals_gather_then_gemm(B, R, N, K, Nt):
    """vqa_gemm_f32_gather: features[image_idx] fused into the GEMM's operand load (vqa/model_vlmap_answer.py:110-129).
    The gathered block it leaves behind must be bit-identical to np.take, the product must match float64, and
    out-of-range / repeated indices behave like vqa_gather_features (clamped)."""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(B * 31 + K)
    table = np.maximum(rng.standard_normal((Nt, R, K)), 0).astype(np.float32)
    idx = rng.integers(0, Nt, size=B).astype(np.int64)
    idx[0] = idx[-1]                                                       # repeated image
    if B > 4:
        idx[1], idx[2] = -3, Nt + 5                                        # clamped to 0 / Nt-1
    W = rng.standard_normal((K, N)).astype(np.float32) * 0.05
    bias = rng.standard_normal(N).astype(np.float32)
    t_d, i_d, w_d, b_d = dev(table), dev(idx), dev(W), dev(bias)
    out = torch.full((B * R, N), float("nan"), device="cuda")
    gathered = torch.full((B * R, K + 8), float("nan"), device="cuda")     # ld > K: the tail columns stay untouched
    P = lambda t: C.c_void_p(t.data_ptr())
    for tall in (20, 21):
        _lib.check(lib.vqa_gemm_set_tall_config(tall), "tall")
        out.fill_(float("nan")); gathered.fill_(float("nan"))
        _lib.check(lib.vqa_gemm_f32_gather(B * R, N, K, P(t_d), K, P(i_d), R, Nt, P(w_d), N, P(out), N, P(b_d), P(gathered),
                                           K + 8, None), "vqa_gemm_f32_gather")
        torch.cuda.synchronize()
        src = np.clip(idx, 0, Nt - 1)
        want_rows = table[src].reshape(B * R, K)
        g = gathered.cpu().numpy()
        np.testing.assert_array_equal(g[:, :K], want_rows)                 # bit-exact gather
        assert np.all(np.isnan(g[:, K:]))
        want = want_rows.astype(np.float64) @ W.astype(np.float64) + bias
        close(out, want, rtol=1e-5, atol=2e-6 * (np.sqrt(K) + 1) * 4)
        # and without the by-product
        out2 = torch.empty_like(out)
        _lib.check(lib.vqa_gemm_f32_gather(B * R, N, K, P(t_d), K, P(i_d), R, Nt, P(w_d), N, P(out2), N, P(b_d), None, 0, None),
                   "vqa_gemm_f32_gather")
        assert torch.equal(out, out2)
    _lib.check(lib.vqa_gemm_set_tall_config(20), "tall")
    assert lib.vqa_gemm_f32_gather(B * R, N, K - 4, P(t_d), K, P(i_d), R, Nt, P(w_d), N, P(out), N, None, None, 0, None) == -4


@pytest.mark.parametrize("B,rep,R,H,D", [(5, 1, 36, 1024, 2048), (3, 5, 36, 1024, 2048), (4, 1, 20, 512, 4096), (2, 2, 40, 256, 2048),
                                         (2, 5, 40, 256, 4096), (3, 5, 17, 768, 2048), (2, 5, 1, 512, 2048)])
@pytest.mark.parametrize("drop", [False, True])
def test_attention_fast_forward_equals_generic(B, rep, R, H, D, drop):
    """The loads-in-flight forward kernel (H | 256, D | 2048) computes what the generic kernel does: same summation
    order, equal up to the compiler's choice of fused multiply-adds (a few ulp), masked regions exactly zero."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(B + rep + R)
    f = lambda a: dev(a.astype(np.float32))
    v, qv = f(np.maximum(rng.standard_normal((B, R, H)), 0)), f(np.maximum(rng.standard_normal((B * rep, H)), 0))
    V = f(np.maximum(rng.standard_normal((B, R, D)), 0))
    w, bias = f(rng.standard_normal(H) * 0.1), f(np.array([0.2]))
    nbv = rng.integers(1, R + 1, size=B).astype(np.int32); nbv[0] = R
    nb = dev(nbv)
    km = dev((rng.random((B * rep, R, H)) < 0.8).astype(np.uint8)) if drop else None
    res = []
    try:
        for fast in (0, 1, 3):      # generic | default (per-memory kernel at rep 5) | per-query fast kernel
            lib.vqa_attn_set_fast(fast)
            res.append(ops.attn_pool_fwd_rep(v, qv, V, nb, w, bias, rep, km, 0.8))
    finally:
        lib.vqa_attn_set_fast(1)
    for k in (1, 2):
        torch.testing.assert_close(res[k][0], res[0][0], rtol=2e-6, atol=1e-8)
        torch.testing.assert_close(res[k][1], res[0][1], rtol=2e-6, atol=1e-7)
    a = res[1][0].cpu().numpy().reshape(B, rep, R)
    assert np.all(a[np.broadcast_to(np.arange(R)[None, None, :] >= nbv[:, None, None], a.shape)] == 0)


@pytest.mark.parametrize("B,rep,R", [(5, 1, 36), (3, 5, 36), (2, 5, 17), (3, 1, 40), (2, 5, 1), (2, 1, 9)])
@pytest.mark.parametrize("drop", [False, True])
def test_attention_fast_backward_equals_generic(B, rep, R, drop):
    """The loads-in-flight backward kernel (H 1024, D 2048, 1 or 5 queries per memory) against the generic one: same
    per-lane summation order, equal up to the compiler's fused multiply-adds."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    H, D = 1024, 2048
    rng = np.random.default_rng(100 + B + rep + R)
    f = lambda a: dev(a.astype(np.float32))
    v, qv = f(np.maximum(rng.standard_normal((B, R, H)), 0)), f(np.maximum(rng.standard_normal((B * rep, H)), 0))
    V = f(np.maximum(rng.standard_normal((B, R, D)), 0))
    w, bias = f(rng.standard_normal(H) * 0.1), f(np.array([0.2]))
    nbv = rng.integers(1, R + 1, size=B).astype(np.int32); nbv[0] = R
    km = dev((rng.random((B * rep, R, H)) < 0.8).astype(np.uint8)) if drop else None
    dp = f(rng.standard_normal((B * rep, D)))
    att, _ = ops.attn_pool_fwd_rep(v, qv, V, dev(nbv), w, bias, rep, km, 0.8)
    res = []
    try:
        for fast in (0, 1):
            lib.vqa_attn_set_fast(fast)
            res.append(ops.attn_pool_bwd_rep(dp, v, qv, V, att, w, rep, km, 0.8))
    finally:
        lib.vqa_attn_set_fast(1)
    for a, b, name in zip(res[0], res[1], ("dv", "dqv", "dw")):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6 * float(a.abs().max()) + 1e-9, msg=lambda m: name + ": " + m)
    # the score bias gradient is analytically zero (a softmax gradient sums to zero): rounding noise on both sides
    assert float(res[0][3].abs().max()) < 1e-3 and float(res[1][3].abs().max()) < 1e-3


@pytest.mark.parametrize("G,rows,N", [(9, 36, 1024), (7, 5, 1024), (6, 5, 2048), (4, 8, 2048), (3, 2, 1024), (5, 7, 1024)])
@pytest.mark.parametrize("drop", [False, True])
def test_ln_register_resident_kernels_equal_generic(drop, G, rows, N):
    """Groups of 36 x 1024 (v_linear_v's block per sample) and of <= 8 rows x 1024 / 2048 (the pre-training model's 5 key
    boxes per image) take the register-resident LN kernels; they compute what the generic kernels do (same per-thread
    order; equal up to the compiler's fused multiply-adds)."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    f = lambda a: dev(a.astype(np.float32))
    pre = f(rng.standard_normal((G * rows, N)) * 2 + 0.5)
    gamma, beta = f(1 + 0.2 * rng.standard_normal(N)), f(0.2 * rng.standard_normal(N))
    dy = f(rng.standard_normal((G * rows, N)))
    km = dev((rng.random((G * rows, N)) < 0.5).astype(np.uint8)) if drop else None
    res = []
    try:
        for fast in (0, 1):
            lib.vqa_ln_set_fast(fast)
            y, mean, rs = ops.ln_relu_fwd(pre, gamma, beta, rows, km, 0.5)
            res.append((y, mean, rs) + tuple(ops.ln_relu_bwd(dy, pre, mean, rs, gamma, beta, rows, km, 0.5)))
    finally:
        lib.vqa_ln_set_fast(1)
    for a, b in zip(res[0], res[1]):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6 * float(a.abs().max()))


@pytest.mark.parametrize("mode", ["1", "2"])
def test_persistent_gru_xcd_local_chains_in_a_fresh_process(mode):
    """VQA_GRU_PERSIST_XCD (read once per process): eight chains of rows, one per XCD, 32-row tiles with 8 waves (1) or
    64-row tiles with 16 waves (2); same results as the per-step kernels, also with empty chains and a ragged last tile
    (B 70 -> chains of 32 / 64 rows, most of them empty)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQA_GRU_PERSIST_XCD=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_ops.py", "-m", "gpu", "-q", "-x", "-k",
                        "test_persistent_gru_forward_equals_stepwise", "-p", "no:cacheprovider"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "3 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("T,B,H", [(5, 96, 512), (14, 512, 1024), (3, 70, 512)])
def test_persistent_gru_forward_equals_stepwise(T, B, H):
    """vqa_gru_seq_fwd_persistent (one launch, two chains, per-chain grid barriers with write-through hand-offs)
    computes the recurrence of the per-step kernels: hs, r, u, c, r*h to rounding, no barrier time-out."""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    if lib.vqa_gru_fwd_persistent_supported(T, B, H) != 1:
        pytest.skip("persistent recurrence does not apply on this device")
    g = torch.Generator(device="cuda").manual_seed(T * 1000 + B)
    xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.3
    Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.04
    Wc = torch.randn(H, H, device="cuda", generator=g) * 0.04
    ln = torch.randint(0, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
    ln[0], ln[1] = T, 0
    P = lambda t: C.c_void_p(t.data_ptr())
    outs = []
    for persistent in (False, True):
        hs = torch.zeros(T + 1, B, H, device="cuda")
        hs[0] = torch.randn(B, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 0.1
        r, u, c, rh = (torch.full((T, B, H), float("nan"), device="cuda") for _ in range(4))
        hs[1:] = float("nan")
        if persistent:
            sync = torch.full((int(lib.vqa_gru_persistent_sync_bytes()) // 4,), 7, dtype=torch.int32, device="cuda")
            _lib.check(lib.vqa_gru_seq_fwd_persistent(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H,
                                                      P(sync), None), "persistent")
            torch.cuda.synchronize()
            import os
            if os.environ.get("VQA_GRU_PERSIST_XCD", "0") in ("1", "2"):      # eight XCD-local chains (experiment)
                per_chain = int(sync[0]) // (2 * T)
                assert int(sync[192]) == 0 and per_chain >= 8
                assert all(int(sync[16 * x]) == 2 * T * per_chain for x in range(8))
            else:
                slots = int(sync[0]) // (2 * T)
                assert int(sync[32]) == 0 and int(sync[0]) == int(sync[16]) == 2 * T * slots and slots >= 64
        else:
            _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "stepwise")
            torch.cuda.synchronize()
        outs.append((hs, r, u, c, rh))
    for name, a, b in zip(("hs", "r", "u", "c", "rh"), outs[0], outs[1]):
        assert not torch.isnan(b).any(), name
        torch.testing.assert_close(b, a, rtol=1e-5, atol=2e-6, msg=lambda m: name + ": " + m)
    assert lib.vqa_gru_fwd_persistent_supported(T, B, 300) == 0
    assert lib.vqa_gru_seq_fwd_persistent(P(xp), P(Wg), P(Wc), P(ln), P(outs[0][0]), P(outs[0][1]), P(outs[0][2]), P(outs[0][3]),
                                          P(outs[0][4]), T, B, 300, P(torch.zeros(64, dtype=torch.int32, device="cuda")), None) == -4


@pytest.mark.parametrize("T,B", [(14, 512), (5, 300), (3, 40), (4, 20), (1, 512), (7, 481), (6, 256), (2, 257)])
def test_weight_stationary_gru_forward_equals_stepwise(T, B):
    """vqa_gru_seq_fwd_ws (csrc/gru_ws.hip: one launch, recurrent weights resident in registers and LDS, eight
    XCD-local chains of two 32-row half-chains in anti-phase) computes the recurrence of the per-step kernels -- hs, r, u,
    c, r*h to rounding, every flag at its final epoch, no barrier time-out -- with full chains, ragged last chains,
    a single half-chain (B <= 32 rows in a chain), empty chains, rows of length 0 and T, and twice in a row on the same
    workspace (the fragment buffers and counters of one call must not leak into the next)."""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    H = 1024
    if lib.vqa_gru_ws_supported(T, B, H) != 1:
        pytest.skip("the weight-stationary recurrence does not apply on this device")
    g = torch.Generator(device="cuda").manual_seed(T * 1000 + B)
    Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.04
    Wc = torch.randn(H, H, device="cuda", generator=g) * 0.04
    P = lambda t: C.c_void_p(t.data_ptr())
    ws = torch.full((int(lib.vqa_gru_ws_workspace_bytes(T)) // 4,), float("nan"), device="cuda")
    for rep in range(2):
        xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.3
        ln = torch.randint(0, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
        ln[0], ln[1] = T, 0
        h0 = torch.randn(B, H, device="cuda", generator=g) * 0.1
        outs = []
        for stationary in (False, True):
            hs = torch.full((T + 1, B, H), float("nan"), device="cuda")
            hs[0] = h0
            r, u, c, rh = (torch.full((T, B, H), float("nan"), device="cuda") for _ in range(4))
            if stationary:
                _lib.check(lib.vqa_gru_seq_fwd_ws(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, P(ws), None), "ws")
                torch.cuda.synchronize()
                words = ws[:1024].view(torch.int32)
                assert int(words[512]) == 0                                   # no barrier time-out
                assert int(words[1023]) == (0x7FC00000 if rep == 0 else 5)    # the sticky word: launches never touch it (NaN fill, then 5)
                words[1023] = 5
                assert words[576:584].tolist() == [32] * 8                    # every XCD hosted 32 of the 256 workgroups
                for chain in range(8):      # a flag per (half-chain, CU): h_0 + 2 sub-phases per step handed off
                    if B > 256:             # chains of 64 rows; a live chain runs both of its half-chains
                        want = [1 + 2 * T if B > 64 * chain else 0] * 2
                    else:                   # chains of 32 rows, one half-chain each
                        want = [1 + 2 * T if B > 32 * chain else 0, 0]
                    for half in range(2):
                        line = words[32 * (2 * chain + half): 32 * (2 * chain + half) + 32].tolist()
                        assert line == [want[half]] * 32, (chain, half, line)
            else:
                _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "stepwise")
                torch.cuda.synchronize()
            outs.append((hs, r, u, c, rh))
        for name, a, b in zip(("hs", "r", "u", "c", "rh"), outs[0], outs[1]):
            assert not torch.isnan(b).any(), name
            torch.testing.assert_close(b, a, rtol=1e-5, atol=2e-6, msg=lambda m: name + ": " + m)
    assert lib.vqa_gru_ws_supported(T, B, 512) == 0 and lib.vqa_gru_ws_supported(T, 513, H) == 0 and lib.vqa_gru_ws_supported(0, B, H) == 0
    assert lib.vqa_gru_seq_fwd_ws(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, 513, H, P(ws), None) == -4
    assert lib.vqa_gru_seq_fwd_ws(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None, None) == -1


@pytest.mark.parametrize("T,B,outs", [(14, 512, False), (5, 300, True), (1, 512, False), (7, 481, True), (3, 257, False)])
def test_weight_stationary_gru_backward_equals_stepwise(T, B, outs):
    """vqa_gru_seq_bwd_ws (csrc/gru_ws.hip: back-propagation through time in one launch, W_g^T / W_c^T slabs resident)
    against the per-step kernels on the same tape: dxp = (dr_pre | du_pre | dc_pre) to rounding -- zero where a row is
    past its length -- with and without per-step output gradients (vqa_gru_seq_bwd_outs), ragged last chains, rows of
    length 0 and T, twice on the same workspace; dh_T is left untouched."""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    H = 1024
    if lib.vqa_gru_ws_bwd_supported(T, B, H) != 1:
        pytest.skip("the weight-stationary back-propagation does not apply on this device")
    g = torch.Generator(device="cuda").manual_seed(T * 1000 + B + 7)
    Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.04
    Wc = torch.randn(H, H, device="cuda", generator=g) * 0.04
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    ws = torch.full((int(lib.vqa_gru_ws_workspace_bytes(T)) // 4,), float("nan"), device="cuda")
    for rep in range(2):
        xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.3
        ln = torch.randint(0, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
        ln[0], ln[1] = T, 0
        hs = torch.zeros(T + 1, B, H, device="cuda")
        hs[0] = torch.randn(B, H, device="cuda", generator=g) * 0.1
        r, u, c, rh = (torch.empty(T, B, H, device="cuda") for _ in range(4))
        _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "fwd")
        dh = torch.randn(B, H, device="cuda", generator=g)
        d_outs = None
        if outs:
            d_outs = torch.randn(T, B, H, device="cuda", generator=g) * 0.5
            d_outs *= (torch.arange(T, device="cuda")[:, None] < ln[None, :]).float()[:, :, None]      # zero past a row's length
        dxp_a = torch.full((T, B, 3 * H), float("nan"), device="cuda")
        dxp_b = torch.full((T, B, 3 * H), float("nan"), device="cuda")
        dh_a, scratch = dh.clone(), torch.empty(B, H, device="cuda")
        if outs:
            _lib.check(lib.vqa_gru_seq_bwd_outs(P(dh_a), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(d_outs), P(dxp_a), P(scratch),
                                                T, B, H, None), "bwd_outs")
        else:
            _lib.check(lib.vqa_gru_seq_bwd(P(dh_a), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp_a), P(scratch), T, B, H, None), "bwd")
        dh_b = dh.clone()
        _lib.check(lib.vqa_gru_seq_bwd_ws(P(dh_b), P(d_outs), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp_b), T, B, H, P(ws), None), "bwd_ws")
        torch.cuda.synchronize()
        words = ws[:1024].view(torch.int32)
        assert int(words[512]) == 0 and words[576:584].tolist() == [32] * 8
        for chain in range(8):
            want = 2 * T if B > 64 * chain else 0
            for half in range(2):
                assert words[32 * (2 * chain + half): 32 * (2 * chain + half) + 32].tolist() == [want] * 32, (chain, half)
        assert torch.equal(dh_b, dh)
        assert not torch.isnan(dxp_b).any()
        scale = float(dxp_a.abs().max())
        torch.testing.assert_close(dxp_b, dxp_a, rtol=1e-5, atol=2e-6 * max(scale, 1.0))
        past = (torch.arange(T, device="cuda")[:, None] >= ln[None, :])
        assert float(dxp_b[past].abs().max()) == 0.0
    assert lib.vqa_gru_ws_bwd_supported(T, 256, H) == 0 and lib.vqa_gru_ws_bwd_supported(T, B, 512) == 0
    assert lib.vqa_gru_seq_bwd_ws(P(dh), None, P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp_b), T, 200, H, P(ws), None) == -4


def test_weight_stationary_gru_waits_for_busy_cus_and_still_agrees():
    """A launch whose 256 workgroups cannot all be placed at once (another stream's kernels hold CUs) waits for them -- its
    bounded spins are long enough for that -- and computes the same numbers: the forward runs while large matrix products
    keep a side stream busy, no error word, results equal to the quiet run bit for bit."""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    T, B, H = 6, 512, 1024
    if lib.vqa_gru_ws_supported(T, B, H) != 1:
        pytest.skip("the weight-stationary recurrence does not apply on this device")
    g = torch.Generator(device="cuda").manual_seed(77)
    xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.3
    Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.04
    Wc = torch.randn(H, H, device="cuda", generator=g) * 0.04
    ln = torch.randint(1, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
    P = lambda t: C.c_void_p(t.data_ptr())
    ws = torch.zeros(int(lib.vqa_gru_ws_workspace_bytes(T)) // 4, device="cuda")
    big = torch.randn(8192, 8192, device="cuda", generator=g)
    side = torch.cuda.Stream()
    outs = []
    for busy in (False, True):
        hs = torch.zeros(T + 1, B, H, device="cuda")
        r, u, c, rh = (torch.empty(T, B, H, device="cuda") for _ in range(4))
        torch.cuda.synchronize()
        if busy:
            with torch.cuda.stream(side):
                for _ in range(6):
                    big @ big                      # ~7 ms each: the recurrence's workgroups find the CUs taken
        _lib.check(lib.vqa_gru_seq_fwd_ws(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, P(ws), None), "ws")
        torch.cuda.synchronize()
        words = ws[:1024].view(torch.int32)
        assert int(words[512]) == 0 and int(words[1023]) == 0
        outs.append((hs, r, u, c, rh))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_experimental_bf16x3_gemm_is_f32_equivalent():
    """csrc/gemm_bf16x3.hip (experiment, not on the default path): three-way bf16 splits + six bf16 MFMA products per
    a*b.  Its error against float64 must be of the order of the exact-f32 MFMA kernel's own (a few f32 ulps of the
    accumulated magnitude), and shapes that are not whole tiles must be refused."""
    from vqa_transfer_externaldata_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 256, 256, 2048
    A = torch.randn(M, K, device="cuda", generator=g).relu_()
    B = (torch.rand(K, N, device="cuda", generator=g) * 2 - 1) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    ref = (A.double().cpu() @ B.double().cpu() + bias.double().cpu()).numpy()
    e32 = np.abs(ops.gemm(A, B, bias=bias).double().cpu().numpy() - ref).max()
    e3 = np.abs(ops.gemm_bf16x3(A, B, bias=bias).double().cpu().numpy() - ref).max()
    scale = np.abs(ref).max()
    assert e3 <= 4e-6 * scale and e3 <= 4 * e32 + 1e-6 * scale, (e3, e32, scale)
    lib = _lib.load()
    assert lib.vqa_gemm_bf16x3_supported(256, 256, 2048) == 1 and lib.vqa_gemm_bf16x3_supported(250, 256, 2048) == 0
    with pytest.raises(_lib.VqaHotError):
        ops.gemm_bf16x3(A[:250], B)


SHORTK_SHAPES = [
    # (M, N, K, lda): every template instance (K <= 64 / 128 / 256 / 304), ragged row panels, padded and unpadded rows
    (7168, 3072, 300, 304), (100, 32, 4, 4), (129, 96, 60, 64), (1000, 256, 64, 64), (257, 64, 128, 128), (640, 128, 100, 104),
    (333, 160, 256, 256), (128, 32, 252, 252), (515, 224, 296, 296), (2048, 512, 304, 304), (31, 64, 300, 300),
    (300, 96, 512, 512), (1000, 256, 400, 400), (129, 64, 308, 312),
]


@pytest.mark.parametrize("M,N,K,lda", SHORTK_SHAPES)
def test_shortk_gemm_matches_float64(M, N, K, lda):
    """csrc/gemm_shortk.hip (A stationary in registers, K <= 304) against float64 on every template instance, with ragged
    row panels, NaN in the padding columns of A (k >= K must never be read into a product) and every epilogue form."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    Afull = torch.full((M, lda), float("nan"), device="cuda")
    Afull[:, :K] = torch.randn(M, K, device="cuda", generator=g)
    A = Afull[:, :K]
    B = torch.randn(K, N, device="cuda", generator=g) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    scale = torch.rand(N, device="cuda", generator=g) + 0.5
    res = torch.randn(M, N, device="cuda", generator=g)
    prod = A.double().cpu() @ B.double().cpu()
    assert lib.vqa_gemm_shortk_supported(M, N, K, lda, N, N) == 1
    tol = dict(rtol=2e-5, atol=2e-5)
    close(ops.gemm_shortk(A, B), prod.numpy(), **tol)
    close(ops.gemm_shortk(A, B, bias=bias), (prod + bias.double().cpu()).numpy(), **tol)
    want = torch.relu(prod * scale.double().cpu() + bias.double().cpu() + res.double().cpu()).numpy()
    close(ops.gemm_shortk(A, B, bias=bias, scale=scale, residual=res, relu=True), want, **tol)
    # the unit split does not change a single bit: one workgroup, an odd count, one unit per workgroup
    ref = ops.gemm_shortk(A, B, bias=bias)
    try:
        for grid in (1, 3, 37, 1 << 20):
            _lib.check(lib.vqa_gemm_shortk_set_grid(grid), "grid")
            assert torch.equal(ops.gemm_shortk(A, B, bias=bias), ref), grid
    finally:
        _lib.check(lib.vqa_gemm_shortk_set_grid(0), "grid")
    # the other workgroup shape (8 waves, one workgroup per CU / 4 waves, two per CU) computes the same bits
    try:
        for waves in (4, 8):
            _lib.check(lib.vqa_gemm_shortk_set_waves(waves), "waves")
            assert torch.equal(ops.gemm_shortk(A, B, bias=bias), ref), waves
            got = ops.gemm_shortk(A, B, bias=bias, scale=scale, residual=res, relu=True)
            close(got, want, **tol)
    finally:
        _lib.check(lib.vqa_gemm_shortk_set_waves(0), "waves")
    # the general kernel agrees to rounding (different summation order)
    close(ops.gemm(A, B, bias=bias), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("transA,M,N,K,split", [(0, 256, 256, 2048, 1), (0, 128, 384, 4096, 4), (1, 256, 128, 4096, 1),
                                                  (1, 384, 256, 6144, 3), (1, 128, 128, 7168, 8)])
def test_experimental_bf16x3_gemm_transposed_and_split_k(transA, M, N, K, split):
    """The general entry of the bf16 x 3 experiment: transposed left operand (dW = X^T dY) and split k with its
    deterministic slab sum, against float64; error of the order of the exact f32 kernel's own."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K + split)
    A = torch.randn(K, M, device="cuda", generator=g).relu_() if transA else torch.randn(M, K, device="cuda", generator=g).relu_()
    B = (torch.rand(K, N, device="cuda", generator=g) * 2 - 1) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    Ad = A.double().cpu()
    ref = ((Ad.t() if transA else Ad) @ B.double().cpu() + bias.double().cpu()).numpy()
    got = ops.gemm_bf16x3_ex(A, B, transA=bool(transA), bias=bias, split_k=split)
    again = ops.gemm_bf16x3_ex(A, B, transA=bool(transA), bias=bias, split_k=split)
    assert torch.equal(got, again)                                   # deterministic (no atomics in the slab sum)
    e3 = np.abs(got.double().cpu().numpy() - ref).max()
    e32 = np.abs(ops.gemm(A, B, transA=bool(transA), bias=bias).double().cpu().numpy() - ref).max()
    scale = np.abs(ref).max()
    assert e3 <= 4e-6 * scale and e3 <= 4 * e32 + 1e-6 * scale, (e3, e32, scale)


def test_shortk_gemm_refuses_what_it_cannot_do():
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    assert lib.vqa_gemm_shortk_supported(128, 64, 516, 516, 64, 64) == 0      # K > 512
    assert lib.vqa_gemm_shortk_supported(128, 48, 64, 64, 48, 48) == 0        # N not whole 32-column tiles
    assert lib.vqa_gemm_shortk_supported(128, 64, 62, 64, 64, 64) == 0        # K % 4
    assert lib.vqa_gemm_shortk_supported(128, 64, 64, 66, 64, 64) == 0        # lda % 4
    A = torch.randn(128, 516, device="cuda")
    B = torch.randn(516, 64, device="cuda")
    with pytest.raises(_lib.VqaHotError):
        ops.gemm_shortk(A, B)


@pytest.mark.parametrize("T,B", [(4, 512), (3, 96), (2, 70), (2, 2560)])
def test_register_streamed_gru_step_kernels_equal_the_lds_tiled_ones(T, B):
    """csrc/gru_stream.hip (gru config 30: one wave per 32 x 32 tile and all of k, operands streamed from L2 into MFMA
    fragment registers; kept as an experiment with its numbers, profiles/r3_gru_stream.txt) computes the recurrence and
    its back-propagation of the default step kernels: ragged row tiles, k split over wave pairs (few rows) and the plain
    form (many rows)."""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    H = 1024
    g = torch.Generator(device="cuda").manual_seed(T * 100 + B)
    xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.3
    Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.04
    Wc = torch.randn(H, H, device="cuda", generator=g) * 0.04
    ln = torch.randint(0, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
    ln[0], ln[1] = T, 0
    dhT0 = torch.randn(B, H, device="cuda", generator=g)
    h0 = torch.randn(B, H, device="cuda", generator=g) * 0.1
    P = lambda t: C.c_void_p(t.data_ptr())
    outs = []
    try:
        for cfg in (-1, 30):
            _lib.check(lib.vqa_gemm_set_gru_config(cfg), "cfg")
            hs = torch.full((T + 1, B, H), float("nan"), device="cuda")
            hs[0] = h0
            r, u, c, rh = (torch.full((T, B, H), float("nan"), device="cuda") for _ in range(4))
            _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "fwd")
            dhT = dhT0.clone()
            dxp = torch.full((T, B, 3 * H), float("nan"), device="cuda")
            dhs = torch.full((B, H), float("nan"), device="cuda")
            _lib.check(lib.vqa_gru_seq_bwd(P(dhT), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H, None),
                       "bwd")
            torch.cuda.synchronize()
            outs.append((hs, r, u, c, rh, dxp))
    finally:
        lib.vqa_gemm_set_gru_config(-1)
    for name, a, b in zip(("hs", "r", "u", "c", "rh", "dxp"), outs[0], outs[1]):
        assert not torch.isnan(b).any(), name
        torch.testing.assert_close(b, a, rtol=2e-5, atol=5e-6, msg=lambda m: name + ": " + m)


@pytest.mark.parametrize("G,N", [(7, 1024), (3, 2048), (5, 1000), (4, 16), (2, 4096), (6, 2052)])
@pytest.mark.parametrize("params,extra", [(True, True), (False, False), (True, False)])
def test_ln_pair_mul_fwd_bwd(G, N, params, extra):
    """vqa_ln_pair_mul_*: LayerNorm + ReLU of two pre-activations and their product in one launch (pooled_linear_l x
    l_linear_l, vqa/model_vlmap_answer.py:163-177), and its backward with an optional extra gradient on the second output"""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(G * 7 + N)
    a, b = rng.standard_normal((G, N)) * 2 + 0.3, rng.standard_normal((G, N)) * 0.5 - 0.1
    ga, ba = 1 + 0.2 * rng.standard_normal(N), 0.2 * rng.standard_normal(N)
    gb, bb = 1 + 0.2 * rng.standard_normal(N), 0.2 * rng.standard_normal(N)
    lna, xa, ra = O.layer_norm_forward(a, ga, ba)
    lnb, xb, rb = O.layer_norm_forward(b, gb, bb)
    ya, yb = np.maximum(lna, 0), np.maximum(lnb, 0)
    f = lambda x: dev(np.ascontiguousarray(x, np.float32))
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    d = {k: f(v) for k, v in dict(a=a, b=b, ga=ga, ba=ba, gb=gb, bb=bb).items()}
    out = {k: torch.empty(G, N, device="cuda") for k in ("ya", "yb", "z", "da", "db", "pga", "pba", "pca", "pgb", "pbb", "pcb")}
    st = {k: torch.empty(G, device="cuda") for k in ("ma", "ra", "mb", "rb")}
    _lib.check(lib.vqa_ln_pair_mul_fwd(P(d["a"]), P(d["b"]), P(d["ga"]), P(d["ba"]), P(d["gb"]), P(d["bb"]), P(out["ya"]), P(out["yb"]),
                                       P(out["z"]), P(st["ma"]), P(st["ra"]), P(st["mb"]), P(st["rb"]), G, N, None), "pair fwd")
    torch.cuda.synchronize()
    close(out["ya"], ya, 1e-4, 2e-5); close(out["yb"], yb, 1e-4, 2e-5); close(out["z"], ya * yb, 1e-4, 5e-5)
    close(st["ra"], ra.reshape(G), 1e-5, 0); close(st["mb"], b.mean(1), 1e-5, 1e-6)
    dz = rng.standard_normal((G, N))
    add = rng.standard_normal((G, N)) if extra else None
    dya, dyb = dz * yb, dz * ya + (add if extra else 0)

    def ln_bwd(dy, ln, xh, rs, gam):
        dln = dy * (ln > 0)
        dxh = dln * gam
        return rs * (dxh - dxh.mean(1, keepdims=True) - xh * (dxh * xh).mean(1, keepdims=True)), dln * xh, dln
    da64, pga64, pba64 = ln_bwd(dya, lna, xa, ra, ga)
    db64, pgb64, pbb64 = ln_bwd(dyb, lnb, xb, rb, gb)
    pp = lambda k: P(out[k]) if params else None
    dz_t, add_t = f(dz), (f(add) if extra else None)            # (kept alive: the call takes raw pointers)
    _lib.check(lib.vqa_ln_pair_mul_bwd(P(dz_t), P(add_t), P(d["a"]), P(d["b"]), P(st["ma"]), P(st["ra"]), P(st["mb"]),
                                       P(st["rb"]), P(d["ga"]), P(d["ba"]), P(d["gb"]), P(d["bb"]), P(out["da"]), P(out["db"]),
                                       pp("pga"), pp("pba"), pp("pca"), pp("pgb"), pp("pbb"), pp("pcb"), G, N, None), "pair bwd")
    torch.cuda.synchronize()
    for got, want in ((out["da"], da64), (out["db"], db64)):
        close(got, want, 1e-3, 2e-4 * np.abs(want).max())
    if params:
        close(out["pga"], pga64, 1e-3, 1e-4 * max(1, np.abs(pga64).max())); close(out["pba"], pba64, 1e-3, 1e-5)
        close(out["pgb"], pgb64, 1e-3, 1e-4 * max(1, np.abs(pgb64).max())); close(out["pbb"], pbb64, 1e-3, 1e-5)
        assert torch.equal(out["pca"], out["da"]) and torch.equal(out["pcb"], out["db"])          # d(bias) partial = d(pre)
    # unsupported shapes are refused, not mis-computed
    assert lib.vqa_ln_pair_mul_fwd(P(d["a"]), P(d["b"]), P(d["ga"]), P(d["ba"]), P(d["gb"]), P(d["bb"]), P(out["ya"]), P(out["yb"]),
                                   P(out["z"]), P(st["ma"]), P(st["ra"]), P(st["mb"]), P(st["rb"]), 1, 4100, None) == -2
    assert lib.vqa_ln_pair_mul_supported(1022, None, 0) == 0 and lib.vqa_ln_pair_mul_supported(1024, None, 0) == 1


@pytest.mark.parametrize("M", [8, 31, 129, 515])
@pytest.mark.parametrize("K,N", [(300, 96), (64, 64), (128, 160)])
def test_shortk_gemm_writes_nothing_outside_its_output_view(M, K, N):
    """The short-K kernel's edges rest on buffer-descriptor range checks (dropped first-unit stores, row offsets past M on the
    last panel): C and the residual are views INSIDE a larger canary-filled allocation -- guard rows before and after, a row
    stride wider than N with guard columns -- and every canary must survive all epilogue forms."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M * 31 + K + N)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(K, N, device="cuda", generator=g) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    scale = torch.rand(N, device="cuda", generator=g) + 0.5
    guard, ld, CANARY = 40, N + 32, -7777.25
    big = torch.full((M + 2 * guard, ld), CANARY, device="cuda")
    resbig = torch.full((M + 2 * guard, ld), 0.5, device="cuda")
    C, res = big[guard:guard + M, 16:16 + N], resbig[guard:guard + M, 16:16 + N]
    assert C.stride(0) == ld and (C.data_ptr() % 16) == 0
    if lib.vqa_gemm_shortk_supported(M, N, K, K, N, ld) != 1:
        pytest.skip("shape not routed to the short-K kernel")
    prod = A.double().cpu() @ B.double().cpu()
    for kw, want in ((dict(), prod), (dict(bias=bias), prod + bias.double().cpu()),
                     (dict(bias=bias, scale=scale, residual=res, relu=True),
                      torch.relu(prod * scale.double().cpu() + bias.double().cpu() + 0.5))):
        big.fill_(CANARY)
        ops.gemm_shortk(A, B, out=C, **kw)
        torch.cuda.synchronize()
        close(C, want.numpy(), rtol=2e-5, atol=2e-5)
        mask = torch.ones_like(big, dtype=torch.bool)
        mask[guard:guard + M, 16:16 + N] = False
        assert bool((big[mask] == CANARY).all()), "a store landed outside the output view (%s)" % sorted(kw)
        assert bool((resbig == 0.5).all())
