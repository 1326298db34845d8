"""Pins of the cfg-5 pre-training oracle: NumPy forward == independent torch forward (float64), top-k
tie semantics, LayerNorm-per-call-site variable contract."""
import numpy as np

from oracle import pretrain_oracle as PO


def _case(seed=0, B=3, n=5, R=6, D=16, H=8, L=4, Vq=20, n_ws=7, A=11, dtype=np.float64):
    rng = np.random.default_rng(seed)
    p = PO.init_params(rng, Vq, n_ws, A, W=12, D=D, H=H, dtype=dtype)
    batch = PO.make_batch(rng, B, n, R, D, L, Vq, n_ws, A, dtype)
    masks = PO.make_masks(rng, B, n, R, H, dtype)
    return p, batch, masks, n


def test_numpy_forward_matches_torch_forward():
    p, batch, masks, n = _case()
    total, report, mid = PO.forward(p, batch, masks, n)
    ttotal, tlosses, grads, slices = PO.torch_loss_and_grads(p, batch, masks, n)
    assert abs(total - ttotal) < 1e-10 * max(1, abs(ttotal))
    for k, v in tlosses.items():
        assert abs(report[k + "_loss"] - v) < 1e-10 * max(1, abs(v)), k
    # variables without a path to the loss get no gradient (V_GloVe, LearnAnswerGloVe exist only for export)
    for n_ in PO.NO_GRAD_VARS:
        assert np.all(grads[n_] == 0)
    assert np.abs(grads["classifier/fc/weights"]).max() > 0 and np.abs(grads["L_GloVe/embed_map"]).max() > 0


def test_layernorm_variables_per_call_site():
    s = PO.variable_shapes(20, 7, 11, W=12, D=16, H=8)
    for scope, cnt in (("pooled_linear_l", 4), ("q_linear_l", 4), ("joint_fc", 4), ("wordset_ft", 2),
                       ("spat_v_linear_v", 2), ("spat_q_linear_v", 2)):
        names = sorted(k for k in s if k.startswith(scope + "/LayerNorm"))
        assert len(names) == 2 * cnt, (scope, names)
    assert "pooled_linear_l/LayerNorm_3/gamma" in s and "pooled_linear_l/LayerNorm/gamma" in s
    # un-suffixed LayerNorm = object blank-fill: the set the VQA model restores (filter_transfer_vars)
    p, batch, masks, n = _case(1)
    t0, _, _ = PO.forward(p, batch, masks, n)
    p2 = dict(p); p2["pooled_linear_l/LayerNorm/gamma"] = p["pooled_linear_l/LayerNorm/gamma"] * 2
    _, r2, _ = PO.forward(p2, batch, masks, n)
    _, r1, _ = PO.forward(p, batch, masks, n)
    assert r2["obj_blank_fill_loss"] != r1["obj_blank_fill_loss"]
    for k in ("attr_blank_fill_loss", "obj_wordset_loss", "attr_wordset_loss"):
        assert r2[k] == r1[k]


def test_n_way_loss_topk_ties_and_mask():
    logits = np.zeros((1, 2, 8)); labels = np.array([[6, 2]]); valid = np.array([[1.0, 0.0]])
    loss, acc, topk = PO.n_way_classification_loss(logits, labels, valid)
    np.testing.assert_allclose(loss, np.log(8))
    assert acc == 0.0          # argmax of an all-tie row is index 0 != 6
    assert topk == 0.0         # tf.nn.top_k keeps the LOWEST indices on ties: 0..4, label 6 is out
    labels = np.array([[4, 2]])
    assert PO.n_way_classification_loss(logits, labels, valid)[2] == 1.0
