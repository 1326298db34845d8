"""Pins of the cfg-5 pre-training oracle: NumPy forward == independent torch forward (float64), top-k
tie semantics, the LayerNorm variable contract in both readings (one per shared scope = TF 1.x / one per call site)."""
import numpy as np
import pytest

from oracle import pretrain_oracle as PO


def _case(seed=0, B=3, n=5, R=6, D=16, H=8, L=4, Vq=20, n_ws=7, A=11, dtype=np.float64, ln_shared=True):
    rng = np.random.default_rng(seed)
    p = PO.init_params(rng, Vq, n_ws, A, W=12, D=D, H=H, dtype=dtype, ln_shared=ln_shared)
    batch = PO.make_batch(rng, B, n, R, D, L, Vq, n_ws, A, dtype)
    masks = PO.make_masks(rng, B, n, R, H, dtype)
    return p, batch, masks, n


@pytest.mark.parametrize("ln_shared", [True, False])
def test_numpy_forward_matches_torch_forward(ln_shared):
    p, batch, masks, n = _case(ln_shared=ln_shared)
    assert PO.ln_shared_in(p) == ln_shared
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
    s = PO.variable_shapes(20, 7, 11, W=12, D=16, H=8, ln_shared=False)
    for scope, cnt in (("pooled_linear_l", 4), ("q_linear_l", 4), ("joint_fc", 4), ("wordset_ft", 2),
                       ("spat_v_linear_v", 2), ("spat_q_linear_v", 2)):
        names = sorted(k for k in s if k.startswith(scope + "/LayerNorm"))
        assert len(names) == 2 * cnt, (scope, names)
    assert "pooled_linear_l/LayerNorm_3/gamma" in s and "pooled_linear_l/LayerNorm/gamma" in s
    # un-suffixed LayerNorm = object blank-fill: the set the VQA model restores (filter_transfer_vars)
    p, batch, masks, n = _case(1, ln_shared=False)
    t0, _, _ = PO.forward(p, batch, masks, n)
    p2 = dict(p); p2["pooled_linear_l/LayerNorm/gamma"] = p["pooled_linear_l/LayerNorm/gamma"] * 2
    _, r2, _ = PO.forward(p2, batch, masks, n)
    _, r1, _ = PO.forward(p, batch, masks, n)
    assert r2["obj_blank_fill_loss"] != r1["obj_blank_fill_loss"]
    for k in ("attr_blank_fill_loss", "obj_wordset_loss", "attr_wordset_loss"):
        assert r2[k] == r1[k]


def test_layernorm_variables_shared_per_scope():
    """TF 1.x: leaving the string-named fc_layer scope zeroes its sub-scope counts, so the un-scoped layer_norm is
    `LayerNorm` at every call site and AUTO_REUSE shares it (vlmap/modules.py:630-650): one beta / gamma per scope,
    every head's loss depends on it, and its gradient is the sum of the per-call-site gradients."""
    s = PO.variable_shapes(20, 7, 11, W=12, D=16, H=8)
    assert PO.ln_shared_in(s)
    for scope in ("pooled_linear_l", "q_linear_l", "joint_fc", "wordset_ft", "spat_v_linear_v", "spat_q_linear_v"):
        assert sorted(k for k in s if k.startswith(scope + "/LayerNorm")) == [scope + "/LayerNorm/beta",
                                                                              scope + "/LayerNorm/gamma"]
    p, batch, masks, n = _case(1)
    _, r1, _ = PO.forward(p, batch, masks, n)
    p2 = dict(p); p2["pooled_linear_l/LayerNorm/gamma"] = p["pooled_linear_l/LayerNorm/gamma"] * 2
    _, r2, _ = PO.forward(p2, batch, masks, n)
    for k in ("obj_blank_fill_loss", "attr_blank_fill_loss", "obj_wordset_loss", "attr_wordset_loss"):
        assert r2[k] != r1[k], k
    # shared gradient == sum over the call sites of the per-site model evaluated at tied parameters
    ps = {k: v for k, v in PO.init_params(np.random.default_rng(1), 20, 7, 11, W=12, D=16, H=8, dtype=np.float64,
                                          ln_shared=False).items()}
    for k in ps:
        base = k.replace("/LayerNorm_1/", "/LayerNorm/").replace("/LayerNorm_2/", "/LayerNorm/").replace("/LayerNorm_3/", "/LayerNorm/")
        ps[k] = p[base]
    _, _, g_shared, _ = PO.torch_loss_and_grads(p, batch, masks, n)
    _, _, g_site, _ = PO.torch_loss_and_grads(ps, batch, masks, n)
    for scope, cnt in (("pooled_linear_l", 4), ("joint_fc", 4), ("wordset_ft", 2), ("spat_v_linear_v", 2)):
        for v in ("beta", "gamma"):
            tot = sum(g_site[PO.ln_name(scope, i) + "/" + v] for i in range(cnt))
            np.testing.assert_allclose(g_shared[scope + "/LayerNorm/" + v], tot, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(g_shared["classifier/fc/weights"], g_site["classifier/fc/weights"], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("ln_shared", [True, False])
def test_gate_conditioned_gradient(ln_shared):
    """relu(x) evaluated as x * gate: with the run's OWN sign pattern the gradients are the unconditioned ones bit for bit;
    with one gate flipped at a pre-activation that is not ~0 they are not (the pattern is really what is used)."""
    p, batch, masks, n = _case(3, ln_shared=ln_shared)
    cap = {}
    t0, _, g0, s0 = PO.torch_loss_and_grads(p, batch, masks, n, capture=cap)
    assert sorted(cap) == sorted(PO.RELU_SITES)
    t1, _, g1, s1 = PO.torch_loss_and_grads(p, batch, masks, n, gates=cap)
    assert t0 == t1
    for k in g0:
        np.testing.assert_array_equal(g0[k], g1[k], err_msg=k)
    flipped = {k: v.copy() for k, v in cap.items()}
    flipped["obj/bf/j"][0, 0, :] ^= True
    _, _, g2, _ = PO.torch_loss_and_grads(p, batch, masks, n, gates=flipped)
    assert np.abs(g2["joint_fc/fc/weights"] - g0["joint_fc/fc/weights"]).max() > 0


def test_n_way_loss_topk_ties_and_mask():
    logits = np.zeros((1, 2, 8)); labels = np.array([[6, 2]]); valid = np.array([[1.0, 0.0]])
    loss, acc, topk = PO.n_way_classification_loss(logits, labels, valid)
    np.testing.assert_allclose(loss, np.log(8))
    assert acc == 0.0          # argmax of an all-tie row is index 0 != 6
    assert topk == 0.0         # tf.nn.top_k keeps the LOWEST indices on ties: 0..4, label 6 is out
    labels = np.array([[4, 2]])
    assert PO.n_way_classification_loss(logits, labels, valid)[2] == 1.0
