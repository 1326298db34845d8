"""Known-answer tests pinning the oracle: the reference has no
fixtures, so these closed-form cases are the oracle's pins."""
import numpy as np
import pytest

from oracle import vqa_oracle as O


def test_sigmoid_ce_points():
    z = np.array([0.0, 1.0, -1.0, 100.0, -100.0])
    for t in (0.0, 1.0, 0.3):
        got = O.sigmoid_ce(z, np.full_like(z, t))
        want = -(t * np.log(1 / (1 + np.exp(-z))) + (1 - t) * np.log(1 - 1 / (1 + np.exp(-z)) + 1e-300))
        # closed forms for the extreme points (log(1-sigmoid(100)) underflows above)
        want[3] = 100.0 * (1 - t)
        want[4] = 100.0 * t
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_gru_zero_kernel_stays_zero_and_bias_case():
    B, T, W, H = 3, 4, 5, 6
    x = np.random.default_rng(0).standard_normal((B, T, W))
    lens = np.array([4, 2, 0], np.int32)
    Wg = np.zeros((W + H, 2 * H)); bg = np.ones(2 * H)      # GRUCell default gate bias 1.0
    Wc = np.zeros((W + H, H)); bc = np.zeros(H)
    h, _ = O.gru_forward(x, lens, Wg, bg, Wc, bc)
    assert np.all(h == 0)
    bc = np.full(H, 0.7)
    h, _ = O.gru_forward(x, np.array([1, 1, 0], np.int32), Wg, bg, Wc, bc)
    want = (1 - 1 / (1 + np.exp(-1.0))) * np.tanh(0.7)      # h1 = (1-sigmoid(1)) * tanh(b)
    np.testing.assert_allclose(h[0], want, rtol=1e-14)
    np.testing.assert_allclose(h[1], want, rtol=1e-14)
    assert np.all(h[2] == 0)                                  # len 0 -> zero state


def test_gru_reset_gate_applied_before_candidate_matmul():
    # TF GRUCell: c = tanh([x, r*h] Wc); torch/MIOpen: tanh(Wx + r*(Uh)).  They differ when
    # Wc's h-block mixes units; check the oracle follows the TF form on a hand case.
    W, H = 1, 2
    x = np.zeros((1, 2, W))
    Wg = np.zeros((W + H, 2 * H)); bg = np.array([0.0, 10.0, -50.0, -50.0])  # r=(.5,~1), u~0
    Wc = np.zeros((W + H, H)); Wc[1, 1] = 1.0; Wc[2, 0] = 1.0                # swaps the units
    bc = np.array([1.0, 0.0])
    h, tape = O.gru_forward(x, np.array([2], np.int32), Wg, bg, Wc, bc)
    h1 = np.tanh(np.array([1.0, 0.0]))
    r = 1 / (1 + np.exp(-np.array([0.0, 10.0])))
    c2 = np.tanh(np.array([1.0 + r[1] * h1[1], r[0] * h1[0]]))
    np.testing.assert_allclose(h[0], c2, rtol=1e-9, atol=1e-12)


def test_gru_length_masking_returns_state_at_len():
    rng = np.random.default_rng(1)
    B, T, W, H = 4, 6, 3, 5
    x = rng.standard_normal((B, T, W))
    Wg = rng.standard_normal((W + H, 2 * H)) * 0.3; bg = np.ones(2 * H)
    Wc = rng.standard_normal((W + H, H)) * 0.3; bc = np.zeros(H)
    lens = np.array([6, 3, 1, 0], np.int32)
    h, _ = O.gru_forward(x, lens, Wg, bg, Wc, bc)
    for b in range(B):
        L = int(lens[b])
        hb, _ = O.gru_forward(x[b:b + 1, :max(L, 1)], np.array([L], np.int32), Wg, bg, Wc, bc)
        np.testing.assert_allclose(h[b], hb[0], rtol=1e-13, atol=1e-15)


def test_layer_norm_constant_block_gives_beta_and_joint_stats():
    x = np.full((2, 3, 4), 5.0)
    gamma = np.arange(4.0) + 1; beta = np.arange(4.0) * 0.5
    y, xhat, _ = O.layer_norm_forward(x, gamma, beta)
    np.testing.assert_allclose(y, np.broadcast_to(beta, x.shape))
    # statistics are over ALL non-batch axes jointly (36x1024 for v_linear_v)
    x = np.random.default_rng(2).standard_normal((2, 3, 4))
    y, xhat, rstd = O.layer_norm_forward(x, np.ones(4), np.zeros(4))
    for b in range(2):
        np.testing.assert_allclose(y[b].mean(), 0, atol=1e-14)
        np.testing.assert_allclose(y[b].var(), 1, rtol=1e-9)
    assert abs(y[0, 0].mean()) > 1e-3      # per-row stats would make this 0


def test_attention_single_box_and_pooling():
    rng = np.random.default_rng(3)
    B, R, H, D = 2, 5, 8, 6
    v = rng.standard_normal((B, R, H)); qv = rng.standard_normal((B, H))
    w = rng.standard_normal((H, 1)); b = np.array([0.3])
    m = np.ones((B, R, H))
    att, _ = O.hadamard_attention_forward(v, np.array([1, 5]), qv, w, b, m)
    np.testing.assert_allclose(att[0], [1, 0, 0, 0, 0])
    np.testing.assert_allclose(att[1].sum(), 1.0)
    s = ((v[1] * qv[1]) / 0.8) @ w[:, 0] + 0.3
    e = np.exp(s - s.max())
    np.testing.assert_allclose(att[1], e / e.sum(), rtol=1e-12)


def test_attention_zero_boxes_gives_nan_row_like_tf():
    v = np.ones((1, 3, 2)); qv = np.ones((1, 2))
    att, _ = O.hadamard_attention_forward(v, np.array([0]), qv, np.ones((2, 1)), np.zeros(1), np.ones((1, 3, 2)))
    assert np.all(np.isnan(att))


def test_untrained_head_all_minus_100():
    rng = np.random.default_rng(4)
    dims = dict(Vq=20, W=6, D=8, H=4, A=11)
    p = O.init_params(rng, "vlmap_answer", head="untrained", dtype=np.float64, **dims)
    table, nbox = O.make_table(rng, 5, 3, dims["D"], np.float64)
    batch = O.make_batch(rng, 4, 5, dims["Vq"], dims["A"], 5, np.float64)
    am = O.make_answer_masks(rng, dims["A"], 8, np.float64)
    masks = O.make_dropout_masks(rng, 4, 3, dims["H"], np.float64)
    loss, report, out, mid, _ = O.forward(p, batch, table, nbox, am, masks)
    assert np.all(mid["logit"] == -100.0)
    assert np.all(out["pred"] == 0)                          # argmax ties -> first index
    tgt = batch["answer_target"]
    want = (tgt * 100.0).sum(1).mean() + dims["A"] * np.log1p(np.exp(-100.0))
    np.testing.assert_allclose(report["answer_report_loss"], want, rtol=1e-12)
    want_train = ((tgt * 100.0 + np.log1p(np.exp(-100.0))) * am["train"]).sum(1).mean()
    np.testing.assert_allclose(report["answer_train_loss"], want_train, rtol=1e-12)


def test_report_guarded_ratio_and_keys():
    rng = np.random.default_rng(5)
    A, B = 7, 3
    z = rng.standard_normal((B, A)); tgt = np.zeros((B, A)); tgt[:, 0] = 1.0
    am = {"train": np.ones(A), "obj": np.ones(A), "attr": np.zeros(A), "exist": np.ones(A)}
    loss, report, out, _ = O.loss_and_report(z, tgt, am, "vlmap_answer")
    assert sorted(report.keys()) == sorted(O.REPORT_KEYS)
    assert report["test_max_acc"] == 0 and report["normal_test_acc"] == 0   # where(den==0, den, .)
    np.testing.assert_allclose(report["normal_exist_acc"], report["exist_acc"] / report["max_exist_acc"])


def test_clip_and_adam_first_step_closed_form():
    p = {"a": np.array([1.0, -2.0]), "e": np.zeros((3, 2))}
    g = {"a": np.array([30.0, 40.0]), "e": np.zeros((3, 2))}
    dx = np.zeros((1, 1, 2))
    st = O.new_opt_state()
    norm = O.clip_adam_step(p, g, ["a", "e"], st, 1e-3, dx, "e")
    assert norm == 50.0
    # clipped grad = g*20/50; first Adam step moves by lr*sign(g) (to eps)
    np.testing.assert_allclose(p["a"], [1.0 - 1e-3, -2.0 - 1e-3], rtol=0, atol=1e-9)
    # embedding norm uses un-aggregated slices
    g2 = {"e": np.array([[2.0, 0.0], [0, 0], [0, 0]])}
    dx2 = np.array([[[1.0, 0.0], [1.0, 0.0]]])              # two occurrences of token 0
    assert O.global_norm(g2, ["e"], dx2, "e") == pytest.approx(np.sqrt(2.0))
