"""Oracle pins (SURVEY.md 8c i,iii): NumPy oracle vs the independently composed
torch-autograd restatement (f64), and central finite differences."""
import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O
from oracle import torch_ref as TR

DIMS = dict(Vq=30, W=12, D=24, H=16, A=21)


def _case(seed, model_type, B=5, R=6, T=7, N=9, full_boxes=False, dtype=np.float64):
    rng = np.random.default_rng(seed)
    p = O.perturb_ln_params(O.init_params(rng, model_type, dtype=dtype, **DIMS), rng)
    table, nbox = O.make_table(rng, N, R, DIMS["D"], dtype, full_boxes=full_boxes)
    batch = O.make_batch(rng, B, T, DIMS["Vq"], DIMS["A"], N, dtype)
    am = O.make_answer_masks(rng, DIMS["A"], 15, dtype, exist_all=False)
    masks = O.make_dropout_masks(rng, B, R, DIMS["H"], dtype, model_type=model_type, num_marginal=7)
    return p, table, nbox, batch, am, masks


@pytest.mark.parametrize("model_type", ["vlmap_answer", "standard", "standard_word2vec", "standard_testmask",
                                        "vlmap_answer_vqa_all2", "vlmap_answer_noc", "vlmap_answer_vqa_all",
                                        "vlmap_answer2", "vlmap_answer_no_noise", "vlmap_answer_full", "vlmap_answer_adapt",
                                        "vlmap_answer_ent"])
def test_forward_and_grads_match_torch_autograd(model_type):
    p, table, nbox, batch, am, masks = _case(11, model_type)
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, model_type)
    grads, dx = O.backward(p, batch, am, masks, tape, model_type)
    tloss, tmid, tgrads, tdx = TR.loss_and_grads(p, batch, table, nbox, am, masks, model_type)
    assert abs(loss - tloss) <= 1e-10 * max(1, abs(tloss))
    for k in ("v_linear_v", "condition", "q_linear_v", "att_score", "pooled_V_ft", "pooled_linear_l",
              "l_linear_l", "joint", "v_joint", "l_joint", "logit", "v_adapt", "q_L_mean", "q_L_log_sigma_sq",
              "q_L_mean_noise", "marginal_prob"):
        if k in mid or k in tmid:
            np.testing.assert_allclose(mid[k], tmid[k], rtol=1e-9, atol=1e-11, err_msg=k)
    for k in p:
        if O.is_const(k):
            assert k not in grads and k not in O.train_var_names(p, model_type)    # a tf.constant: no gradient
            continue
        np.testing.assert_allclose(grads[k], tgrads[k], rtol=1e-7, atol=1e-11, err_msg=k)
    np.testing.assert_allclose(dx, tdx, rtol=1e-7, atol=1e-12)


def test_vqa_all2_known_answers():
    """vqa/model_vlmap_answer_vqa_all2.py:196-244: a trainable second head on the FIXED joint, logits summed, loss
    = ce(fixed) * train_mask + ce(tuned), pred = argmax(fixed * test_mask + tuned * train_mask); the tuned_q_linear_l /
    tuned_joint_fc branch is built but feeds nothing (the tuned head reads `joint`, :216-217)."""
    mt = "vlmap_answer_vqa_all2"
    p, table, nbox, batch, am, masks = _case(21, mt)
    sc = O.scope_names(mt)
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, mt)
    z1 = mid["joint"] @ p["WordWeightAnswer/fc/weights"] + p["WordWeightAnswer/fc/biases"]
    z2 = mid["joint"] @ p["TunedWordWeightAnswer/fc/weights"] + p["TunedWordWeightAnswer/fc/biases"]
    np.testing.assert_allclose(mid["logit"], z1 + z2, rtol=1e-12)
    tgt, train = batch["answer_target"], am["train"]
    assert report["answer_train_loss"] == pytest.approx((O.sigmoid_ce(z1, tgt) * train + O.sigmoid_ce(z2, tgt)).sum(1).mean(), rel=1e-12)
    assert report["answer_report_loss"] == pytest.approx((O.sigmoid_ce(z1, tgt) + O.sigmoid_ce(z2, tgt)).sum(1).mean(), rel=1e-12)
    np.testing.assert_array_equal(out["pred"], np.argmax(z1 * (1 - train) + z2 * train, axis=1))
    tloss, tmid, tgrads, _ = TR.loss_and_grads(p, batch, table, nbox, am, masks, mt)
    np.testing.assert_array_equal(out["pred"], tmid["pred"])
    np.testing.assert_allclose(mid["tuned_l_linear_l"], tmid["tuned_l_linear_l"], rtol=1e-9, atol=1e-11)
    # variable contract: the frozen set is model_vlmap_answer's (:85-94), the tuned scopes train but get no gradient
    names = O.train_var_names(p, mt)
    assert "TunedWordWeightAnswer/fc/weights" in names and "tuned_joint_fc/fc/weights" in names
    assert not any(n.split("/")[0] in ("q_linear_l", "pooled_linear_l", "joint_fc", "WordWeightAnswer") for n in names)
    assert O.transfer_var_names(p, mt) == O.transfer_var_names({k: v for k, v in p.items() if "uned" not in k}, "vlmap_answer")
    grads, _ = O.backward(p, batch, am, masks, tape, mt)
    for k in p:
        if k.startswith(("tuned_q_linear_l/", "tuned_joint_fc/")):
            assert not grads[k].any() and not tgrads[k].any(), k
    assert np.abs(grads["TunedWordWeightAnswer/fc/weights"]).max() > 0
    # an untrained fixed head (weights 0, bias -100) leaves the decision on training answers to the tuned head alone
    p2 = dict(p, **{"WordWeightAnswer/fc/weights": np.zeros_like(p["WordWeightAnswer/fc/weights"]),
                    "WordWeightAnswer/fc/biases": np.full_like(p["WordWeightAnswer/fc/biases"], -100.0)})
    out2 = O.forward(p2, batch, table, nbox, am, masks, mt)[2]
    mixed = np.where(train > 0, z2, -100.0)
    np.testing.assert_array_equal(out2["pred"], np.argmax(mixed, axis=1))


def test_vqa_all_row_min_substitution_and_masked_sum_loss():
    """vqa/model_vlmap_answer_vqa_all.py: fixed logits of answers unknown to the word-weight directory sit at the row
    minimum (:192-194), both loss terms are train-masked and the tuned one sees the sum (:234-242), pred = argmax of the sum."""
    mt = "vlmap_answer_vqa_all"
    p, table, nbox, batch, am, masks = _case(41, mt)
    assert (am["exist"] == 0).any() and (am["exist"] == 1).any()
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, mt)
    z1 = mid["joint"] @ p["WordWeightAnswer/fc/weights"] + p["WordWeightAnswer/fc/biases"]
    z2 = mid["joint"] @ p["TunedWordWeightAnswer/fc/weights"] + p["TunedWordWeightAnswer/fc/biases"]
    z1m = np.where(am["exist"] > 0, z1, z1.min(axis=1, keepdims=True))
    np.testing.assert_allclose(mid["logit"], z1m + z2, rtol=1e-12)
    tgt = batch["answer_target"]
    ell = O.sigmoid_ce(z1m, tgt) + O.sigmoid_ce(z1m + z2, tgt)
    assert report["answer_train_loss"] == pytest.approx((ell * am["train"]).sum(1).mean(), rel=1e-12)
    assert report["answer_report_loss"] == pytest.approx(ell.sum(1).mean(), rel=1e-12)
    np.testing.assert_array_equal(out["pred"], np.argmax(z1m + z2, axis=1))
    # with every answer known the substitution is the identity
    am1 = dict(am, exist=np.ones_like(am["exist"]))
    l1 = O.forward(p, batch, table, nbox, am1, masks, mt)[3]["logit"]
    np.testing.assert_allclose(l1, z1 + z2, rtol=1e-12)
    # finite differences through the min (the FROZEN head's bias at the arg-min answer of row 0 moves every unknown answer)
    grads, _ = O.backward(p, batch, am, masks, tape, mt)
    a = int(np.argmin(z1[0]))
    name, eps = "WordWeightAnswer/fc/biases", 1e-6
    pp = {k: v.copy() for k, v in p.items()}
    pp[name][a] += eps; lp = O.forward(pp, batch, table, nbox, am, masks, mt)[0]
    pp[name][a] -= 2 * eps; lm = O.forward(pp, batch, table, nbox, am, masks, mt)[0]
    assert abs((lp - lm) / (2 * eps) - grads[name][a]) <= 1e-6 + 1e-5 * abs(grads[name][a])


def test_noc_variant_known_answers():
    """vqa/model_vlmap_answer_noc.py (= nocarch): two un-composed branches joint_v(pooled_linear_l), joint_l(l_linear_l), their
    transferred heads, logit = v_logit + l_logit; frozen / transfer sets of :80-103."""
    mt = "vlmap_answer_noc"
    p, table, nbox, batch, am, masks = _case(31, mt)
    assert "joint_fc/fc/weights" not in p and "WordWeightAnswer/fc/weights" not in p
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, mt)
    zv = mid["v_joint"] @ p["WordWeightAnswerV/fc/weights"] + p["WordWeightAnswerV/fc/biases"]
    zl = mid["l_joint"] @ p["WordWeightAnswerL/fc/weights"] + p["WordWeightAnswerL/fc/biases"]
    np.testing.assert_allclose(mid["logit"], zv + zl, rtol=1e-12)
    assert np.all(mid["v_joint"][masks["joint"] == 0] == 0) and np.all(mid["l_joint"][masks["joint_l"] == 0] == 0)
    ell = O.sigmoid_ce(mid["logit"], batch["answer_target"])
    assert report["answer_train_loss"] == pytest.approx((ell * am["train"]).sum(1).mean(), rel=1e-12)
    names = O.train_var_names(p, mt)
    assert not any(n.split("/")[0] in O.FROZEN_TOP_SCOPES_NOC for n in names) and "v_linear_v/fc/weights" in names
    assert sorted({n.split("/")[0] for n in O.transfer_var_names(p, mt)}) == ["joint_l", "joint_v", "pooled_linear_l", "q_linear_l"]
    # the identical twin file
    l2 = O.forward(p, batch, table, nbox, am, masks, "vlmap_answer_nocarch")[0]
    assert l2 == loss
    # untrained heads: each -100 bias, summed -> -200 everywhere
    pu = O.init_params(np.random.default_rng(1), mt, dtype=np.float64, head="untrained", **DIMS)
    z = O.forward(pu, batch, table, nbox, am, masks, mt)[3]["logit"]
    assert np.all(z == -200.0)


def test_word2vec_head_known_answers():
    """vqa/model_standard_word2vec.py:180-202: classifier FC to 300-d, logits = joint2 x fixed answer-GloVe matrix,
    train loss masked by the train-answer mask, report loss not."""
    p, table, nbox, batch, am, masks = _case(15, "standard_word2vec")
    sc = O.scope_names("standard_word2vec")
    assert p[sc["head"] + "/fc/weights"].shape == (2 * DIMS["H"], DIMS["W"]) and p[sc["glove"]].shape == (DIMS["W"], DIMS["A"])
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, "standard_word2vec")
    j2 = mid["joint"] @ p[sc["head"] + "/fc/weights"] + p[sc["head"] + "/fc/biases"]
    np.testing.assert_allclose(mid["logit"], j2 @ p[sc["glove"]], rtol=1e-12)
    ell = O.sigmoid_ce(mid["logit"], batch["answer_target"])
    assert report["answer_train_loss"] == pytest.approx((ell * am["train"]).sum(1).mean(), rel=1e-12)
    assert report["answer_report_loss"] == pytest.approx(ell.sum(1).mean(), rel=1e-12)
    assert loss == report["answer_train_loss"] and report["answer_train_loss"] < report["answer_report_loss"]
    # a zero GloVe column gives logit 0 for that answer whatever the input
    p[sc["glove"]][:, 3] = 0
    assert np.all(O.forward(p, batch, table, nbox, am, masks, "standard_word2vec")[3]["logit"][:, 3] == 0)


def test_finite_differences_on_selected_params():
    p, table, nbox, batch, am, masks = _case(12, "vlmap_answer", B=3, R=4, T=5)
    _, _, _, _, tape = O.forward(p, batch, table, nbox, am, masks)
    grads, _ = O.backward(p, batch, am, masks, tape)
    rng = np.random.default_rng(0)
    eps = 1e-6
    for name in p:
        flat = p[name].reshape(-1)
        for i in rng.choice(flat.size, size=min(3, flat.size), replace=False):
            old = flat[i]
            flat[i] = old + eps
            lp = O.forward(p, batch, table, nbox, am, masks)[0]
            flat[i] = old - eps
            lm = O.forward(p, batch, table, nbox, am, masks)[0]
            flat[i] = old
            fd = (lp - lm) / (2 * eps)
            an = grads[name].reshape(-1)[i]
            assert abs(fd - an) <= 1e-6 + 1e-5 * abs(an), (name, i, fd, an)


def test_f32_oracle_close_to_f64_oracle():
    p, table, nbox, batch, am, masks = _case(13, "vlmap_answer")
    l64, _, _, mid64, _ = O.forward(p, batch, table, nbox, am, masks)
    c = lambda d: {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in d.items()}
    l32, _, _, mid32, _ = O.forward(c(p), c(batch), table.astype(np.float32), nbox, c(am), c(masks))
    assert mid32["logit"].dtype == np.float32
    np.testing.assert_allclose(mid32["logit"], mid64["logit"], rtol=0, atol=5e-5)


def test_train_step_matches_torch_cpu_baseline_port():
    p, table, nbox, batch, am, masks = _case(14, "vlmap_answer", dtype=np.float32)
    p_t = {k: v.copy() for k, v in p.items()}
    stepper = TR.CpuTrainStep(p_t, table, nbox, am, "vlmap_answer", lr=1e-3)
    st = O.new_opt_state()
    for it in range(3):
        loss, report, out, mid, grads, norm = O.train_step(p, batch, table, nbox, am, masks, st, 1e-3)
        tl, tn = stepper(batch, masks)
        assert abs(loss - tl) < 1e-4 * max(1, abs(tl))
        assert abs(norm - tn) < 1e-4 * max(1, tn)
    for k in O.train_var_names(p, "vlmap_answer"):
        if k.endswith("score/fc/biases"):
            # d(loss)/d(score bias) == 0 exactly (softmax shift invariance); in f32 it is
            # rounding noise that Adam normalises to +-lr steps -> not comparable.
            continue
        np.testing.assert_allclose(p[k], stepper.P[k].detach().numpy(), rtol=0, atol=2e-4, err_msg=k)
    for k in p:
        if k.split("/")[0] in O.FROZEN_TOP_SCOPES_VLMAP:
            np.testing.assert_array_equal(p[k], p_t[k])     # frozen vars never move


def test_standard_testmask_is_standard_with_the_masked_training_loss():
    """vqa/model_standard_testmask.py:262-268, 295-304: the network of model_standard, the training loss masked by the
    train-answer mask (so head gradients of the test answers vanish), the report loss unmasked, nine report scalars."""
    p, table, nbox, batch, am, masks = _case(17, "standard_testmask")
    assert sorted(p) == sorted(_case(17, "standard")[0])                              # same variables
    assert O.train_var_names(p, "standard_testmask") == sorted(p)                     # all trainable
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, "standard_testmask")
    loss_s, report_s, _, mid_s, _ = O.forward(p, batch, table, nbox, am, masks, "standard")
    np.testing.assert_array_equal(mid["logit"], mid_s["logit"])                       # same forward
    assert report["answer_report_loss"] == report_s["answer_report_loss"] == loss_s
    assert loss == report["answer_train_loss"] < loss_s                               # masked sum of positive terms
    grads, _ = O.backward(p, batch, am, masks, tape, "standard_testmask")
    test_cols = am["train"] == 0
    assert test_cols.any() and np.all(grads["reasoning/classifier/fc/weights"][:, test_cols] == 0)
    assert np.all(grads["reasoning/classifier/fc/biases"][test_cols] == 0)
    r9 = O.testmask_report(report)
    assert sorted(r9) == sorted(["answer_train_loss", "answer_report_loss", "answer_accuracy", "exist_answer_accuracy",
                                 "test_answer_accuracy", "normal_test_answer_accuracy", "max_exist_answer_accuracy",
                                 "test_max_answer_accuracy", "test_max_exist_answer_accuracy"])
    assert r9["answer_accuracy"] == report["answer_acc"] and r9["test_max_exist_answer_accuracy"] == report["test_max_exist_acc"]


# ------------------------------------------------------------------------------------------------------------------
# the five older ablations of model_vlmap_answer: known answers + finite differences
# ------------------------------------------------------------------------------------------------------------------
def _fd(p, name, idx, fwd, eps=1e-6):
    pp = {k: v.copy() for k, v in p.items()}
    pp[name][idx] += eps
    lp = fwd(pp)
    pp[name][idx] -= 2 * eps
    return (lp - fwd(pp)) / (2 * eps)


def test_answer2_tanh_layer_feeds_q_linear_l_only():
    """vqa/model_vlmap_answer2.py:127-131,164: q_L_ft2 = tanh(LN(fc(q_L_ft))) is the `condition` and q_linear_l's input;
    q_linear_v keeps reading the GRU state"""
    mt = "vlmap_answer2"
    p, table, nbox, batch, am, masks = _case(51, mt)
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, mt)
    h = tape["h"]
    pre = h @ p["q_L_ft2/fc/weights"] + p["q_L_ft2/fc/biases"]
    ln = (pre - pre.mean(1, keepdims=True)) / np.sqrt(pre.var(1, keepdims=True) + O.LN_EPS)
    want = np.tanh(ln * p["q_L_ft2/LayerNorm/gamma"] + p["q_L_ft2/LayerNorm/beta"])
    np.testing.assert_allclose(mid["condition"], want, rtol=1e-12, atol=1e-14)
    assert np.abs(mid["condition"]).max() <= 1.0
    base = {k: v for k, v in p.items() if not k.startswith("q_L_ft2/")}
    mid0 = O.forward(base, batch, table, nbox, am, masks, "vlmap_answer")[3]
    np.testing.assert_array_equal(mid0["q_linear_v"], mid["q_linear_v"])          # the attention branch is untouched
    assert np.abs(mid0["l_linear_l"] - mid["l_linear_l"]).max() > 1e-3
    assert sorted(report) == sorted(O.REPORT_KEYS)
    names = O.train_var_names(p, mt)
    assert "q_L_ft2/fc/weights" in names and "q_L_ft2/LayerNorm/gamma" in names and "q_linear_l/fc/weights" not in names
    grads, _ = O.backward(p, batch, am, masks, tape, mt)
    fwd = lambda pp: O.forward(pp, batch, table, nbox, am, masks, mt)[0]
    for name, idx in (("q_L_ft2/fc/weights", (3, 5)), ("q_L_ft2/LayerNorm/beta", (2,)), ("encode_L/rnn/gru_cell/candidate/bias", (4,))):
        assert abs(_fd(p, name, idx, fwd) - grads[name][idx]) <= 1e-7 + 1e-5 * abs(grads[name][idx]), name


def test_no_noise_and_full_known_answers():
    """vqa/model_vlmap_answer_no_noise.py:122-125,157 and _full.py:124-134,166,217-223,272-276"""
    p, table, nbox, batch, am, masks = _case(52, "vlmap_answer_full")
    H = DIMS["H"]
    # identity q_L_mean: no_noise is model_vlmap_answer with the older 9-key report
    pn = {k: v for k, v in p.items() if not k.startswith("q_L_log_sigma_sq/")}
    pn["q_L_mean/fc/weights"], pn["q_L_mean/fc/biases"] = np.eye(H), np.zeros(H)
    ln_, rn, _, mn, _ = O.forward(pn, batch, table, nbox, am, masks, "vlmap_answer_no_noise")
    base = {k: v for k, v in pn.items() if not k.startswith("q_L_mean/")}
    l0, r0, _, m0, _ = O.forward(base, batch, table, nbox, am, masks, "vlmap_answer")
    np.testing.assert_allclose(mn["logit"], m0["logit"], rtol=1e-12)
    assert sorted(rn) == sorted(O.TESTMASK_REPORT) and rn["answer_accuracy"] == r0["answer_acc"] and ln_ == pytest.approx(l0, rel=1e-12)
    # full: zero noise = no_noise's logits; the loss carries 0.1 x the KL term
    z0 = dict(masks, noise=np.zeros_like(masks["noise"]))
    lf, rf, _, mf, tf_ = O.forward(p, batch, table, nbox, am, z0, "vlmap_answer_full")
    pnn = {k: v for k, v in p.items() if not k.startswith("q_L_log_sigma_sq/")}
    lnn, _, _, mnn, _ = O.forward(pnn, batch, table, nbox, am, masks, "vlmap_answer_no_noise")
    np.testing.assert_allclose(mf["logit"], mnn["logit"], rtol=1e-12)
    qm, qs = mf["q_L_mean"], mf["q_L_log_sigma_sq"]
    kl = -0.5 * np.mean(np.sum(1 + qs - qm ** 2 - np.exp(qs), axis=1))
    assert rf["latent_loss"] == pytest.approx(kl, rel=1e-12) and rf["train_latent_loss"] == pytest.approx(0.1 * kl, rel=1e-12)
    assert lf == pytest.approx(lnn + 0.1 * kl, rel=1e-12)
    assert sorted(rf) == sorted(list(O.TESTMASK_REPORT) + ["latent_loss", "train_latent_loss"])
    # a standard-normal posterior (mean 0, log sigma^2 0) has zero KL, and then x = noise exactly
    pz = dict(p)
    for s_ in ("q_L_mean", "q_L_log_sigma_sq"):
        pz[s_ + "/fc/weights"], pz[s_ + "/fc/biases"] = np.zeros((H, H)), np.zeros(H)
    lz, rz, _, mz, _ = O.forward(pz, batch, table, nbox, am, masks, "vlmap_answer_full")
    assert rz["latent_loss"] == 0.0
    np.testing.assert_array_equal(mz["q_L_mean_noise"], masks["noise"])
    # finite differences (with noise): the sigma head and the mean head
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, "vlmap_answer_full")
    grads, _ = O.backward(p, batch, am, masks, tape, "vlmap_answer_full")
    fwd = lambda pp: O.forward(pp, batch, table, nbox, am, masks, "vlmap_answer_full")[0]
    for name, idx in (("q_L_log_sigma_sq/fc/weights", (1, 2)), ("q_L_log_sigma_sq/fc/biases", (7,)), ("q_L_mean/fc/weights", (0, 3)),
                      ("encode_L/rnn/gru_cell/gates/kernel", (14, 9))):
        assert abs(_fd(p, name, idx, fwd) - grads[name][idx]) <= 1e-7 + 1e-5 * abs(grads[name][idx]), name


def test_adapt_pools_the_adapted_features():
    """vqa/model_vlmap_answer_adapt.py:132-142: pooled_V_ft = att . v_adapt with v_adapt = relu(LN_[R,H](fc(V_ft)))"""
    mt = "vlmap_answer_adapt"
    p, table, nbox, batch, am, masks = _case(53, mt)
    assert p["pooled_linear_l/fc/weights"].shape == (DIMS["H"], DIMS["H"])
    nbox[:] = np.maximum(nbox, 1)
    nbox[batch["image_idx"][0]] = 1                                             # a one-box image: pooled = v_adapt[:, 0]
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, mt)
    assert mid["pooled_V_ft"].shape == (5, DIMS["H"]) and mid["v_adapt"].shape == (5, 6, DIMS["H"])
    np.testing.assert_allclose(mid["pooled_V_ft"][0], mid["v_adapt"][0, 0], rtol=1e-12)
    np.testing.assert_allclose(mid["pooled_V_ft"], np.einsum("br,brh->bh", mid["att_score"], mid["v_adapt"]), rtol=1e-12)
    pre = mid["V_ft"] @ p["v_adapt/fc/weights"] + p["v_adapt/fc/biases"]
    mu, var = pre.mean((1, 2), keepdims=True), pre.var((1, 2), keepdims=True)        # LN over the whole [R, H] block
    want = np.maximum((pre - mu) / np.sqrt(var + O.LN_EPS) * p["v_adapt/LayerNorm/gamma"] + p["v_adapt/LayerNorm/beta"], 0)
    np.testing.assert_allclose(mid["v_adapt"], want, rtol=1e-11, atol=1e-13)
    assert sorted(report) == sorted(O.TESTMASK_REPORT)
    grads, _ = O.backward(p, batch, am, masks, tape, mt)
    fwd = lambda pp: O.forward(pp, batch, table, nbox, am, masks, mt)[0]
    for name, idx in (("v_adapt/fc/weights", (5, 2)), ("v_adapt/LayerNorm/gamma", (3,)), ("v_linear_v/fc/weights", (1, 1))):
        assert abs(_fd(p, name, idx, fwd) - grads[name][idx]) <= 1e-7 + 1e-5 * abs(grads[name][idx]), name


def test_ent_marginal_pairing_and_closed_forms():
    """vqa/model_vlmap_answer_ent.py:191-211, 281-292"""
    mt = "vlmap_answer_ent"
    # tf.tile([M, 1]) + reshape [-1, M, L]: pairing (i, m) reads pooled_linear_l[(i * M + m) % B]
    idx = O.marginal_index(5, 7)
    tile = np.tile(np.arange(5)[:, None], (7, 1)).reshape(-1, 7)
    np.testing.assert_array_equal(idx, tile)
    assert idx[0, :6].tolist() == [0, 1, 2, 3, 4, 0] and idx[3, 0] == (3 * 7) % 5
    p, table, nbox, batch, am, masks = _case(54, mt)
    loss, report, out, mid, tape = O.forward(p, batch, table, nbox, am, masks, mt)
    n_sel = int(((am["exist"] * am["train"]) > 0.5).sum())
    assert mid["marginal_prob"].shape == (5, n_sel) and 0 < n_sel < 15
    np.testing.assert_allclose(mid["marginal_prob"].sum(1), 1.0, rtol=1e-12)
    ne = np.mean(np.sum(mid["marginal_prob"] * np.log(mid["marginal_prob"] + 1e-8), axis=1))
    assert report["entropy"] == pytest.approx(ne, rel=1e-12) and report["weighted_entropy"] == pytest.approx(0.1 * ne, rel=1e-12)
    base, rb, _, mb, _ = O.forward(p, batch, table, nbox, am, masks, "vlmap_answer")
    np.testing.assert_array_equal(mb["logit"], mid["logit"])                     # the regulariser does not touch the answer path
    assert loss == pytest.approx(base + 0.1 * ne, rel=1e-12)
    assert sorted(report) == sorted(O.REPORT_KEYS + ["entropy", "weighted_entropy"])
    # an untrained head (weights 0, bias -100): every pairing is uniform over the known training answers
    pu = dict(p, **{"WordWeightAnswer/fc/weights": np.zeros_like(p["WordWeightAnswer/fc/weights"]),
                    "WordWeightAnswer/fc/biases": np.full_like(p["WordWeightAnswer/fc/biases"], -100.0)})
    ru = O.forward(pu, batch, table, nbox, am, masks, mt)[1]
    assert ru["entropy"] == pytest.approx(np.log(1.0 / n_sel + 1e-8), rel=1e-12)
    # gradient: only through l_linear_l (pooled_linear_l is behind stop_gradient); finite differences on train vars
    grads, _ = O.backward(p, batch, am, masks, tape, mt)
    g0, _ = O.backward(p, batch, am, masks, O.forward(p, batch, table, nbox, am, masks, "vlmap_answer")[4], "vlmap_answer")
    np.testing.assert_array_equal(grads["v_linear_v/fc/weights"], g0["v_linear_v/fc/weights"])     # nothing reaches the visual side
    assert np.abs(grads["encode_L/rnn/gru_cell/gates/kernel"] - g0["encode_L/rnn/gru_cell/gates/kernel"]).max() > 0
    # tf.stop_gradient: the value of pooled_linear_l inside the regulariser is a constant of the differentiation
    fwd = lambda pp: O.forward(pp, batch, table, nbox, am, masks, mt, stop_grad_values={"pooled_linear_l": mid["pooled_linear_l"]})[0]
    for name, idx_ in (("encode_L/rnn/gru_cell/candidate/kernel", (13, 4)), ("encode_L/rnn/gru_cell/gates/bias", (6,)),
                       ("q_linear_l/fc/biases", (2,)), ("joint_fc/LayerNorm/gamma", (9,)), ("WordWeightAnswer/fc/weights", (4, 3))):
        # (many more ReLU gates than the base model: a smaller step keeps the stencil on one side of every kink)
        assert abs(_fd(p, name, idx_, fwd, eps=1e-7) - grads[name][idx_]) <= 2e-7 + 1e-4 * abs(grads[name][idx_]), name
