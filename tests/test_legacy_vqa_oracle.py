"""Oracle pins of the oldest registry model, vqa/model_vqa.py (oracle/legacy_vqa_oracle.py): NumPy forward against the
independently composed torch forward, finite differences of the autograd gradients on every variable, known answers of
the BasicLSTMCell recurrence and the broadcast scoring layer, the variable / filter contract."""
import numpy as np

from oracle import legacy_vqa_oracle as LO
from oracle import vqa_oracle as O

DIMS = dict(Vq=30, W=12, D=16, L=16, M=20, A=11)


def _case(seed, B=5, R=6, T=7, N=9, dtype=np.float64):
    rng = np.random.default_rng(seed)
    p = LO.init_params(rng, dtype=dtype, **DIMS)
    for k in p:
        if k.endswith("/biases") or k.endswith("/bias"):
            p[k] = (p[k] + 0.1 * rng.standard_normal(p[k].shape)).astype(dtype)
    table, nbox = O.make_table(rng, N, R, DIMS["D"], dtype, full_boxes=False)
    batch = O.make_batch(rng, B, T, DIMS["Vq"], DIMS["A"], N, dtype, min_len=1)
    batch["q_intseq"][0, 0] = DIMS["Vq"] - 1                       # the trainable GloVe rows are used
    batch["q_intseq"][1, 0] = DIMS["Vq"] - 3
    answers = LO.make_answers(rng, DIMS["A"], DIMS["Vq"])
    answers["intseq"][2, 0] = DIMS["Vq"] - 2
    return p, table, nbox, batch, answers


def test_numpy_forward_matches_torch_and_gradients_match_finite_differences():
    p, table, nbox, batch, answers = _case(2)
    loss, report, out, mid = LO.forward(p, batch, table, nbox, answers)
    tloss, tmid, grads, sq = LO.torch_loss_and_grads(p, batch, table, nbox, answers)
    assert abs(loss - tloss) <= 1e-10 * max(1, abs(tloss))
    for k, v in tmid.items():
        np.testing.assert_allclose(mid[k], v, rtol=1e-9, atol=1e-11, err_msg=k)
    assert sorted(report) == sorted(LO.REPORT_KEYS) and sorted(grads) == sorted(k for k in p if not O.is_const(k))
    fwd = lambda pp: LO.forward(pp, batch, table, nbox, answers)[0]
    rng = np.random.default_rng(0)
    for name in sorted(grads):
        idx = tuple(int(rng.integers(0, s)) for s in p[name].shape)
        pp = {k: v.copy() for k, v in p.items()}
        eps = 1e-6
        pp[name][idx] += eps
        lp = fwd(pp)
        pp[name][idx] -= 2 * eps
        fd = (lp - fwd(pp)) / (2 * eps)
        assert abs(fd - grads[name][idx]) <= 2e-7 + 2e-5 * abs(grads[name][idx]), (name, idx, fd, grads[name][idx])
    assert sq > 0 and np.abs(grads["GloVe/learn"]).max() > 0


def test_lstm_and_scoring_known_answers():
    p, table, nbox, batch, answers = _case(3)
    L, W = DIMS["L"], DIMS["W"]
    x = np.zeros((2, 3, W))
    K0, b0 = np.zeros((W + L, 4 * L)), np.zeros(4 * L)
    # zero kernel, zero bias: j = 0 -> c stays 0 -> h = tanh(0) * 0.5 = 0
    assert not LO.lstm_final_h(x, np.array([3, 1]), K0, b0).any()
    # bias only on j and o: c1 = sigmoid(0) tanh(bj) , h1 = tanh(c1) sigmoid(bo); forget gate sees f + 1
    b = b0.copy(); b[L:2 * L] = 0.7; b[3 * L:] = -0.3
    c1 = 0.5 * np.tanh(0.7)
    h1 = np.tanh(c1) / (1 + np.exp(0.3))
    np.testing.assert_allclose(LO.lstm_final_h(x, np.array([1, 1]), K0, b), h1, rtol=1e-13)
    c2 = c1 / (1 + np.exp(-1.0)) + 0.5 * np.tanh(0.7)               # c sigmoid(0 + forget_bias 1) + ...
    np.testing.assert_allclose(LO.lstm_final_h(x, np.array([2, 1]), K0, b)[0], np.tanh(c2) / (1 + np.exp(0.3)), rtol=1e-13)
    np.testing.assert_allclose(LO.lstm_final_h(x, np.array([2, 1]), K0, b)[1], h1, rtol=1e-13)     # carried through past len 1
    # the scoring layer is a broadcast sum inside tanh
    loss, report, out, mid = LO.forward(p, batch, table, nbox, answers)
    al = mid["answer_ft"] @ p["reasoning/answer_layer1/fc/weights"]
    pl = mid["pooled_map_L"] @ p["reasoning/pooled_layer1/fc/weights"]
    ql = mid["q_L_ft"] @ p["reasoning/q_layer1/fc/weights"] + p["reasoning/q_layer1/fc/biases"]
    z23 = np.tanh(al[3] + pl[2] + ql[2]) @ p["reasoning/classifier/fc/weights"][:, 0] + p["reasoning/classifier/fc/biases"][0]
    assert abs(mid["logit"][2, 3] - z23) < 1e-12
    assert report["answer_loss"] == O.sigmoid_ce(mid["logit"], batch["answer_target"]).sum(1).mean()       # no train mask


def test_variable_and_filter_contract():
    p, *_ = _case(4)
    names = sorted(k for k in p if not O.is_const(k))
    assert LO.train_var_names(p, True) == names
    assert sorted({n.split("/")[0] for n in LO.train_var_names(p, False)}) == ["GloVe", "encode_L", "reasoning"]
    assert sorted({n.split("/")[0] for n in LO.transfer_var_names(p)}) == ["GloVe", "L2V", "V2L", "encode_L"]
    sh = LO.variable_shapes(50)
    assert sh["encode_L/rnn/basic_lstm_cell/kernel"] == (812, 2048) and sh["GloVe/learn"] == (3, 300)
    assert "reasoning/answer_layer1/fc/biases" not in sh and sh["reasoning/classifier/fc/weights"] == (512, 1)
