"""Oracle pins of the bi-directional-GRU models (vqa/model_vlmap_finetune.py, vqa/model_vlmap_only.py; oracle/bi_oracle.py):
NumPy forward against the independently composed torch forward, finite differences of the autograd gradients, known
answers of tf.nn.bidirectional_dynamic_rnn's sequence reversal, the variable / filter contract."""
import numpy as np
import pytest

from oracle import bi_oracle as BO
from oracle import vqa_oracle as O

DIMS = dict(Vq=30, W=12, D=24, H=16, A=21)


def _case(seed, B=5, R=6, T=7, N=9, dtype=np.float64, min_len=1):
    rng = np.random.default_rng(seed)
    p = O.perturb_ln_params(BO.init_params(rng, dtype=dtype, **DIMS), rng)
    table, nbox = O.make_table(rng, N, R, DIMS["D"], dtype, full_boxes=False)
    batch = O.make_batch(rng, B, T, DIMS["Vq"], DIMS["A"], N, dtype, min_len=min_len)
    am = O.make_answer_masks(rng, DIMS["A"], 15, dtype, exist_all=False)
    masks = BO.make_masks(rng, B, R, T, DIMS["H"], dtype)
    return p, table, nbox, batch, am, masks


def test_numpy_forward_matches_torch_forward_and_gradients_match_finite_differences():
    p, table, nbox, batch, am, masks = _case(3)
    loss, report, out, mid = BO.forward(p, batch, table, nbox, am, masks)
    tloss, tmid, grads, slices = BO.torch_loss_and_grads(p, batch, table, nbox, am, masks)
    assert abs(loss - tloss) <= 1e-10 * max(1, abs(tloss))
    for k, v in tmid.items():
        np.testing.assert_allclose(mid[k], v, rtol=1e-9, atol=1e-11, err_msg=k)
    assert sorted(report) == sorted(BO.REPORT_KEYS) and sorted(grads) == sorted(p)
    fwd = lambda pp: BO.forward(pp, batch, table, nbox, am, masks)[0]
    rng = np.random.default_rng(0)
    for name in sorted(p):
        idx = tuple(int(rng.integers(0, s)) for s in p[name].shape)
        if name.endswith("embed_map"):
            idx = (int(batch["q_intseq"][0, 0]), idx[1])                       # a row the batch actually uses
        pp = {k: v.copy() for k, v in p.items()}
        eps = 1e-6
        pp[name][idx] += eps
        lp = fwd(pp)
        pp[name][idx] -= 2 * eps
        fd = (lp - fwd(pp)) / (2 * eps)
        assert abs(fd - grads[name][idx]) <= 2e-7 + 2e-5 * abs(grads[name][idx]), (name, idx, fd, grads[name][idx])
    # the dense embedding gradients are the scatter-add of their slices
    for name, sl in slices.items():
        dense = np.zeros_like(p[name])
        np.add.at(dense, batch["q_intseq"].reshape(-1), sl.reshape(-1, sl.shape[-1]))
        np.testing.assert_allclose(dense, grads[name], rtol=1e-10, atol=1e-14)


def test_bidirectional_rnn_known_answers():
    p, table, nbox, batch, am, masks = _case(4, B=4, T=6)
    lens = np.array([6, 1, 3, 4], np.int32)
    batch["q_intseq_len"] = lens
    x = p["LearnGloVe/embed_map"][batch["q_intseq"]]
    # reverse_sequence: the first len steps reversed, the tail untouched
    r = BO.reverse_sequence(x, lens)
    np.testing.assert_array_equal(r[2, :3], x[2, :3][::-1]); np.testing.assert_array_equal(r[2, 3:], x[2, 3:])
    np.testing.assert_array_equal(BO.reverse_sequence(r, lens), x)
    q_map, q_ft = BO.encode_L_bidirection(x, lens, p)
    h1 = DIMS["H"] // 2
    assert q_map.shape == (4, 6, DIMS["H"]) and q_ft.shape == (4, DIMS["H"])
    for b, n in enumerate(lens):
        assert np.all(q_map[b, n:] == 0)                                        # outputs past the length are zero
        np.testing.assert_array_equal(q_ft[b, :h1], q_map[b, n - 1, :h1])       # fw final state = fw output at the last token
        np.testing.assert_array_equal(q_ft[b, h1:], q_map[b, 0, h1:])           # bw final state = bw output at the FIRST token
    # the bw half is a plain GRU over the reversed tokens
    bw = [p[BO.GRU % ("bw", k) + s] for k, s in (("gates", "/kernel"), ("gates", "/bias"), ("candidate", "/kernel"), ("candidate", "/bias"))]
    for b, n in enumerate(lens):
        _, s = BO.gru_outputs(x[b:b + 1, :n][:, ::-1], np.array([n]), *bw)
        np.testing.assert_allclose(s[0], q_ft[b, h1:], rtol=1e-13)
    # a one-token question: both directions see only x[0] from the zero state
    of, _ = BO.gru_outputs(x[1:2, :1], np.array([1]), *[p[BO.GRU % ("fw", k) + s] for k, s in (("gates", "/kernel"), ("gates", "/bias"), ("candidate", "/kernel"), ("candidate", "/bias"))])
    np.testing.assert_allclose(q_map[1, 0, :h1], of[0, 0], rtol=1e-13)


def test_variable_and_filter_contract():
    p, *_ = _case(5)
    names = sorted(p)
    assert BO.train_var_names(p, "vlmap_finetune") == names                               # :64-68
    only = BO.train_var_names(p, "vlmap_only")
    assert sorted({n.split("/")[0] for n in only}) == ["LearnGloVe", "encode_L_bi", "q_att_key", "q_att_query", "word_attention"]
    for mt in BO.MODEL_TYPES:
        assert sorted({n.split("/")[0] for n in BO.transfer_var_names(p, mt)}) == sorted(BO.TRANSFER_TOP_SCOPES)
    sh = BO.variable_shapes(50, 300, 2048, 1024, 3000)
    assert sh["encode_L_bi/bidirectional_rnn/fw/gru_cell/gates/kernel"] == (812, 1024)
    assert sh["encode_L_bi/bidirectional_rnn/bw/gru_cell/candidate/kernel"] == (812, 512)
    assert sh["v_word_fc/fc/weights"] == (300, 1024) and sh["word_attention/compute/score/fc/weights"] == (1024, 1)
    assert sh["V_WordMap/embed_map"] == (50, 300)


def test_layer_norm_of_sequence_layers_spans_the_padded_block_and_word_attention_masks_padding():
    p, table, nbox, batch, am, masks = _case(6, B=3, T=5)
    batch["q_intseq_len"] = np.array([5, 2, 3], np.int32)
    batch["q_intseq"][1, 2:] = 0
    loss, report, out, mid = BO.forward(p, batch, table, nbox, am, masks)
    assert np.all(mid["w_att_score"][1, 2:] == 0) and abs(mid["w_att_score"][1].sum() - 1) < 1e-12
    pre = mid["q_L_map"] @ p["q_att_key/fc/weights"] + p["q_att_key/fc/biases"]
    mu, var = pre.mean((1, 2), keepdims=True), pre.var((1, 2), keepdims=True)               # the padded rows count
    want = np.maximum((pre - mu) / np.sqrt(var + O.LN_EPS) * p["q_att_key/LayerNorm/gamma"] + p["q_att_key/LayerNorm/beta"], 0)
    np.testing.assert_allclose(mid["q_att_key"], want, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(mid["pooled_q_v"], np.einsum("bt,bth->bh", mid["w_att_score"], mid["q_v_ft"]), rtol=1e-12)
    assert report["answer_accuracy"] == batch["answer_target"][np.arange(3), out["pred"]].mean()
