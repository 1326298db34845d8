"""GPU parity of the registry's oldest model (vqa/model_vqa.py; `model_type` 13 of vqa_fusion_forward / _backward; LSTM
kernels of csrc/lstm_ops.hip) against the float64 oracle (oracle/legacy_vqa_oracle.py), through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import legacy_vqa_oracle as LO
from oracle import vqa_oracle as O
from tests.gpu_util import dev, dev_batch, to64

pytestmark = pytest.mark.gpu

SMALL = dict(Vq=30, W=12, D=16, L=16, M=20, A=11)
MED = dict(Vq=500, W=300, D=128, L=96, M=64, A=300)
FULL = dict(Vq=2000, W=300, D=512, L=512, M=512, A=3000)            # the reference's dimensions (vqa/model_vqa.py:10-13)
MID = ["q_L_ft", "q_map_V", "att_score", "pooled_V_ft", "pooled_map_L", "answer_ft", "logit"]


def make_case(seed, B, R, T, N, dims, La=4):
    rng = np.random.default_rng(seed)
    p = LO.init_params(rng, **dims)
    for k in p:
        if k.endswith("/biases") or k.endswith("/bias"):
            p[k] = (p[k] + 0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    table, nbox = O.make_table(rng, N, R, dims["D"], full_boxes=False)
    batch = O.make_batch(rng, B, T, dims["Vq"], dims["A"], N, min_len=1)
    batch["q_intseq"][0, 0] = dims["Vq"] - 1
    batch["q_intseq"][B - 1, 0] = dims["Vq"] - 3
    answers = LO.make_answers(rng, dims["A"], dims["Vq"], La)
    answers["intseq"][2, 0] = dims["Vq"] - 2
    return p, table, nbox, batch, answers


def make_engine(p, table, nbox, answers, B, R, T, dims, ft_vlmap, **kw):
    from vqa_transfer_externaldata_amd import fusion as F
    A = dims["A"]
    eng = F.FusionEngine(model_type="vqa", B=B, R=R, T=T, N_img=table.shape[0], Vq=dims["Vq"], W=dims["W"], D=dims["D"],
                         H=dims["L"], A=A, map_dim=dims["M"], ft_vlmap=ft_vlmap, glove_fixed=p[LO.FIXED], answers=answers,
                         params={k: v for k, v in p.items() if not O.is_const(k)}, **kw)
    ones, zeros = np.ones(A, np.float32), np.zeros(A, np.float32)
    eng.bind_inputs(table=dev(table), nbox_table=dev(nbox),
                    answer_masks={"train": dev(ones), "obj": dev(zeros), "attr": dev(zeros), "exist": dev(ones)})
    return eng


def run_engine(eng, batch, lr=None):
    eng.forward(dev_batch(batch), None, None, want_dz=True)
    eng.backward()
    if lr is not None:
        eng.optimizer_step(lr)
    torch.cuda.synchronize()


def grad_close(got, want, name, tol=5e-4):
    got = got.detach().cpu().numpy().astype(np.float64)
    sc = max(np.abs(want).max(), 1e-12)
    err = np.abs(got - want).max()
    assert err <= tol * sc + 1e-9, "%s: max err %.3e vs scale %.3e" % (name, err, sc)


@pytest.mark.parametrize("ft_vlmap", [False, True])
@pytest.mark.parametrize("cfg", [("small", SMALL, 5, 6, 7, 9), ("med", MED, 32, 36, 14, 64), ("full_dims", FULL, 8, 36, 14, 24)])
def test_forward_backward_match_oracle(cfg, ft_vlmap):
    name, dims, B, R, T, N = cfg
    p, table, nbox, batch, answers = make_case(91, B, R, T, N, dims)
    eng = make_engine(p, table, nbox, answers, B, R, T, dims, ft_vlmap)
    run_engine(eng, batch)
    p64, b64 = to64(p), to64(batch)
    loss, report, out, mid = LO.forward(p64, b64, table.astype(np.float64), nbox, answers)
    _, _, grads, sq = LO.torch_loss_and_grads(p64, b64, table.astype(np.float64), nbox, answers)
    for k in MID:
        got = eng.tensor(k).cpu().numpy().reshape(mid[k].shape)
        tol = 1e-3 if k == "logit" else 2e-4 * max(1.0, np.abs(mid[k]).max())
        assert np.abs(got - mid[k]).max() <= tol, (k, np.abs(got - mid[k]).max())
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])
    np.testing.assert_array_equal(eng.tensor("num_V_ft").cpu().numpy(), mid["num_V_ft"])
    rep = eng.report()
    assert abs(rep["answer_train_loss"] - report["answer_loss"]) <= 1e-4 * max(1.0, abs(report["answer_loss"]))
    assert abs(rep["answer_acc"] - report["answer_accuracy"]) <= 1e-6
    assert set(eng.train_names) == set(LO.train_var_names(p, ft_vlmap))
    for n in eng.train_names:
        grad_close(eng.grads[n], grads[n], n)
    got_sq = float(eng.grad_flat[eng.n_train])
    assert abs(got_sq - sq) <= 1e-3 * sq + 1e-14, (got_sq, sq)


@pytest.mark.parametrize("ft_vlmap", [False, True])
def test_train_steps_match_oracle_f32(ft_vlmap):
    dims, B, R, T, N = MED, 32, 36, 14, 64
    p, table, nbox, batch, answers = make_case(92, B, R, T, N, dims)
    eng = make_engine(p, table, nbox, answers, B, R, T, dims, ft_vlmap)
    frozen_before = {n: eng.params[n].clone() for n in eng.frozen_names}
    st = O.new_opt_state()
    for it in range(3):
        run_engine(eng, batch, lr=1e-3)
        loss, grads, norm = LO.train_step(p, batch, table, nbox, answers, st, 1e-3, ft_vlmap)
        assert abs(float(eng.norm_sq[0]) ** 0.5 - norm) <= 1e-3 * norm
        assert abs(float(eng.loss()) - loss) <= 2e-4 * max(1, abs(loss))
    for n in eng.train_names:
        got = eng.params[n].cpu().numpy()
        assert np.abs(got - p[n]).max() <= 4.5e-4 + 1e-4 * np.abs(p[n]).max(), n
        assert np.mean(np.abs(got - p[n]) > 1e-4) < 0.02, n
    for n in eng.frozen_names:
        assert torch.equal(eng.params[n], frozen_before[n])
    assert {n.split("/")[0] for n in eng.frozen_names} == (set() if ft_vlmap else {"L2V", "V2L"})


def test_lstm_step_kernels_known_answers():
    """zero kernel: the cell only sees its bias; forget_bias 1.0; rows past their length carry (c, h) through"""
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    N, L = 3, 8
    b = np.zeros(4 * L, np.float32); b[L:2 * L] = 0.7; b[3 * L:] = -0.3
    g = dev(np.tile(b, (N, 1)))
    c0, h0 = dev(np.full((N, L), 0.25, np.float32)), dev(np.full((N, L), -0.5, np.float32))
    c1, h1 = torch.empty_like(c0), torch.empty_like(h0)
    lens = dev(np.array([2, 1, 0], np.int32))
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.vqa_lstm_step_fwd(P(g), P(c0), P(h0), P(lens), 1, P(c1), P(h1), N, L, None), "lstm fwd")
    torch.cuda.synchronize()
    cn = 0.25 / (1 + np.exp(-1.0)) + 0.5 * np.tanh(0.7)
    np.testing.assert_allclose(c1[0].cpu().numpy(), cn, rtol=1e-6)
    np.testing.assert_allclose(h1[0].cpu().numpy(), np.tanh(cn) / (1 + np.exp(0.3)), rtol=1e-6)
    assert torch.equal(c1[1:], c0[1:]) and torch.equal(h1[1:], h0[1:])                       # t = 1 >= len: carried
    np.testing.assert_allclose(g[0, 2 * L:3 * L].cpu().numpy(), 1 / (1 + np.exp(-1.0)), rtol=1e-6)   # activated f' left in place


def test_model_class_trains_through_the_trainer(tmp_path):
    """`python vqa/trainer.py` with its default --model_type vqa: answers from data_info.hdf5 (in memory here), 512-d
    region features, two report scalars, V2L / L2V frozen unless --ft_vlmap, checkpoints, loss goes down"""
    import os
    from tests.test_gpu_trainer import _config, _datasets
    from vqa_transfer_externaldata_amd import importer, trainer
    assert importer.get_model_class("vqa").MODEL_TYPE == "vqa"
    rng = np.random.default_rng(8)
    c, Vq, A = _config(tmp_path, "vqa", learning_rate=1e-3)
    c.ft_vlmap = False
    c.answer_intseq = rng.integers(0, Vq, size=(A, 3)).astype(np.int32)
    c.answer_intseq_len = rng.integers(1, 4, size=A).astype(np.int32)
    feats = {"features": np.maximum(rng.standard_normal((24, 36, 512)), 0).astype(np.float32),
             "spatials": np.zeros((24, 36, 6), np.float32), "normal_boxes": np.zeros((24, 36, 4), np.float32),
             "num_boxes": np.full(24, 36, np.int32), "max_box_num": 36, "vfeat_dim": 512}
    t = trainer.Trainer(c, datasets=_datasets(Vq, A), image_features=feats)
    m, eng = t.model, t.model.engine
    assert set(m.report) == {"answer_loss", "answer_accuracy"} and set(m.mid_result) >= {"num_V_ft", "att_score", "pred"}
    assert {v.split("/")[0] for v in t.train_vars} == {"GloVe", "encode_L", "reasoning"}
    assert sorted({v.split("/")[0] for v in t.transfer_vars}) == ["GloVe", "L2V", "V2L", "encode_L"]
    assert tuple(eng.params["encode_L/rnn/basic_lstm_cell/kernel"].shape) == (812, 2048)
    frozen = {k: v.clone() for k, v in eng.params.items() if k.split("/")[0] in ("L2V", "V2L")}
    step, summary, loss0, report, dt = t.run_train_step(True)
    assert step == 1 and set(report) == {"answer_loss", "answer_accuracy"} and abs(loss0 - report["answer_loss"]) <= 1e-6 * max(1, loss0)
    losses = [t.run_train_step(False)[2] for _ in range(11)]
    assert np.mean(losses[-3:]) < loss0, (loss0, losses)
    t.train()
    assert any(f.startswith("model-") for f in os.listdir(c.train_dir))                      # checkpoints by global step
    for k, v in frozen.items():
        assert torch.equal(eng.params[k], v), k
    # wrong feature width: the dot-product attention cannot work
    bad = dict(feats, features=feats["features"][:, :, :64].copy(), vfeat_dim=64)
    c2, _, _ = _config(tmp_path, "vqa")
    c2.answer_intseq, c2.answer_intseq_len, c2.train_dir = c.answer_intseq, c.answer_intseq_len, str(tmp_path / "run2")
    with pytest.raises(ValueError, match="512-d region features"):
        trainer.Trainer(c2, datasets=_datasets(Vq, A), image_features=bad)
