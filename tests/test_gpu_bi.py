"""GPU parity of the bi-directional-GRU models (vqa/model_vlmap_finetune.py, vqa/model_vlmap_only.py; `model_type` 12 of
vqa_fusion_forward / _backward) against the float64 oracle (oracle/bi_oracle.py), through the C ABI.  Bars as in
tests/test_gpu_fusion.py: logits 1e-3 abs, pred bit-exact, activations 2e-4, gradients 5e-4 of max|g|."""
import numpy as np
import pytest
import torch

from oracle import bi_oracle as BO
from oracle import vqa_oracle as O
from tests.gpu_util import dev, dev_batch, to64

pytestmark = pytest.mark.gpu

SMALL = dict(Vq=30, W=12, D=24, H=16, A=21)
MED = dict(Vq=500, W=300, D=256, H=128, A=300)
FULL = dict(Vq=2000, W=300, D=2048, H=1024, A=3000)
MID = ["v_linear_v", "q_L_map", "q_L_ft", "q_att_key", "q_att_query", "w_att_score", "q_v_ft", "pooled_q_v", "q_linear_v",
       "att_score", "pooled_V_ft", "pooled_linear_l", "l_linear_l", "joint", "logit"]


def make_case(seed, B, R, T, N, dims, min_len=1, num_train=None):
    rng = np.random.default_rng(seed)
    p = O.perturb_ln_params(BO.init_params(rng, **dims), rng)
    table, nbox = O.make_table(rng, N, R, dims["D"], full_boxes=False)
    batch = O.make_batch(rng, B, T, dims["Vq"], dims["A"], N, min_len=min_len)
    am = O.make_answer_masks(rng, dims["A"], num_train or int(dims["A"] * 0.75), exist_all=False)
    masks = BO.make_masks(rng, B, R, T, dims["H"])
    return p, table, nbox, batch, am, masks


def make_engine(model_type, p, table, nbox, am, B, R, T, dims, **kw):
    from vqa_transfer_externaldata_amd import fusion as F
    eng = F.FusionEngine(model_type=model_type, B=B, R=R, T=T, N_img=table.shape[0],
                         params={k: v.astype(np.float32) for k, v in p.items()}, **dims, **kw)
    eng.bind_inputs(table=dev(table), nbox_table=dev(nbox), answer_masks={k: dev(v) for k, v in am.items()})
    return eng


def run_engine(eng, batch, masks, lr=None):
    u8 = lambda k: dev(masks[k].astype(np.uint8))
    eng.forward(dev_batch(batch), u8("att"), u8("joint"), want_dz=True, keep_word=u8("word"))
    eng.backward()
    if lr is not None:
        eng.optimizer_step(lr)
    torch.cuda.synchronize()


def grad_close(got, want, name, tol=5e-4):
    got = got.detach().cpu().numpy().astype(np.float64)
    sc = max(np.abs(want).max(), 1e-12)
    err = np.abs(got - want).max()
    assert err <= tol * sc + 1e-9, "%s: max err %.3e vs scale %.3e" % (name, err, sc)


@pytest.mark.parametrize("model_type", BO.MODEL_TYPES)
@pytest.mark.parametrize("cfg", [("small", SMALL, 5, 6, 7, 9), ("med", MED, 32, 36, 14, 64), ("cfg1_full_dims", FULL, 8, 36, 14, 24)])
def test_forward_backward_match_oracle(model_type, cfg):
    name, dims, B, R, T, N = cfg
    p, table, nbox, batch, am, masks = make_case(81, B, R, T, N, dims)
    eng = make_engine(model_type, p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    p64, b64, am64, m64 = to64(p), to64(batch), to64(am), to64(masks)
    loss, report, out, mid = BO.forward(p64, b64, table.astype(np.float64), nbox, am64, m64)
    tloss, tmid, grads, slices = BO.torch_loss_and_grads(p64, b64, table.astype(np.float64), nbox, am64, m64)
    for k in MID:
        got = eng.tensor(k).cpu().numpy().reshape(mid[k].shape)
        tol = 1e-3 if k == "logit" else 2e-4 * max(1.0, np.abs(mid[k]).max())
        assert np.abs(got - mid[k]).max() <= tol, (k, np.abs(got - mid[k]).max())
    np.testing.assert_array_equal(eng.tensor("condition").cpu().numpy().reshape(B, -1), eng.tensor("q_L_ft").cpu().numpy().reshape(B, -1))
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])
    np.testing.assert_array_equal(eng.tensor("num_V_ft").cpu().numpy(), mid["num_V_ft"])
    rep = eng.report()
    for new, old in (("answer_train_loss", "answer_train_loss"), ("answer_report_loss", "answer_report_loss"), ("answer_accuracy", "answer_acc")):
        assert abs(rep[old] - report[new]) <= 1e-4 * max(1.0, abs(report[new])), (new, rep[old], report[new])
    assert set(eng.train_names) == set(BO.train_var_names(p, model_type))
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):             # softmax shift invariance: analytically zero (both attentions)
            assert abs(float(eng.grads[n][0])) <= 1e-5
            continue
        grad_close(eng.grads[n], grads[n], n)
    # the un-aggregated slices of the two tables and their joint norm in the tail slot
    W = dims["W"]
    grad_close(eng.tensor("dx_embed").view(T, B, W).transpose(0, 1), slices["LearnGloVe/embed_map"], "dx_embed")
    want_sq = (slices["LearnGloVe/embed_map"] ** 2).sum()
    if model_type == "vlmap_finetune":
        grad_close(eng.tensor("d_e2").view(B, T, W), slices["V_WordMap/embed_map"], "d_e2")
        want_sq += (slices["V_WordMap/embed_map"] ** 2).sum()
    sq = float(eng.grad_flat[eng.n_train])
    assert abs(sq - want_sq) <= 1e-3 * want_sq + 1e-12


@pytest.mark.parametrize("model_type", BO.MODEL_TYPES)
def test_train_steps_match_oracle_f32(model_type):
    dims, B, R, T, N = MED, 32, 36, 14, 64
    p, table, nbox, batch, am, masks = make_case(82, B, R, T, N, dims)
    eng = make_engine(model_type, p, table, nbox, am, B, R, T, dims)
    frozen_before = {n: eng.params[n].clone() for n in eng.frozen_names}
    st = O.new_opt_state()
    for it in range(3):
        run_engine(eng, batch, masks, lr=1e-3)
        loss, grads, norm = BO.train_step(p, batch, table, nbox, am, masks, st, 1e-3, model_type)
        assert abs(float(eng.norm_sq[0]) ** 0.5 - norm) <= 1e-3 * norm
        assert abs(float(eng.loss()) - loss) <= 2e-4 * max(1, abs(loss))
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            continue
        got = eng.params[n].cpu().numpy()
        assert np.abs(got - p[n]).max() <= 4.5e-4 + 1e-4 * np.abs(p[n]).max(), n
        assert np.mean(np.abs(got - p[n]) > 1e-4) < 0.02, n
    for n in eng.frozen_names:
        assert torch.equal(eng.params[n], frozen_before[n])
    if model_type == "vlmap_only":
        assert {n.split("/")[0] for n in eng.frozen_names} == set(BO.FROZEN_TOP_SCOPES_ONLY)
    else:
        assert eng.frozen_names == []


def test_lengths_one_full_and_repeated_tokens():
    """one-token questions, full-length questions, a batch whose padded width exceeds every length but one, repeated
    tokens (scatter-add of both tables), one-box images"""
    dims, B, R, T, N = MED, 12, 36, 10, 24
    p, table, nbox, batch, am, masks = make_case(83, B, R, T, N, dims)
    batch["q_intseq_len"][:] = [1, 10, 1, 2, 3, 1, 10, 4, 2, 1, 5, 1]
    batch["q_intseq"][:, :] = batch["q_intseq"][:, :] % 7                        # heavy token repetition
    batch["q_intseq"][np.arange(T)[None, :] >= batch["q_intseq_len"][:, None]] = 0
    nbox[batch["image_idx"][0]] = 1
    eng = make_engine("vlmap_finetune", p, table, nbox, am, B, R, T, dims, deterministic=True)
    run_engine(eng, batch, masks)
    p64, b64, am64, m64 = to64(p), to64(batch), to64(am), to64(masks)
    loss, report, out, mid = BO.forward(p64, b64, table.astype(np.float64), nbox, am64, m64)
    _, _, grads, slices = BO.torch_loss_and_grads(p64, b64, table.astype(np.float64), nbox, am64, m64)
    for k in ("q_L_map", "q_L_ft", "w_att_score", "pooled_q_v", "att_score", "logit"):
        got = eng.tensor(k).cpu().numpy().reshape(mid[k].shape)
        assert np.abs(got - mid[k]).max() <= (1e-3 if k == "logit" else 2e-4 * max(1.0, np.abs(mid[k]).max())), k
    qm = eng.tensor("q_L_map").view(B, T, -1).cpu().numpy()
    for b, n in enumerate(batch["q_intseq_len"]):
        assert np.all(qm[b, n:] == 0)
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])
    for n in eng.train_names:
        if not n.endswith("score/fc/biases"):
            grad_close(eng.grads[n], grads[n], n)
    g1 = eng.grad_flat.clone()
    run_engine(eng, batch, masks)                                               # deterministic mode: bitwise repeatable
    assert torch.equal(g1, eng.grad_flat)


def test_phased_backward_equals_monolithic():
    dims, B, R, T, N = MED, 16, 36, 14, 32
    p, table, nbox, batch, am, masks = make_case(84, B, R, T, N, dims)
    a = make_engine("vlmap_finetune", p, table, nbox, am, B, R, T, dims, deterministic=True)
    b = make_engine("vlmap_finetune", p, table, nbox, am, B, R, T, dims, deterministic=True)
    run_engine(a, batch, masks)

    class Rec:
        def __init__(self): self.seen = []
        def start(self, bucket): self.seen.append((bucket.data_ptr(), bucket.numel()))
        def finish(self): pass
    u8 = lambda k: dev(masks[k].astype(np.uint8))
    b.forward(dev_batch(batch), u8("att"), u8("joint"), want_dz=True, keep_word=u8("word"))
    r = Rec()
    b.backward(reducer=r)
    torch.cuda.synchronize()
    assert torch.equal(a.grad_flat, b.grad_flat)
    assert sum(n for _, n in r.seen) == b.n_train + 4 and len(r.seen) == 5


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model_type", BO.MODEL_TYPES)
def test_models_train_through_the_trainer(tmp_path, model_type):
    """`python vqa/trainer.py --model_type vlmap_finetune|vlmap_only --pretrained_param_path ... --vlmap_word_weight_dir ...`:
    mandatory arguments, V_WordMap and the head initialised from the word-weight directory, the transfer set restored
    from the pre-trained checkpoint, train sets of the two files, three report scalars, checkpoints"""
    import os
    import pickle
    from tests.test_gpu_trainer import _config, _datasets, _features
    from vqa_transfer_externaldata_amd import hdf5_io, importer, trainer
    Model = importer.get_model_class(model_type)
    c, Vq, A = _config(tmp_path, model_type, learning_rate=5e-4)
    with pytest.raises(ValueError, match="pretrained_param_path is mendatory"):
        Model({}, c, is_train=True)
    rng = np.random.default_rng(6)
    wdir = tmp_path / "word_weights"
    os.makedirs(str(wdir))
    src = ["a%d" % i for i in range(A) if i % 3]
    words = ["w%d" % i for i in range(0, Vq, 2)] + ["unseen"]                       # every other question word is known
    ww = {"class_weights": (0.05 * rng.standard_normal((2048, len(src)))).astype(np.float32),
          "class_biases": (0.1 * rng.standard_normal(len(src))).astype(np.float32),
          "v_word": rng.standard_normal((len(words), 300)).astype(np.float32)}
    hdf5_io.write(str(wdir / "weights.hdf5"), ww)
    with open(str(wdir / "answer_dict.pkl"), "wb") as f:
        pickle.dump({"vocab": src, "dict": {a: i for i, a in enumerate(src)}}, f)
    with open(str(wdir / "vocab.pkl"), "wb") as f:
        pickle.dump({"vocab": words, "dict": {w: i for i, w in enumerate(words)}}, f)
    c.pretrained_param_path = "x"
    with pytest.raises(ValueError, match="word_weight_dir is mendatory"):
        Model({}, c, is_train=True)
    c.vlmap_word_weight_dir = str(wdir)
    # a "pre-trained" checkpoint holding the transfer set (vlmap_memft's variables of the same names)
    from vqa_transfer_externaldata_amd import fusion as F
    shapes = F.variable_shapes(model_type, Vq, 300, 64, 1024, A)
    pre = {n: torch.from_numpy(rng.standard_normal(shapes[n]).astype(np.float32) * 0.05)
           for n in F.filter_transfer_vars(sorted(shapes), model_type)}
    c.pretrained_param_path = str(tmp_path / "pretrained.pt")
    torch.save(pre, c.pretrained_param_path)
    t = trainer.Trainer(c, datasets=_datasets(Vq, A), image_features=_features())
    m, eng = t.model, t.model.engine
    assert m.MODEL_TYPE == model_type and set(m.report) == {"answer_train_loss", "answer_report_loss", "answer_accuracy"}
    P = eng.params
    np.testing.assert_array_equal(P["V_WordMap/embed_map"][4].cpu().numpy(), ww["v_word"][2])       # w4 is row 2 of the directory
    assert not P["V_WordMap/embed_map"][5].any()                                                     # w5 is unknown: zero row
    np.testing.assert_array_equal(P["WordWeightAnswer/fc/weights"][:, 1].cpu().numpy(), ww["class_weights"][:, 0])
    for n, v in pre.items():
        assert torch.equal(P[n].cpu(), v), n
    assert sorted({v.split("/")[0] for v in t.transfer_vars}) == sorted(F.TRANSFER_TOP_SCOPES_BI)
    tops = {v.split("/")[0] for v in t.train_vars}
    if model_type == "vlmap_only":
        assert tops == {"LearnGloVe", "encode_L_bi", "q_att_key", "q_att_query", "word_attention"}
    else:
        assert len(t.train_vars) == len(eng.shapes)
    assert m.mid_result["q_L_map"].shape[0] == 32 and m.mid_result["q_L_map"].shape[2] == 1024
    before = {k: v.clone() for k, v in P.items()}
    step, summary, loss0, report, dt = t.run_train_step(True)
    assert step == 1 and set(report) == set(m.report) and abs(loss0 - report["answer_train_loss"]) <= 1e-6 * max(1, abs(loss0))
    t.train()
    assert t.global_step == 13 and os.path.exists(os.path.join(c.train_dir, "model-8"))
    for k, v in before.items():
        moved = not torch.equal(P[k], v)
        if k.endswith("score/fc/biases"):
            continue
        assert moved == (k in eng.train_names), k
    _, _, loss1, vreport, _ = t.run_val_step(False, "val")
    assert set(vreport) == set(m.report) and np.isfinite(loss1)
