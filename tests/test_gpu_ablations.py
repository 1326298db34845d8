"""GPU parity of the five older model_vlmap_answer ablations (vlmap_answer2, _no_noise, _adapt, _full,
_ent; `model_type` 7..11 of vqa_fusion_forward / _backward) against the float64 oracle, through the C ABI.
Same bars as tests/test_gpu_fusion.py: logits 1e-3 abs, pred bit-exact, activations 2e-4, gradients 5e-4 of max|g|."""
import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O
from tests.gpu_util import dev, dev_batch, make_case, make_engine, to64, variant_inputs

pytestmark = pytest.mark.gpu

SMALL = dict(Vq=30, W=12, D=24, H=16, A=21)
MED = dict(Vq=500, W=300, D=256, H=128, A=300)
FULL = dict(Vq=2000, W=300, D=2048, H=1024, A=3000)
TYPES = list(O.ABLATION_FAMILY)
MID_KEYS = ["v_linear_v", "condition", "q_linear_v", "att_score", "pooled_V_ft", "pooled_linear_l", "l_linear_l", "joint", "logit"]
EXTRA_MID = {"vlmap_answer_adapt": ["v_adapt"], "vlmap_answer_no_noise": ["q_L_mean"],
             "vlmap_answer_full": ["q_L_mean", "q_L_log_sigma_sq", "q_L_mean_noise"], "vlmap_answer_ent": ["marginal_prob"],
             "vlmap_answer2": []}


def run_engine(eng, batch, masks, lr=None, want_dz=True):
    ka, kj = dev(masks["att"].astype(np.uint8)), dev(masks["joint"].astype(np.uint8))
    eng.forward(dev_batch(batch), ka, kj, want_dz=want_dz, **variant_inputs(masks))
    if want_dz:
        eng.backward()
    if lr is not None:
        eng.optimizer_step(lr)
    torch.cuda.synchronize()


def grad_close(got, want, name, tol=5e-4):
    got = got.detach().cpu().numpy().astype(np.float64)
    sc = max(np.abs(want).max(), 1e-12)
    err = np.abs(got - want).max()
    assert err <= tol * sc + 1e-9, "%s: max err %.3e vs scale %.3e" % (name, err, sc)


def _engine_kw(model_type, M):
    return {"num_marginal": M} if model_type == "vlmap_answer_ent" else {}


def _check_against_oracle(eng, model_type, p, table, nbox, batch, am, masks, B, R, T, dims):
    p64 = to64(p)
    loss, report, out, mid, tape = O.forward(p64, to64(batch), table.astype(np.float64), nbox, to64(am), to64(masks), model_type)
    grads, dx = O.backward(p64, to64(batch), to64(am), to64(masks), tape, model_type)
    for k in MID_KEYS + EXTRA_MID[model_type]:
        want = mid[k]
        got = eng.tensor(k).cpu().numpy()
        if k == "marginal_prob":          # [B, ent_cols] on the device with zeros at the excluded answers
            sel = (am["exist"] * am["train"]) > 0.5
            got = got.reshape(B, -1)
            assert got.shape[1] >= int(np.nonzero(sel)[0].max()) + 1
            dense = np.zeros((B, dims["A"]))
            dense[:, :got.shape[1]] = got
            assert np.all(dense[:, ~sel] == 0)
            got = dense[:, sel]
            assert np.abs(got - want).max() <= 1e-5, (k, np.abs(got - want).max())
            continue
        got = got[:want.size].reshape(want.shape)
        tol = 1e-3 if k == "logit" else 2e-4 * max(1.0, np.abs(want).max())
        assert np.abs(got - want).max() <= tol, (k, np.abs(got - want).max())
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])
    rep = eng.report()
    if model_type in O.OLD_REPORT_TYPES:
        for new, old in O.TESTMASK_REPORT.items():
            assert abs(rep[old] - report[new]) <= 1e-4 * max(1.0, abs(report[new])), (new, rep[old], report[new])
    else:
        for k in O.REPORT_KEYS:
            assert abs(rep[k] - report[k]) <= 1e-4 * max(1.0, abs(report[k])), (k, rep[k], report[k])
    ex = eng.extra_report()
    if model_type == "vlmap_answer_full":
        assert abs(ex["latent_loss"] - report["latent_loss"]) <= 1e-4 * max(1.0, abs(report["latent_loss"]))
        assert abs(ex["train_latent_loss"] - report["train_latent_loss"]) <= 1e-4 * max(1.0, abs(report["train_latent_loss"]))
    elif model_type == "vlmap_answer_ent":
        assert abs(ex["entropy"] - report["entropy"]) <= 1e-4 * max(1.0, abs(report["entropy"]))
        assert abs(ex["weighted_entropy"] - report["weighted_entropy"]) <= 1e-4 * max(1.0, abs(report["weighted_entropy"]))
    else:
        assert ex == {}
    assert abs(float(eng.loss()) - loss) <= 1e-4 * max(1.0, abs(loss))              # self.loss = sum of self.losses
    assert set(eng.train_names) == set(O.train_var_names(p, model_type))
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            assert abs(float(eng.grads[n][0])) <= 1e-5
            continue
        grad_close(eng.grads[n], grads[n], n)
    grad_close(eng.tensor("dx_embed").view(T, B, dims["W"]).transpose(0, 1), dx, "dx_embed")
    return loss, report, mid


@pytest.mark.parametrize("model_type", TYPES)
@pytest.mark.parametrize("cfg", [("small", SMALL, 5, 6, 7, 9, 7), ("med", MED, 32, 36, 14, 64, 200),
                                 ("cfg1_full_dims", FULL, 8, 36, 14, 24, 200)])
def test_forward_backward_match_oracle(model_type, cfg):
    name, dims, B, R, T, N, M = cfg
    p, table, nbox, batch, am, masks = make_case(61, model_type, B, R, T, N, dims, num_marginal=M)
    eng = make_engine(model_type, p, table, nbox, am, B, R, T, dims, **_engine_kw(model_type, M))
    run_engine(eng, batch, masks)
    _check_against_oracle(eng, model_type, p, table, nbox, batch, am, masks, B, R, T, dims)


@pytest.mark.parametrize("model_type", TYPES)
def test_train_steps_match_oracle_f32(model_type):
    dims, B, R, T, N, M = MED, 32, 36, 14, 64, 16
    p, table, nbox, batch, am, masks = make_case(62, model_type, B, R, T, N, dims, num_marginal=M)
    eng = make_engine(model_type, p, table, nbox, am, B, R, T, dims, **_engine_kw(model_type, M))
    frozen_before = {n: eng.params[n].clone() for n in eng.frozen_names}
    st = O.new_opt_state()
    for it in range(3):
        run_engine(eng, batch, masks, lr=1e-3)
        loss, report, out, mid, grads, norm = O.train_step(p, batch, table, nbox, am, masks, st, 1e-3, model_type=model_type)
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
    assert {n.split("/")[0] for n in eng.frozen_names} == {"q_linear_l", "pooled_linear_l", "joint_fc", "WordWeightAnswer"}


def test_ent_edge_cases_and_eval_mode():
    """B not a divisor of M, an answer set that excludes leading columns, no dropout on the pairings, want_dz = 0 (the
    pairings' probabilities stay in the workspace), and the untrained head's closed form"""
    mt, dims, B, R, T, N, M = "vlmap_answer_ent", MED, 7, 36, 14, 16, 10
    p, table, nbox, batch, am, masks = make_case(63, mt, B, R, T, N, dims, num_marginal=M)
    am["exist"][:5] = 0
    eng = make_engine(mt, p, table, nbox, am, B, R, T, dims, num_marginal=M)
    sel = (am["exist"] * am["train"]) > 0.5
    assert eng.dims.ent_cols == min(dims["A"], (int(np.nonzero(sel)[0].max()) + 1 + 31) // 32 * 32)
    run_engine(eng, batch, masks)
    _check_against_oracle(eng, mt, p, table, nbox, batch, am, masks, B, R, T, dims)
    # eval mode + no pairing dropout: probabilities of every pairing
    nodrop = dict(masks, tile_joint=np.full_like(masks["tile_joint"], O.KEEP_JOINT))      # mask / keep = 1: dropout off
    ka, kj = dev(masks["att"].astype(np.uint8)), dev(masks["joint"].astype(np.uint8))
    eng.forward(dev_batch(batch), ka, kj, want_dz=False)              # keep_tile = None
    torch.cuda.synchronize()
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am), to64(nodrop), mt)
    C = eng.dims.ent_cols
    prob = eng.tensor("tile_z").cpu().numpy().reshape(B, M, C)
    dense = np.zeros((B, M, dims["A"])); dense[:, :, :C] = prob
    np.testing.assert_allclose(dense[:, :, sel], tape["t_ent"]["prob"], atol=1e-5)
    np.testing.assert_allclose(dense[:, :, sel].sum(-1), 1.0, atol=1e-5)
    assert abs(eng.extra_report()["entropy"] - report["entropy"]) <= 1e-4 * abs(report["entropy"])
    # untrained head: uniform over the known training answers
    pu = dict(p, **{"WordWeightAnswer/fc/weights": np.zeros_like(p["WordWeightAnswer/fc/weights"]),
                    "WordWeightAnswer/fc/biases": np.full_like(p["WordWeightAnswer/fc/biases"], -100.0)})
    eng2 = make_engine(mt, pu, table, nbox, am, B, R, T, dims, num_marginal=M)
    run_engine(eng2, batch, masks)
    want = np.log(1.0 / sel.sum() + 1e-8)
    assert abs(eng2.extra_report()["entropy"] - want) <= 1e-5 * abs(want)
    z = eng2.tensor("logit").cpu().numpy()
    assert np.all(z == -100.0)


def test_full_noise_generator_and_zero_noise():
    """vqa_normal_noise: reproducible, keyed by (seed, offset), standard normal; zero noise reduces _full to _no_noise + KL"""
    mt, dims, B, R, T, N = "vlmap_answer_full", MED, 16, 36, 14, 32
    p, table, nbox, batch, am, masks = make_case(64, mt, B, R, T, N, dims)
    eng = make_engine(mt, p, table, nbox, am, B, R, T, dims)
    a = eng.make_noise(7, 3).clone()
    b = eng.make_noise(7, 3).clone()
    c = eng.make_noise(7, 4).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    shard = eng.make_noise(7, 3, row_offset=0, global_rows=B)            # one shard covering the batch = the batch
    assert torch.equal(shard, a)
    big = torch.empty(1 << 20, dtype=torch.float32, device="cuda")
    from vqa_transfer_externaldata_amd import _lib
    import ctypes as C
    _lib.check(eng.lib.vqa_normal_noise(C.c_void_p(big.data_ptr()), big.numel(), 123, 0, None), "noise")
    torch.cuda.synchronize()
    x = big.double().cpu().numpy()
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1) < 5e-3 and abs((x ** 3).mean()) < 2e-2 and abs((x ** 4).mean() - 3) < 5e-2
    assert np.isfinite(x).all() and np.abs(x).max() < 7
    zero = dict(masks, noise=np.zeros_like(masks["noise"]))
    run_engine(eng, batch, zero)
    pn = {k: v for k, v in p.items() if not k.startswith("q_L_log_sigma_sq/")}
    eng_n = make_engine("vlmap_answer_no_noise", pn, table, nbox, am, B, R, T, dims)
    run_engine(eng_n, batch, {k: v for k, v in masks.items() if k != "noise"})
    assert torch.equal(eng.tensor("logit"), eng_n.tensor("logit"))
    ex = eng.extra_report()
    assert abs(ex["total_loss"] - (eng_n.report()["answer_train_loss"] + 0.1 * ex["latent_loss"])) <= 1e-4 * abs(ex["total_loss"])


def test_adapt_one_box_and_short_boxes():
    mt, dims, B, R, T, N = "vlmap_answer_adapt", MED, 12, 36, 14, 24
    p, table, nbox, batch, am, masks = make_case(65, mt, B, R, T, N, dims)
    nbox[batch["image_idx"][0]] = 1
    nbox[batch["image_idx"][1]] = 5
    eng = make_engine(mt, p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    _check_against_oracle(eng, mt, p, table, nbox, batch, am, masks, B, R, T, dims)
    pooled = eng.tensor("pooled_V_ft").view(B, -1)
    va = eng.tensor("v_adapt").view(B, R, -1)
    if nbox[batch["image_idx"][0]] == 1:
        assert torch.allclose(pooled[0], va[0, 0], atol=1e-6)


# ------------------------------------------------------------------------------------------------------------------
# Model classes through the Trainer mirror (vqa/trainer.py:84 builds every one of them: Model(batch, config, is_train))
# ------------------------------------------------------------------------------------------------------------------
OLD_KEYS = {"answer_train_loss", "answer_report_loss", "answer_accuracy", "exist_answer_accuracy", "test_answer_accuracy",
            "normal_test_answer_accuracy", "max_exist_answer_accuracy", "test_max_answer_accuracy",
            "test_max_exist_answer_accuracy"}
REPORT_OF = {"vlmap_answer2": set(O.REPORT_KEYS), "vlmap_answer_ent": set(O.REPORT_KEYS) | {"entropy", "weighted_entropy"},
             "vlmap_answer_no_noise": OLD_KEYS | {"model_step"}, "vlmap_answer_adapt": OLD_KEYS,
             "vlmap_answer_full": OLD_KEYS | {"model_step", "latent_loss_weight", "latent_loss", "train_latent_loss"}}
NEW_SCOPES = {"vlmap_answer2": {"q_L_ft2"}, "vlmap_answer_no_noise": {"q_L_mean"}, "vlmap_answer_adapt": {"v_adapt"},
              "vlmap_answer_full": {"q_L_mean", "q_L_log_sigma_sq"}, "vlmap_answer_ent": set()}


@pytest.mark.parametrize("model_type", TYPES)
def test_variant_trains_through_the_trainer(tmp_path, model_type):
    """`python vqa/trainer.py --model_type <variant>`: importer entry, constructor, frozen / transfer sets (the same four /
    three scopes in all five files), a transferred head from a word-weight directory, report keys of the variant's
    generation, loss = sum of the variant's losses, checkpoints, and only trainable variables move"""
    import os
    import pickle
    from tests.test_gpu_trainer import _config, _datasets, _features
    from vqa_transfer_externaldata_amd import hdf5_io, importer, trainer
    assert model_type in importer.get_model_types()
    c, Vq, A = _config(tmp_path, model_type, learning_rate=5e-4)
    c.num_marginal = 12                                           # (vlmap_answer_ent; 200 in the reference, :16)
    rng = np.random.default_rng(5)
    wdir = tmp_path / "word_weights"
    os.makedirs(str(wdir))
    src = ["a%d" % i for i in range(A) if i % 3]                  # two of three answers are known to the word-weight dir
    ww = {"class_weights": (0.05 * rng.standard_normal((2048, len(src)))).astype(np.float32),
          "class_biases": (0.1 * rng.standard_normal(len(src))).astype(np.float32)}
    hdf5_io.write(str(wdir / "weights.hdf5"), ww)
    with open(str(wdir / "answer_dict.pkl"), "wb") as f:
        pickle.dump({"vocab": src, "dict": {a: i for i, a in enumerate(src)}}, f)
    c.vlmap_word_weight_dir = str(wdir)
    t = trainer.Trainer(c, datasets=_datasets(Vq, A), image_features=_features())
    m, eng = t.model, t.model.engine
    assert m.MODEL_TYPE == model_type and set(m.report) == REPORT_OF[model_type]
    frozen_scopes = ("q_linear_l", "pooled_linear_l", "joint_fc", "WordWeightAnswer")
    assert not any(v.split("/")[0] in frozen_scopes for v in t.train_vars)
    assert NEW_SCOPES[model_type] <= {v.split("/")[0] for v in t.train_vars}
    assert sorted({v.split("/")[0] for v in t.transfer_vars}) == ["joint_fc", "pooled_linear_l", "q_linear_l"]
    np.testing.assert_array_equal(eng.params["WordWeightAnswer/fc/weights"][:, 1].cpu().numpy(), ww["class_weights"][:, 0])
    assert float(eng.params["WordWeightAnswer/fc/biases"][0]) == -100.0
    if model_type == "vlmap_answer_adapt":
        assert tuple(eng.params["pooled_linear_l/fc/weights"].shape) == (1024, 1024) and m.mid_result["pooled_V_ft"].shape == (32, 1024)
    if model_type == "vlmap_answer2":
        assert float(m.heavy_output["condition"].abs().max()) <= 1.0                # tanh output, not the GRU state
    if model_type == "vlmap_answer_ent":
        assert eng.dims.num_marginal == 12 and eng.dims.ent_cols == 32 and m.mid_result["marginal_prob"].shape == (32, 32)
        sel = (np.arange(A) < 30) & (np.arange(A) % 3 != 0)
        mp = m.mid_result["marginal_prob"].cpu().numpy()
        np.testing.assert_allclose(mp[:, :30][:, sel[:30]].sum(1), 1.0, atol=1e-5)
        assert np.all(mp[:, :30][:, ~sel[:30]] == 0)
    frozen = {k: v.clone() for k, v in eng.params.items() if k.split("/")[0] in frozen_scopes}
    moving = {k: eng.params[k].clone() for k in eng.train_names if k.split("/")[0] in NEW_SCOPES[model_type] | {"v_linear_v"}}
    step, summary, loss0, report, dt = t.run_train_step(True)
    assert step == 1 and set(report) == REPORT_OF[model_type] and np.isfinite(loss0)
    if model_type == "vlmap_answer_full":
        assert report["latent_loss_weight"] == 0.1 and report["model_step"] >= 0
        assert abs(loss0 - (report["answer_train_loss"] + report["train_latent_loss"])) <= 1e-4 * abs(loss0)
        assert abs(report["train_latent_loss"] - 0.1 * report["latent_loss"]) <= 1e-6 + 1e-5 * abs(report["latent_loss"])
    elif model_type == "vlmap_answer_ent":
        assert abs(loss0 - (report["answer_train_loss"] + report["weighted_entropy"])) <= 1e-4 * abs(loss0)
        assert report["entropy"] < 0 and abs(report["weighted_entropy"] - 0.1 * report["entropy"]) <= 1e-6
    else:
        assert abs(loss0 - report["answer_train_loss"]) <= 1e-6 * max(1.0, abs(loss0))
    t.train()
    assert t.global_step == 13 and os.path.exists(os.path.join(c.train_dir, "model-8"))
    for k, v in frozen.items():
        assert torch.equal(eng.params[k], v), k
    for k, v in moving.items():
        assert not torch.equal(eng.params[k], v), k
    _, _, loss1, vreport, _ = t.run_val_step(False, "val")
    assert set(vreport) == REPORT_OF[model_type] and np.isfinite(loss1)
    # checkpoint round trip: a fresh trainer restored from model-8 holds the same variables and step
    c2, _, _ = _config(tmp_path, model_type, learning_rate=5e-4)
    c2.num_marginal, c2.vlmap_word_weight_dir, c2.train_dir = 12, str(wdir), str(tmp_path / "run2")
    c2.checkpoint = os.path.join(c.train_dir, "model-8")
    t2 = trainer.Trainer(c2, datasets=_datasets(Vq, A), image_features=_features())
    sd = torch.load(c2.checkpoint, map_location="cpu")
    assert t2.global_step == 8
    for k in t2.model.engine.shapes:
        assert torch.equal(t2.model.engine.params[k].cpu(), sd[k]), k
