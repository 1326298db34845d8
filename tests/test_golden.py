"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must reproduce
them on CPU, and the HIP path must match them on the GPU."""
import os

import numpy as np
import pytest

from oracle import conv_oracle as CO
from oracle import vqa_oracle as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FUSION = ["fusion_vlmap_answer_b8.npz", "fusion_standard_b8.npz", "fusion_standard_word2vec_b4.npz",
          "fusion_vlmap_answer_vqa_all2_b8.npz"] + ["fusion_%s_b8.npz" % mt for mt in O.ABLATION_FAMILY]
# one LayerNorm per shared fc_layer scope (TF 1.x; default) / one per call site: the parameter names say which
PRETRAIN = ["pretrain_cfg5_toy.npz", "pretrain_cfg5_toy_persite.npz"]


def _load(name):
    z = np.load(os.path.join(HERE, name))
    sub = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return z, sub


@pytest.mark.parametrize("name", FUSION)
def test_oracle_reproduces_fusion_golden(name):
    z, sub = _load(name)
    mt = str(z["model_type"])
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    masks = {k: v.astype(np.float64) for k, v in sub("keep/").items()}
    if "noise" in z.files:
        masks["noise"] = z["noise"].astype(np.float64)
    loss, report, out, mid, tape = O.forward(to64(sub("param/")), to64(sub("batch/")), z["table"].astype(np.float64),
                                             z["nbox"], to64(sub("amask/")), masks, mt)
    for k, v in sub("mid/").items():
        np.testing.assert_allclose(np.asarray(mid[k], np.float64), v, rtol=1e-12, atol=1e-12, err_msg=k)
    for k, v in sub("report/").items():
        assert abs(report[k] - float(v)) <= 1e-12 * max(1, abs(float(v))), k
    assert abs(loss - float(z["loss"])) <= 1e-12 * max(1, abs(float(z["loss"])))
    if mt != "vlmap_answer2":                                           # (its `condition` is tanh(LN(.)) of the GRU state)
        assert np.all(sub("mid/")["condition"][0] == 0)                 # the fixture keeps the len-0 question
    np.testing.assert_array_equal(sub("mid/")["att_score"][1], np.eye(int(z["R"]))[0])   # and the 1-box image
    # float32 oracle stays within the published tolerance of its float64 self
    f32 = lambda d: {k: (v.astype(np.float32) if v.dtype.kind == "f" else v) for k, v in d.items()}
    _, _, _, mid32, _ = O.forward(f32(sub("param/")), f32(sub("batch/")), z["table"], z["nbox"], f32(sub("amask/")),
                                  {k: v.astype(np.float32) for k, v in masks.items()}, mt)
    grads, dx = O.backward(to64(sub("param/")), to64(sub("batch/")), to64(sub("amask/")), masks, tape, mt)
    for k, v in sub("grad/").items():
        np.testing.assert_allclose(grads[k], v, rtol=2e-6, atol=1e-9, err_msg=k)       # stored as float32
    assert np.abs(mid32["logit"] - sub("mid/")["logit"]).max() < 1e-3
    np.testing.assert_array_equal(mid32["pred"], sub("mid/")["pred"])


def _pretrain_fixture(name):
    z, sub = _load(name)
    cfg = {k: int(z[k]) for k in ("B", "n", "R", "D", "H", "L", "W", "Vq", "n_ws", "A")}
    return z, sub, cfg


@pytest.mark.parametrize("name", PRETRAIN)
def test_oracle_reproduces_pretrain_golden(name):
    from oracle import pretrain_oracle as PO
    z, sub, cfg = _pretrain_fixture(name)
    assert PO.ln_shared_in(sub("param/")) == ("persite" not in name)
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    masks = {k: v.astype(np.float64) for k, v in sub("keep/").items()}
    total, report, mid = PO.forward(to64(sub("param/")), to64(sub("batch/")), masks, cfg["n"])
    assert abs(total - float(z["total_loss"])) <= 1e-12 * max(1.0, abs(total))
    assert sorted(report) == sorted(sub("report/"))
    for k, v in sub("report/").items():
        assert abs(report[k] - float(v)) <= 1e-12 * max(1.0, abs(float(v))), k
    for k, v in sub("mid/").items():
        np.testing.assert_allclose(np.asarray(mid[k], np.float64), v, rtol=1e-12, atol=1e-12, err_msg=k)
    _, _, grads, slices = PO.torch_loss_and_grads(to64(sub("param/")), to64(sub("batch/")), masks, cfg["n"])
    for k, v in sub("grad/").items():
        np.testing.assert_allclose(grads[k], v, rtol=1e-9, atol=1e-12, err_msg=k)


def test_oracle_reproduces_conv_golden():
    z, sub = _load("vfeat_resnet_narrow.npz")
    blocks = [("block%d" % (i + 1), int(b), int(u), int(s)) for i, (b, u, s) in enumerate(z["blocks"])]
    p64 = {k: v.astype(np.float64) for k, v in sub("param/").items()}
    v, enc = CO.model_vfeat_resnet(z["image"].astype(np.float64), z["normal_box"].astype(np.float64), p64, blocks)
    np.testing.assert_allclose(v, z["V_ft"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(enc, z["enc_I"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FUSION)
def test_hip_matches_fusion_golden(name):
    import torch
    from vqa_transfer_externaldata_amd import fusion as F
    z, sub = _load(name)
    mt = str(z["model_type"])
    B, R, T, N = int(z["B"]), int(z["R"]), int(z["T"]), int(z["N"])
    dims = {k: int(z["dim_" + k]) for k in ("Vq", "W", "D", "H", "A")}
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    params = sub("param/")
    kw = {}
    if mt == "standard_word2vec":      # the constant answer-GloVe matrix travels beside the variables
        kw["answer_glove"] = params[O.OUTPUT_GLOVE]
    if mt == "vlmap_answer_ent":
        kw["num_marginal"] = int(sub("keep/")["tile_joint"].shape[1])
    eng = F.FusionEngine(model_type=mt, B=B, R=R, T=T, N_img=N,
                         params={k: v for k, v in params.items() if not O.is_const(k)}, **dims, **kw)
    eng.bind_inputs(table=dev(z["table"]), nbox_table=dev(z["nbox"]),
                    answer_masks={k: dev(v) for k, v in sub("amask/").items()})
    batch = {k: dev(v) for k, v in sub("batch/").items()}
    keep = sub("keep/")
    kw = {}
    if "noise" in z.files:
        kw["noise"] = dev(z["noise"].astype(np.float32))
    if "tile_joint" in keep:
        kw["keep_tile"] = dev(keep["tile_joint"])
    eng.forward(batch, dev(keep["att"]), dev(keep["joint"]), want_dz=True, **kw)
    eng.backward()
    torch.cuda.synchronize()
    mid = sub("mid/")
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), mid["pred"])            # bit exact
    np.testing.assert_array_equal(eng.tensor("num_V_ft").cpu().numpy(), mid["num_V_ft"])
    assert np.abs(eng.tensor("logit").cpu().numpy().reshape(mid["logit"].shape) - mid["logit"]).max() < 1e-3
    for k in ("v_linear_v", "condition", "q_linear_v", "att_score", "pooled_V_ft", "joint", "v_adapt", "q_L_mean",
              "q_L_log_sigma_sq", "q_L_mean_noise"):
        if k not in mid:
            continue
        got = eng.tensor(k).cpu().numpy()[:mid[k].size].reshape(mid[k].shape)
        assert np.abs(got - mid[k]).max() <= 2e-4 * max(1.0, np.abs(mid[k]).max()), k
    rep = eng.report()
    rep.update(eng.extra_report())
    old = {new: rep[o] for new, o in O.TESTMASK_REPORT.items()}            # the oldest variants' key names
    for k, v in sub("report/").items():
        got = rep[k] if k in rep else old[k]
        assert abs(got - float(v)) <= 1e-4 * max(1.0, abs(float(v))), k
    assert abs(float(eng.loss()) - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    grads = sub("grad/")
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            continue
        g = eng.grads[n].cpu().numpy()
        assert np.abs(g - grads[n]).max() <= 5e-4 * max(np.abs(grads[n]).max(), 1e-12) + 1e-9, n


@pytest.mark.gpu
def test_hip_matches_conv_golden():
    import torch
    from vqa_transfer_externaldata_amd import vfeat as VF
    z, sub = _load("vfeat_resnet_narrow.npz")
    blocks = [("block%d" % (i + 1), int(b), int(u), int(s)) for i, (b, u, s) in enumerate(z["blocks"])]
    model = VF.VfeatResnetModel(sub("param/"), blocks)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    v = model.build({"image": dev(z["image"]), "normal_box": dev(z["normal_box"])}).cpu().numpy()
    assert np.abs(v - z["V_ft"]).max() <= 1e-4 * np.abs(z["V_ft"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", PRETRAIN)
def test_hip_matches_pretrain_golden(name):
    import torch
    from vqa_transfer_externaldata_amd import pretrain as PT
    z, sub, cfg = _pretrain_fixture(name)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    eng = PT.PretrainEngine(n=cfg["n"], R=cfg["R"], D=cfg["D"], H=cfg["H"], W=cfg["W"], A=cfg["A"], Vq=cfg["Vq"],
                            n_ws=cfg["n_ws"], params=sub("param/"))
    assert eng.ln_shared == ("persite" not in name)                   # the variable names decide
    eng.forward({k: dev(v) for k, v in sub("batch/").items()}, {k: dev(v) for k, v in sub("keep/").items()})
    eng.backward()
    torch.cuda.synchronize()
    rep = eng.fetch_report()
    for k, v in sub("report/").items():
        assert abs(rep[k] - float(v)) <= 2e-4 * max(1.0, abs(float(v))), (k, rep[k], float(v))
    mid = sub("mid/")
    for kind in ("obj", "attr"):
        if kind + "/att" not in mid:
            continue
        t = eng._tape["kinds"][kind]
        assert np.abs(t["att"].cpu().numpy() - mid[kind + "/att"]).max() < 1e-5
        assert np.abs(t["blank_fill"]["z"].cpu().numpy().reshape(mid[kind + "/bf_logit"].shape) - mid[kind + "/bf_logit"]).max() < 1e-3
        assert np.abs(t["wordset"]["z"].cpu().numpy().reshape(mid[kind + "/ws_logit"].shape) - mid[kind + "/ws_logit"]).max() < 1e-3
    grads = sub("grad/")
    for name in eng.train_names:
        g = eng.grads[name].cpu().numpy().astype(np.float64)
        if name.endswith("score/fc/biases"):
            assert np.abs(g).max() < 1e-5                              # analytically zero
            continue
        sc = max(np.abs(grads[name]).max(), 1e-12)
        assert np.abs(g - grads[name]).max() <= 1e-3 * sc + 1e-8, name
    sq = float(z["slice_sq"])
    assert abs(float(eng.grad_flat[eng.n_train]) - sq) <= 1e-3 * sq + 1e-12
