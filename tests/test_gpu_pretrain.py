"""GPU parity of the cfg-5 pre-training model against its oracle: forward report,
logits, every gradient (torch-autograd reference), clip+Adam step, export bridge."""
import numpy as np
import pytest
import torch

from oracle import pretrain_oracle as PO

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _setup(seed, B, n, R, D, H, L, W, Vq, n_ws, A, ln_shared=True):
    from vqa_transfer_externaldata_amd import pretrain as PT
    rng = np.random.default_rng(seed)
    p = PO.init_params(rng, Vq, n_ws, A, W=W, D=D, H=H, ln_shared=ln_shared)
    batch = PO.make_batch(rng, B, n, R, D, L, Vq, n_ws, A)
    masks = PO.make_masks(rng, B, n, R, H)
    eng = PT.PretrainEngine(n=n, R=R, D=D, H=H, W=W, A=A, Vq=Vq, n_ws=n_ws, params=p)
    assert eng.ln_shared == ln_shared                  # the variable names of `p` decide
    db = {k: dev(v) for k, v in batch.items()}
    dm = {k: dev(v.astype(np.uint8)) for k, v in masks.items()}
    return PT, eng, p, batch, masks, db, dm


@pytest.mark.parametrize("ln_shared", [True, False])
@pytest.mark.parametrize("sort", [False, True])
@pytest.mark.parametrize("cfg", [dict(B=3, n=5, R=6, D=16, H=8, L=4, W=12, Vq=20, n_ws=7, A=12),
                                 dict(B=16, n=5, R=36, D=256, H=128, L=10, W=300, Vq=200, n_ws=50, A=400)])
def test_forward_backward_match_oracle(cfg, sort, ln_shared):
    PT, eng, p, batch, masks, db, dm = _setup(5, ln_shared=ln_shared, **cfg)
    if sort:       # captions encoded in length order, finished ones skipped by the recurrence: same results
        db.update({k: v for k, v in PT.add_length_sort(dict(batch)).items() if k.endswith("/sort")})
    eng.forward(db, dm)
    eng.backward()
    torch.cuda.synchronize()
    rep = eng.fetch_report()
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    total, report, mid = PO.forward(to64(p), to64(batch), to64(masks), cfg["n"])
    assert sorted(rep) == sorted(report)                              # the 13 report keys of the reference
    for k in report:
        assert abs(rep[k] - report[k]) <= 2e-4 * max(1.0, abs(report[k])), (k, rep[k], report[k])
    for k in PO.KINDS:
        z = eng._tape["kinds"][k]["blank_fill"]["z"].cpu().numpy().reshape(mid[k + "/bf_logit"].shape)
        assert np.abs(z - mid[k + "/bf_logit"]).max() < 1e-3
        z = eng._tape["kinds"][k]["wordset"]["z"].cpu().numpy().reshape(mid[k + "/ws_logit"].shape)
        assert np.abs(z - mid[k + "/ws_logit"]).max() < 1e-3
        att = eng._tape["kinds"][k]["att"].cpu().numpy()
        assert np.abs(att - mid[k + "/att"]).max() < 1e-5
    _, _, grads, slices = PO.torch_loss_and_grads(to64(p), to64(batch), to64(masks), cfg["n"])
    for name in eng.train_names:
        g = eng.grads[name].cpu().numpy().astype(np.float64)
        sc = max(np.abs(grads[name]).max(), 1e-12)
        if name.endswith("score/fc/biases"):
            assert np.abs(g).max() < 1e-5                              # analytically zero
            continue
        assert np.abs(g - grads[name]).max() <= 1e-3 * sc + 1e-8, (name, np.abs(g - grads[name]).max(), sc)
    sq = sum(float((v ** 2).sum()) for v in slices.values())
    assert abs(float(eng.grad_flat[eng.n_train]) - sq) <= 1e-3 * sq + 1e-12


def test_train_steps_reduce_loss_and_export_bridge(tmp_path):
    cfg = dict(B=16, n=5, R=36, D=128, H=64, L=8, W=300, Vq=100, n_ws=30, A=60)
    PT, eng, p, batch, masks, db, dm = _setup(6, **cfg)
    losses = []
    for it in range(12):
        eng.train_step(db, dm, 2e-3)
        losses.append(eng.fetch_report()["total_loss"])
    assert losses[-1] < losses[0] - 0.5, losses
    assert np.isfinite(losses).all()
    # V_GloVe / LearnAnswerGloVe never move (no gradient path in this model)
    for k in PT.NO_GRAD_VARS:
        np.testing.assert_array_equal(eng.params[k].cpu().numpy(), p[k])
    vocab = {"vocab": ["w%d" % i for i in range(cfg["Vq"])]}
    adict = {"vocab": ["a%d" % i for i in range(cfg["A"])], "dict": {"a%d" % i: i for i in range(cfg["A"])}}
    d = PT.export_word_weights(eng.state_dict(), vocab, adict, str(tmp_path / "word_weights_model-12"))
    # ... and the VQA model's WordWeightAnswer init consumes it (vlmap/modules.py:589-627)
    from vqa_transfer_externaldata_amd import model_vlmap_answer as MV
    ww = MV.load_word_weight_dir(d)
    vqa_answers = {"vocab": ["a3", "zzz", "a7"]}
    w, b = MV.word_weight_answer_init(vqa_answers, 2 * cfg["H"], ww)
    cw = eng.params["classifier/fc/weights"].cpu().numpy()
    np.testing.assert_array_equal(w[:, 0], cw[:, 3]); np.testing.assert_array_equal(w[:, 2], cw[:, 7])
    assert np.all(w[:, 1] == 0) and b[1] == -100.0
    with pytest.raises(ValueError, match="Do not overwrite"):
        PT.export_word_weights(eng.state_dict(), vocab, adict, d)


def hip_relu_gates(eng, B):
    """sign pattern of every ReLU of the HIP forward (oracle.pretrain_oracle.RELU_SITES), read from its activations"""
    n, R, H = eng.n, eng.R, eng.H
    g = {}
    for k in PO.KINDS:
        g[k + "/v"] = (eng.tensor(k + "/v").view(B, R, H) > 0).cpu().numpy()
        g[k + "/qv"] = (eng.tensor(k + "/qv").view(B, n, H) > 0).cpu().numpy()
        for hd in ("bf", "ws"):
            for t, w in (("vl", H), ("ll", H), ("j", 2 * H)):     # j is stored after dropout: dropped positions carry no gradient either way
                g["%s/%s/%s" % (k, hd, t)] = (eng.tensor("%s/%s/%s" % (k, hd, t)).view(B, n, w) > 0).cpu().numpy()
    return g


@pytest.mark.parametrize("ln_shared", [True, False])
def test_full_size_cfg5_bs512_matches_oracle_f64(ln_shared):
    """BASELINE configs[4] as configured: bs 512 images = 2560 blank-fill rows per category, D 2048, H 1024, A 4000
    (3000 objects + 1000 attributes), captions <= 10 tokens -- forward report, logits and attention against the
    float64 NumPy oracle; EVERY gradient against the float64 torch-autograd restatement evaluated with the ReLU sign
    pattern of the HIP forward (gate-conditioned: 5e-4 of max|g| element-wise), and against the unconditioned float64
    gradient norm-wise (vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:54-94, 323-609, 675-706)."""
    cfg = dict(B=512, n=5, R=36, D=2048, H=1024, L=10, W=300, Vq=5000, n_ws=2000, A=4000)
    PT, eng, p, batch, masks, db, dm = _setup(9, ln_shared=ln_shared, **cfg)
    db.update({k: v for k, v in PT.add_length_sort(dict(batch)).items() if k.endswith("/sort")})
    eng.forward(db, dm)
    eng.backward()
    torch.cuda.synchronize()
    rep = eng.fetch_report()
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    p64, b64, m64 = to64(p), to64(batch), to64(masks)
    total, report, mid = PO.forward(p64, b64, m64, cfg["n"])
    for k in report:
        assert abs(rep[k] - report[k]) <= 2e-4 * max(1.0, abs(report[k])), (k, rep[k], report[k])
    for k in PO.KINDS:
        for head, key in (("blank_fill", "/bf_logit"), ("wordset", "/ws_logit")):
            z = eng._tape["kinds"][k][head]["z"].cpu().numpy().reshape(mid[k + key].shape)
            assert np.abs(z - mid[k + key]).max() < 1e-3, (k, head, np.abs(z - mid[k + key]).max())
            # top-1 bit-exact, except where float64 itself sees a tie within the logit tolerance (10 240 rows x 4000)
            a, b = z.argmax(-1), mid[k + key].argmax(-1)
            gap = np.take_along_axis(mid[k + key], b[..., None], -1)[..., 0] - np.take_along_axis(mid[k + key], a[..., None], -1)[..., 0]
            assert np.all((a == b) | (gap < 2e-4)) and np.mean(a != b) < 2e-3, (k, head, int((a != b).sum()))
        att = eng._tape["kinds"][k]["att"].cpu().numpy()
        assert np.abs(att - mid[k + "/att"]).max() < 1e-5
    del mid
    hip = {name: eng.grads[name].cpu().numpy().astype(np.float64) for name in eng.train_names}
    # (1) the discriminating bar.  With the HIP forward's ReLU gates fixed the loss is smooth in the parameters, so a
    # correct float32 backward agrees with float64 to accumulated rounding: every trainable tensor, element-wise.
    cap = {}
    gates = hip_relu_gates(eng, cfg["B"])
    _, _, gc, slices = PO.torch_loss_and_grads(p64, b64, m64, cfg["n"], gates=gates)
    worst = {}
    for name in eng.train_names:
        if name.endswith("score/fc/biases"):
            assert np.abs(hip[name]).max() < 1e-5                      # analytically zero (softmax shift invariance)
            continue
        sc = max(np.abs(gc[name]).max(), 1e-30)
        worst[name] = np.abs(hip[name] - gc[name]).max() / sc
    bad = {k: v for k, v in worst.items() if v > 5e-4}
    assert not bad, bad
    sq = sum(float((v ** 2).sum()) for v in slices.values())
    assert abs(float(eng.grad_flat[eng.n_train]) - sq) <= 1e-3 * sq + 1e-12
    # (2) the unconditioned float64 gradient differs from (1) exactly by the gates on which float32 and float64
    # disagree (pre-activations within rounding of 0, a handful among 1e8); each flip moves one column's gradient and
    # what is upstream of it by ~1e-3 of the tensor norm, so this second check is norm-wise
    _, _, gu, _ = PO.torch_loss_and_grads(p64, b64, m64, cfg["n"], capture=cap)
    flips = sum(int((cap[s] != gates[s]).sum()) for s in PO.RELU_SITES if not s.endswith("/j"))
    total_gates = sum(cap[s].size for s in PO.RELU_SITES if not s.endswith("/j"))
    assert flips <= 1e-5 * total_gates, (flips, total_gates)
    for name in eng.train_names:
        if name.endswith("score/fc/biases"):
            continue
        fro = np.linalg.norm(hip[name] - gu[name]) / max(np.linalg.norm(gu[name]), 1e-30)
        assert fro <= 5e-3, (name, fro)


def test_checkpoint_variable_names_choose_the_layernorm_mode():
    """A checkpoint with `<scope>/LayerNorm_1/...` makes the engine per-call-site, one without makes it shared,
    whatever it was built as (PretrainEngine.load_state_dict); a checkpoint that lacks model variables raises."""
    cfg = dict(B=4, n=5, R=6, D=16, H=8, L=4, W=12, Vq=20, n_ws=7, A=12)
    PT, eng_site, p_site, batch, masks, db, dm = _setup(3, ln_shared=False, **cfg)
    eng_site.train_step(db, dm, 1e-3)
    sd_site = eng_site.state_dict()
    assert "joint_fc/LayerNorm_3/gamma" in sd_site and "joint_fc/LayerNorm_3/gamma/Adam" in sd_site
    _, eng, p, _, _, _, _ = _setup(4, ln_shared=True, **cfg)
    assert eng.ln_shared and "joint_fc/LayerNorm_3/gamma" not in eng.state_dict()
    n_shared = eng.n_train
    eng.load_state_dict(sd_site)
    assert not eng.ln_shared and eng.n_train > n_shared and eng.step_count == 1
    for k, v in eng_site.params.items():
        assert torch.equal(v, eng.params[k]), k
    eng.train_step(db, dm, 1e-3); eng_site.train_step(db, dm, 1e-3)
    torch.cuda.synchronize()
    for k in ("joint_fc/LayerNorm_3/gamma", "classifier/fc/weights", "pooled_linear_l/LayerNorm/beta"):
        np.testing.assert_allclose(eng.params[k].cpu().numpy(), eng_site.params[k].cpu().numpy(), rtol=0, atol=2e-6, err_msg=k)
    # and back: a shared-LayerNorm checkpoint into the per-call-site engine
    _, eng_sh, _, _, _, _, _ = _setup(4, ln_shared=True, **cfg)
    eng.load_state_dict(eng_sh.state_dict())
    assert eng.ln_shared and eng.n_train == n_shared
    eng.forward(db, dm); eng_sh.forward(db, dm)
    assert eng.fetch_report() == eng_sh.fetch_report()
    sd = eng_sh.state_dict()
    del sd["q_linear_l/LayerNorm/beta"]
    with pytest.raises(KeyError, match="q_linear_l/LayerNorm/beta"):
        eng.load_state_dict(sd)


def test_checkpoint_resume_continues_the_adam_trajectory():
    """state_dict keeps the Adam moments and the step count: 2 steps + save/load + 2 steps == 4 uninterrupted steps."""
    cfg = dict(B=8, n=5, R=36, D=128, H=64, L=8, W=300, Vq=100, n_ws=30, A=60)
    PT, eng, p, batch, masks, db, dm = _setup(12, **cfg)
    for _ in range(4):
        eng.train_step(db, dm, 2e-3)
    want = {k: v.cpu().numpy().copy() for k, v in eng.params.items()}
    _, eng2, _, _, _, _, _ = _setup(12, **cfg)
    for _ in range(2):
        eng2.train_step(db, dm, 2e-3)
    sd = eng2.state_dict()
    assert "classifier/fc/weights/Adam" in sd and "classifier/fc/weights/Adam_1" in sd and int(sd["global_step"]) == 2
    zeros = {k: np.zeros_like(np.asarray(v)) for k, v in p.items()}
    eng3 = PT.PretrainEngine(n=cfg["n"], R=cfg["R"], D=cfg["D"], H=cfg["H"], W=cfg["W"], A=cfg["A"], Vq=cfg["Vq"],
                             n_ws=cfg["n_ws"], params=zeros)
    eng3.load_state_dict(sd)
    assert eng3.step_count == 2
    for _ in range(2):
        eng3.train_step(db, dm, 2e-3)
    torch.cuda.synchronize()
    for k, v in eng3.params.items():
        if k.endswith("score/fc/biases"):
            continue       # analytically zero gradient (softmax shift invariance): Adam amplifies run-to-run rounding noise
        # bitwise up to the atomically scatter-added embedding gradients (order of the adds varies run to run)
        np.testing.assert_allclose(v.cpu().numpy(), want[k], rtol=0, atol=2e-6, err_msg=k)


def test_resident_feature_tables_give_the_dense_batch_results():
    """PretrainEngine.bind_tables + batches that carry image_idx: the rows gathered on the device are the dense
    batch's, so report, logits and gradients are bit for bit those of the dense path"""
    from vqa_transfer_externaldata_amd import pretrain as PT
    cfg = dict(B=6, n=5, R=36, D=64, H=32, L=5, W=12, Vq=30, n_ws=9, A=20)
    rng = np.random.default_rng(11)
    p = PO.init_params(rng, cfg["Vq"], cfg["n_ws"], cfg["A"], W=cfg["W"], D=cfg["D"], H=cfg["H"])
    batch = PO.make_batch(rng, cfg["B"], cfg["n"], cfg["R"], cfg["D"], cfg["L"], cfg["Vq"], cfg["n_ws"], cfg["A"])
    masks = {k: dev(v.astype(np.uint8)) for k, v in PO.make_masks(rng, cfg["B"], cfg["n"], cfg["R"], cfg["H"]).items()}
    N = 17
    idx = rng.permutation(N)[:cfg["B"]].astype(np.int64)
    table = rng.standard_normal((N, cfg["R"], cfg["D"])).astype(np.float32)
    spat = rng.random((N, cfg["R"], 6)).astype(np.float32)
    nbox = rng.integers(1, cfg["R"] + 1, size=N).astype(np.int32)
    dense = dict(batch, image_ft=table[idx], spatial_ft=spat[idx], num_boxes=nbox[idx])
    res = {k: v for k, v in batch.items() if k not in ("image_ft", "spatial_ft", "num_boxes")}
    res["image_idx"] = idx
    out = []
    for b, bind in ((dense, False), (res, True)):
        eng = PT.PretrainEngine(n=cfg["n"], R=cfg["R"], D=cfg["D"], H=cfg["H"], W=cfg["W"], A=cfg["A"], Vq=cfg["Vq"],
                                n_ws=cfg["n_ws"], params=p, deterministic=True)      # atomic-free embedding gradients
        if bind:
            eng.bind_tables(table, spat, nbox)
        eng.forward({k: dev(v) for k, v in b.items()}, masks)
        eng.backward()
        torch.cuda.synchronize()
        out.append((eng.fetch_report(), eng.grad_flat.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])
    eng = PT.PretrainEngine(n=cfg["n"], R=cfg["R"], D=cfg["D"], H=cfg["H"], W=cfg["W"], A=cfg["A"], Vq=cfg["Vq"],
                            n_ws=cfg["n_ws"], params=p)
    with pytest.raises(ValueError, match="bind_tables"):
        eng.forward({k: dev(v) for k, v in res.items()}, masks)


def test_trainer_loop_with_tables_in_hbm_and_forked_producers_in_a_fresh_process():
    """tools/pretrain_trainer_bench.py resident-workers: the Trainer mirror with both splits' tables on the device and
    4 forked batch producers started before the GPU is touched (a fresh process, as `python -m ...pretrain_trainer`)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pretrain_trainer_bench.py"), "resident-workers", "4",
                        "64", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "resident-workers" in r.stdout and "images/s" in r.stdout, r.stdout


def test_trainer_step_with_prefetched_batch_and_deferred_report_equals_the_plain_sequence():
    """Trainer.run_train_step draws and uploads batch i+1 on a side stream while step i runs and fetches the 13 report
    scalars once per step (Model.prepare / build(prepared=, defer_report=True) / finish_report).  Same batches, same
    dropout masks, same reports and parameters as set_batch -> build -> backward -> apply_gradients, one step at a time
    (vlmap_memft/trainer.py:202-263)."""
    import tempfile
    from vqa_transfer_externaldata_amd import dataset_vlmap as DV, pretrain_trainer as PTT
    R, D, L, Vq, n_ws, A, B = 36, 64, 6, 40, 12, 30, 8

    def make():
        data = DV.synthetic_dataset(40, Vq, n_ws, A, R=R, D=D, max_len=L, seed=5)
        ds = {"train": DV.Dataset(split="train", data=data, seed=1), "val": DV.Dataset(split="val", data=data, seed=2)}
        cfg = PTT.build_parser().parse_args(["--batch_size", str(B), "--max_train_iter", "4", "--learning_rate", "0.002",
                                             "--features_on_device", "1", "--input_workers", "0", "--input_prefetch", "0"])
        cfg.data_cfg = ds["train"].get_config()
        cfg.vocab = {"vocab": ["w%d" % i for i in range(Vq)], "dict": {"w%d" % i: i for i in range(Vq)}}
        cfg.answer_dict, cfg.ws_dict = data["answer_dict"], data["ws_dict"]
        cfg.synthetic, cfg.train_dir = 1, tempfile.mkdtemp()
        cfg.deterministic = 1      # atomic-free embedding gradients: run-to-run noise (1e-6, which Adam can amplify on
        return PTT.Trainer(cfg, ds)  # near-zero gradients within a few steps) does not blur the comparison

    ta, tb = make(), make()
    reports_a, reports_b = [], []
    for _ in range(4):
        step, _, loss, report, _ = ta.run_train_step(False)
        reports_a.append(report)
        tb.model.set_batch(tb._next("train"))
        tb.model.build()
        tb.model.backward()
        tb.model.apply_gradients(tb._lr())
        torch.cuda.synchronize()
        reports_b.append(dict(tb.model.report))
    assert step == 4
    for ra, rb in zip(reports_a, reports_b):
        assert sorted(ra) == sorted(rb)
        for k in ra:
            assert abs(ra[k] - rb[k]) <= 1e-5 * max(1.0, abs(rb[k])), (k, ra[k], rb[k])
    pa, pb = ta.model.engine.params, tb.model.engine.params
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k        # deterministic engines: the two sequences are bit for bit the same
    # a validation step in between leaves the prefetched training batch in place
    ta.run_val_step(False)
    assert ta._prepared is not None
    ta.run_train_step(False)


def test_two_prepared_batches_in_flight_without_a_device_sync():
    """Model.prepare twice in a row (side-stream uploads, no synchronisation in between), then both batches built on the
    compute stream: each gives the report of a plain build of the same batch (the prepared tensors are recorded on the
    compute stream, so the caching allocator cannot recycle the first batch's blocks for the second)."""
    import tempfile
    from vqa_transfer_externaldata_amd import dataset_vlmap as DV, pretrain_trainer as PTT
    R, D, L, Vq, n_ws, A, B = 36, 64, 6, 40, 12, 30, 8
    data = DV.synthetic_dataset(40, Vq, n_ws, A, R=R, D=D, max_len=L, seed=5)
    ds = {"train": DV.Dataset(split="train", data=data, seed=1), "val": DV.Dataset(split="val", data=data, seed=2)}
    cfg = PTT.build_parser().parse_args(["--batch_size", str(B), "--features_on_device", "1", "--input_workers", "0",
                                         "--input_prefetch", "0"])
    cfg.data_cfg = ds["train"].get_config()
    cfg.vocab = {"vocab": ["w%d" % i for i in range(Vq)], "dict": {"w%d" % i: i for i in range(Vq)}}
    cfg.answer_dict, cfg.ws_dict = data["answer_dict"], data["ws_dict"]
    cfg.synthetic, cfg.train_dir, cfg.dropout_off = 1, tempfile.mkdtemp(), True
    t = PTT.Trainer(cfg, ds)
    b1, b2 = t._next("train"), t._next("train")
    want = []
    for b in (b1, b2):
        t.model.set_batch(b)
        t.model.build()
        want.append(dict(t.model.report))
    p1 = t.model.prepare(b1)
    p2 = t.model.prepare(b2)               # no torch.cuda.synchronize() between the two
    got = []
    for p in (p1, p2):
        t.model.build(prepared=p)
        got.append(dict(t.model.report))
    del p1, p2
    torch.cuda.synchronize()
    assert got == want
