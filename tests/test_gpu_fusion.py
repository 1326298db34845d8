"""GPU parity of the whole fusion model (forward, backward, clip+Adam) against
the CPU oracle, through the C ABI (vqa_fusion_forward / vqa_fusion_backward /
vqa_sumsq / vqa_clip_adam).  Tolerances: logits 1e-3 abs (BASELINE north_star),
argmax/pred bit-exact; gradients 5e-4 of the tensor's max-abs."""
import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O
from tests.gpu_util import dev, dev_batch, make_case, make_engine, to64

pytestmark = pytest.mark.gpu

SMALL = dict(Vq=30, W=12, D=24, H=16, A=21)
MED = dict(Vq=500, W=300, D=256, H=128, A=300)
FULL = dict(Vq=2000, W=300, D=2048, H=1024, A=3000)

MID_KEYS = ["v_linear_v", "condition", "q_linear_v", "att_score", "pooled_V_ft", "pooled_linear_l", "l_linear_l",
            "joint", "logit"]


def run_engine(eng, batch, masks, lr=None):
    ka, kj = dev(masks["att"].astype(np.uint8)), dev(masks["joint"].astype(np.uint8))
    kj2 = dev(masks["joint_l"].astype(np.uint8)) if "joint_l" in masks else None        # vlmap_answer_noc's second dropout site
    db = dev_batch(batch)
    eng.forward(db, ka, kj, want_dz=True, keep_joint2=kj2)
    eng.backward()
    if lr is not None:
        eng.optimizer_step(lr)
    torch.cuda.synchronize()


def grad_close(got, want, name, tol=5e-4):
    got = got.detach().cpu().numpy().astype(np.float64)
    sc = max(np.abs(want).max(), 1e-12)
    err = np.abs(got - want).max()
    assert err <= tol * sc + 1e-9, "%s: max err %.3e vs scale %.3e" % (name, err, sc)


@pytest.mark.parametrize("model_type", ["vlmap_answer", "standard", "standard_word2vec", "standard_testmask",
                                        "vlmap_answer_vqa_all2", "vlmap_answer_noc", "vlmap_answer_vqa_all"])
@pytest.mark.parametrize("cfg", [("small", SMALL, 5, 6, 7, 9), ("med", MED, 32, 36, 14, 64),
                                 ("cfg1_full_dims", FULL, 8, 36, 14, 24)])
def test_forward_backward_match_oracle(model_type, cfg):
    name, dims, B, R, T, N = cfg
    p, table, nbox, batch, am, masks = make_case(21, model_type, B, R, T, N, dims)
    eng = make_engine(model_type, p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    # oracle in float64 on the same f32 inputs
    p64 = to64(p)
    loss, report, out, mid, tape = O.forward(p64, to64(batch), table.astype(np.float64), nbox, to64(am), to64(masks),
                                             model_type)
    grads, dx = O.backward(p64, to64(batch), to64(am), to64(masks), tape, model_type)
    shapes = {"v_linear_v": (B, R, dims["H"]), "att_score": (B, R), "logit": (B, dims["A"])}
    if model_type in O.NOC_FAMILY:      # "joint" holds v_joint there; l_joint is the second branch
        mid = dict(mid, joint=mid["v_joint"])
        got = eng.tensor("l_joint").cpu().numpy().reshape(mid["l_joint"].shape)
        assert np.abs(got - mid["l_joint"]).max() <= 2e-4 * max(1.0, np.abs(mid["l_joint"]).max())
    for k in MID_KEYS:
        got = eng.tensor(k).cpu().numpy().reshape(mid[k].shape)
        tol = 1e-3 if k == "logit" else 2e-4 * max(1.0, np.abs(mid[k]).max())
        assert np.abs(got - mid[k]).max() <= tol, (k, np.abs(got - mid[k]).max())
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])          # bit-exact argmax
    np.testing.assert_array_equal(eng.tensor("num_V_ft").cpu().numpy(), mid["num_V_ft"])
    rep = eng.report()
    for k in O.REPORT_KEYS:
        assert abs(rep[k] - report[k]) <= 1e-4 * max(1.0, abs(report[k])), (k, rep[k], report[k])
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            # analytically zero (softmax shift invariance): only rounding noise on both sides
            assert abs(float(eng.grads[n][0])) <= 1e-5
            continue
        grad_close(eng.grads[n], grads[n], n)
    grad_close(eng.tensor("dx_embed").view(T, B, dims["W"]).transpose(0, 1), dx, "dx_embed")
    sq = float(eng.grad_flat[eng.n_train])
    assert abs(sq - (dx ** 2).sum()) <= 1e-3 * (dx ** 2).sum() + 1e-12


def test_vqa_all2_two_heads_and_dead_branch():
    """model_type 4 (vqa/model_vlmap_answer_vqa_all2.py:196-244): both heads' logits, their sum, the mixed-mask argmax on
    a case where the two heads disagree, the frozen set, zero gradients on the tuned layers that feed nothing, two train
    steps against the oracle, and the dead branch's mid results on request."""
    mt, dims, B, R, T, N = "vlmap_answer_vqa_all2", MED, 24, 36, 14, 32
    p, table, nbox, batch, am, masks = make_case(61, mt, B, R, T, N, dims)
    eng = make_engine(mt, p, table, nbox, am, B, R, T, dims)
    assert eng.frozen_names == sorted(n for n in p if n.split("/")[0] in ("q_linear_l", "pooled_linear_l", "joint_fc", "WordWeightAnswer"))
    assert "TunedWordWeightAnswer/fc/weights" in eng.train_names and "tuned_joint_fc/fc/weights" in eng.train_names
    run_engine(eng, batch, masks)
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am), to64(masks), mt)
    z1 = eng.tensor("logit_fixed").view(B, -1).cpu().numpy()
    z2 = eng.tensor("logit_tuned").view(B, -1).cpu().numpy()
    assert np.abs(z1 - tape["z1"]).max() < 1e-3 and np.abs(z2 - tape["z2"]).max() < 1e-3
    np.testing.assert_array_equal(eng.tensor("logit").view(B, -1).cpu().numpy(), z1 + z2)       # output['logit'] (:226-227)
    pred = eng.tensor("pred").cpu().numpy()
    np.testing.assert_array_equal(pred, out["pred"])
    train = am["train"]
    np.testing.assert_array_equal(pred, np.argmax(z1 * (1 - train) + z2 * train, axis=1))
    assert (pred != np.argmax(z1 + z2, axis=1)).any()               # the mixed-mask rule is not the argmax of the sum
    for n in eng.train_names:
        if n.startswith(("tuned_q_linear_l/", "tuned_joint_fc/")):
            assert not eng.grads[n].any(), n                        # no path to the loss (:216-217 reads `joint`)
    before = {n: eng.params[n].clone() for n in eng.params if n.startswith(("tuned_q_linear_l/", "tuned_joint_fc/", "joint_fc/"))}
    st = O.new_opt_state()
    pp = {k: v.copy() for k, v in p.items()}
    for it in range(2):
        run_engine(eng, batch, masks, lr=1e-3)
        loss, report, out, mid, grads, norm = O.train_step(pp, batch, table, nbox, am, masks, st, 1e-3, mt)
        assert abs(float(eng.norm_sq[0]) ** 0.5 - norm) <= 1e-3 * norm
        assert abs(eng.report()["answer_train_loss"] - loss) <= 2e-4 * max(1, abs(loss))
    for n, v in before.items():
        assert torch.equal(eng.params[n], v), n                      # frozen, or trainable with a zero gradient: unchanged
    got = eng.params["TunedWordWeightAnswer/fc/weights"].cpu().numpy()
    assert np.abs(got - p["TunedWordWeightAnswer/fc/weights"]).max() > 1e-4
    assert np.mean(np.abs(got - pp["TunedWordWeightAnswer/fc/weights"]) > 1e-4) < 0.02


def test_vqa_all_row_minimum_substitution_on_the_gpu():
    """model_type 6 (vqa/model_vlmap_answer_vqa_all.py:192-194, 234-244): unknown answers' fixed logits at the row minimum
    (bit for bit the raw logits' minimum), the summed logits, their argmax, and the frozen head's bias gradient -- which
    exists only through the minimum -- against the oracle."""
    mt, dims, B, R, T, N = "vlmap_answer_vqa_all", MED, 24, 36, 14, 32
    p, table, nbox, batch, am, masks = make_case(71, mt, B, R, T, N, dims)
    eng = make_engine(mt, p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    raw = eng.tensor("logit_raw").view(B, -1).cpu().numpy()
    fixed = eng.tensor("logit_fixed").view(B, -1).cpu().numpy()
    ex = am["exist"]
    assert (ex == 0).any()
    np.testing.assert_array_equal(fixed, np.where(ex > 0, raw, raw.min(axis=1, keepdims=True)))
    np.testing.assert_array_equal(eng.tensor("rowmin").cpu().numpy(), raw.min(axis=1))
    z = eng.tensor("logit").view(B, -1).cpu().numpy()
    np.testing.assert_array_equal(z, fixed + eng.tensor("logit_tuned").view(B, -1).cpu().numpy())
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), z.argmax(1))
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am), to64(masks), mt)
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])
    # d loss / d raw fixed logits after the in-place backward of the substitution
    grads, _ = O.backward(to64(p), to64(batch), to64(am), to64(masks), tape, mt)
    dl = eng.tensor("dlogit").view(B, -1).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(dl.sum(0), grads["WordWeightAnswer/fc/biases"], rtol=0,
                               atol=5e-4 * np.abs(grads["WordWeightAnswer/fc/biases"]).max())


def test_train_steps_match_oracle_f32():
    dims, B, R, T, N = MED, 32, 36, 14, 64
    p, table, nbox, batch, am, masks = make_case(22, "vlmap_answer", B, R, T, N, dims)
    eng = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
    frozen_before = {n: eng.params[n].clone() for n in eng.frozen_names}
    st = O.new_opt_state()
    for it in range(3):
        run_engine(eng, batch, masks, lr=1e-3)
        loss, report, out, mid, grads, norm = O.train_step(p, batch, table, nbox, am, masks, st, 1e-3)
        assert abs(float(eng.norm_sq[0]) ** 0.5 - norm) <= 1e-3 * norm
        assert abs(eng.report()["answer_train_loss"] - loss) <= 2e-4 * max(1, abs(loss))
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            continue       # exact-zero gradient -> Adam amplifies rounding noise (see tests/test_oracle_crosscheck.py)
        got = eng.params[n].cpu().numpy()
        # Adam moves each weight by ~lr per step; allow 15 % of the 3-step travel on sign-ambiguous entries
        assert np.abs(got - p[n]).max() <= 4.5e-4 + 1e-4 * np.abs(p[n]).max(), n
        assert np.mean(np.abs(got - p[n]) > 1e-4) < 0.02, n
    for n in eng.frozen_names:
        assert torch.equal(eng.params[n], frozen_before[n])                        # filter_train_vars honoured


def test_untrained_head_known_answer():
    # WordWeightAnswer with word_weight_dir=None: weights 0, bias -100 -> logits -100, pred 0
    dims, B, R, T, N = MED, 8, 36, 14, 16
    p, table, nbox, batch, am, masks = make_case(23, "vlmap_answer", B, R, T, N, dims, head="untrained")
    eng = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    z = eng.tensor("logit").cpu().numpy()
    assert np.all(z == -100.0)
    assert np.all(eng.tensor("pred").cpu().numpy() == 0)
    tgt = batch["answer_target"].astype(np.float64)
    want = (tgt * 100.0).sum(1).mean() + dims["A"] * np.log1p(np.exp(-100.0))
    assert abs(eng.report()["answer_report_loss"] - want) <= 1e-5 * want


def test_ragged_lengths_including_zero_and_short_boxes():
    dims, B, R, T, N = MED, 16, 36, 14, 32
    p, table, nbox, batch, am, masks = make_case(24, "vlmap_answer", B, R, T, N, dims)
    batch["q_intseq_len"][:3] = [0, 1, 14]
    batch["q_intseq"][0, :] = 0
    nbox[batch["image_idx"][1]] = 1
    eng = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am),
                                             to64(masks))
    h = eng.tensor("condition").view(B, dims["H"]).cpu().numpy()
    assert np.all(h[0] == 0)                                                        # len 0 -> zero state
    assert np.abs(h - mid["condition"]).max() < 1e-5
    att = eng.tensor("att_score").view(B, R).cpu().numpy()
    np.testing.assert_array_equal(att[1], np.eye(R, dtype=np.float32)[0])           # one box -> [1,0,...]
    assert np.abs(eng.tensor("logit").view(B, -1).cpu().numpy() - mid["logit"]).max() < 1e-3


@pytest.mark.parametrize("model_type", ["vlmap_answer", "standard"])
def test_length_sorted_live_recurrence_matches_masked_recurrence_and_oracle(model_type):
    """Rows sorted by length + live_rows: the recurrence runs on the still-running prefix only; forward, gradients
    and the Adam step must equal the masked (every row, every step) recurrence on the same batch, and the oracle."""
    from vqa_transfer_externaldata_amd import input_ops_vqa as io
    dims, B, R, T, N = MED, 70, 36, 14, 32
    p, table, nbox, batch, am, masks = make_case(41, model_type, B, R, T, N, dims)
    rng = np.random.default_rng(5)
    batch["q_intseq_len"] = rng.integers(0, T + 1, size=B).astype(np.int32)
    batch["q_intseq_len"][:4] = [0, 0, T, 1]
    sb = io.sort_by_length(batch)
    order = sb.pop("sort_order")
    live = sb["live_rows"]
    assert live[0] == (batch["q_intseq_len"] > 0).sum() and live[-1] == (batch["q_intseq_len"] == T).sum()
    smasks = {"att": masks["att"][order], "joint": masks["joint"][order]}
    plain = {k: v for k, v in sb.items() if k != "live_rows"}

    def run(b):
        eng = make_engine(model_type, p, table, nbox, am, B, R, T, dims)
        ka, kj = dev(smasks["att"].astype(np.uint8)), dev(smasks["joint"].astype(np.uint8))
        db = dev_batch(plain)
        if "live_rows" in b:
            db["live_rows"] = b["live_rows"]
        eng.forward(db, ka, kj, want_dz=True)
        eng.backward()
        torch.cuda.synchronize()
        return eng

    e_mask, e_live = run(plain), run(sb)
    for k in ("condition", "logit", "att_score"):
        assert torch.equal(e_mask.tensor(k), e_live.tensor(k)), k
    emb = e_mask.embed_floats
    assert torch.equal(e_mask.grad_flat[emb:], e_live.grad_flat[emb:])           # bit for bit (embedding: float atomics)
    assert (e_mask.grad_flat[:emb] - e_live.grad_flat[:emb]).abs().max() <= 1e-6 * e_mask.grad_flat[:emb].abs().max() + 1e-12
    hs = e_live.tensor("hs").view(T + 1, B, dims["H"])
    assert torch.equal(hs, e_mask.tensor("hs").view(T + 1, B, dims["H"]))        # finished rows carry their state
    loss, report, out, mid, tape = O.forward(to64(p), to64(plain), table.astype(np.float64), nbox, to64(am),
                                             to64(smasks), model_type)
    grads, dx = O.backward(to64(p), to64(plain), to64(am), to64(smasks), tape, model_type)
    assert np.abs(e_live.tensor("logit").view(B, -1).cpu().numpy() - mid["logit"]).max() < 1e-3
    for n in e_live.train_names:
        if not n.endswith("score/fc/biases"):
            grad_close(e_live.grads[n], grads[n], n)


def test_full_size_properties_bs512():
    """BASELINE config 2 sizes: size-independent properties instead of an oracle run."""
    dims = dict(Vq=16384, W=300, D=2048, H=1024, A=3000)
    B, R, T, N = 512, 36, 14, 1024
    p, table, nbox, batch, am, masks = make_case(25, "vlmap_answer", B, R, T, N, dims, full_boxes=True)
    eng = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    att = eng.tensor("att_score").view(B, R)
    assert torch.allclose(att.sum(1), torch.ones(B, device="cuda"), atol=1e-5)       # softmax rows sum to 1
    assert bool((att >= 0).all())
    z = eng.tensor("logit").view(B, dims["A"])
    assert torch.equal(eng.tensor("pred").long(), z.argmax(1))                       # argmax consistent
    assert bool(torch.isfinite(eng.grad_flat).all())
    # sample independence (LN is per sample): a sub-batch of the first 8 samples gives the same logits
    sub = {k: v[:8].copy() for k, v in batch.items()}
    eng8 = make_engine("vlmap_answer", p, table, nbox, am, 8, R, T, dims)
    m8 = {"att": masks["att"][:8], "joint": masks["joint"][:8]}
    eng8.forward(dev_batch(sub), dev(m8["att"].astype(np.uint8)), dev(m8["joint"].astype(np.uint8)))
    torch.cuda.synchronize()
    assert (eng8.tensor("logit").view(8, -1) - z[:8]).abs().max() < 1e-4
    # linearity of the backward in dlogit: gradient of the score weights scales with 1/global_batch
    g1 = eng.grads["v_linear_v/fc/weights"].clone()
    eng2 = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims, global_batch=2 * B)
    run_engine(eng2, batch, masks)
    assert (eng2.grads["v_linear_v/fc/weights"] * 2 - g1).abs().max() <= 1e-5 * g1.abs().max()
    # and the embedding scatter-add preserves mass: sum(dE) == sum(dx)
    dE, dx = eng.grads["LearnGloVe/embed_map"], eng.tensor("dx_embed")
    assert abs(float(dE.double().sum()) - float(dx.double().sum())) <= 1e-6 * float(dx.double().abs().sum()) + 1e-9
    # ... and row by row at the bench's occupancy (Vq 16384, 512 x 14 tokens): dE[v] = sum of the dx rows of the live
    # positions holding token v, in float64, for the atomic form and the atomic-free (deterministic) one
    W = dims["W"]
    tok, ln = batch["q_intseq"], batch["q_intseq_len"]
    dxh = dx.view(T, B, W).cpu().numpy().astype(np.float64)
    want = np.zeros((dims["Vq"], W))
    live = np.arange(T)[:, None] < ln[None, :]                                       # [T, B]
    np.add.at(want, tok.T[live], dxh[live])
    got = dE.cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale, np.abs(got - want).max() / scale
    assert not got[np.setdiff1d(np.arange(dims["Vq"]), tok.T[live])].any()          # untouched rows stay zero
    eng_det = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims, deterministic=True)
    run_engine(eng_det, batch, masks)
    got_det = eng_det.grads["LearnGloVe/embed_map"].cpu().numpy()
    assert np.abs(got_det - want).max() <= 2e-6 * scale
    run_engine(eng_det, batch, masks)
    assert np.array_equal(eng_det.grads["LearnGloVe/embed_map"].cpu().numpy(), got_det)   # run to run bit-identical


def test_gather_from_the_bench_sized_table_is_bit_exact():
    """vqa_gather_features on a table of the bench's size (8192 images x 36 x 2048 = 2.4 GB, built on the device):
    every gathered row equals the table row bit for bit, box counts included, clamping at both ends like the host"""
    from vqa_transfer_externaldata_amd import ops
    N, R, D, B = 8192, 36, 2048, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    table = torch.rand(N, R, D, device="cuda", generator=g)
    nbox = torch.randint(1, R + 1, (N,), dtype=torch.int32, device="cuda", generator=g)
    idx = torch.randint(0, N, (B,), dtype=torch.int64, device="cuda", generator=g)
    idx[0], idx[1], idx[2] = 0, N - 1, N - 1
    V, nb = ops.gather_features(table, nbox, idx)
    assert torch.equal(V, table[idx]) and torch.equal(nb, nbox[idx])


@pytest.mark.parametrize("recurrence", ["weight_stationary", "per_step"])
def test_full_size_bs512_forward_matches_oracle_f64(recurrence):
    """BASELINE config 2 (bs 512, full dims) forward against the float64 oracle: logits within 1e-3
    (north_star), argmax bit-exact, report scalars, and every trainable gradient (the big-tile GEMM paths) -- with the
    GRU recurrence as one weight-stationary launch per direction (the shipped path at this size, csrc/gru_ws.hip) and as
    the per-step kernels (what other shapes, length-sorted batches and RCCL runs use)."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    dims = dict(Vq=4096, W=300, D=2048, H=1024, A=3000)
    B, R, T, N = 512, 36, 14, 256
    p, table, nbox, batch, am, masks = make_case(27, "vlmap_answer", B, R, T, N, dims, full_boxes=True)
    eng = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
    lib.vqa_gru_ws_set_mode(3 if recurrence == "weight_stationary" else 0)
    try:
        if recurrence == "weight_stationary" and lib.vqa_gru_ws_bwd_supported(T, B, dims["H"]) != 1:
            pytest.skip("the weight-stationary recurrence does not apply on this device")
        run_engine(eng, batch, masks)
    finally:
        lib.vqa_gru_ws_set_mode(-1)
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am),
                                             to64(masks), "vlmap_answer")
    z = eng.tensor("logit").view(B, dims["A"]).cpu().numpy()
    assert np.abs(z - mid["logit"]).max() <= 1e-3, np.abs(z - mid["logit"]).max()
    np.testing.assert_array_equal(eng.tensor("pred").cpu().numpy(), out["pred"])
    rep = eng.report()
    for k in O.REPORT_KEYS:
        assert abs(rep[k] - report[k]) <= 1e-4 * max(1.0, abs(report[k])), (k, rep[k], report[k])
    grads, dx = O.backward(to64(p), to64(batch), to64(am), to64(masks), tape, "vlmap_answer")
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            continue                      # analytically zero: rounding noise on both sides
        grad_close(eng.grads[n], grads[n], n)


@pytest.mark.parametrize("gru_cfg", [4, 7, 8, 9, 10, 11, 12, 13, 16, 17, 18, 20, 21])
def test_fused_gru_tile_configs_match_oracle(gru_cfg):
    """Every tile configuration of the fused GRU-step GEMMs (in-block split-k 1/2/4) gives the oracle's
    final state and GRU gradients."""
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    dims, B, R, T, N = MED, 40, 36, 14, 64
    p, table, nbox, batch, am, masks = make_case(31, "vlmap_answer", B, R, T, N, dims)
    try:
        assert lib.vqa_gemm_set_gru_config(gru_cfg) == 0
        eng = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
        run_engine(eng, batch, masks)
    finally:
        lib.vqa_gemm_set_gru_config(-1)       # back to the library defaults
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am),
                                             to64(masks))
    grads, dx = O.backward(to64(p), to64(batch), to64(am), to64(masks), tape)
    h = eng.tensor("condition").view(B, dims["H"]).cpu().numpy()
    assert np.abs(h - mid["condition"]).max() < 1e-5
    for n in eng.train_names:
        if n.startswith("encode_L") or n.startswith("LearnGloVe"):
            grad_close(eng.grads[n], grads[n], n)


def test_repeated_runs_are_bitwise_identical():
    """Run-to-run bitwise equality (a race between streams or an unordered reduction would show; run the
    suite with VQA_HOT_OVERLAP=1 to screen the optional side-stream overlap the same way)."""
    dims, B, R, T, N = MED, 64, 36, 14, 64
    p, table, nbox, batch, am, masks = make_case(32, "standard", B, R, T, N, dims)
    eng = make_engine("standard", p, table, nbox, am, B, R, T, dims)
    run_engine(eng, batch, masks)
    g1, z1 = eng.grad_flat.clone(), eng.tensor("logit").clone()
    for _ in range(5):
        run_engine(eng, batch, masks)
        assert torch.equal(z1, eng.tensor("logit"))
        emb = eng.embed_floats                     # the embedding scatter-add uses float atomics (order varies)
        assert torch.equal(g1[emb:], eng.grad_flat[emb:])
    # deterministic mode is PER ENGINE (vqa_dims_t.flags): a second engine in the same process with the atomic-free
    # scatter-add reproduces its WHOLE gradient buffer bit for bit, while the first keeps using atomics beside it
    eng_det = make_engine("standard", p, table, nbox, am, B, R, T, dims, deterministic=True)
    run_engine(eng_det, batch, masks)
    g2 = eng_det.grad_flat.clone()
    assert torch.equal(g1[emb:], g2[emb:])
    assert (g1[:emb] - g2[:emb]).abs().max() <= 1e-5 * g1[:emb].abs().max()
    for _ in range(3):
        run_engine(eng, batch, masks)                  # interleaved: must not disturb the other engine's setting
        run_engine(eng_det, batch, masks)
        assert torch.equal(g2, eng_det.grad_flat)
        assert torch.equal(g1[emb:], eng.grad_flat[emb:])


def test_phased_backward_equals_monolithic_and_bucket_layout():
    """vqa_fusion_backward_phases(1|2|4|8 in order) == vqa_fusion_backward, and the flat layout puts the
    buckets where FusionEngine.backward(reducer=...) slices them."""
    dims, B, R, T, N = MED, 48, 36, 14, 64
    for mt in ("vlmap_answer", "standard"):
        p, table, nbox, batch, am, masks = make_case(41, mt, B, R, T, N, dims)
        eng = make_engine(mt, p, table, nbox, am, B, R, T, dims)
        run_engine(eng, batch, masks)
        ref = eng.grad_flat.clone()
        seen = []

        class Rec:
            def start(self, bucket):
                seen.append((bucket.data_ptr() - eng.grad_flat.data_ptr()) // 4)
                seen.append(bucket.numel())

            def finish(self):
                pass
        eng.grad_flat.fill_(float("nan"))
        eng.backward(reducer=Rec())
        torch.cuda.synchronize()
        emb = eng.embed_floats
        for n in eng.train_names[1:]:                                            # dense grads bit-identical
            off, cnt = eng._train_tab[n]
            assert torch.equal(ref[off:off + cnt], eng.grad_flat[off:off + cnt]), n
        assert torch.equal(ref[eng.n_train], eng.grad_flat[eng.n_train])         # slice sum of squares
        cnt = eng._train_tab[eng.train_names[0]][1]
        assert torch.allclose(ref[:cnt], eng.grad_flat[:cnt], rtol=1e-4, atol=1e-7)   # atomics: order varies
        # buckets: rest, embed, tail, GRU gates, GRU candidate -- disjoint and covering the whole buffer
        offs, lens = seen[0::2], seen[1::2]
        assert offs == [eng.gru_end, 0, eng.n_train, eng.gru_mid, emb] and sum(lens) == eng.grad_flat.numel()
        assert emb < eng.gru_mid < eng.gru_end
        gru_names = [n for n in eng.train_names if n.startswith("encode_L/")]
        for n in gru_names:
            off = eng._train_tab[n][0]
            assert emb <= off < eng.gru_end


def test_standard_word2vec_train_steps_and_model_class(tmp_path):
    """vqa/model_standard_word2vec.py: three clip+Adam steps against the oracle; the constant answer-GloVe matrix never
    moves and is not a variable; the Model mirror builds it from GloVe with the mean-of-words rule for phrases."""
    dims, B, R, T, N = MED, 32, 36, 14, 64
    p, table, nbox, batch, am, masks = make_case(52, "standard_word2vec", B, R, T, N, dims)
    eng = make_engine("standard_word2vec", p, table, nbox, am, B, R, T, dims)
    assert O.OUTPUT_GLOVE not in eng.params and eng.shapes["reasoning/classifier/fc/weights"] == (2 * dims["H"], dims["W"])
    g0 = eng.answer_glove.clone()
    st = O.new_opt_state()
    for it in range(3):
        run_engine(eng, batch, masks, lr=1e-3)
        loss, report, out, mid, grads, norm = O.train_step(p, batch, table, nbox, am, masks, st, 1e-3, "standard_word2vec")
        assert abs(float(eng.norm_sq[0]) ** 0.5 - norm) <= 1e-3 * norm
        rep = eng.report()
        assert abs(rep["answer_train_loss"] - loss) <= 2e-4 * max(1, abs(loss))
        assert abs(rep["answer_report_loss"] - report["answer_report_loss"]) <= 2e-4 * max(1, report["answer_report_loss"])
        assert rep["answer_train_loss"] < rep["answer_report_loss"]            # train loss is masked, report loss is not
    for n in eng.train_names:
        if n.endswith("score/fc/biases"):
            continue
        got = eng.params[n].cpu().numpy()
        assert np.abs(got - p[n]).max() <= 4.5e-4 + 1e-4 * np.abs(p[n]).max(), n
    assert torch.equal(eng.answer_glove, g0) and "reasoning/output_glove" not in " ".join(eng.state_dict())
    # Model mirror: GloVe lookup with oov_mean_initialize for multi-word answers
    from vqa_transfer_externaldata_amd import importer, trainer
    from vqa_transfer_externaldata_amd import input_ops_vqa as io
    c = trainer.parse_config(["--batch_size", "8", "--model_type", "standard_word2vec"])
    words = ["w%d" % i for i in range(20)]
    c.vocab = {"vocab": words, "dict": {w: i for i, w in enumerate(words)}}
    answers = ["red", "blue", "traffic light", "dog", ""]
    c.answer_dict = {"vocab": answers, "dict": {a: i for i, a in enumerate(answers)}, "num_train_answer": 3,
                     "is_object": [0, 0, 1, 1, 0], "is_attribute": [1, 1, 0, 0, 0]}
    rng = np.random.default_rng(3)
    gl = ["red", "blue", "traffic", "light", "dog"]
    c.glove = {"dict": {w: i for i, w in enumerate(gl)}, "param": rng.standard_normal((5, 300)).astype(np.float32)}
    c.train_dir, c.tf_record_dir = str(tmp_path / "run"), str(tmp_path / "data")
    feats = {"features": np.maximum(rng.standard_normal((6, 36, 64)), 0).astype(np.float32),
             "spatials": np.zeros((6, 36, 6), np.float32), "normal_boxes": np.zeros((6, 36, 4), np.float32),
             "num_boxes": np.full(6, 36, np.int32), "max_box_num": 36, "vfeat_dim": 64}
    b = next(io.create(8, None, "train", data=io.synthetic_split(8, 6, 20, 5, seed=1)))
    m = importer.get_model_class("standard_word2vec")(b, c, is_train=True, image_features=feats)
    G = m.engine.answer_glove.cpu().numpy()
    np.testing.assert_array_equal(G[:, 0], c.glove["param"][0])
    np.testing.assert_allclose(G[:, 2], (c.glove["param"][2] + c.glove["param"][3]) / 2, rtol=1e-6)   # "traffic light"
    assert np.all(G[:, 4] == 0)                                                                       # "" stays zero
    assert m.output["logit"].shape == (8, 5) and np.all(m.output["logit"][:, 4].cpu().numpy() == 0)
    c.answer_dict["vocab"][1] = "fire hydrant"
    with pytest.raises(Exception, match="Unkown words"):
        importer.get_model_class("standard_word2vec")(b, c, is_train=True, image_features=feats)


@pytest.mark.parametrize("cfg", [("med", MED, 32, 36, 14, 64), ("cfg1_full_dims", FULL, 8, 36, 14, 24)])
def test_fused_feature_gather_equals_gather_pass(cfg):
    """VQA_FLAG_FUSED_GATHER: V_ft = features[image_idx] read inside v_linear_v's GEMM instead of a pass of its own --
    the gathered block left in the workspace is bit-identical, everything downstream agrees to rounding."""
    name, dims, B, R, T, N = cfg
    p, table, nbox, batch, am, masks = make_case(61, "vlmap_answer", B, R, T, N, dims)
    a = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims)
    b = make_engine("vlmap_answer", p, table, nbox, am, B, R, T, dims, fused_gather=True)
    run_engine(a, batch, masks)
    run_engine(b, batch, masks)
    assert torch.equal(a.tensor("V_ft"), b.tensor("V_ft")) and torch.equal(a.tensor("num_V_ft"), b.tensor("num_V_ft"))
    want = table[batch["image_idx"]].astype(np.float32).reshape(-1)
    np.testing.assert_array_equal(b.tensor("V_ft").cpu().numpy(), want)
    for k in MID_KEYS:
        x, y = a.tensor(k).cpu().numpy(), b.tensor(k).cpu().numpy()
        assert np.abs(x - y).max() <= 1e-5 * max(1.0, np.abs(x).max()), k
    assert torch.equal(a.tensor("pred"), b.tensor("pred"))
    ga, gb = a.grad_flat.cpu().numpy(), b.grad_flat.cpu().numpy()
    assert np.abs(ga - gb).max() <= 1e-5 * np.abs(ga).max()


def test_tuning_switches_of_the_step_keep_parity():
    """The A/B switches of the fused step are read once per process: the two-GEMM form of the x-projection
    (VQA_HOT_XCAT=0: per-kernel GEMMs + column sums for the GRU bias gradients) and the round-1 forward order
    (VQA_HOT_VISUAL_LATE=0), the three-launch form of pooled_linear_l x l_linear_l (VQA_HOT_LN_PAIR=0) and the second packing of
    the x rows in backward (VQA_HOT_REPACK=1) must reproduce the golden vectors (forward, report, every gradient) like the shipped path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQA_HOT_XCAT="0", VQA_HOT_VISUAL_LATE="0", VQA_HOT_LN_PAIR="0", VQA_HOT_REPACK="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_golden.py", "-m", "gpu", "-q", "-x", "-k",
                        "test_hip_matches_fusion_golden", "-p", "no:cacheprovider"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "9 passed" in r.stdout, r.stdout[-500:]       # the nine fusion fixtures


def test_recurrence_time_out_is_raised_where_results_are_fetched():
    """FusionEngine.check_recurrence: the sticky error word of the weight-stationary GRU launches (set when their bounded waits
    give up, csrc/gru_ws.hip) turns report() into an exception instead of numbers; a healthy step leaves it at zero."""
    from vqa_transfer_externaldata_amd import _lib
    dims = dict(Vq=300, W=300, D=128, H=1024, A=40)
    p, table, nbox, batch, am, masks = make_case(91, "vlmap_answer", 8, 4, 5, 16, dims)
    eng = make_engine("vlmap_answer", p, table, nbox, am, 8, 4, 5, dims)
    db = dev_batch(batch)
    ka, kj = eng.make_keep_masks(3, 0)
    eng.train_step(db, ka, kj, 1e-3)
    assert np.isfinite(eng.report()["answer_train_loss"])
    if eng._ws_err_word is False:
        pytest.skip("no weight-stationary recurrence buffer for this shape")
    eng._ws_err_word.fill_(1)
    with pytest.raises(_lib.VqaHotError, match="recurrence timed out"):
        eng.report()
    eng._ws_err_word.zero_()
    eng.train_step(db, ka, kj, 1e-3)
    assert np.isfinite(eng.report()["answer_train_loss"])
