"""Whole-step hipGraph replay (FusionEngine.train_step_graph, csrc/graph.hip) against the eager step: same kernels, same
arithmetic -- parameters, Adam moments and reports equal after several steps in deterministic mode; shapes and
variants get their own graphs; the device-side Adam step count follows the host's."""
import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O
from tests.gpu_util import dev, dev_batch, make_case, make_engine

pytestmark = pytest.mark.gpu

MED = dict(Vq=500, W=300, D=256, H=128, A=300)


def _pair(model_type, B, R, T, N, dims, seed, **kw):
    p, table, nbox, batch, am, masks = make_case(seed, model_type, B, R, T, N, dims, num_marginal=6)
    a = make_engine(model_type, p, table, nbox, am, B, R, T, dims, deterministic=True, **kw)
    b = make_engine(model_type, p, table, nbox, am, B, R, T, dims, deterministic=True, **kw)
    return a, b, dev_batch(batch)


def _eager(eng, db, lr, seed, step):
    ka, kj = eng.make_keep_masks(seed, step)
    kw = {}
    if eng.model_type in ("vlmap_answer_noc", "vlmap_answer_nocarch"):
        kw["keep_joint2"] = eng.make_keep_mask_joint2(seed, step)
    if eng.model_type == "vlmap_answer_full":
        kw["noise"] = eng.make_noise(seed, step)
    if eng.model_type == "vlmap_answer_ent":
        kw["keep_tile"] = eng.make_keep_mask_tile(seed, step)
    eng.train_step(db, ka, kj, lr, **kw)


def _close(x, y, what):
    """same kernels and the same arithmetic, so the replay is bitwise equal in every run observed; the bar here is the
    optimiser's resolution, so that a reordered atomic in some future runtime is not read as a wrong graph"""
    d = float((x - y).abs().max())
    assert d <= 1e-6 * max(1.0, float(x.abs().max())), (what, d)


@pytest.mark.parametrize("model_type", ["vlmap_answer", "standard", "vlmap_answer_noc", "vlmap_answer_full", "vlmap_answer_ent"])
def test_graph_replay_equals_eager_steps(model_type):
    dims, B, R, T, N = MED, 64, 36, 14, 64
    kw = {"num_marginal": 6} if model_type == "vlmap_answer_ent" else {}
    a, b, db = _pair(model_type, B, R, T, N, dims, 71, **kw)
    lrs = [1e-3] * 5 + [5e-4] * 2      # the changed rate reaches the replayed Adam: it lives in device memory, no re-capture
    # the eager engine runs all its steps first, the replayed one after it (no eager step between two replays)
    want = []
    for step, lr in enumerate(lrs):
        _eager(a, db, lr, 5, step)
        torch.cuda.synchronize()
        want.append((a.tensor("report")[:16].clone(), a.train_flat.clone()))
    nodes = None
    for step, lr in enumerate(lrs):
        nodes = b.train_step_graph(db, lr, 5, step)
        torch.cuda.synchronize()
        _close(want[step][0], b.tensor("report")[:16], ("report", step))
        _close(want[step][1], b.train_flat, ("params", step))
    assert nodes is not None and nodes > 60                      # one graph holds the whole step
    assert a.step_count == b.step_count == 7 and int(b._g_step.item()) == 7
    _close(a.m_flat, b.m_flat, "m")
    _close(a.v_flat, b.v_flat, "v")
    _close(a.grad_flat, b.grad_flat, "grad")


def test_graphs_per_shape_and_live_rows():
    """a second batch shape captures its own graph after resize(); length-sorted batches (live_rows) key their own"""
    from vqa_transfer_externaldata_amd import input_ops_vqa
    dims, R, N = MED, 36, 64
    p, table, nbox, batch, am, masks = make_case(72, "vlmap_answer", 48, R, 14, N, dims)
    a = make_engine("vlmap_answer", p, table, nbox, am, 48, R, 14, dims, deterministic=True)
    b = make_engine("vlmap_answer", p, table, nbox, am, 48, R, 14, dims, deterministic=True)
    sb = input_ops_vqa.sort_by_length({k: v for k, v in batch.items()})
    full = {k: (dev(v) if k != "live_rows" else v) for k, v in sb.items()}
    full["answer_target"] = full["answer_target"].float()
    short = {k: (v[:20].contiguous() if torch.is_tensor(v) else v) for k, v in full.items() if k != "live_rows"}
    short["q_intseq"] = short["q_intseq"][:, :9].contiguous()
    short["q_intseq_len"] = short["q_intseq_len"].clamp(max=9)
    seq = [(full, 48, 14), (full, 48, 14), (short, 20, 9), (short, 20, 9), (full, 48, 14)]
    want = []
    for step, (db, B, T) in enumerate(seq):
        a.resize(B, T)
        _eager(a, db, 1e-3, 3, step)
        torch.cuda.synchronize()
        want.append(a.tensor("report")[:13].clone())
    for step, (db, B, T) in enumerate(seq):
        b.resize(B, T)
        b.train_step_graph(db, 1e-3, 3, step)
        torch.cuda.synchronize()
        _close(want[step], b.tensor("report")[:13], ("report", step))
    _close(a.train_flat, b.train_flat, "params")


def test_graph_api_rejects_bad_arguments():
    import ctypes as C
    from vqa_transfer_externaldata_amd import _lib
    lib = _lib.load()
    assert lib.vqa_graph_capture_begin(None) == -1 and lib.vqa_graph_launch(None, None) == -1
    assert lib.vqa_graph_destroy(None) == 0 and lib.vqa_stream_is_capturing(None) == 0
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    assert lib.vqa_graph_capture_begin(sp) == 0 and lib.vqa_stream_is_capturing(sp) == 1
    x = torch.zeros(1024, device="cuda")
    with torch.cuda.stream(s):
        assert lib.vqa_tanh_fwd(C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 1024, sp) == 0
    ex, n = C.c_void_p(), C.c_int()
    assert lib.vqa_graph_capture_end(sp, C.byref(ex), C.byref(n)) == 0 and n.value == 1 and lib.vqa_stream_is_capturing(sp) == 0
    x.fill_(1.0)
    torch.cuda.synchronize()
    assert lib.vqa_graph_launch(ex, sp) == 0 and lib.vqa_graph_launch(ex, sp) == 0
    torch.cuda.synchronize()
    assert abs(float(x[0]) - np.tanh(np.tanh(1.0))) < 1e-6
    assert lib.vqa_graph_destroy(ex) == 0
