import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_graph import _pair, _eager, MED
def nan_report(e):
    names = ["condition", "q_L_mean", "q_L_log_sigma_sq", "q_L_mean_noise", "l_linear_l", "joint", "logit", "dlogit", "d_ll", "d_lin", "d_qm", "d_qs", "d_h0", "dxp", "dx_embed", "wx_cat", "xp"]
    bad = [k for k in names if not torch.isfinite(e.tensor(k)).all()]
    return bad, bool(torch.isfinite(e.train_flat).all()), bool(torch.isfinite(e.grad_flat).all()), float(e._g_lr_t[0]) if hasattr(e, "_g_lr_t") else None, int(e._g_step[0]) if hasattr(e, "_g_step") else None, float(e.norm_sq[0])
def run(mt):
    dims, B, R, T, N = MED, 64, 36, 14, 64
    a, b, db = _pair(mt, B, R, T, N, dims, 71)
    for step in range(5):
        _eager(a, db, 1e-3, 5, step); b.train_step_graph(db, 1e-3, 5, step); torch.cuda.synchronize()
    print("after graph steps", nan_report(b), torch.equal(a.train_flat, b.train_flat))
    _eager(a, db, 5e-4, 5, 5); torch.cuda.synchronize()
    _eager(b, db, 5e-4, 5, 5); torch.cuda.synchronize()
    print("after eager on b", nan_report(b), torch.equal(a.train_flat, b.train_flat), torch.equal(a.grad_flat, b.grad_flat), torch.equal(a.tensor("report")[:16], b.tensor("report")[:16]))
    _eager(a, db, 5e-4, 5, 6); torch.cuda.synchronize()
    b.train_step_graph(db, 5e-4, 5, 6); torch.cuda.synchronize()
    print("after graph on b", nan_report(b), torch.equal(a.train_flat, b.train_flat), torch.equal(a.grad_flat, b.grad_flat), torch.equal(a.tensor("report")[:16], b.tensor("report")[:16]))
    d = (a.grad_flat - b.grad_flat).abs()
    tab = b._train_tab
    worst = sorted(((float(d[o:o+c].max()) if torch.isfinite(d[o:o+c]).all() else float("nan"), n) for n, (o, c) in tab.items()), key=lambda x: -1 if x[0] != x[0] else x[0])
    print("grad diffs", [w for w in worst if w[0] != 0][:8])
run("vlmap_answer_full")
run("vlmap_answer_noc")
