import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_graph import _pair, _eager, MED
def run(mt, steps, between):
    dims, B, R, T, N = MED, 64, 36, 14, 64
    a, b, db = _pair(mt, B, R, T, N, dims, 71)
    out = []
    for step in steps:
        _eager(a, db, 1e-3, 5, step); b.train_step_graph(db, 1e-3, 5, step); torch.cuda.synchronize()
        ok = (torch.equal(a.tensor("report")[:3], b.tensor("report")[:3]), torch.equal(a.grad_flat, b.grad_flat), torch.equal(a.train_flat, b.train_flat))
        masks = torch.equal(a._keep_att, b._keep_att) and torch.equal(a._keep_joint, b._keep_joint) and (not hasattr(a, "_keep_joint2") or torch.equal(a._keep_joint2, b._keep_joint2))
        mids = {k: torch.equal(a.tensor(k), b.tensor(k)) for k in ("v_linear_v", "condition", "att_score", "joint", "logit", "dlogit", "d_joint", "d_pl", "d_ll", "d_pooled", "d_v", "d_qv", "dxp", "dx_embed")}
        bad = [k for k, v in mids.items() if not v]
        out.append((step, ok, masks, bad[:4]))
        if between == "alloc":
            x = torch.empty(1 << 20, device="cuda"); del x
    return out
for mt in ("vlmap_answer_noc",):
    print(mt, "contig", run(mt, range(6), None), flush=True)
    print(mt, "skip3", run(mt, (0, 1, 2, 4, 5), None), flush=True)
    print(mt, "contig+alloc", run(mt, range(6), "alloc"), flush=True)
