"""Diagnostic: cfg-5 full-size gradient error of the HIP path vs the f64 autograd oracle, next to the error a plain
float32 CPU autograd run of the same model makes against the same f64 numbers."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pretrain_oracle as PO
from tests.test_gpu_pretrain import _setup
cfg = dict(B=512, n=5, R=36, D=2048, H=1024, L=10, W=300, Vq=5000, n_ws=2000, A=4000)
PT, eng, p, batch, masks, db, dm = _setup(9, **cfg)
if os.environ.get("SORT", "1") == "1":
    db.update({k: v for k, v in PT.add_length_sort(dict(batch)).items() if k.endswith("/sort")})
eng.forward(db, dm); eng.backward(); torch.cuda.synchronize()
to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
_, _, g64, _ = PO.torch_loss_and_grads(to64(p), to64(batch), to64(masks), cfg["n"])
_, _, g32, _ = PO.torch_loss_and_grads(p, batch, masks, cfg["n"], dtype=torch.float32)
for name in eng.train_names:
    g = eng.grads[name].cpu().numpy().astype(np.float64)
    sc = max(np.abs(g64[name]).max(), 1e-30)
    e_hip = np.abs(g - g64[name]); e_cpu = np.abs(g32[name].astype(np.float64) - g64[name])
    i = np.unravel_index(e_hip.argmax(), e_hip.shape)
    print("%-52s scale %.3e  hip err/scale %.2e  cpu-f32 err/scale %.2e  argmax %s  fro-rel hip %.2e cpu %.2e" % (
        name, sc, e_hip.max() / sc, e_cpu.max() / sc, i, np.linalg.norm(e_hip) / max(np.linalg.norm(g64[name]), 1e-30),
        np.linalg.norm(e_cpu) / max(np.linalg.norm(g64[name]), 1e-30)))
