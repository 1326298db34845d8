"""Tile-config sweep for the fused GRU recurrence (vqa_gru_seq_fwd / _bwd) at B 512, H 1024, T 14; min of repeats."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib  # noqa: E402

lib = _lib.load()
T, B, H = int(os.environ.get("GRU_T", 14)), int(os.environ.get("GRU_B", 512)), 1024
g = torch.Generator(device="cuda").manual_seed(0)
xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.1
Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.03
Wc = torch.randn(H, H, device="cuda", generator=g) * 0.03
ln = torch.randint(1, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
hs = torch.zeros(T + 1, B, H, device="cuda")
r = torch.empty(T, B, H, device="cuda"); u = torch.empty_like(r); c = torch.empty_like(r); rh = torch.empty_like(r)
dhT0 = torch.randn(B, H, device="cuda", generator=g); dhT = dhT0.clone()
dxp = torch.empty(T, B, 3 * H, device="cuda"); dhs = torch.empty(B, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())


def fwd():
    _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "fwd")


def bwd():
    dhT.copy_(dhT0)
    _lib.check(lib.vqa_gru_seq_bwd(P(dhT), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H, None), "bwd")


def tm(f, n=30):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


cfgs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "11,16,17,18").split(",")]
best = {k: [1e9, 1e9] for k in cfgs}
ref = None
for rep in range(3):
    for k in cfgs:
        _lib.check(lib.vqa_gemm_set_gru_config(k), "cfg")
        tf, tb = tm(fwd), tm(bwd)
        best[k][0] = min(best[k][0], tf); best[k][1] = min(best[k][1], tb)
        if rep == 0:
            out = torch.cat([hs[T].flatten(), dxp.flatten()[::97]]).double()
            if ref is None:
                ref = out
            print("cfg %2d max |diff| vs first cfg: %.3e" % (k, float((out - ref).abs().max())), flush=True)
for k, (tf, tb) in best.items():
    print("gru cfg %2d: fwd %.1f us  bwd %.1f us  (min of 3 x 30)" % (k, tf, tb), flush=True)
_lib.check(lib.vqa_gemm_set_gru_config(-1), "cfg")
print("defaults (fwd 18 / bwd 18): fwd %.1f us  bwd %.1f us" % (tm(fwd), tm(bwd)))
