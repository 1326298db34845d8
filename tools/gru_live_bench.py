"""Masked recurrence (every row, every step) against the live-prefix recurrence on length-sorted rows."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib
lib = _lib.load()
T, H = 14, 1024
P = lambda t: C.c_void_p(t.data_ptr())
def tm(f, n=20):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B in (512, 2560, 8192):
    g = torch.Generator(device="cuda").manual_seed(0)
    xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.1
    Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.03
    Wc = torch.randn(H, H, device="cuda", generator=g) * 0.03
    lens = np.sort(np.random.default_rng(1).integers(3, T + 1, size=B))[::-1].astype(np.int32).copy()
    live = (lens[None, :] > np.arange(T)[:, None]).sum(1).astype(np.int32)
    ln = torch.from_numpy(lens).cuda()
    hs = torch.zeros(T + 1, B, H, device="cuda"); r = torch.empty(T, B, H, device="cuda"); u = torch.empty_like(r); c = torch.empty_like(r); rh = torch.empty_like(r)
    dhT0 = torch.randn(B, H, device="cuda", generator=g); dhT = dhT0.clone(); dxp = torch.empty(T, B, 3 * H, device="cuda"); dhs = torch.empty(B, H, device="cuda")
    f_mask = lambda: _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "f")
    f_live = lambda: _lib.check(lib.vqa_gru_seq_fwd_live(P(xp), P(Wg), P(Wc), P(ln), live.ctypes.data, P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "fl")
    def b_mask():
        dhT.copy_(dhT0); _lib.check(lib.vqa_gru_seq_bwd(P(dhT), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H, None), "b")
    def b_live():
        dhT.copy_(dhT0); _lib.check(lib.vqa_gru_seq_bwd_live(P(dhT), P(Wg), P(Wc), P(ln), live.ctypes.data, P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H, None), "bl")
    print("B %5d (lengths U{3..14}, mean live fraction %.2f): forward masked %.0f us / live %.0f us; backward masked %.0f us / live %.0f us"
          % (B, live.sum() / (T * B), tm(f_mask), tm(f_live), tm(b_mask), tm(b_live)), flush=True)
