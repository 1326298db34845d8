"""Where the time of ONE fused GRU-step launch goes (a -DVQA_DBG_STAMPS build, tools/dbg/libvqahot_GSTAMPS.so): thread 0 of
every workgroup records the 100 MHz real-time counter at start / first tile in LDS / k loop done / partial tiles exchanged /
epilogue issued.  Runs the forward (or backward) recurrence once per config and prints, for the LAST launch of each epilogue
kind, the spread of those points over the workgroups, relative to the first workgroup's start."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vqa_transfer_externaldata_amd import _lib  # noqa: E402

_lib._LIB_PATH = os.path.join(ROOT, "tools", "dbg", "libvqahot_GSTAMPS.so")
lib = _lib.load()
T, B, H = 14, int(os.environ.get("GRU_B", 512)), 1024
g = torch.Generator(device="cuda").manual_seed(0)
xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.1
Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.03
Wc = torch.randn(H, H, device="cuda", generator=g) * 0.03
ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
hs = torch.zeros(T + 1, B, H, device="cuda")
r = torch.empty(T, B, H, device="cuda"); u = torch.empty_like(r); c = torch.empty_like(r); rh = torch.empty_like(r)
dhT = torch.randn(B, H, device="cuda", generator=g)
dxp = torch.empty(T, B, 3 * H, device="cuda"); dhs = torch.empty(B, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
st = torch.zeros(5, 4096, 8, dtype=torch.int64, device="cuda")
lib.vqa_gemm_dbg_stamps.argtypes = [C.c_void_p]
_lib.check(lib.vqa_gemm_dbg_stamps(P(st)), "stamps")
names = {1: "gates", 2: "candidate", 3: "bwd rh", 4: "bwd dh"}
for cfg in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,18").split(",")]:
    _lib.check(lib.vqa_gemm_set_gru_config(cfg), "cfg")
    for _ in range(3):
        st.zero_()
        _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "fwd")
        _lib.check(lib.vqa_gru_seq_bwd(P(dhT.clone()), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H, None), "bwd")
        torch.cuda.synchronize()
    s = st.cpu().numpy()
    print("== gru config %d (B %d): times in us relative to the first workgroup's start; median [min .. max] over workgroups" % (cfg, B))
    for kind in (1, 2, 3, 4):
        a = s[kind]
        a = a[a[:, 0] != 0]
        if len(a) == 0:
            continue
        t0 = a[:, 0].min()
        rel = (a[:, :5] - t0) / 100.0
        pts = ["start", "first tile in LDS", "k loop done", "tiles exchanged", "epilogue issued"]
        print("  %-9s %4d workgroups: " % (names[kind], len(a)) + " | ".join(
            "%s %.1f [%.1f .. %.1f]" % (pts[i], np.median(rel[:, i]), rel[:, i].min(), rel[:, i].max()) for i in range(5)))
        d = np.diff(rel, axis=1)
        print("            per workgroup (median): prologue %.2f, k loop %.2f, exchange %.2f, epilogue %.2f; whole launch first start -> last end %.1f us" % (
            np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(d[:, 3]), rel[:, 4].max()))
        clk = (a[:, 6] - a[:, 5]) / np.maximum(a[:, 4] - a[:, 0], 1) * 0.1
        print("            shader clock over the workgroup's life: median %.2f GHz [%.2f .. %.2f]" % (np.median(clk), clk.min(), clk.max()))
_lib.check(lib.vqa_gemm_set_gru_config(-1), "cfg")
# the roofline GEMM (v_linear_v forward, plain epilogue: slot 0), 20 launches back to back
from vqa_transfer_externaldata_amd import ops  # noqa: E402
A = torch.randn(18432, 2048, device="cuda", generator=g).relu_()
Bm = torch.randn(2048, 1024, device="cuda", generator=g) * 0.03
out = torch.empty(18432, 1024, device="cuda")
for _ in range(20):
    ops.gemm(A, Bm, out=out)
torch.cuda.synchronize()
a = st.cpu().numpy()[0]
a = a[a[:, 0] != 0]
clk = (a[:, 6] - a[:, 5]) / np.maximum(a[:, 4] - a[:, 0], 1) * 0.1
print("roofline GEMM (M 18432 N 1024 K 2048), 20th launch: %d workgroups, first start -> last end %.1f us, shader clock median %.2f GHz [%.2f .. %.2f]" % (
    len(a), (a[:, 4].max() - a[:, 0].min()) / 100.0, np.median(clk), clk.min(), clk.max()))
