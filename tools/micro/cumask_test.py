"""Feasibility of CU-partitioned streams (hipExtStreamCreateWithCUMask): time the roofline GEMM on a stream that owns the
first n CU bits (bit i = XCC i % 8, CU slot i / 8 on multi-XCC parts, so a prefix of the mask keeps all eight XCDs), alone and
beside the GRU recurrence on the complementary mask."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vqa_transfer_externaldata_amd import _lib, ops  # noqa: E402

hip = C.CDLL("libamdhip64.so")
lib = _lib.load()


def masked_stream(lo, hi):
    words = (C.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(s.value)


g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(18432, 2048, device="cuda", generator=g).relu_()
B = torch.randn(2048, 1024, device="cuda", generator=g) * 0.03
out = torch.empty(18432, 1024, device="cuda")
T, Bz, H = 14, 512, 1024
xp = torch.randn(T, Bz, 3 * H, device="cuda", generator=g) * 0.1
Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.03
Wc = torch.randn(H, H, device="cuda", generator=g) * 0.03
ln = torch.full((Bz,), T, dtype=torch.int32, device="cuda")
hs = torch.zeros(T + 1, Bz, H, device="cuda")
r = torch.empty(T, Bz, H, device="cuda"); u = torch.empty_like(r); c = torch.empty_like(r); rh = torch.empty_like(r)
P = lambda t: C.c_void_p(t.data_ptr())


def gru(st):
    _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, Bz, H, C.c_void_p(st.cuda_stream)), "gru")


def gemm(st):
    with torch.cuda.stream(st):
        ops.gemm(A, B, out=out)


def tm(fns, n=10):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        main = torch.cuda.current_stream()
        e0.record(main)
        ends = []
        for st, f in fns:
            st.wait_event(e0)
            f(st)
            e = torch.cuda.Event(); e.record(st); ends.append(e)
        for e in ends:
            main.wait_event(e)
        e1.record(main)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


full = masked_stream(0, 256)
print("GEMM alone, full mask: %.1f us; GRU forward alone, full mask: %.1f us" % (tm([(full, gemm)]), tm([(full, gru)])), flush=True)
for cfg in (16, 18):
    _lib.check(lib.vqa_gemm_set_gru_config(cfg), "cfg")
    for n in (128, 144, 160, 176):
        sa, sb = masked_stream(0, n), masked_stream(n, 256)
        ta, tb = tm([(sa, gemm)]), tm([(sb, gru)])
        both = tm([(sa, gemm), (sb, gru)])
        print("gru cfg %d, GEMM on %3d CU bits: alone %.1f us | GRU on the other %3d: alone %.1f us | both together %.1f us (serial on the full chip: see first line)" % (
            cfg, n, ta, 256 - n, tb, both), flush=True)
_lib.check(lib.vqa_gemm_set_gru_config(-1), "cfg")
un = torch.cuda.Stream(), torch.cuda.Stream()
print("both together on two UNMASKED streams: %.1f us" % tm([(un[0], gemm), (un[1], gru)]), flush=True)
