"""v_linear_v forward: separate gather + GEMM against the gather fused into the GEMM's operand load
(vqa_gemm_f32_gather), at the bench shape (B 512, R 36, D 2048, H 1024, table of 8192 images)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib  # noqa: E402

lib = _lib.load()
B, R, D, H, N = 512, 36, 2048, 1024, 8192
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.randn(N, R, D, device="cuda", generator=g).relu_()
nbox = torch.full((N,), R, dtype=torch.int32, device="cuda")
idx = torch.randint(0, N, (B,), device="cuda", generator=g)
W = torch.randn(D, H, device="cuda", generator=g) * 0.02
bias = torch.zeros(H, device="cuda")
V = torch.empty(B * R, D, device="cuda"); nb = torch.empty(B, dtype=torch.int32, device="cuda")
out = torch.empty(B * R, H, device="cuda"); out2 = torch.empty_like(out); V2 = torch.empty_like(V)
P = lambda t: C.c_void_p(t.data_ptr())


def gather():
    _lib.check(lib.vqa_gather_features(P(table), P(nbox), P(idx), P(V), P(nb), B, R, D, N, None), "gather")


def gemm():
    _lib.check(lib.vqa_gemm_f32(0, 0, B * R, H, D, P(V), D, P(W), H, P(out), H, P(bias), None, 0, 1, None, 0, None), "gemm")


def fused(byproduct):
    _lib.check(lib.vqa_gemm_f32_gather(B * R, H, D, P(table), D, P(idx), R, N, P(W), H, P(out2), H, P(bias),
                                       P(V2) if byproduct else None, D, None), "fused")


def tm(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for cfg in (20, 21):
    _lib.check(lib.vqa_gemm_set_tall_config(cfg), "cfg")
    gather(); gemm(); fused(True); torch.cuda.synchronize()
    assert torch.equal(V, V2)
    print("tall cfg %d: max |fused - separate| = %.3e" % (cfg, float((out - out2).abs().max())))
    for rep in range(2):
        print("  gather %.1f us | gemm %.1f us | gather+gemm %.1f us | fused with V_ft by-product %.1f us | fused, no by-product %.1f us"
              % (tm(gather), tm(gemm), tm(lambda: (gather(), gemm())), tm(lambda: fused(True)), tm(lambda: fused(False))), flush=True)
