"""EXPERIMENT: the roofline GEMM (v_linear_v forward: M 18432, N 1024, K 2048) on the bf16 matrix pipe with three-way operand
splits (csrc/gemm_bf16x3.hip) against the shipped exact-f32 MFMA kernel: time, and error of both against float64."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import ops  # noqa: E402

M, N, K = 18432, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).relu_()                 # post-ReLU-like region features
B = (torch.rand(K, N, device="cuda", generator=g) * 2 - 1) * (6.0 / (K + N)) ** 0.5
bias = torch.randn(N, device="cuda", generator=g) * 0.1


def tm(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


c32 = ops.gemm(A, B, bias=bias)
c3 = ops.gemm_bf16x3(A, B, bias=bias)
torch.cuda.synchronize()
rows = slice(0, 2048)                                                       # float64 reference on a slab of rows
ref = (A[rows].double().cpu() @ B.double().cpu() + bias.double().cpu()).numpy()
for name, c in (("f32 MFMA (32x32x2 f32)", c32), ("bf16 x 3 (6 x 32x32x16 bf16)", c3)):
    d = np.abs(c[rows].double().cpu().numpy() - ref)
    print("%-30s max |err| %.3e  rms err %.3e  (max |C| %.2f, rms |C| %.3f)" % (name, d.max(), np.sqrt((d ** 2).mean()), np.abs(ref).max(),
                                                                              np.sqrt((ref ** 2).mean())), flush=True)
print("bf16x3 vs f32 MFMA: max |diff| %.3e" % float((c3 - c32).abs().max()), flush=True)
fl = 2.0 * M * N * K
for name, f in (("f32 MFMA", lambda: ops.gemm(A, B, bias=bias, out=c32)), ("bf16 x 3", lambda: ops.gemm_bf16x3(A, B, bias=bias, out=c3))):
    best, med = tm(f)
    print("%-10s %.1f us (median %.1f) = %.1f algorithmic TFLOP/s" % (name, best, med, fl / best / 1e6), flush=True)
