"""Does the existence / use of CU-masked streams slow down kernels on ordinary streams?"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vqa_transfer_externaldata_amd import ops  # noqa: E402

hip = C.CDLL("libamdhip64.so")
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(18432, 2048, device="cuda", generator=g).relu_()
B = torch.randn(2048, 1024, device="cuda", generator=g) * 0.03
out = torch.empty(18432, 1024, device="cuda")
x = torch.randn(64 << 20, device="cuda")
y = torch.empty_like(x)


def tm(f, n=10):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def report(tag):
    print("%-52s GEMM %.1f us, 256 MB copy %.1f us" % (tag, tm(lambda: ops.gemm(A, B, out=out)), tm(lambda: y.copy_(x))), flush=True)


torch.cuda.set_stream(torch.cuda.Stream())
report("ordinary stream, no masked stream exists")
words = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 4 + [0] * 4))
s = C.c_void_p()
assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
report("ordinary stream, a masked stream (128 CUs) exists")
ms = torch.cuda.ExternalStream(s.value)
with torch.cuda.stream(ms):
    ops.gemm(A, B, out=out)
torch.cuda.synchronize()
report("ordinary stream, after the masked stream ran a GEMM")
with torch.cuda.stream(ms):
    print("masked stream itself: GEMM %.1f us" % tm(lambda: ops.gemm(A, B, out=out)), flush=True)
report("ordinary stream again")
assert hip.hipStreamDestroy(s) == 0
report("ordinary stream, masked stream destroyed")
