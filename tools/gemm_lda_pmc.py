"""Does the power-of-two row stride of the left operand (2048 floats = 8 KB) cost L2 set conflicts?  Runs the
v_linear_v forward GEMM (18432 x 1024 x 2048) on the same matrix stored with lda = 2048 and with lda = 2048 + 32,
10 launches each (run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and read the CSV in dispatch order with
--summarise <counter_collection.csv>)."""
import csv
import ctypes as C
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    csv.field_size_limit(1 << 30)
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "gemm_f32_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    vals = [float(r["Counter_Value"]) * 2 * 1024 / 1e6 for r in rows]          # KiB, x2 (gfx950) -> MB
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    n = len(vals) // 2
    for name, lo in (("lda 2048", 0), ("lda 2080", n)):
        v, d = vals[lo + 2:lo + n], durs[lo + 2:lo + n]
        print("%s: fetch %.1f MB / launch, %.1f us (n=%d)" % (name, sum(v) / len(v), sum(d) / len(d), len(v)))
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib  # noqa: E402
lib = _lib.load()
M, N, K = 18432, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)
W = torch.randn(K, N, device="cuda", generator=g) * 0.02
out = torch.empty(M, N, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
for lda in (K, K + 32):
    A = torch.randn(M, lda, device="cuda", generator=g).relu_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for i in range(12):
        if i == 2:
            torch.cuda.synchronize(); e0.record()
        _lib.check(lib.vqa_gemm_f32(0, 0, M, N, K, P(A), lda, P(W), N, P(out), N, None, None, 0, 1, None, 0, None), "gemm")
    e1.record(); torch.cuda.synchronize()
    print("lda %d: %.1f us" % (lda, e0.elapsed_time(e1) * 100), flush=True)
