"""Big-GEMM K sweep (fixed overhead vs steady state) with the vendor library beside it; usage: gemm_big_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib, ops
lib = _lib.load()
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, N = 18432, 1024
for K in (512, 1024, 2048, 4096, 8192):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    row = []
    for cfg in (5, 20, 21, 22):
        lib.vqa_gemm_set_config(cfg)
        us = bench(lambda: ops.gemm(A, B, out=out, split_k=1))
        row.append("cfg%d %.0fus %.1fTF" % (cfg, us, 2.0 * M * N * K / us / 1e6))
    us = bench(lambda: torch.matmul(A, B, out=out))
    row.append("rocblas %.0fus %.1fTF" % (us, 2.0 * M * N * K / us / 1e6))
    print("NN K=%d: " % K + "  ".join(row), flush=True)
# dW shape (TN): K = B*R
Kd, Md, Nd = 18432, 2048, 1024
A = torch.randn(Kd, Md, device="cuda"); B = torch.randn(Kd, Nd, device="cuda"); out = torch.empty(Md, Nd, device="cuda")
for cfg in (0, 19, 20, 21):
    lib.vqa_gemm_set_config(cfg)
    for sk in (0, 4, 8):
        try:
            us = bench(lambda: ops.gemm(A, B, transA=True, out=out, split_k=sk))
            print("TN cfg%d split%d %.0fus %.1fTF" % (cfg, sk, us, 2.0 * Md * Nd * Kd / us / 1e6), flush=True)
        except Exception as ex:
            print("TN cfg%d split%d failed %s" % (cfg, sk, ex))
us = bench(lambda: torch.matmul(A.t(), B, out=out))
print("TN rocblas %.0fus %.1fTF" % (us, 2.0 * Md * Nd * Kd / us / 1e6))
