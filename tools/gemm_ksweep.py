import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vqa_transfer_externaldata_amd import _lib, ops
lib = _lib.load()
def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for cfg in (11, 14, 15):
    lib.vqa_gemm_set_config(cfg)
    for (M, N) in ((512, 2048), (512, 1024)):
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
            row.append("K%d %.1f" % (K, bench(lambda: ops.gemm(A, B, out=out, split_k=1))))
        print("cfg%d %dx%d: " % (cfg, M, N) + "  ".join(row))
