"""GPU tuning sweep for vqa_gemm_f32: every tile config x the GEMM shapes of the
bs-512 train step.  Run on the GPU box:  python tools/gemm_tune.py [--split]"""
import ctypes as C
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib, ops  # noqa: E402

SHAPES = [
    # name, layout, M, N, K
    ("v_fwd", "nn", 18432, 1024, 2048),
    ("xp_g", "nn", 7168, 2048, 300),
    ("gru_gates_fwd", "nn", 512, 2048, 1024),
    ("gru_cand_fwd", "nn", 512, 1024, 1024),
    ("head_fwd", "nn", 512, 3000, 2048),
    ("fc_1024", "nn", 512, 1024, 1024),
    ("fc_2048_1024", "nn", 512, 1024, 2048),
    ("fc_1024_2048", "nn", 512, 2048, 1024),
    ("fc_bwd_dx", "nt", 512, 1024, 1024),
    ("fc_dw", "tn", 1024, 1024, 512),
    ("fc_dw2", "tn", 2048, 3000, 512),
    ("gru_bwd_dh", "nt", 512, 1024, 2048),
    ("gru_bwd_drh", "nt", 512, 1024, 1024),
    ("dj", "nt", 512, 2048, 3000),
    ("dx", "nt", 7168, 300, 2048),
    ("dWv", "tn", 2048, 1024, 18432),
    ("dWg_h", "tn", 1024, 2048, 7168),
    ("dWc_h", "tn", 1024, 1024, 7168),
    ("dWg_x", "tn", 300, 2048, 7168),
    ("xp_cat", "nn", 7168, 3072, 300), ("dx_cat", "nt", 7168, 300, 3072), ("dWx_cat", "tn", 300, 3072, 7168),   # packed x rows
]

if os.environ.get("TUNE_SET") == "pretrain":      # cfg-5 pre-training step: 2560 rows per category
    SHAPES = [
        ("pl_fwd", "nn", 2560, 1024, 2048), ("q_fwd", "nn", 2560, 1024, 1024), ("j_fwd", "nn", 2560, 2048, 1024),
        ("cls_fwd", "nn", 2560, 4000, 2048), ("d_j", "nt", 2560, 2048, 4000), ("d_jin", "nt", 2560, 1024, 2048),
        ("d_lft", "nt", 2560, 1024, 1024), ("d_pooled", "nt", 2560, 2048, 1024), ("xp_g", "nn", 25600, 2048, 300),
        ("xp_c", "nn", 25600, 1024, 300), ("dx_g", "nt", 25600, 300, 2048), ("dWc", "tn", 2048, 4000, 2560),
        ("dWj", "tn", 1024, 2048, 2560), ("dWq", "tn", 1024, 1024, 2560), ("dWpl", "tn", 2048, 1024, 2560),
        ("dWg_x", "tn", 300, 2048, 25600), ("dWg_h", "tn", 1024, 2048, 25600),
    ]


if os.environ.get("TUNE_ONLY"):       # a subset of the shapes by name
    SHAPES = [s for s in SHAPES if s[0] in os.environ["TUNE_ONLY"].split(",")]

CFGS = [int(x) for x in os.environ.get('TUNE_CFGS', '0,1,3,4,6,7,10,11,12,13').split(',')]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    lib = _lib.load()
    # bring the device to its loaded clock first: the first measurements of a cold process read 10-15 % slow
    wa = torch.randn(8192, 4096, device="cuda"); wb = torch.randn(4096, 4096, device="cuda")
    for _ in range(300):
        ops.gemm(wa, wb)
    torch.cuda.synchronize()
    splits = [0, 1, 2, 4, 8] if "--split" in sys.argv else [0]
    if "--persist-only" in sys.argv:
        splits = []
    print("%-14s %-3s %6s %5s %6s | " % ("shape", "lay", "M", "N", "K") +
          " ".join("cfg%d" % c for c in CFGS) + "   (us / TFLOP/s, split_k auto)")
    for name, lay, M, N, K in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        A = torch.randn((K, M) if lay == "tn" else (M, K), device="cuda", generator=g)
        B = torch.randn((N, K) if lay == "nt" else (K, N), device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda")
        for sk in splits:
            cells = []
            for cfg in CFGS:
                lib.vqa_gemm_set_config(cfg)
                f = lambda: ops.gemm(A, B, transA=(lay == "tn"), transB=(lay == "nt"), split_k=sk, out=out)
                us = min(bench(f), bench(f))
                cells.append("%7.1f/%5.1f" % (us, 2.0 * M * N * K / us / 1e6))
            print("%-14s %-3s %6d %5d %6d sk%d | " % (name, lay, M, N, K, sk) + " ".join(cells))
    lib.vqa_gemm_set_config(-1)
    if "--order" in sys.argv:
        print("tile order sweep (auto / n-fastest / m-fastest): us")
        for name, lay, M, N, K in SHAPES:
            g = torch.Generator(device="cuda").manual_seed(0)
            A = torch.randn((K, M) if lay == "tn" else (M, K), device="cuda", generator=g)
            B = torch.randn((N, K) if lay == "nt" else (K, N), device="cuda", generator=g)
            out = torch.empty(M, N, device="cuda")
            cells = []
            for order in (-1, 0, 1):
                lib.vqa_gemm_set_order(order)
                f = lambda: ops.gemm(A, B, transA=(lay == "tn"), transB=(lay == "nt"), split_k=0, out=out)
                us = bench(f)
                cells.append("%7.1f/%5.1f" % (us, 2.0 * M * N * K / us / 1e6))
            print("%-14s %-3s | " % (name, lay) + "  ".join(cells))
        lib.vqa_gemm_set_order(-1)
    if "--persist" in sys.argv:
        print("persistent (max_blocks) sweep: us / TFLOP/s")
        for name, lay, M, N, K in [s_ for s_ in SHAPES if s_[0] in ("v_fwd", "dWv", "xp_g", "dWg_h")]:
            g = torch.Generator(device="cuda").manual_seed(0)
            A = torch.randn((K, M) if lay == "tn" else (M, K), device="cuda", generator=g)
            B = torch.randn((N, K) if lay == "nt" else (K, N), device="cuda", generator=g)
            out = torch.empty(M, N, device="cuda")
            for cfg in (0, 1, 5, 6):
                cells = []
                for mb in (0, 256, 512, 768):
                    lib.vqa_gemm_set_config(cfg)
                    lib.vqa_gemm_set_max_blocks(mb)
                    f = lambda: ops.gemm(A, B, transA=(lay == "tn"), transB=(lay == "nt"), split_k=0, out=out)
                    us = bench(f)
                    cells.append("mb%d %7.1f/%5.1f" % (mb, us, 2.0 * M * N * K / us / 1e6))
                print("%-8s cfg%d | " % (name, cfg) + "  ".join(cells))
        lib.vqa_gemm_set_config(-1)
        lib.vqa_gemm_set_max_blocks(0)


if __name__ == "__main__":
    main()
