"""Runs one GEMM shape repeatedly (for rocprofv3 --pmc passes).  usage: one_gemm.py lay M N K cfg [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib, ops  # noqa: E402

lay, M, N, K, cfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
lib = _lib.load()
lib.vqa_gemm_set_config(cfg)
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn((K, M) if lay == "tn" else (M, K), device="cuda", generator=g)
B = torch.randn((N, K) if lay == "nt" else (K, N), device="cuda", generator=g)
out = torch.empty(M, N, device="cuda")
for _ in range(iters):
    ops.gemm(A, B, transA=(lay == "tn"), transB=(lay == "nt"), out=out)
torch.cuda.synchronize()
