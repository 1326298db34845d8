"""Attention + pooling kernels alone at the models' shapes: forward / backward, 1 query per image (VQA step, B 512) and
5 queries per image (pre-training step), fast and generic kernels on the same box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib, ops  # noqa: E402

if os.environ.get("VQA_HOT_LIB"):      # another build of the library (same-box A/B of compile-time choices)
    _lib._LIB_PATH = os.path.abspath(os.environ["VQA_HOT_LIB"])
lib = _lib.load()
B, R, H, D = 512, 36, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)


def tm(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for rep in [int(x) for x in os.environ.get("ATTN_REPS", "1,5").split(",")]:
    v = torch.relu(torch.randn(B, R, H, device="cuda", generator=g))
    qv = torch.relu(torch.randn(B * rep, H, device="cuda", generator=g))
    V = torch.relu(torch.randn(B, R, D, device="cuda", generator=g))
    w = torch.randn(H, device="cuda", generator=g) * 0.1
    bias = torch.zeros(1, device="cuda")
    nb = torch.full((B,), R, dtype=torch.int32, device="cuda")
    km = (torch.rand(B * rep, R, H, device="cuda", generator=g) < 0.8).to(torch.uint8)
    dp = torch.randn(B * rep, D, device="cuda", generator=g)
    att, _ = ops.attn_pool_fwd_rep(v, qv, V, nb, w, bias, rep, km, 0.8)
    for fast in (0, 1):
        lib.vqa_attn_set_fast(fast)
        tf = tm(lambda: ops.attn_pool_fwd_rep(v, qv, V, nb, w, bias, rep, km, 0.8))
        tb = tm(lambda: ops.attn_pool_bwd_rep(dp, v, qv, V, att, w, rep, km, 0.8))
        print("rep %d  %s kernels: forward %6.1f us   backward (+ 2 column sums) %6.1f us" % (rep, "fast   " if fast else "generic", tf, tb),
              flush=True)
    lib.vqa_attn_set_fast(1)
