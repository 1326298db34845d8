"""Per-layer-shape timing of the extractor's convolutions (ResNet-101 blocks 1-4 @448) over tile configurations.

usage: conv_tune.py [batch] ; prints, for every distinct (Hi, Ci, k, Co, stride) of the network with its count, the
time per launch under each tile configuration, and the total of the per-shape best against the defaults.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib, vfeat as VF  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
PLAIN = [int(c) for c in os.environ.get("PLAIN_CFGS", "3,5,20,21,22,16").split(",")]
CONV = [int(c) for c in os.environ.get("CONV_CFGS", "0,1,2,3").split(",")]
lib = _lib.load()
dev = torch.device("cuda:0")

# (Hi, Ci, k, Co, stride, residual) -> count
shapes = {}
H = 112
cin = 64
for name, base, n, stride in VF.BLOCKS_R101_FULL:
    for depth, db, s in VF.block_units(base, n, stride):
        Ho = (H - 1) // s + 1
        if depth != cin:
            key = (H, cin, 1, depth, s, 0)
            shapes[key] = shapes.get(key, 0) + 1
        for key in ((H, cin, 1, db, 1, 0), (H, db, 3, db, s, 0), (Ho, db, 1, depth, 1, 1)):
            shapes[key] = shapes.get(key, 0) + 1
        H, cin = Ho, depth


def p(t):
    return _lib.C.c_void_p(t.data_ptr()) if t is not None else None


def time_shape(key, reps=5):
    Hi, Ci, k, Co, s, res = key
    Ho = (Hi - 1) // s + 1
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(batch, Hi, Hi, Ci, generator=g, device=dev)
    w = torch.rand(k, k, Ci, Co, generator=g, device=dev) * 0.01
    sc = torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    r = torch.rand(batch, Ho, Ho, Co, generator=g, device=dev) if res else None
    y = torch.empty(batch, Ho, Ho, Co, device=dev)
    st = _lib.C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = (k - 1) // 2          # conv2d_same: explicit (k-1)/2 padding, then VALID

    def run():
        _lib.check(lib.vqa_conv2d_nhwc(p(x), batch, Hi, Hi, Ci, p(w), k, k, Co, s, pad, pad, Ho, Ho, p(sc), p(sh), p(r), 1,
                                       p(y), st), "conv")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best, y


tot_def = tot_best = 0.0
for key, cnt in sorted(shapes.items(), key=lambda kv: -kv[1]):
    Hi, Ci, k, Co, s, res = key
    Ho = (Hi - 1) // s + 1
    fl = 2.0 * batch * Ho * Ho * Co * k * k * Ci
    plain = (k == 1 and s == 1)
    res_t = {}
    ref = None
    for c in (PLAIN if plain else CONV):
        if plain:
            lib.vqa_gemm_set_config(c)
        else:
            lib.vqa_conv_set_config(c)
        t, y = time_shape(key)
        res_t[c] = t
        if ref is None:
            ref = y.clone()
        else:
            d = (y - ref).abs().max().item()
            if d > 1e-3 * ref.abs().max().item():
                print("  !! cfg %d differs from first cfg by %.3e" % (c, d))
    lib.vqa_gemm_set_config(-1)
    lib.vqa_conv_set_config(-1)
    d0, _ = time_shape(key)               # the shipped per-shape choice
    b = min(res_t, key=res_t.get)
    tot_def += cnt * d0
    tot_best += cnt * res_t[b]
    print("Hi %3d Ci %4d k %d Co %4d s %d res %d x%2d | %s | best cfg %d  %.0f TF/s (default %.0f)" % (
        Hi, Ci, k, Co, s, res, cnt, "  ".join("%d: %7.1f" % (c, t) for c, t in res_t.items()), b,
        fl / res_t[b] / 1e6, fl / d0 / 1e6), flush=True)
print("sum over the network: default %.1f ms, per-shape best %.1f ms (batch %d)" % (tot_def / 1e3, tot_best / 1e3, batch))
