"""Short-K GEMM with A stationary in registers (csrc/gemm_shortk.hip) against the general f32 MFMA kernel (vqa_gemm_f32) on
the shapes it was written for: the packed x-projection of the GRU (M = T*B = 7168, K = 300 in rows of 304, N = 3072) and the
extractor's 1x1 expansion convolutions (128 images: conv2 M 1605632 K 64, conv3 M 401408 K 128, conv4 M 100352 K 256).
Prints time, fraction of the f32 MFMA peak (157.3 TFLOP/s) and the difference between the two kernels' results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib, ops  # noqa: E402

if os.environ.get("SK_LIB"):      # another build of the library (timing variants, tools/dbg)
    _lib._LIB_PATH = os.path.abspath(os.environ["SK_LIB"])
PEAK = 157.3e12
g = torch.Generator(device="cuda").manual_seed(0)


def tm(f, n=12):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


shapes = [("xp  (GRU x-projection)", 7168, 3072, 300, 304, False),
          ("conv5 expand 512->2048", 25088, 2048, 512, 512, True),
          ("conv4 expand 256->1024", 100352, 1024, 256, 256, True),
          ("conv3 expand 128->512", 401408, 512, 128, 128, True),
          ("conv2 expand 64->256", 401408 * 2, 256, 64, 64, True)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0].split()[0] in sys.argv[1].split(",")]
grids = [int(x) for x in os.environ.get("SK_GRIDS", "0,256,768,1024").split(",")]
if os.environ.get("SK_WAVES"):
    _lib.check(_lib.load().vqa_gemm_shortk_set_waves(int(os.environ["SK_WAVES"])), "waves")
lib = _lib.load()
for name, M, N, K, lda, epi in shapes:
    Afull = torch.randn(M, lda, device="cuda", generator=g)
    A = Afull[:, :K]
    B = torch.randn(K, N, device="cuda", generator=g) * (2.0 / K) ** 0.5
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    fl = 2.0 * M * N * K
    ref = ops.gemm(A, B, bias=bias)
    out = torch.empty_like(ref)
    b0, m0 = tm(lambda: ops.gemm(A, B, bias=bias, out=ref))
    print("%-26s M %7d N %4d K %3d: general kernel %7.1f us (median %7.1f) = %.3f of peak" % (name, M, N, K, b0, m0, fl / b0 / 1e-6 / PEAK),
          flush=True)
    for grid in grids:
        _lib.check(lib.vqa_gemm_shortk_set_grid(grid), "grid")
        ops.gemm_shortk(A, B, bias=bias, out=out)
        torch.cuda.synchronize()
        diff = float((out - ref).abs().max())
        b1, m1 = tm(lambda: ops.gemm_shortk(A, B, bias=bias, out=out))
        print("    A-stationary, grid %4s: %7.1f us (median %7.1f) = %.3f of peak   max |diff| %.2e" % (
            grid if grid else "2/CU", b1, m1, fl / b1 / 1e-6 / PEAK, diff), flush=True)
    _lib.check(lib.vqa_gemm_shortk_set_grid(0), "grid")
    if epi:     # the convolution's epilogue: folded BatchNorm scale / shift, residual, ReLU
        scale = torch.rand(N, device="cuda", generator=g) + 0.5
        res = torch.randn(M, N, device="cuda", generator=g)
        want = torch.relu((A @ B) * scale + bias + res)
        got = ops.gemm_shortk(A, B, bias=bias, scale=scale, residual=res, relu=True, out=out)
        torch.cuda.synchronize()
        print("    with scale / shift / residual / ReLU: max |diff to torch| %.2e" % float((got - want).abs().max()), flush=True)
        b1, m1 = tm(lambda: ops.gemm_shortk(A, B, bias=bias, scale=scale, residual=res, relu=True, out=out))
        print("    ... %7.1f us (median %7.1f) = %.3f of peak" % (b1, m1, fl / b1 / 1e-6 / PEAK), flush=True)
        del want, got, res
    del Afull, A, B, ref, out
    torch.cuda.empty_cache()
