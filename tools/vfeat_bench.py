"""ResNet-101 (blocks 1-4) @448 region-feature extraction throughput; usage: vfeat_bench.py [batch] [iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VQA_HOT_LIB"):      # another build of the library (same-box A/B of compile-time choices)
    from vqa_transfer_externaldata_amd import _lib as _l0
    _l0._LIB_PATH = os.path.abspath(os.environ["VQA_HOT_LIB"])
from vqa_transfer_externaldata_amd import vfeat as VF  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(1234)
if os.environ.get("VQA_GEMM_CFG"):   # tuning only
    from vqa_transfer_externaldata_amd import _lib
    _lib.load().vqa_gemm_set_config(int(os.environ["VQA_GEMM_CFG"]))
if os.environ.get("VQA_CONV_CFG"):   # tuning only: one tile config for every implicit-GEMM convolution
    from vqa_transfer_externaldata_amd import _lib
    _lib.load().vqa_conv_set_config(int(os.environ["VQA_CONV_CFG"]))
model = VF.VfeatResnetModel(VF.init_random_params(rng, VF.BLOCKS_R101_FULL), VF.BLOCKS_R101_FULL)
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.rand(batch, 448, 448, 3, generator=g, device="cuda") * 255.0
ys = torch.sort(torch.rand(batch, 36, 2, generator=g, device="cuda"), dim=-1).values
xs = torch.sort(torch.rand(batch, 36, 2, generator=g, device="cuda"), dim=-1).values
b = {"image": img, "normal_box": torch.stack([ys[..., 0], xs[..., 0], ys[..., 1], xs[..., 1]], -1).contiguous()}
model.build(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    model.build(b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = VF.conv_flops_per_image(VF.BLOCKS_R101_FULL, 448, 448)
print("batch %d: %.1f imgs/s  %.1f TFLOP/s (%.1f%% of f32 MFMA peak)" % (batch, batch / dt, batch * fl / dt / 1e12,
                                                                       100 * batch * fl / dt / 1e12 / 157.3))
