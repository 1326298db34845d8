"""Throughput of the cfg-5 pre-training step (BASELINE configs[4], stage 1) at bs 512 on one MI355X."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VQA_HOT_LIB"):      # A/B of two builds of the library on one box (tuning only)
    from vqa_transfer_externaldata_amd import _lib as _l0
    _l0._LIB_PATH = os.environ["VQA_HOT_LIB"]
from vqa_transfer_externaldata_amd import pretrain as PT  # noqa: E402

B, n, R, D, H, L, W, Vq, n_ws, A = 512, 5, 36, 2048, 1024, 10, 300, 5000, 2000, 4000
rng = np.random.default_rng(0)
p = PT.init_random_params(rng, Vq, n_ws, A, W=W, D=D, H=H)
from vqa_transfer_externaldata_amd import dataset_vlmap as DV  # noqa: E402
ds = DV.Dataset(split="train", data=DV.synthetic_dataset(B, Vq, n_ws, A, R=R, D=D, max_len=L, seed=0), seed=0)
batch = next(DV.create_ops(B, ds, is_train=True, shuffle=False))
batch = {k: v for k, v in batch.items() if v.dtype.kind in "fi" and k != "image_id"}
sort_info = {} if os.environ.get("SORT", "1") == "0" else {k: v for k, v in PT.add_length_sort(dict(batch)).items() if k.endswith("/sort")}
eng = PT.PretrainEngine(n=n, R=R, D=D, H=H, W=W, A=A, Vq=Vq, n_ws=n_ws, params=p)
db = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
db.update(sort_info)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if any(os.environ.get(k) for k in ("VQA_LN_FAST", "VQA_GRU_CFG", "VQA_ATTN_FAST", "VQA_SOFTMAX_FAST")):   # A/B switches
    from vqa_transfer_externaldata_amd import _lib
    _l = _lib.load()
    if os.environ.get("VQA_LN_FAST"):
        _l.vqa_ln_set_fast(int(os.environ["VQA_LN_FAST"]))
    if os.environ.get("VQA_GRU_CFG"):
        _lib.check(_l.vqa_gemm_set_gru_config(int(os.environ["VQA_GRU_CFG"])), "gru cfg")
    if os.environ.get("VQA_ATTN_FAST"):
        _l.vqa_attn_set_fast(int(os.environ["VQA_ATTN_FAST"]))
    if os.environ.get("VQA_SOFTMAX_FAST"):
        _l.vqa_softmax_set_fast(int(os.environ["VQA_SOFTMAX_FAST"]))
for i in range(3):
    eng.train_step(db, eng.make_keep_masks(B, int(os.environ.get("MASK_SEED", "1")), i), 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    eng.train_step(db, eng.make_keep_masks(B, int(os.environ.get("MASK_SEED", "1")), 3 + i), 1e-3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
rep = eng.fetch_report()
print("pretrain step %.2f ms  -> %.0f images/s (%.0f blank-fill entries/s); total_loss %.3f"
      % (dt * 1e3, B / dt, 2 * B * n / dt, rep["total_loss"]))
