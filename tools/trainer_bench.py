"""Throughput of the Trainer mirror's own loop (data iterator + H2D + step + report fetch) at bs 512, full model
dimensions, synthetic split -- next to the kernel-only number of bench.py."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import input_ops_vqa as io, trainer

Vq, A, N_img, R, D, B = 16384, 3000, 2048, 36, 2048, 512
tmp = tempfile.mkdtemp()
c = trainer.parse_config(["--batch_size", str(B), "--max_train_iter", "100000", "--model_type", "vlmap_answer",
                          "--sort_by_length", os.environ.get("SORT", "1")])
c.vocab = {"vocab": ["w%d" % i for i in range(Vq)], "dict": {"w%d" % i: i for i in range(Vq)}}
c.answer_dict = {"vocab": ["a%d" % i for i in range(A)], "dict": {"a%d" % i: i for i in range(A)}, "num_train_answer": 2250,
                 "is_object": [i % 2 for i in range(A)], "is_attribute": [1 - i % 2 for i in range(A)]}
c.synthetic = 1
c.train_dir = os.path.join(tmp, "run"); c.tf_record_dir = os.path.join(tmp, "data")
rng = np.random.default_rng(0)
feats = {"features": np.maximum(rng.standard_normal((N_img, R, D), dtype=np.float32), 0), "spatials": np.zeros((N_img, R, 6), np.float32),
         "normal_boxes": np.zeros((N_img, R, 4), np.float32), "num_boxes": np.full(N_img, R, np.int32), "max_box_num": R, "vfeat_dim": D}
ds = {"train": io.synthetic_split(B * 64, N_img, Vq, A, seed=1), "val": io.synthetic_split(B * 2, N_img, Vq, A, seed=2),
      "testval": io.synthetic_split(B * 2, N_img, Vq, A, seed=3)}
t = trainer.Trainer(c, datasets=ds, image_features=feats)
t0 = time.perf_counter()
for _ in range(64):                      # first epoch: every batch is assembled on the host and uploaded once
    t.run_train_step(False)
torch.cuda.synchronize()
print("first epoch (host assembly + upload overlapped with the previous step): %.2f ms/step" % ((time.perf_counter() - t0) / 64 * 1e3))
for _ in range(5):
    t.run_train_step(False)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.perf_counter()
for _ in range(n):
    t.run_train_step(False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("Trainer.run_train_step, batches cached on the device: %.2f ms/step = %.0f samples/s (bench.py kernel-only step: ~3.8 ms)" % (dt * 1e3, B / dt))

# host-side breakdown of one step (all batches cached on the device by now)
import collections
acc = collections.defaultdict(float)
m = t.model
for _ in range(50):
    a = time.perf_counter(); m.set_batch(t._next("train")); m.build(); b = time.perf_counter()
    m.backward(reducer=t._allreduce); c_ = time.perf_counter()
    m.apply_gradients(t._lr()); d = time.perf_counter()
    torch.cuda.synchronize(); e = time.perf_counter()
    rep = m.engine.report(); f = time.perf_counter()
    for k, v in (("next+build enqueue", b - a), ("backward enqueue", c_ - b), ("optimizer enqueue", d - c_),
                 ("wait for GPU", e - d), ("report fetch", f - e)):
        acc[k] += v / 50
print("host breakdown (ms):", {k: round(v * 1e3, 3) for k, v in acc.items()})
