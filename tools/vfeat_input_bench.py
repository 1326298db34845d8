"""Throughput of the extractor's input side (JPEG decode + resize + box normalisation + padded batches,
dataset_vfeat / input_ops_vfeat) on the host cores: usage vfeat_input_bench.py [n_images] [workers,...]"""
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import dataset_vfeat as DV, input_ops_vfeat as IO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
workers = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,16").split(",")]
rng = np.random.default_rng(0)
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "VG_100K"))
    paths, boxes = [], {}
    yy, xx = np.mgrid[0:480, 0:640]
    for i in range(n):                                   # photo-like content: smooth gradients + some texture
        img = np.stack([(np.sin(xx / (20.0 + i % 7)) + np.cos(yy / (15.0 + i % 5))) * 60 + 128 + rng.normal(0, 8, xx.shape)
                        for _ in range(3)], -1).clip(0, 255).astype(np.uint8)
        p = "VG_100K/%d.jpg" % i
        Image.fromarray(img).save(os.path.join(d, p), quality=90)
        paths.append(p)
        b = rng.uniform(0, 300, size=(36, 4)).astype(np.float32); b[:, 2:] += 20
        boxes[p.replace("/", "-")] = b
    ds = DV.create_dataset(paths, d, None, is_train=False, boxes=boxes)
    for w in workers:
        for dt_img in (np.float32, np.uint8):
            t0 = time.perf_counter()
            cnt = 0
            for b in IO.create(ds, 96, is_train=False, shuffle=False, num_parallel_calls=w, prefetch=3, reuse_buffers=True,
                               image_dtype=dt_img):
                cnt += len(b["id"])
            dt = time.perf_counter() - t0
            print("%d workers, %s pixels: %d images (640x480 JPEG -> %s) in %.2f s = %.0f imgs/s on %d host cores" % (
                w, np.dtype(dt_img).name, cnt, "x".join(str(x) for x in b["image"].shape[1:]), dt, cnt / dt, os.cpu_count()),
                flush=True)
