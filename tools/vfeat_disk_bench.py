"""The extractor END TO END from JPEG files: decode + resize on the host thread pool -> upload -> conv stack + ROI crop ->
rows of the dense feature arrays (vfeat.Extractor.extract, as `python -m ...vfeat_extractor` runs it), images/s.
usage: vfeat_disk_bench.py [n_images] [batch] [workers]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import dataset_vfeat as DV, input_ops_vfeat as IO, vfeat as VF  # noqa: E402
from vqa_transfer_externaldata_amd.vfeat_extractor import device_batches  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.default_rng(0)
model = None if os.environ.get("LOADER_PROCESSES") else VF.VfeatResnetModel(VF.init_random_params(rng, VF.BLOCKS_R101_FULL),
                                                                            VF.BLOCKS_R101_FULL)
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "VG_100K"))
    paths, boxes = [], {}
    yy, xx = np.mgrid[0:480, 0:640]
    for i in range(n):
        img = np.stack([(np.sin(xx / (20.0 + i % 7)) + np.cos(yy / (15.0 + i % 5))) * 60 + 128 + rng.normal(0, 8, xx.shape)
                        for _ in range(3)], -1).clip(0, 255).astype(np.uint8)
        p = "VG_100K/%d.jpg" % i
        Image.fromarray(img).save(os.path.join(d, p), quality=90)
        paths.append(p)
        b = rng.uniform(0, 300, size=(36, 4)).astype(np.float32); b[:, 2:] += 20
        boxes[p.replace("/", "-")] = b
    id2idx = {p.replace("/", "-"): i for i, p in enumerate(paths)}
    paths = paths * int(os.environ.get("REPEAT", "6"))            # steady state: the files are read several times
    n = len(paths)
    ds = DV.create_dataset(paths, d, None, is_train=False, boxes=boxes)
    if os.environ.get("LOADER_PROCESSES"):       # forked decoders: every pipeline is created BEFORE the GPU is touched
        nproc = int(os.environ["LOADER_PROCESSES"])
        pipes = [("%d decoding processes, shared pinned ring, byte pixels" % nproc,
                  IO.create(ds, B, is_train=False, shuffle=False, prefetch=3, reuse_buffers=True, pinned=True, image_dtype=np.uint8,
                            processes=nproc)) for _ in range(2)]
        model = VF.VfeatResnetModel(VF.init_random_params(rng, VF.BLOCKS_R101_FULL), VF.BLOCKS_R101_FULL)
        for mode, batches in pipes:
            ex = VF.Extractor(model, id2idx, ds.get_config().max_roi_num)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = ex.extract(device_batches(batches, "cuda:0"))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("%-24s %d JPEGs 640x480 -> 540x540 -> resnet_v1_101 b1-4 + 36-box crop: %.2f s = %.0f images/s (batch %d)" % (
                mode, n, dt, n / dt, B), flush=True)
        sys.exit(0)
    for mode, kw in (("fresh pageable batches", dict(reuse_buffers=False)),
                     ("ring of pageable blocks", dict(reuse_buffers=True)),
                     ("ring of pinned blocks", dict(reuse_buffers=True, pinned=True)),
                     ("pinned ring, byte pixels", dict(reuse_buffers=True, pinned=True, image_dtype=np.uint8))):
        ex = VF.Extractor(model, id2idx, ds.get_config().max_roi_num)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ex.extract(device_batches(IO.create(ds, B, is_train=False, shuffle=False, num_parallel_calls=workers,
                                                  prefetch=3, **kw), "cuda:0"))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-24s %d JPEGs 640x480 -> 540x540 -> resnet_v1_101 b1-4 + 36-box crop: %.2f s = %.0f images/s (batch %d, %d loader "
              "threads)" % (mode, n, dt, n / dt, B, workers), flush=True)
