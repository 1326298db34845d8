"""Extractor input pipeline (vqa/datasets/dataset_vfeat.py:51-98, input_ops_vfeat.py:15-81): PIL resize to 540x540,
DenseCap xywh boxes -> scaled x1y1x2y2 -> normalised clipped [y1,x1,y2,x2], padded batches, one ordered pass."""
import json
import os

import numpy as np
import pytest

from vqa_transfer_externaldata_amd import dataset_vfeat as DV, hdf5_io, input_ops_vfeat as IO


def _make_images(root, n=5):
    from PIL import Image
    rng = np.random.default_rng(0)
    paths, sizes = [], []
    os.makedirs(os.path.join(root, "train2014"), exist_ok=True)
    for i in range(n):
        w, h = int(rng.integers(60, 200)), int(rng.integers(50, 180))
        arr = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        p = "train2014/COCO_train2014_%012d.jpg" % i
        Image.fromarray(arr).save(os.path.join(root, p), quality=95)
        paths.append(p); sizes.append((w, h))
    return paths, sizes


def test_box_helpers_known_answers():
    b = np.array([[10., 20., 30., 40.]], np.float32)                       # x, y, w, h
    np.testing.assert_array_equal(DV.xywh_to_x1y1x2y2(b), [[10, 20, 40, 60]])
    np.testing.assert_array_equal(DV.scale_boxes_xywh(b, [2.0, 0.5]), [[20, 10, 60, 20]])
    np.testing.assert_array_equal(DV.scale_boxes_xywh(b, 0.5), [[5, 10, 15, 20]])
    n = DV.normalize_boxes_x1y1x2y2(np.array([[54., 108., 270., 600.]]), 540, 540)   # y2 beyond the image -> clipped
    np.testing.assert_allclose(n, [[0.2, 0.1, 1.0, 0.5]], rtol=1e-6)


def test_dataset_and_batches_from_densecap_hdf5(tmp_path):
    img_dir, dc_dir = str(tmp_path / "images"), str(tmp_path / "densecap")
    paths, sizes = _make_images(img_dir)
    rng = np.random.default_rng(1)
    tree = {}
    nbx = [3, 60, 1, 7, 12]                                                   # one image has more than MAX_ROI_NUM boxes
    for p, (w, h), n in zip(paths, sizes, nbx):
        xy = rng.random((n, 2)) * [w / 2, h / 2]
        wh = rng.random((n, 2)) * [w / 2, h / 2] + 1
        tree[p.replace("/", "-")] = {"boxes": np.concatenate([xy, wh], 1).astype(np.float32)}
    os.makedirs(os.path.join(dc_dir, "train2014"))
    hdf5_io.write(os.path.join(dc_dir, "train2014", DV.DENSECAP_FILENAME), tree)
    ds = DV.create_dataset(paths, img_dir, dc_dir)
    cfg = ds.get_config()
    assert (cfg.image_width, cfg.image_height, cfg.max_roi_num) == (540, 540, 50) and len(ds) == 5
    d = ds.get_data(1)
    assert d["image"].shape == (540, 540, 3) and d["image"].dtype == np.float32 and 0 <= d["image"].min() <= d["image"].max() <= 255
    assert int(d["num_box"]) == 50 and d["box"].shape == (50, 4)               # truncated to MAX_ROI_NUM
    assert d["image_id"] == "train2014-COCO_train2014_000000000001.jpg" and int(d["image_id_len"]) == len(d["image_id"])
    w, h = sizes[1]
    raw = tree[d["image_id"]]["boxes"][:50]
    fx, fy = 540.0 / w, 540.0 / h
    want = np.stack([raw[:, 0] * fx, raw[:, 1] * fy, (raw[:, 0] + raw[:, 2]) * fx, (raw[:, 1] + raw[:, 3]) * fy], 1)
    np.testing.assert_allclose(d["box"], want, rtol=1e-5)
    np.testing.assert_allclose(d["normal_box"], np.clip(np.stack([want[:, 1], want[:, 0], want[:, 3], want[:, 2]], 1) / 540, 0, 1),
                               rtol=1e-5)
    assert np.all(d["normal_box"][:, 0] <= d["normal_box"][:, 2]) and np.all(d["normal_box"][:, 1] <= d["normal_box"][:, 3])
    # batches: one ordered pass, padded to the batch's longest box list, short last batch
    bs = list(IO.create(ds, 2, is_train=False, shuffle=False, num_parallel_calls=3, prefetch=2))
    assert [len(b["id"]) for b in bs] == [2, 2, 1] and [int(i) for b in bs for i in b["id"]] == [0, 1, 2, 3, 4]
    assert bs[0]["box"].shape == (2, 50, 4) and bs[1]["box"].shape == (2, 7, 4) and bs[2]["box"].shape == (1, 12, 4)
    np.testing.assert_array_equal(bs[0]["num_box"], [3, 50])
    assert np.all(bs[0]["normal_box"][0, 3:] == 0)                             # zero padding after the 3 real boxes
    np.testing.assert_array_equal(bs[0]["image"][1], d["image"])
    assert bs[1]["image_id"][1] == paths[3].replace("/", "-")
    # a training pipeline shuffles once and repeats
    it = IO.create(ds, 5, is_train=True, shuffle=True, repeat=2, num_parallel_calls=2)
    a, b = next(it), next(it)
    assert sorted(a["id"].tolist()) == [0, 1, 2, 3, 4] and a["id"].tolist() == b["id"].tolist() != [0, 1, 2, 3, 4]
    with pytest.raises(StopIteration):
        next(it)


def test_ring_buffers_give_the_same_batches_when_consumed_in_order(tmp_path):
    """reuse_buffers=True (the extractor's mode): same batches as the allocating mode when each one is consumed before
    the next is drawn; the image blocks come from a ring of prefetch + 2 buffers"""
    from PIL import Image
    rng = np.random.default_rng(5)
    d = tmp_path / "img" / "VG_100K"
    d.mkdir(parents=True)
    paths, boxes = [], {}
    for i in range(11):
        Image.fromarray(rng.integers(0, 255, size=(40 + i, 50, 3), dtype=np.uint8)).save(str(d / ("%d.png" % i)))
        paths.append("VG_100K/%d.png" % i)
        boxes["VG_100K-%d.png" % i] = rng.uniform(1, 20, size=(3 + i % 4, 4)).astype(np.float32)
    ds = DV.create_dataset(paths, str(tmp_path / "img"), None, boxes=boxes)
    want = [{k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}
            for b in IO.create(ds, 4, is_train=False, shuffle=False, num_parallel_calls=2, prefetch=2)]
    seen = set()
    n = 0
    for got, w in zip(IO.create(ds, 4, is_train=False, shuffle=False, num_parallel_calls=3, prefetch=2, reuse_buffers=True), want):
        for k in ("id", "image", "box", "normal_box", "num_box"):
            np.testing.assert_array_equal(got[k], w[k], err_msg=k)
        assert got["image_id"] == w["image_id"]
        seen.add(got["image"].__array_interface__["data"][0])
        n += 1
    assert n == len(want) == 3 and [len(w["id"]) for w in want] == [4, 4, 3] and len(seen) <= 4


def test_decoding_processes_give_the_same_batches_as_the_thread_pool(tmp_path):
    """processes=N: forked decoders writing into a shared-memory ring -- same batches, same order, byte pixels and float
    pixels; a failing image surfaces as an error in the consumer instead of a hang."""
    from PIL import Image
    rng = np.random.default_rng(6)
    d = tmp_path / "img" / "VG_100K"
    d.mkdir(parents=True)
    paths, boxes = [], {}
    for i in range(13):
        Image.fromarray(rng.integers(0, 255, size=(30 + i, 44, 3), dtype=np.uint8)).save(str(d / ("%d.png" % i)))
        paths.append("VG_100K/%d.png" % i)
        boxes["VG_100K-%d.png" % i] = rng.uniform(1, 20, size=(2 + i % 5, 4)).astype(np.float32)
    ds = DV.create_dataset(paths, str(tmp_path / "img"), None, boxes=boxes)
    for dt in (np.float32, np.uint8):
        want = [{k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}
                for b in IO.create(ds, 4, is_train=False, shuffle=False, num_parallel_calls=2, prefetch=2, reuse_buffers=True,
                                   image_dtype=dt)]
        n = 0
        for got, w in zip(IO.create(ds, 4, is_train=False, shuffle=False, prefetch=2, reuse_buffers=True, image_dtype=dt,
                                    processes=3), want):
            for k in ("id", "image", "box", "normal_box", "num_box", "image_id_len"):
                np.testing.assert_array_equal(got[k], w[k], err_msg=k)
            assert got["image_id"] == w["image_id"] and got["image"].dtype == dt
            n += 1
        assert n == len(want) == 4
    with pytest.raises(ValueError):
        IO.create(ds, 4, is_train=False, shuffle=False, reuse_buffers=False, processes=2)
    bad = DV.create_dataset(paths[:3] + ["VG_100K/missing.png"], str(tmp_path / "img"), None,
                            boxes=dict(boxes, **{"VG_100K-missing.png": boxes["VG_100K-0.png"]}))
    with pytest.raises(RuntimeError, match="decoding failed"):
        list(IO.create(bad, 2, is_train=False, shuffle=False, prefetch=2, reuse_buffers=True, processes=2))
