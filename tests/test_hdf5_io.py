"""hdf5_io: the TF-free / h5py-free reader and writer of the reference's HDF5 files.

Pins, strongest first:
  1. files written through the REAL libhdf5 C library with the calls h5py makes for the reference's scripts
     (tests/h5ref.py; contiguous datasets, symbol-table groups, scalar int64, variable-length string) are read
     back value-for-value by hdf5_io.File;
  2. files written by hdf5_io.write are read back by the real libhdf5 (H5Dread) value-for-value;
  3. round trip through our own writer + reader, byte-level superblock checks, error behaviour.
1 and 2 skip when the image has no libhdf5 (this one has /opt/conda/lib/libhdf5.so.103 = HDF5 1.10.6)."""
import os
import struct

import numpy as np
import pytest

from tests import h5ref
from vqa_transfer_externaldata_amd import hdf5_io as H

needs_libhdf5 = pytest.mark.skipif(h5ref.load() is None, reason="no libhdf5 in this image")


def _feature_tree(rng, N=7, R=36, D=48):
    """The feature file of data/tools/vqa_v2/process_bottom_up_attention_36.py:47-53,95-100."""
    nb = rng.random((N, R, 4)).astype(np.float32)
    return {"image_features": rng.standard_normal((N, R, D)).astype(np.float32), "normal_boxes": nb,
            "spatial_features": rng.random((N, R, 6)).astype(np.float32),
            "num_boxes": rng.integers(10, R + 1, N).astype(np.int32),
            "data_info": {"vfeat_dim": D, "max_box_num": R, "pretrained_param_path": "bottom_up_attention_36"}}


def _check_tree(f, tree, prefix=""):
    assert sorted(f.keys()) == sorted(tree)
    for k, v in tree.items():
        if isinstance(v, dict):
            assert isinstance(f[k], H.Group)
            _check_tree(f[k], v, prefix + k + "/")
        elif isinstance(v, str):
            assert f[k][()] == v and f[k].value == v
        else:
            a = np.asarray(v)
            got = f[k]
            assert tuple(got.shape) == a.shape, (prefix + k, got.shape, a.shape)
            np.testing.assert_array_equal(np.asarray(got), a)
            if a.ndim == 0:
                assert int(got[()]) == int(a) and int(got.value) == int(a)


@needs_libhdf5
def test_reads_files_written_by_the_real_libhdf5_like_h5py_does(tmp_path):
    rng = np.random.default_rng(0)
    h5 = h5ref.H5()
    # feature file: contiguous f32 / i32 datasets, a sub-group with scalar int64 + variable-length string datasets
    tree = _feature_tree(rng)
    p = str(tmp_path / "vfeat_bottomup_36.hdf5")
    h5.write(p, tree)
    assert H.is_hdf5(p)
    with H.File(p) as f:
        assert f.superblock["version"] == 0 and f.superblock["size_of_offsets"] == 8
        _check_tree(f, tree)
        # the access pattern of vqa/model_vlmap_answer.py:59-70
        feats = np.array(f.get("image_features"))
        assert feats.dtype == np.float32 and feats.shape == (7, 36, 48)
        assert int(f["data_info"]["max_box_num"].value) == 36 and int(f["data_info"]["vfeat_dim"].value) == 48
        assert isinstance(f["image_features"].read(), np.memmap)          # zero-copy for contiguous storage
        np.testing.assert_array_equal(f["image_features"][3, 5:7], tree["image_features"][3, 5:7])
    # weights.hdf5 of vlmap_memft/export_word_weights.py:60-73 (f[key] = ndarray)
    w = {k: rng.standard_normal(s).astype(np.float32) for k, s in
         [("v_word", (50, 300)), ("l_word", (50, 300)), ("l_answer_word", (40, 300)), ("class_weights", (2048, 40)),
          ("class_biases", (40,))]}
    p2 = str(tmp_path / "weights.hdf5")
    h5.write(p2, w)
    got = H.load_tree(p2)
    assert sorted(got) == sorted(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k])
    # data_info.hdf5 of generator_tf_record_memft_genome.py:103-107 (np.array(x, dtype=np.int32) scalars + tables)
    di = {"data_info": {"intseq_ans": rng.integers(0, 99, (30, 4)).astype(np.int32),
                        "intseq_ans_len": rng.integers(1, 5, 30).astype(np.int32),
                        "max_ans_len": np.array(4, np.int32), "num_answers": np.array(30, np.int32)}}
    p3 = str(tmp_path / "data_info.hdf5")
    h5.write(p3, di)
    with H.File(p3) as f:
        assert int(f["data_info"]["num_answers"].value) == 30
        _check_tree(f, di)


@needs_libhdf5
def test_reads_many_entries_and_chunked_compressed_files_of_libhdf5(tmp_path):
    """More group entries than one symbol-table node holds (B-tree with several leaves) and, beyond what the
    reference writes, chunked + shuffled + deflated storage as other tools produce it."""
    rng = np.random.default_rng(1)
    h5 = h5ref.H5()
    tree = {"ds_%03d" % i: rng.standard_normal((3, i + 1)).astype(np.float32) for i in range(41)}
    tree["grp"] = {"x_%d" % i: np.arange(i + 2, dtype=np.int64) for i in range(11)}
    p = str(tmp_path / "many.h5")
    h5.write(p, tree)
    with H.File(p) as f:
        _check_tree(f, tree)
    big = {"table": rng.standard_normal((37, 29)).astype(np.float32), "ids": np.arange(1000, dtype=np.int32)}
    p2 = str(tmp_path / "chunked.h5")
    h5.write(p2, big, chunks={"table": (8, 16), "ids": (128,)}, deflate=4, shuffle=True)
    with H.File(p2) as f:
        _check_tree(f, big)
    p3 = str(tmp_path / "chunked_plain.h5")
    h5.write(p3, big, chunks={"table": (10, 10)})
    with H.File(p3) as f:
        _check_tree(f, big)


@needs_libhdf5
def test_files_we_write_are_read_by_the_real_libhdf5(tmp_path):
    rng = np.random.default_rng(2)
    h5 = h5ref.H5()
    tree = _feature_tree(rng)
    tree["data_info"]["vfeat_dim"] = np.array(48, np.int32)
    tree["many"] = {"k%02d" % i: rng.standard_normal(5).astype(np.float64) for i in range(20)}   # > one leaf node
    p = str(tmp_path / "ours.hdf5")
    H.write(p, tree)
    for name, dt in [("image_features", "<f4"), ("normal_boxes", "<f4"), ("spatial_features", "<f4"), ("num_boxes", "<i4"),
                     ("data_info/vfeat_dim", "<i4"), ("data_info/max_box_num", "<i8")]:
        want = tree
        for part in name.split("/"):
            want = want[part]
        got = h5.read(p, name, dt)
        np.testing.assert_array_equal(got, np.asarray(want))
        assert got.shape == np.asarray(want).shape
    for k, v in tree["many"].items():
        np.testing.assert_array_equal(h5.read(p, "many/" + k, "<f8"), v)
    s = h5.read(p, "data_info/pretrained_param_path", "S")
    assert bytes(s[()]).rstrip(b"\0") == b"bottom_up_attention_36"


def test_round_trip_and_superblock_bytes(tmp_path):
    rng = np.random.default_rng(3)
    tree = _feature_tree(rng)
    tree["empty"] = np.zeros((0, 4), np.float32)
    tree["nested"] = {"deeper": {"x": np.arange(6, dtype=np.int64).reshape(2, 3), "f64": np.float64(2.5)}}
    p = str(tmp_path / "rt.hdf5")
    H.write(p, tree)
    raw = open(p, "rb").read()
    # format signature and superblock v0 fields (HDF5 File Format Specification, "Disk Format: Level 0A")
    assert raw[:8] == b"\x89HDF\r\n\x1a\n"
    assert raw[8] == 0 and raw[9] == 0 and raw[10] == 0 and raw[12] == 0          # the four version numbers
    assert raw[13] == 8 and raw[14] == 8                                           # size of offsets / lengths
    leaf_k, internal_k = struct.unpack_from("<HH", raw, 16)
    assert (leaf_k, internal_k) == (4, 16)
    base, free, eof, drv = struct.unpack_from("<QQQQ", raw, 24)
    assert base == 0 and free == H.UNDEF and drv == H.UNDEF and eof == len(raw)
    name_off, root_hdr, cache_type, _ = struct.unpack_from("<QQII", raw, 56)
    assert name_off == 0 and cache_type == 1 and root_hdr % 8 == 0
    btree, heap = struct.unpack_from("<QQ", raw, 80)
    assert raw[btree:btree + 4] == b"TREE" and raw[heap:heap + 4] == b"HEAP"
    assert raw[root_hdr] == 1                                                      # version-1 object header
    with H.File(p) as f:
        _check_tree(f, tree)
        assert f["nested/deeper/x"].shape == (2, 3) and float(f["/nested/deeper/f64"][()]) == 2.5
        assert "image_features" in f and "nope" not in f and f.get("nope") is None
        with pytest.raises(KeyError):
            f["data_info/nope"]
    assert H.load_tree(p)["nested"]["deeper"]["x"].tolist() == [[0, 1, 2], [3, 4, 5]]


def test_rejects_what_it_does_not_understand(tmp_path):
    p = str(tmp_path / "not.hdf5")
    with open(p, "wb") as f:
        f.write(b"PK\x03\x04" + b"\0" * 200)
    assert not H.is_hdf5(p)
    with pytest.raises(H.Hdf5FormatError, match="not an HDF5 file"):
        H.File(p)
    q = str(tmp_path / "v2.hdf5")
    H.write(q, {"a": np.arange(3)})
    raw = bytearray(open(q, "rb").read())
    raw[8] = 2                                                                     # pretend libver='latest'
    open(q, "wb").write(bytes(raw))
    with pytest.raises(H.Hdf5FormatError, match="superblock version 2"):
        H.File(q)
    with pytest.raises(H.Hdf5FormatError):
        H.write(str(tmp_path / "bad.hdf5"), {"o": np.array([object()])})


def _writers():
    """ways to produce a reference-format file: our writer, and the real libhdf5 (as h5py would) when present"""
    out = [("hdf5_io", H.write)]
    if h5ref.load() is not None:
        out.append(("libhdf5", h5ref.H5().write))
    return out


@pytest.mark.parametrize("who,writer", _writers())
def test_package_entry_points_read_the_reference_files(tmp_path, who, writer):
    """model_vlmap_answer.load_image_features / load_word_weight_dir and input_ops_vqa.read_num_answers on files in
    the reference's layout (vqa/model_vlmap_answer.py:59-70, vlmap/modules.py:598-601,
    vqa/datasets/input_ops_vqa_tf_record_memft.py:13-15)."""
    import pickle
    from vqa_transfer_externaldata_amd import input_ops_vqa, model_vlmap_answer as MV
    rng = np.random.default_rng(4)
    tree = _feature_tree(rng, N=5, R=36, D=32)
    p = str(tmp_path / "vfeat_bottomup_36_my.hdf5")
    writer(p, tree)
    feats, spat, boxes, nb, max_box, dim = MV.load_image_features(p)
    assert (max_box, dim) == (36, 32) and feats.shape == (5, 36, 32) and feats.dtype == np.float32
    np.testing.assert_array_equal(np.asarray(feats), tree["image_features"])
    np.testing.assert_array_equal(np.asarray(spat), tree["spatial_features"])
    np.testing.assert_array_equal(np.asarray(boxes), tree["normal_boxes"])
    np.testing.assert_array_equal(nb, tree["num_boxes"])
    # the .npz alternative and the error for anything else
    q = str(tmp_path / "feats.npz")
    np.savez(q, max_box_num=36, vfeat_dim=32, **{k: v for k, v in tree.items() if k != "data_info"})
    assert MV.load_image_features(q)[0].shape == (5, 36, 32)
    bad = tmp_path / "feats.bin"
    bad.write_bytes(b"\0" * 64)
    with pytest.raises(ValueError, match="neither an HDF5 file nor"):
        MV.load_image_features(str(bad))
    # word weights
    wd = tmp_path / "word_weights_model-4001"
    wd.mkdir()
    cw, cb = rng.standard_normal((64, 9)).astype(np.float32), rng.standard_normal(9).astype(np.float32)
    writer(str(wd / "weights.hdf5"), {"class_weights": cw, "class_biases": cb, "v_word": np.zeros((3, 4), np.float32)})
    ad = {"vocab": ["a%d" % i for i in range(9)], "dict": {"a%d" % i: i for i in range(9)}}
    pickle.dump(ad, open(wd / "answer_dict.pkl", "wb"), protocol=2)
    ww = MV.load_word_weight_dir(str(wd))
    np.testing.assert_array_equal(ww["class_weights"], cw)
    np.testing.assert_array_equal(ww["class_biases"], cb)
    w, b = MV.word_weight_answer_init({"vocab": ["a7", "unseen"]}, 64, ww)
    np.testing.assert_array_equal(w[:, 0], cw[:, 7])
    assert b[0] == cb[7] and b[1] == -100.0
    # data_info.hdf5
    writer(str(tmp_path / "data_info.hdf5"), {"data_info": {"num_answers": np.array(3000, np.int32),
                                                            "max_ans_len": np.array(4, np.int32)}})
    assert input_ops_vqa.read_num_answers(str(tmp_path)) == 3000


def test_extractor_and_export_write_the_reference_layout(tmp_path):
    import torch
    from vqa_transfer_externaldata_amd import model_vlmap_answer as MV, pretrain as PT, vfeat as VF

    class FakeModel:                       # stands in for the HIP conv stack: the file layout is what is under test
        def build(self, batch):
            return batch["image"].mean(dim=(1, 2))[:, None, :].repeat(1, batch["normal_box"].shape[1], 4)

    rng = np.random.default_rng(5)
    ids = ["COCO_%d.jpg" % i for i in range(3)]
    nbx = np.sort(rng.random((3, 4, 4)).astype(np.float32), -1)
    batches = [{"image": torch.rand(3, 8, 8, 3), "normal_box": torch.from_numpy(nbx), "num_box": [4, 3, 4], "image_id": ids}]
    path = str(tmp_path / "used_vfeat.hdf5")
    out = VF.Extractor(FakeModel(), {k: i for i, k in enumerate(ids)}, 4, "data/nets/resnet_v1_50.ckpt").extract(batches, path)
    feats, spat, boxes, nb, max_box, dim = MV.load_image_features(path)
    assert (max_box, dim) == (4, 12)
    np.testing.assert_array_equal(np.asarray(feats), out["image_features"])
    np.testing.assert_array_equal(np.asarray(spat), out["spatial_features"])
    assert np.all(np.asarray(feats)[1, 3] == 0)                                  # image 1 has 3 boxes: row 3 stays zero
    with H.File(path) as f:
        assert f["data_info"]["pretrained_param_path"][()] == "data-nets-resnet_v1_50.ckpt"
    sd = {k: np.full(s, i, np.float32) for i, (k, s) in enumerate(PT.variable_shapes(11, 5, 7, 8, 16, 4).items())}
    d = PT.export_word_weights(sd, {"vocab": ["w"]}, {"vocab": ["a%d" % i for i in range(7)],
                                                      "dict": {"a%d" % i: i for i in range(7)}},
                               str(tmp_path / "word_weights_model-1"))
    got = H.load_tree(os.path.join(d, "weights.hdf5"))
    assert sorted(got) == ["class_biases", "class_weights", "l_answer_word", "l_word", "v_word"]
    np.testing.assert_array_equal(got["class_weights"], sd["classifier/fc/weights"])
    assert MV.load_word_weight_dir(d)["class_biases"].shape == (7,)


def test_create_streams_rows_into_a_complete_file(tmp_path):
    """hdf5_io.create: Empty placeholders are allocated as holes, the file is readable at once, rows written through the
    memmaps appear in it; libhdf5 reads the result when the image has it"""
    p = str(tmp_path / "stream.hdf5")
    mm = H.create(p, {"image_features": H.Empty((5, 3, 4)), "num_boxes": np.arange(5, dtype=np.int32),
                      "data_info": {"vfeat_dim": np.array(4, np.int32), "ids": H.Empty((2,), np.int32)}})
    assert sorted(mm) == ["/data_info/ids", "/image_features"]
    t = H.load_tree(p)
    assert np.all(np.asarray(t["image_features"]) == 0) and t["image_features"].shape == (5, 3, 4)
    mm["/image_features"][2] = 7.0
    mm["/image_features"][4, 1] = [1, 2, 3, 4]
    mm["/data_info/ids"][:] = [3, 4]
    for m in mm.values():
        m.flush()
    t = H.load_tree(p)
    want = np.zeros((5, 3, 4), np.float32); want[2] = 7; want[4, 1] = [1, 2, 3, 4]
    np.testing.assert_array_equal(np.asarray(t["image_features"]), want)
    np.testing.assert_array_equal(np.asarray(t["data_info"]["ids"]), [3, 4])
    np.testing.assert_array_equal(np.asarray(t["num_boxes"]), np.arange(5))
    if h5ref.load() is not None:
        np.testing.assert_array_equal(h5ref.H5().read(p, "image_features", "f4").reshape(5, 3, 4), want)


def test_sharded_extraction_parts_merge_into_the_single_run_table(tmp_path):
    """Extractor(part=(rank, world)) + merge_parts: three ranks over 7 images (3 + 2 + 2, no collective on the data path)
    give the file one process writes, num_boxes quirk included (every entry = the first image's count, :118-121)"""
    import torch
    from vqa_transfer_externaldata_amd import vfeat as VF, vfeat_extractor as VX

    class FakeModel:
        def build(self, batch):
            return batch["image"].mean(dim=(1, 2))[:, None, :].repeat(1, batch["normal_box"].shape[1], 4)

    rng = np.random.default_rng(9)
    N, R = 7, 4
    ids = ["COCO_%d.jpg" % i for i in range(N)]
    id2idx = {k: (i * 3) % N for i, k in enumerate(ids)}                         # a non-trivial id -> row map
    img = torch.from_numpy(rng.random((N, 8, 8, 3)).astype(np.float32))
    nbx = torch.from_numpy(np.sort(rng.random((N, R, 4)).astype(np.float32), -1))
    nums = [3, 4, 2, 4, 1, 4, 3]

    def batches(lo, hi, bs=2):
        for a in range(lo, hi, bs):
            b = min(a + bs, hi)
            yield {"image": img[a:b], "normal_box": nbx[a:b], "num_box": nums[a:b], "image_id": ids[a:b]}

    single = str(tmp_path / "single.hdf5")
    VF.Extractor(FakeModel(), id2idx, R, "w.ckpt").extract(batches(0, N), single)
    sharded = str(tmp_path / "sharded.hdf5")
    bounds = [VX.shard_of(N, r, 3) for r in range(3)]
    assert bounds == [(0, 3), (3, 5), (5, 7)]
    for r, (lo, hi) in enumerate(bounds):
        out = VF.Extractor(FakeModel(), id2idx, R, "w.ckpt").extract(batches(lo, hi), sharded, part=(r, 3), n_part_rows=hi - lo)
        assert out["image_features"].shape[0] == hi - lo and sorted(out["image_idx"]) == sorted(id2idx[i] for i in ids[lo:hi])
        assert os.path.exists(VF.Extractor.part_path(sharded, r, 3))
    VF.Extractor.merge_parts(sharded, 3, N)
    assert not os.path.exists(VF.Extractor.part_path(sharded, 0, 3))
    a, b = H.load_tree(single), H.load_tree(sharded)
    for k in ("image_features", "normal_boxes", "spatial_features", "num_boxes"):
        np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)
    assert np.all(np.asarray(b["num_boxes"]) == 3)
    for k in ("vfeat_dim", "max_box_num"):
        assert int(np.asarray(a["data_info"][k])) == int(np.asarray(b["data_info"][k]))
    with H.File(sharded) as f:
        assert f["data_info"]["pretrained_param_path"][()] == "w.ckpt"


def test_sharded_extraction_with_more_ranks_than_images_and_partial_files(tmp_path):
    """world > images: ranks with an empty shard write no part and merge_parts skips them; tables are streamed into
    `<path>.partial` and only a finished one carries the final name"""
    import torch
    from vqa_transfer_externaldata_amd import vfeat as VF, vfeat_extractor as VX

    class FakeModel:
        def __init__(self, fail_at=None):
            self.calls, self.fail_at = 0, fail_at

        def build(self, batch):
            self.calls += 1
            if self.fail_at is not None and self.calls > self.fail_at:
                raise RuntimeError("GPU went away")
            return batch["image"].mean(dim=(1, 2))[:, None, :].repeat(1, batch["normal_box"].shape[1], 2)

    rng = np.random.default_rng(4)
    N, R, world = 2, 3, 4
    ids = ["i%d.jpg" % i for i in range(N)]
    id2idx = {k: i for i, k in enumerate(ids)}
    img = torch.from_numpy(rng.random((N, 4, 4, 3)).astype(np.float32))
    nbx = torch.from_numpy(np.sort(rng.random((N, R, 4)).astype(np.float32), -1))

    def batches(lo, hi):
        for a in range(lo, hi):
            yield {"image": img[a:a + 1], "normal_box": nbx[a:a + 1], "num_box": [R], "image_id": ids[a:a + 1]}

    single = str(tmp_path / "single.hdf5")
    VF.Extractor(FakeModel(), id2idx, R).extract(batches(0, N), single)
    assert os.path.exists(single) and not os.path.exists(single + ".partial")
    sharded = str(tmp_path / "sharded.hdf5")
    for r in range(world):
        lo, hi = VX.shard_of(N, r, world)
        out = VF.Extractor(FakeModel(), id2idx, R).extract(batches(lo, hi), sharded, part=(r, world), n_part_rows=hi - lo)
        assert os.path.exists(VF.Extractor.part_path(sharded, r, world)) == (hi > lo)
        assert len(out["image_idx"]) == hi - lo
    VF.Extractor.merge_parts(sharded, world, N)
    a, b = H.load_tree(single), H.load_tree(sharded)
    for k in ("image_features", "normal_boxes", "spatial_features", "num_boxes"):
        np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)
    # a run that dies half-way leaves only the .partial file
    dead = str(tmp_path / "dead.hdf5")
    with pytest.raises(RuntimeError, match="went away"):
        VF.Extractor(FakeModel(fail_at=1), id2idx, R).extract(batches(0, N), dead)
    assert not os.path.exists(dead)


def _random_tree(rng, depth=0):
    dts = [np.float32, np.float64, np.int32, np.int64, np.uint8, np.int8, np.int16, np.uint16, np.uint32]
    tree = {}
    for i in range(int(rng.integers(1, 6))):
        name = "n%d_%s" % (i, "".join(rng.choice(list("abcXYZ_09"), size=int(rng.integers(1, 9)))))
        kind = rng.integers(0, 10)
        if kind < 2 and depth < 3:
            tree[name] = _random_tree(rng, depth + 1)
        elif kind == 2:
            tree[name] = "".join(rng.choice(list("abc /-_.0123"), size=int(rng.integers(1, 30))))
        else:
            dt = dts[int(rng.integers(0, len(dts)))]
            shape = tuple(int(x) for x in rng.integers(0, 7, size=int(rng.integers(0, 4))))
            a = rng.integers(-100, 100, size=shape) if np.issubdtype(dt, np.integer) else rng.standard_normal(shape) * 10
            if np.issubdtype(dt, np.unsignedinteger):
                a = np.abs(a)
            tree[name] = np.asarray(a).astype(dt)
    return tree


@pytest.mark.parametrize("seed", range(12))
def test_random_trees_round_trip_and_are_read_by_libhdf5(tmp_path, seed):
    """seeded random trees (nested groups, 9 dtypes, 0-d to 3-d shapes with empty axes, strings): our writer ->
    our reader value for value and dtype for dtype; every numeric dataset is also read back by the real libhdf5"""
    rng = np.random.default_rng(1000 + seed)
    tree = _random_tree(rng)
    p = str(tmp_path / "r.hdf5")
    H.write(p, tree)
    with H.File(p) as f:
        _check_tree(f, tree)

        def walk(t, pre=""):
            for k, v in t.items():
                if isinstance(v, dict):
                    yield from walk(v, pre + k + "/")
                elif not isinstance(v, str):
                    yield pre + k, v
        for name, want in walk(tree):
            assert np.asarray(f[name]).dtype == want.dtype, name
    h5 = h5ref.H5() if h5ref.load() is not None else None
    if h5 is not None:
        for name, want in walk(tree):
            if want.size == 0:
                continue
            got = h5.read(p, name, want.dtype.newbyteorder("<").str)
            np.testing.assert_array_equal(got, want, err_msg=name)
