"""Region-feature extractor on MI355X: slim-style ResNet-v1 bottleneck stack + ROI crop
(SURVEY rows a13-a16).

Counterpart of modules.encode_I_block3 / encode_I_full / roi_pool / I_reduce_dim / I2V
(vlmap/modules.py:143-239) and of vqa/model_vfeat.py, vqa/model_vfeat_resnet.py.  Inference
only, as in the reference extraction path (is_training=False -> BatchNorm uses moving
statistics, folded here to per-channel scale/shift).  Variable names are slim's
(`resnet_v1_50/block1/unit_1/bottleneck_v1/conv1/weights`, `.../BatchNorm/gamma` ...), the
contract of data/nets/resnet_v1_50.ckpt.  Layout NHWC, fp32, convolutions on the f32 MFMA.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib

ENC_I_MEAN = (123.68, 116.78, 103.94)      # vlmap/modules.py:18-20
SLIM_BN_EPS = 1e-5                          # resnet_arg_scope
LAYERS_BN_EPS = 1e-3                        # tf.contrib.layers.batch_norm default (modules.conv2d)

BLOCKS_R50_B3 = [("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2)]
BLOCKS_R50_FULL = BLOCKS_R50_B3 + [("block4", 512, 3, 1)]
BLOCKS_R101_FULL = [("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 23, 2), ("block4", 512, 3, 1)]


def block_units(base_depth, num_units, stride):
    """slim resnet_v1_block: stride on the LAST unit."""
    return [(base_depth * 4, base_depth, 1)] * (num_units - 1) + [(base_depth * 4, base_depth, stride)]


def init_random_params(rng, blocks, scope="resnet_v1_50", dtype=np.float32):
    """Random-init weights of the architecture under slim's variable names (He-normal convs, BatchNorm
    gamma ~ 1, small beta / moving_mean, moving_variance ~ 1): stands in for data/nets/resnet_v1_50.ckpt,
    which is download-only."""
    p = {}

    def conv(name, k, ci, co):
        p[name + "/weights"] = (rng.standard_normal((k, k, ci, co)) * np.sqrt(2.0 / (k * k * ci))).astype(dtype)
        p[name + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(co)).astype(dtype)
        p[name + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(co)).astype(dtype)
        p[name + "/BatchNorm/moving_mean"] = (0.1 * rng.standard_normal(co)).astype(dtype)
        p[name + "/BatchNorm/moving_variance"] = (1 + 0.2 * rng.random(co)).astype(dtype)

    conv(scope + "/conv1", 7, 3, 64)
    cin = 64
    for name, base, n, stride in blocks:
        for i, (depth, db, s) in enumerate(block_units(base, n, stride)):
            pre = "%s/%s/unit_%d/bottleneck_v1" % (scope, name, i + 1)
            if depth != cin:
                conv(pre + "/shortcut", 1, cin, depth)
            conv(pre + "/conv1", 1, cin, db)
            conv(pre + "/conv2", 3, db, db)
            conv(pre + "/conv3", 1, db, depth)
            cin = depth
    return p


def init_random_head_params(rng, p, enc_dim, v_dim, dtype=np.float32):
    """random-init I_reduce_dim / I2V variables of model_vfeat (vlmap/modules.py:219-239; layers.conv2d + batch_norm)"""
    for scope, k, ci in (("I_reduce_dim/conv2d", 1, enc_dim), ("I2V/conv2d_1", 3, v_dim)):
        p[scope + "/conv2d/weights"] = (rng.standard_normal((k, k, ci, v_dim)) * np.sqrt(2.0 / (k * k * ci))).astype(dtype)
        p[scope + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(v_dim)).astype(dtype)
        p[scope + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(v_dim)).astype(dtype)
        p[scope + "/BatchNorm/moving_mean"] = (0.1 * rng.standard_normal(v_dim)).astype(dtype)
        p[scope + "/BatchNorm/moving_variance"] = (1 + 0.2 * rng.random(v_dim)).astype(dtype)
    return p


def conv_flops_per_image(blocks, H, W, c0=64):
    """Algorithmic 2*MAC of conv1 + the bottleneck blocks for one HxW image."""
    fl = 0
    h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    fl += 2 * h * w * 49 * 3 * c0
    h, w = (h + 1) // 2, (w + 1) // 2
    cin = c0
    for name, base, n, stride in blocks:
        for depth, db, s in block_units(base, n, stride):
            ho, wo = ((h - 1) // s + 1, (w - 1) // s + 1) if s > 1 else (h, w)
            if depth != cin:
                fl += 2 * ho * wo * cin * depth
            fl += 2 * h * w * cin * db + 2 * ho * wo * 9 * db * db + 2 * ho * wo * db * depth
            h, w, cin = ho, wo, depth
    return fl


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class ConvBN:
    """conv weights (HWIO) + BatchNorm folded to scale/shift, resident on the device."""

    def __init__(self, params, prefix, eps, device, wname="/weights", bn="/BatchNorm/", pad_k_to=None, pad_taps_to=None):
        w = np.asarray(params[prefix + wname], np.float32)
        if pad_taps_to is not None:       # zero taps / channels: [kh,kw,ci,co] -> [KH,KW,CI,co]
            wp = np.zeros(tuple(pad_taps_to) + (w.shape[3],), np.float32)
            wp[:w.shape[0], :w.shape[1], :w.shape[2]] = w
            w = wp
        self.kh, self.kw, self.ci, self.co = w.shape
        g, b, m, v = (np.asarray(params[prefix + bn + k], np.float64)
                      for k in ("gamma", "beta", "moving_mean", "moving_variance"))
        scale = g / np.sqrt(v + eps)
        shift = b - m * scale
        w2 = w.reshape(self.kh * self.kw * self.ci, self.co)
        if pad_k_to is not None and pad_k_to > w2.shape[0]:
            w2 = np.concatenate([w2, np.zeros((pad_k_to - w2.shape[0], self.co), np.float32)], 0)
        self.w = torch.from_numpy(np.ascontiguousarray(w2)).to(device)
        self.scale = torch.from_numpy(scale.astype(np.float32)).to(device)
        self.shift = torch.from_numpy(shift.astype(np.float32)).to(device)


def conv2d(x, cb, stride=1, pad=(0, 0), out_hw=None, residual=None, relu=True):
    """y = [relu](conv(x)*scale + shift [+ residual]) through vqa_conv2d_nhwc."""
    lib = _lib.load()
    B, Hi, Wi, Ci = x.shape
    assert Ci * cb.kh * cb.kw <= cb.w.shape[0] and x.is_contiguous()
    Ho, Wo = out_hw if out_hw is not None else (Hi, Wi)
    y = torch.empty(B, Ho, Wo, cb.co, dtype=torch.float32, device=x.device)
    _lib.check(lib.vqa_conv2d_nhwc(_p(x), B, Hi, Wi, Ci, _p(cb.w), cb.kh, cb.kw, cb.co, stride, pad[0], pad[1], Ho, Wo,
                                   _p(cb.scale), _p(cb.shift), _p(residual), int(relu), _p(y), _st(x)),
               "vqa_conv2d_nhwc")
    return y


def conv2d_backward(x, cb, y, dy, stride=1, pad=(0, 0), relu=True, need_dx=True, need_dw=True, chunk_images=None):
    """Gradients of conv2d() (vqa_conv2d_nhwc_bwd; the reference's only consumer is the legacy CNN fine-tune,
    vlmap/model_vlmap.py:675-690): returns (dx, dw [kh*kw*Ci, Co], dshift [Co], dresidual) for dy = d loss / d y.
    chunk_images bounds the im2col scratch (default: as many images at once as 1 GiB of scratch holds)."""
    lib = _lib.load()
    B, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    K = cb.kh * cb.kw * Ci
    if chunk_images is None:
        chunk_images = max(1, min(B, int((1 << 28) // max(1, 2 * Ho * Wo * K))))
    n = int(lib.vqa_conv2d_bwd_workspace_floats(B, Ho, Wo, Ci, cb.kh, cb.kw, Co, chunk_images))
    if n <= 0:
        raise _lib.VqaHotError("vqa_conv2d_bwd_workspace_floats rejected the shape")
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty(K, Co, dtype=torch.float32, device=x.device) if need_dw else None
    dshift = torch.empty(Co, dtype=torch.float32, device=x.device)
    dres = torch.empty_like(dy)
    w = cb.w[:K]                                       # (conv1's rows beyond kh*kw*Ci are zero padding of the forward)
    _lib.check(lib.vqa_conv2d_nhwc_bwd(_p(x), B, Hi, Wi, Ci, _p(w), cb.kh, cb.kw, Co, stride, pad[0], pad[1], Ho, Wo,
                                       _p(cb.scale), _p(y), int(relu), _p(dy.contiguous()), _p(dx), _p(dw), _p(dshift),
                                       _p(dres), _p(ws), ws.numel(), _st(x)), "vqa_conv2d_nhwc_bwd")
    return dx, dw, dshift, dres


def max_pool_3x3_s2_same(x):
    lib = _lib.load()
    B, H, W, Cc = x.shape
    y = torch.empty(B, (H + 1) // 2, (W + 1) // 2, Cc, dtype=torch.float32, device=x.device)
    _lib.check(lib.vqa_maxpool3x3s2_same_nhwc(_p(x), B, H, W, Cc, _p(y), _st(x)), "vqa_maxpool3x3s2_same_nhwc")
    return y


def subsample(x, factor):
    if factor == 1:
        return x
    lib = _lib.load()
    B, H, W, Cc = x.shape
    y = torch.empty(B, (H - 1) // factor + 1, (W - 1) // factor + 1, Cc, dtype=torch.float32, device=x.device)
    _lib.check(lib.vqa_subsample_nhwc(_p(x), B, H, W, Cc, factor, _p(y), _st(x)), "vqa_subsample_nhwc")
    return y


def roi_pool(ftmap, box, height, width):
    """modules.roi_pool (vlmap/modules.py:204-216): box [B,n,4] normalised [y1,x1,y2,x2]."""
    lib = _lib.load()
    B, H, W, Cc = ftmap.shape
    n = box.shape[1]
    boxes = box.reshape(-1, 4).contiguous().float()
    ids = torch.arange(B, device=ftmap.device, dtype=torch.int32).repeat_interleave(n).contiguous()
    out = torch.empty(B * n, height, width, Cc, dtype=torch.float32, device=ftmap.device)
    _lib.check(lib.vqa_crop_and_resize_nhwc(_p(ftmap), B, H, W, Cc, _p(boxes), _p(ids), B * n, height, width, _p(out),
                                            _st(ftmap)), "vqa_crop_and_resize_nhwc")
    return out.view(B, n, height, width, Cc)


class ResNetV1:
    """modules.encode_I_block3 / encode_I_full: mean subtraction + conv1 + pool1 + bottleneck blocks."""

    def __init__(self, params, blocks=BLOCKS_R50_B3, scope="resnet_v1_50", device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.VqaHotError("ResNetV1 needs a GPU (no CPU fallback)")
        self.device = torch.device(device)
        self.blocks = blocks
        self.conv1 = ConvBN(params, scope + "/conv1", SLIM_BN_EPS, self.device, pad_taps_to=(7, 8, 4))   # [7,7,3,Co] -> [7,8,4,Co]
        self.units = []
        cin = self.conv1.co
        for name, base, n, stride in blocks:
            for i, (depth, db, s) in enumerate(block_units(base, n, stride)):
                pre = "%s/%s/unit_%d/bottleneck_v1" % (scope, name, i + 1)
                u = {"depth": depth, "stride": s,
                     "shortcut": ConvBN(params, pre + "/shortcut", SLIM_BN_EPS, self.device) if depth != cin else None,
                     "conv1": ConvBN(params, pre + "/conv1", SLIM_BN_EPS, self.device),
                     "conv2": ConvBN(params, pre + "/conv2", SLIM_BN_EPS, self.device),
                     "conv3": ConvBN(params, pre + "/conv3", SLIM_BN_EPS, self.device)}
                self.units.append(u)
                cin = depth
        self.out_channels = cin
        self._mean = (C.c_float * 3)(*ENC_I_MEAN)

    def stem(self, images):
        lib = _lib.load()
        B, H, W, Cc = images.shape
        assert Cc == 3 and images.dtype == torch.float32 and images.is_contiguous()
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1          # conv2d_same(7, stride 2): pad 3/3, VALID
        # conv1 as an implicit GEMM: mean-subtracted RGB padded to 16-byte pixels (the zero padding of the
        # convolution is the loader's out-of-range zero, so the mean is only subtracted inside the image), filter
        # rows padded from 7 to 8 taps -> one k tile = one filter row, K = 7 * 8 * 4 = 224
        x4 = torch.empty(B, H, W, 4, dtype=torch.float32, device=images.device)
        _lib.check(lib.vqa_pad_c3c4_nhwc(_p(images), B, H, W, self._mean, _p(x4), _st(images)), "vqa_pad_c3c4_nhwc")
        y = conv2d(x4, self.conv1, stride=2, pad=(3, 3), out_hw=(Ho, Wo), relu=True)
        return max_pool_3x3_s2_same(y)

    def bottleneck(self, x, u):
        s = u["stride"]
        B, H, W, _ = x.shape
        Ho, Wo = ((H - 1) // s + 1, (W - 1) // s + 1) if s > 1 else (H, W)
        if u["shortcut"] is None:
            shortcut = subsample(x, s)
        else:
            shortcut = conv2d(x, u["shortcut"], stride=s, out_hw=(Ho, Wo), relu=False)
        r = conv2d(x, u["conv1"], relu=True)
        r = conv2d(r, u["conv2"], stride=s, pad=(1, 1), out_hw=(Ho, Wo), relu=True)    # conv2d_same(3, s)
        return conv2d(r, u["conv3"], residual=shortcut, relu=True)

    def __call__(self, images):
        x = self.stem(images)
        for u in self.units:
            x = self.bottleneck(x, u)
        return x


class VfeatResnetModel:
    """vqa/model_vfeat_resnet.py:28-40: conv map -> 1x1 crop_and_resize -> outputs['V_ft'] [B,n_box,C]."""

    def __init__(self, params, blocks=BLOCKS_R50_B3, device="cuda:0"):
        self.net = ResNetV1(params, blocks, device=device)
        self.outputs = {}

    def build(self, batch):
        enc = self.net(batch["image"])
        nb = batch["normal_box"].shape[1]
        self.outputs["enc_I"] = enc
        self.outputs["V_ft"] = roi_pool(enc, batch["normal_box"], 1, 1).view(-1, nb, enc.shape[3])
        return self.outputs["V_ft"]


class VfeatModel:
    """vqa/model_vfeat.py:28-51: block3 -> I_reduce_dim -> 5x5 ROI crop -> I2V (two 3x3 VALID convs sharing
    one weight tensor and one BatchNorm, vlmap/modules.py:232-237) -> [B,n_box,512]."""

    ROI_SZ = 5

    def __init__(self, params, blocks=BLOCKS_R50_B3, device="cuda:0"):
        self.net = ResNetV1(params, blocks, device=device)
        dev = self.net.device
        self.reduce = ConvBN(params, "I_reduce_dim/conv2d", LAYERS_BN_EPS, dev, wname="/conv2d/weights")
        self.i2v = ConvBN(params, "I2V/conv2d_1", LAYERS_BN_EPS, dev, wname="/conv2d/weights")
        self.outputs = {}

    def build(self, batch):
        enc = self.net(batch["image"])
        low = conv2d(enc, self.reduce, relu=True)
        B, nb = batch["normal_box"].shape[:2]
        roi = roi_pool(low, batch["normal_box"], self.ROI_SZ, self.ROI_SZ)
        flat = roi.view(B * nb, self.ROI_SZ, self.ROI_SZ, self.reduce.co)
        v = conv2d(flat, self.i2v, out_hw=(3, 3), relu=True)            # VALID 5x5 -> 3x3
        v = conv2d(v, self.i2v, out_hw=(1, 1), relu=True)               # VALID 3x3 -> 1x1, same weights
        self.outputs["V_ft"] = v.view(B, nb, self.i2v.co)
        return self.outputs["V_ft"]


def spatial_features(normal_box):
    """The 6-d box feature the extractor stores (vqa/vfeat_extractor_tf_record_memft.py:132-139).  The
    reference labels column 0 'x1' although normal_box is [y1,x1,y2,x2]; the stored tuple is therefore
    (b0, b1, b2, b3, b2-b0, b3-b1) -- reproduced as is."""
    b = normal_box
    return np.stack([b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1)


class Extractor:
    """vqa/vfeat_extractor_tf_record_memft.py:26-147: run the model over batches and fill dense
    [N, max_roi, D] arrays (image_features, normal_boxes, spatial_features, num_boxes, data_info).  save_path
    ending in .hdf5 / .h5 writes the reference's HDF5 layout (hdf5_io, no h5py: the four datasets + group data_info
    with max_box_num, vfeat_dim, pretrained_param_path); any other name writes an .npz with the same keys.

    HDF5 output is STREAMED like the reference's h5py datasets (:118-139): the file is created with the three tables
    as holes when the first batch comes back, and every batch's rows go straight into np.memmap views of the file --
    the 36 GB table of the real feature set is never held in host RAM, and rows extracted before a failure are on disk.
    `part=(rank, world)`: this process extracts a SHARD of the images (feature extraction shards by image with no
    collective); it writes a compact part file `<save_path>.part<rank>of<world>` (only its rows + their
    image indices) that `merge_parts` folds into the final table."""

    def __init__(self, model, image_id2idx, max_roi_num, pretrained_param_path="random_init"):
        self.model, self.image_id2idx, self.max_roi_num = model, image_id2idx, max_roi_num
        self.pretrained_param_path = pretrained_param_path

    @staticmethod
    def part_path(save_path, rank, world):
        return "%s.part%dof%d" % (save_path, rank, world)

    def extract(self, batches, save_path=None, part=None, n_part_rows=None):
        """One batch in flight: while the GPU runs batch i, the host writes the rows of batch i-1 (its features come back
        through an asynchronous copy into pinned memory, awaited through an event) and the loader threads decode batch
        i+1 -- the reference's loop is session.run, then the Python row loop, serially (:99-139).
        part / n_part_rows: shard mode, see the class docstring (n_part_rows = number of images this shard will see)."""
        from . import hdf5_io
        N = len(self.image_id2idx)
        stream = save_path is not None and save_path.endswith((".hdf5", ".h5"))
        if part is not None and not stream:
            raise ValueError("sharded extraction writes HDF5 parts: save_path must end in .hdf5 / .h5")
        rows_total = N if part is None else int(n_part_rows)
        state = {"feats": None, "boxes": None, "spat": None, "next": 0, "first_n": None}
        num_boxes = np.zeros([N], np.int32)
        part_idx = np.full([rows_total], -1, np.int32) if part is not None else None
        pool = {}

        def pinned(t, slot):
            key = (slot, tuple(t.shape), t.dtype)
            if key not in pool:
                pool[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            return pool[key]

        def data_info(D):
            return {"pretrained_param_path": self.pretrained_param_path.replace("/", "-"),
                    "max_box_num": np.array(self.max_roi_num, np.int32), "vfeat_dim": np.array(D, np.int32)}

        def init_tables(D, n):
            state["first_n"] = n
            num_boxes[:] += n                            # reference initialises every entry to the first n
            if not stream:
                state["feats"] = np.zeros((rows_total, self.max_roi_num, D), np.float32)
                state["boxes"] = np.zeros((rows_total, self.max_roi_num, 4), np.float32)
                state["spat"] = np.zeros((rows_total, self.max_roi_num, 6), np.float32)
                return
            tree = {"image_features": hdf5_io.Empty((rows_total, self.max_roi_num, D)),
                    "normal_boxes": hdf5_io.Empty((rows_total, self.max_roi_num, 4)),
                    "spatial_features": hdf5_io.Empty((rows_total, self.max_roi_num, 6)),
                    "data_info": data_info(D)}
            if part is None:
                tree["num_boxes"] = num_boxes.copy()
                path = save_path
            else:
                tree["image_idx"] = hdf5_io.Empty((rows_total,), np.int32)
                tree["first_num_box"] = np.array(n, np.int32)
                path = self.part_path(save_path, *part)
            # streamed into `<path>.partial`, renamed when the last row is on disk: a run that dies half-way never leaves
            # something a trainer could take for a finished table
            state["final_path"] = path
            mm = hdf5_io.create(path + ".partial", tree)
            state["feats"], state["boxes"], state["spat"] = mm["/image_features"], mm["/normal_boxes"], mm["/spatial_features"]
            state["idx_mm"] = mm.get("/image_idx")

        def finish(p):
            v_host, nb_host, nums, ids, ev = p
            if ev is not None:
                ev.synchronize()
            v, nbx = v_host.numpy(), nb_host.numpy()
            for b in range(v.shape[0]):
                n = min(int(nums[b]), self.max_roi_num)
                if state["feats"] is None:
                    init_tables(v.shape[2], n)
                idx = self.image_id2idx[ids[b]]
                if part is not None:                     # compact rows in arrival order + where they belong
                    part_idx[state["next"]] = idx
                    idx, state["next"] = state["next"], state["next"] + 1
                state["feats"][idx, :n] = v[b, :n]
                state["boxes"][idx, :n] = nbx[b, :n]
                state["spat"][idx, :n] = spatial_features(nbx[b, :n])

        pending, k = None, 0
        for batch in batches:
            vg = self.model.build({"image": batch["image"], "normal_box": batch["normal_box"]})
            if vg.is_cuda:
                v_host = pinned(vg, k & 1)
                v_host.copy_(vg, non_blocking=True)
                nb_host = pinned(batch["normal_box"], k & 1)
                nb_host.copy_(batch["normal_box"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:       # host tensors (the file-layout tests drive this class with a stand-in model)
                v_host, nb_host, ev = vg, batch["normal_box"].cpu(), None
            cur = (v_host, nb_host, np.array(batch["num_box"]), list(batch["image_id"]), ev)
            if pending is not None:
                finish(pending)
            pending, k = cur, k + 1
        if pending is not None:
            finish(pending)
        feats, boxes, spat = state["feats"], state["boxes"], state["spat"]
        if part is not None and state.get("idx_mm") is not None:
            state["idx_mm"][:] = part_idx
            state["idx_mm"].flush()
        if stream and feats is not None:
            for m in (feats, boxes, spat):
                m.flush()
            os.replace(state["final_path"] + ".partial", state["final_path"])      # the mappings follow the inode
        if feats is None:
            if part is None:
                raise ValueError("no images to extract")
            return {"image_idx": part_idx, "num_boxes": num_boxes}      # an empty shard (world > images): no part file
        out = {"image_features": feats, "normal_boxes": boxes, "spatial_features": spat, "num_boxes": num_boxes,
               "max_box_num": np.int32(self.max_roi_num), "vfeat_dim": np.int32(feats.shape[2])}
        if part is not None:
            out["image_idx"] = part_idx
        if save_path is not None and not stream:
            np.savez(save_path, **out)
        return out

    @staticmethod
    def merge_parts(save_path, world, n_images, remove=True):
        """Rank 0, after every rank finished its part: the final `<save_path>` in the reference's layout, rows copied from
        the parts (memmap to memmap, one part at a time).  num_boxes follows the reference: every entry = the count of the
        FIRST image of the run (:118-121), i.e. of rank 0's first image."""
        from . import hdf5_io
        # a rank whose shard is empty (world > images) writes no part
        present = [r for r in range(world) if os.path.exists(Extractor.part_path(save_path, r, world))]
        if not present:
            raise FileNotFoundError("no part files of %s" % save_path)
        parts = [hdf5_io.File(Extractor.part_path(save_path, r, world)) for r in present]
        first = parts[0]
        _, R, D = first["image_features"].shape
        di = first["data_info"]
        ppp = di["pretrained_param_path"][()]
        ppp = ppp.decode() if isinstance(ppp, bytes) else str(ppp)
        n0 = int(np.asarray(first["first_num_box"].read()))
        mm = hdf5_io.create(save_path + ".partial", {
            "image_features": hdf5_io.Empty((n_images, R, D)), "normal_boxes": hdf5_io.Empty((n_images, R, 4)),
            "spatial_features": hdf5_io.Empty((n_images, R, 6)), "num_boxes": np.zeros([n_images], np.int32) + n0,
            "data_info": {"pretrained_param_path": ppp, "max_box_num": np.array(R, np.int32), "vfeat_dim": np.array(D, np.int32)}})
        for f in parts:
            idx = np.asarray(f["image_idx"].read())
            ok = idx >= 0
            for name in ("image_features", "normal_boxes", "spatial_features"):
                src = f[name].read()
                dst = mm["/" + name]
                for lo in range(0, len(idx), 256):         # bounded host memory per copy
                    sel = ok[lo:lo + 256]
                    dst[idx[lo:lo + 256][sel]] = np.asarray(src[lo:lo + 256])[sel]
        for m in mm.values():
            m.flush()
        os.replace(save_path + ".partial", save_path)
        for r, f in zip(present, parts):
            f.close()
            if remove:
                os.remove(Extractor.part_path(save_path, r, world))
        return save_path
