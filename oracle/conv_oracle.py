"""CPU oracle for the region-feature extractor.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference ships no fixtures for this path and its arithmetic
lives in tensorflow-gpu==1.6.0 / tf.contrib.slim (requirements.txt:1), absent here;
the slim checkpoint (data/nets/README.md:6-7) is download-only.  This NumPy
restatement follows the published slim `resnet_v1` definition and TF op semantics
 and is pinned by known-answer tests plus an independent
torch (F.conv2d / F.max_pool2d) composition in tests/test_conv_oracle.py.

Reference lines restated (relative to /root/reference):
  * encode_I_block3 / encode_I_full   vlmap/modules.py:143-191  (mean subtraction :18-20,170-174;
                                      slim resnet_v1 with resnet_arg_scope, is_training=False)
  * roi_pool (tf.image.crop_and_resize) vlmap/modules.py:204-216
  * I_reduce_dim, I2V, conv2d         vlmap/modules.py:219-239, 552-572
  * Model (vfeat / vfeat_resnet)      vqa/model_vfeat.py:28-51, vqa/model_vfeat_resnet.py:28-40
"""
from __future__ import annotations

import numpy as np

ENC_I_MEAN = (123.68, 116.78, 103.94)          # vlmap/modules.py:18-20 (R, G, B)
SLIM_BN_EPS = 1e-5                              # resnet_arg_scope(batch_norm_epsilon=1e-5)
LAYERS_BN_EPS = 1e-3                            # tf.contrib.layers.batch_norm default (modules.conv2d)


# ----------------------------------------------------------------------------- primitives
def conv2d_nhwc(x, w, stride=1, pad=((0, 0), (0, 0))):
    """x [B,H,W,Ci], w [kh,kw,Ci,Co] (HWIO, TF layout), explicit zero padding, VALID after padding."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    xp = np.pad(x, ((0, 0), pad[0], pad[1], (0, 0)))
    Ho = (xp.shape[1] - kh) // stride + 1
    Wo = (xp.shape[2] - kw) // stride + 1
    cols = np.empty((B, Ho, Wo, kh, kw, Ci), x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, :, ky, kx, :] = xp[:, ky:ky + (Ho - 1) * stride + 1:stride,
                                          kx:kx + (Wo - 1) * stride + 1:stride, :]
    return (cols.reshape(B * Ho * Wo, kh * kw * Ci) @ w.reshape(kh * kw * Ci, Co)).reshape(B, Ho, Wo, Co)


def same_pad(size, k, stride):
    """TF 'SAME': out = ceil(size/stride); the odd padding element goes at the END."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return (total // 2, total - total // 2)


def conv2d_same(x, w, stride):
    """slim resnet_utils.conv2d_same: stride 1 -> 'SAME'; stride > 1 -> explicit (k-1) padding
    split beg/end then 'VALID' (so the result does not depend on the input size parity)."""
    k = w.shape[0]
    if stride == 1:
        return conv2d_nhwc(x, w, 1, (same_pad(x.shape[1], k, 1), same_pad(x.shape[2], k, 1)))
    beg = (k - 1) // 2
    return conv2d_nhwc(x, w, stride, ((beg, k - 1 - beg), (beg, k - 1 - beg)))


def bn_inference(x, bn, eps):
    """Inference batch norm with moving statistics: gamma*(x-mean)/sqrt(var+eps)+beta."""
    return (x - bn["moving_mean"]) / np.sqrt(bn["moving_variance"] + eps) * bn["gamma"] + bn["beta"]


def fold_bn(bn, eps):
    """scale/shift form consumed by the HIP kernels (y = conv*scale + shift)."""
    scale = bn["gamma"] / np.sqrt(bn["moving_variance"] + eps)
    return scale, bn["beta"] - bn["moving_mean"] * scale


def max_pool_3x3_s2_same(x):
    """slim pool1: max_pool2d([3,3], stride=2, padding='SAME') (padding never wins the max)."""
    B, H, W, C = x.shape
    ph, pw = same_pad(H, 3, 2), same_pad(W, 3, 2)
    xp = np.pad(x, ((0, 0), ph, pw, (0, 0)), constant_values=-np.inf)
    Ho, Wo = -(-H // 2), -(-W // 2)
    out = np.full((B, Ho, Wo, C), -np.inf, x.dtype)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, xp[:, ky:ky + (Ho - 1) * 2 + 1:2, kx:kx + (Wo - 1) * 2 + 1:2, :])
    return out


def subsample(x, factor):
    """resnet_utils.subsample: max_pool2d([1,1], stride=factor) == strided slicing."""
    return x if factor == 1 else x[:, ::factor, ::factor, :]


# ----------------------------------------------------------------------------- slim resnet_v1
def bottleneck(x, p, prefix, depth, depth_bottleneck, stride):
    """slim resnet_v1.bottleneck (post-activation v1 unit)."""
    def cbn(inp, name, s, relu, same=False):
        w = p[prefix + "/" + name + "/weights"]
        y = conv2d_same(inp, w, s) if same else conv2d_nhwc(subsample(inp, s), w, 1)
        y = bn_inference(y, {k: p[prefix + "/" + name + "/BatchNorm/" + k]
                             for k in ("gamma", "beta", "moving_mean", "moving_variance")}, SLIM_BN_EPS)
        return np.maximum(y, 0) if relu else y

    if depth == x.shape[-1]:
        shortcut = subsample(x, stride)
    else:
        shortcut = cbn(x, "shortcut", stride, relu=False)       # 1x1 conv, stride s == subsample then 1x1
    r = cbn(x, "conv1", 1, relu=True)
    r = cbn(r, "conv2", stride, relu=True, same=True)
    r = cbn(r, "conv3", 1, relu=False)
    return np.maximum(shortcut + r, 0)


def block_units(base_depth, num_units, stride):
    """resnet_v1_block: the stride sits on the LAST unit."""
    return [(base_depth * 4, base_depth, 1)] * (num_units - 1) + [(base_depth * 4, base_depth, stride)]


BLOCKS_R50_B3 = [("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2)]          # encode_I_block3
BLOCKS_R50_FULL = BLOCKS_R50_B3 + [("block4", 512, 3, 1)]                                      # encode_I_full
BLOCKS_R101_FULL = [("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 23, 2), ("block4", 512, 3, 1)]


def resnet_v1(images, p, blocks, scope="resnet_v1_50"):
    """encode_I_*: RGB mean subtraction, conv1 7x7/2 (conv2d_same) + BN + ReLU, pool1 3x3/2 SAME,
    bottleneck blocks; inference statistics."""
    x = images - np.asarray(ENC_I_MEAN, images.dtype)
    w = p[scope + "/conv1/weights"]
    x = conv2d_same(x, w, 2)
    x = bn_inference(x, {k: p[scope + "/conv1/BatchNorm/" + k]
                         for k in ("gamma", "beta", "moving_mean", "moving_variance")}, SLIM_BN_EPS)
    x = np.maximum(x, 0)
    x = max_pool_3x3_s2_same(x)
    for name, base, n, stride in blocks:
        for i, (depth, db, s) in enumerate(block_units(base, n, stride)):
            x = bottleneck(x, p, "%s/%s/unit_%d/bottleneck_v1" % (scope, name, i + 1), depth, db, s)
    return x


def init_resnet_params(rng, blocks, scope="resnet_v1_50", dtype=np.float32, width_div=1):
    """He-normal conv weights and non-trivial BN statistics (so folding errors would show).
    width_div shrinks every depth (tests run a narrow network)."""
    p = {}

    def conv(name, k, ci, co):
        p[name + "/weights"] = (rng.standard_normal((k, k, ci, co)) * np.sqrt(2.0 / (k * k * ci))).astype(dtype)
        p[name + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(co)).astype(dtype)
        p[name + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(co)).astype(dtype)
        p[name + "/BatchNorm/moving_mean"] = (0.1 * rng.standard_normal(co)).astype(dtype)
        p[name + "/BatchNorm/moving_variance"] = (1 + 0.2 * rng.random(co)).astype(dtype)

    c0 = 64 // width_div
    conv(scope + "/conv1", 7, 3, c0)
    cin = c0
    for name, base, n, stride in blocks:
        base = base // width_div
        for i, (depth, db, s) in enumerate(block_units(base, n, stride)):
            pre = "%s/%s/unit_%d/bottleneck_v1" % (scope, name, i + 1)
            if depth != cin:
                conv(pre + "/shortcut", 1, cin, depth)
            conv(pre + "/conv1", 1, cin, db)
            conv(pre + "/conv2", 3, db, db)
            conv(pre + "/conv3", 1, db, depth)
            cin = depth
    return p


def scaled_blocks(blocks, width_div):
    return [(n, b // width_div, u, s) for (n, b, u, s) in blocks]


# ----------------------------------------------------------------------------- crop_and_resize
def crop_and_resize(fmap, boxes, box_ind, ch, cw, extrapolation_value=0.0):
    """tf.image.crop_and_resize (bilinear).  boxes [n,4] = normalised [y1,x1,y2,x2]."""
    B, H, W, C = fmap.shape
    n = boxes.shape[0]
    out = np.full((n, ch, cw, C), extrapolation_value, fmap.dtype)
    dt = fmap.dtype.type
    for i in range(n):
        y1, x1, y2, x2 = [dt(v) for v in boxes[i]]
        b = int(box_ind[i])
        hs = (y2 - y1) * dt(H - 1) / dt(ch - 1) if ch > 1 else dt(0)
        ws = (x2 - x1) * dt(W - 1) / dt(cw - 1) if cw > 1 else dt(0)
        for yy in range(ch):
            in_y = y1 * dt(H - 1) + dt(yy) * hs if ch > 1 else dt(0.5) * (y1 + y2) * dt(H - 1)
            if in_y < 0 or in_y > H - 1:
                continue
            top, bot = int(np.floor(in_y)), int(np.ceil(in_y))
            ly = in_y - dt(top)
            for xx in range(cw):
                in_x = x1 * dt(W - 1) + dt(xx) * ws if cw > 1 else dt(0.5) * (x1 + x2) * dt(W - 1)
                if in_x < 0 or in_x > W - 1:
                    continue
                left, right = int(np.floor(in_x)), int(np.ceil(in_x))
                lx = in_x - dt(left)
                tl, tr = fmap[b, top, left], fmap[b, top, right]
                bl, br = fmap[b, bot, left], fmap[b, bot, right]
                t = tl + (tr - tl) * lx
                bt = bl + (br - bl) * lx
                out[i, yy, xx] = t + (bt - t) * ly
    return out


def roi_pool(fmap, box, height, width):
    """modules.roi_pool: every image's boxes, batch ids tiled (vlmap/modules.py:204-216)."""
    B, n = box.shape[:2]
    ids = np.repeat(np.arange(B), n)
    return crop_and_resize(fmap, box.reshape(-1, 4), ids, height, width).reshape(B, n, height, width, -1)


# ----------------------------------------------------------------------------- vfeat models
def model_vfeat_resnet(images, normal_box, p, blocks=BLOCKS_R50_B3):
    """vqa/model_vfeat_resnet.py:28-40: block3 map -> 1x1 crop_and_resize -> [B, n_box, C]."""
    enc = resnet_v1(images, p, blocks)
    roi = roi_pool(enc, normal_box, 1, 1)
    return roi.reshape(images.shape[0], normal_box.shape[1], enc.shape[-1]), enc


def model_vfeat(images, normal_box, p, blocks=BLOCKS_R50_B3, v_dim=512, roi_sz=5):
    """vqa/model_vfeat.py:28-51: block3 -> I_reduce_dim (1x1 conv+BN+ReLU) -> 5x5 crop_and_resize
    -> I2V: two 3x3 VALID convs that SHARE one weight tensor and one BN ('conv2d_1' twice,
    vlmap/modules.py:232-237)."""
    enc = resnet_v1(images, p, blocks)

    def conv_bn_relu(x, scope, pad_same):
        w = p[scope + "/conv2d/weights"]
        k = w.shape[0]
        pad = (same_pad(x.shape[1], k, 1), same_pad(x.shape[2], k, 1)) if pad_same else ((0, 0), (0, 0))
        y = conv2d_nhwc(x, w, 1, pad)
        y = bn_inference(y, {kk: p[scope + "/BatchNorm/" + kk]
                             for kk in ("gamma", "beta", "moving_mean", "moving_variance")}, LAYERS_BN_EPS)
        return np.maximum(y, 0)

    low = conv_bn_relu(enc, "I_reduce_dim/conv2d", True)
    roi = roi_pool(low, normal_box, roi_sz, roi_sz)
    B, n = normal_box.shape[:2]
    flat = roi.reshape(B * n, roi_sz, roi_sz, v_dim)
    v = conv_bn_relu(flat, "I2V/conv2d_1", False)
    v = conv_bn_relu(v, "I2V/conv2d_1", False)
    return v.reshape(B, n, v_dim), enc


def init_vfeat_head_params(rng, p, enc_dim, v_dim, dtype=np.float32):
    for scope, k, ci in (("I_reduce_dim/conv2d", 1, enc_dim), ("I2V/conv2d_1", 3, v_dim)):
        p[scope + "/conv2d/weights"] = (rng.standard_normal((k, k, ci, v_dim)) * np.sqrt(2.0 / (k * k * ci))).astype(dtype)
        p[scope + "/BatchNorm/gamma"] = (1 + 0.1 * rng.standard_normal(v_dim)).astype(dtype)
        p[scope + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(v_dim)).astype(dtype)
        p[scope + "/BatchNorm/moving_mean"] = (0.1 * rng.standard_normal(v_dim)).astype(dtype)
        p[scope + "/BatchNorm/moving_variance"] = (1 + 0.2 * rng.random(v_dim)).astype(dtype)
    return p


def make_boxes(rng, B, n, dtype=np.float32):
    """normalised [y1,x1,y2,x2] with y1<=y2, x1<=x2 (util/box_utils.py:140-147 ordering)."""
    ys = np.sort(rng.random((B, n, 2)), axis=-1)
    xs = np.sort(rng.random((B, n, 2)), axis=-1)
    return np.stack([ys[..., 0], xs[..., 0], ys[..., 1], xs[..., 1]], axis=-1).astype(dtype)


def conv_flops_per_image(blocks, H, W, c0=64):
    """Algorithmic MAC*2 of the stack, slim stride placement."""
    def out_same(s):
        return -(-s // 2)
    fl = 0
    h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    fl += 2 * h * w * 49 * 3 * c0
    h, w = out_same(h), out_same(w)
    cin = c0
    for name, base, n, stride in blocks:
        for depth, db, s in block_units(base, n, stride):
            ho, wo = ((h - 1) // s + 1, (w - 1) // s + 1) if s > 1 else (h, w)
            if depth != cin:
                fl += 2 * ho * wo * cin * depth
            fl += 2 * h * w * cin * db
            fl += 2 * ho * wo * 9 * db * db
            fl += 2 * ho * wo * db * depth
            h, w, cin = ho, wo, depth
    return fl
