"""Pins of the conv-stack oracle: known answers + an independent torch composition
(F.conv2d / F.max_pool2d / manual BN) in float64."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import conv_oracle as CO


def test_stack_output_sizes_match_survey():
    # 540 -> 270 -> 135 -> 68 -> 34 -> 17 ; 448 -> 224 -> 112 -> 56 -> 28 -> 14
    for H, sizes in ((540, [270, 135, 68, 34, 17]), (448, [224, 112, 56, 28, 14])):
        h = (H + 6 - 7) // 2 + 1
        got = [h]
        h = -(-h // 2); got.append(h)
        for _ in range(3):
            h = (h - 1) // 2 + 1; got.append(h)
        assert got == sizes


def test_flop_counts_match_survey():
    gf = lambda b, s: CO.conv_flops_per_image(b, s, s) / 1e9
    assert abs(gf(CO.BLOCKS_R50_B3, 540) - 32.3) < 0.4
    assert abs(gf(CO.BLOCKS_R101_FULL, 448) - 57.5) < 0.6
    assert abs(gf(CO.BLOCKS_R50_FULL, 448) - 27.8) < 0.4


def test_conv_identity_bn_and_same_padding():
    x = np.random.default_rng(0).standard_normal((2, 5, 6, 3))
    w = np.zeros((3, 3, 3, 3)); w[1, 1] = np.eye(3)                       # centre-tap identity
    np.testing.assert_allclose(CO.conv2d_same(x, w, 1), x)
    np.testing.assert_allclose(CO.conv2d_same(x, w, 2), x[:, ::2, ::2])   # explicit pad 1/1 then VALID
    bn = {"gamma": np.ones(3), "beta": np.zeros(3), "moving_mean": np.zeros(3),
          "moving_variance": np.ones(3) - CO.SLIM_BN_EPS}
    np.testing.assert_allclose(CO.bn_inference(x, bn, CO.SLIM_BN_EPS), x, rtol=1e-12)
    s, t = CO.fold_bn({"gamma": np.array([2.0]), "beta": np.array([1.0]), "moving_mean": np.array([3.0]),
                       "moving_variance": np.array([4.0 - 1e-5])}, 1e-5)
    np.testing.assert_allclose([s[0], t[0]], [1.0, -2.0])


def test_max_pool_same_puts_extra_padding_at_the_end():
    x = np.arange(16.0).reshape(1, 4, 4, 1)
    y = CO.max_pool_3x3_s2_same(x)[0, :, :, 0]          # windows rows {0,1,2},{2,3,pad}
    np.testing.assert_array_equal(y, [[10, 11], [14, 15]])
    x = np.arange(25.0).reshape(1, 5, 5, 1)              # odd size: pad 1 at both ends
    np.testing.assert_array_equal(CO.max_pool_3x3_s2_same(x)[0, :, :, 0], [[6, 8, 9], [16, 18, 19], [21, 23, 24]])


def test_crop_and_resize_known_answers():
    f = np.arange(12.0).reshape(1, 3, 4, 1)              # f[y,x] = 4y + x
    c = CO.crop_and_resize(f, np.array([[0, 0, 1, 1.0]]), [0], 1, 1)
    np.testing.assert_allclose(c[0, 0, 0, 0], 4 * 1.0 + 1.5)             # centre of the map, bilinear
    c = CO.crop_and_resize(f, np.array([[0, 0, 1, 1.0]]), [0], 3, 4)
    np.testing.assert_allclose(c[0, :, :, 0], f[0, :, :, 0])             # full box at native size = identity
    c = CO.crop_and_resize(f, np.array([[-1.0, 0, -0.5, 1.0]]), [0], 1, 1)
    assert c[0, 0, 0, 0] == 0.0                                           # outside -> extrapolation value 0


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2)


def test_resnet_stack_matches_torch_composition():
    rng = np.random.default_rng(1)
    blocks = CO.scaled_blocks(CO.BLOCKS_R50_FULL, 8)
    blocks = [(n, b, 2, s) for (n, b, u, s) in blocks]                    # 2 units per block keeps it quick
    p = CO.init_resnet_params(rng, [(n, b * 8, u, s) for (n, b, u, s) in blocks], dtype=np.float64, width_div=8)
    img = rng.uniform(0, 255, size=(2, 67, 75, 3))
    want = CO.resnet_v1(img, p, blocks)

    def bn(x, pre):
        g, b, m, v = (torch.from_numpy(p[pre + "/BatchNorm/" + k]).view(1, -1, 1, 1)
                      for k in ("gamma", "beta", "moving_mean", "moving_variance"))
        return (x - m) / torch.sqrt(v + CO.SLIM_BN_EPS) * g + b

    def conv(x, pre, stride, same):
        w = torch.from_numpy(p[pre + "/weights"]).permute(3, 2, 0, 1)
        k = w.shape[-1]
        if same and stride > 1:
            x = F.pad(x, ((k - 1) // 2, k - 1 - (k - 1) // 2) * 2)
            return F.conv2d(x, w, stride=stride)
        return F.conv2d(x, w, stride=stride, padding=(k - 1) // 2 if same else 0)

    x = _t(img) - torch.tensor(CO.ENC_I_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    x = torch.relu(bn(conv(x, "resnet_v1_50/conv1", 2, True), "resnet_v1_50/conv1"))
    H, W = x.shape[-2:]
    x = F.max_pool2d(F.pad(x, (0, CO.same_pad(W, 3, 2)[1], 0, CO.same_pad(H, 3, 2)[1]), value=-1e300)
                     if CO.same_pad(W, 3, 2)[0] == 0 else
                     F.pad(x, CO.same_pad(W, 3, 2) + CO.same_pad(H, 3, 2), value=-1e300), 3, 2)
    cin = x.shape[1]
    for name, base, n, stride in blocks:
        for i, (depth, db, s) in enumerate(CO.block_units(base, n, stride)):
            pre = "resnet_v1_50/%s/unit_%d/bottleneck_v1" % (name, i + 1)
            sc = x[:, :, ::s, ::s] if depth == cin else bn(conv(x, pre + "/shortcut", s, False), pre + "/shortcut")
            r = torch.relu(bn(conv(x, pre + "/conv1", 1, False), pre + "/conv1"))
            r = torch.relu(bn(conv(r, pre + "/conv2", s, True), pre + "/conv2"))
            r = bn(conv(r, pre + "/conv3", 1, False), pre + "/conv3")
            x = torch.relu(sc + r)
            cin = depth
    got = x.permute(0, 2, 3, 1).numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)


def test_vfeat_models_shapes_and_shared_i2v_weights():
    rng = np.random.default_rng(2)
    blocks = [(n, b, 1, s) for (n, b, u, s) in CO.scaled_blocks(CO.BLOCKS_R50_B3, 8)]
    p = CO.init_resnet_params(rng, [(n, b * 8, u, s) for (n, b, u, s) in blocks], dtype=np.float64, width_div=8)
    enc_dim = blocks[-1][1] * 4
    p = CO.init_vfeat_head_params(rng, p, enc_dim, 16, np.float64)
    img = rng.uniform(0, 255, size=(2, 128, 128, 3))
    box = CO.make_boxes(rng, 2, 5, np.float64)
    v, enc = CO.model_vfeat_resnet(img, box, p, blocks)
    assert v.shape == (2, 5, enc_dim) and enc.shape == (2, 4, 4, enc_dim)
    # 1x1 crop == bilinear sample at the box centre
    b0 = box[0, 0]
    cy, cx = 0.5 * (b0[0] + b0[2]) * 3, 0.5 * (b0[1] + b0[3]) * 3
    y0, x0 = int(np.floor(cy)), int(np.floor(cx)); y1, x1 = int(np.ceil(cy)), int(np.ceil(cx))
    top = enc[0, y0, x0] + (enc[0, y0, x1] - enc[0, y0, x0]) * (cx - x0)
    bot = enc[0, y1, x0] + (enc[0, y1, x1] - enc[0, y1, x0]) * (cx - x0)
    np.testing.assert_allclose(v[0, 0], top + (bot - top) * (cy - y0), rtol=1e-12, atol=1e-12)
    v2, _ = CO.model_vfeat(img, box, p, blocks, v_dim=16)
    assert v2.shape == (2, 5, 16) and np.all(v2 >= 0)
