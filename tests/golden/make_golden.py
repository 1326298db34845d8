"""Generates the golden vectors under tests/golden/ from the CPU oracle (float64 on float32 inputs).

The reference cannot be imported here (Python 2 + TensorFlow 1.6), so these vectors
freeze the ORACLE, not the reference: they are regression pins for the oracle and seeded parity
cases for the HIP path.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import conv_oracle as CO  # noqa: E402
from oracle import vqa_oracle as O  # noqa: E402

DIMS = dict(Vq=60, W=300, D=64, H=32, A=50)
MID = ["num_V_ft", "v_linear_v", "condition", "q_linear_v", "att_score", "pooled_V_ft", "pooled_linear_l",
       "l_linear_l", "joint", "logit", "pred"]


ABLATION_MID = {"vlmap_answer2": [], "vlmap_answer_no_noise": ["q_L_mean"], "vlmap_answer_adapt": ["v_adapt"],
                "vlmap_answer_full": ["q_L_mean", "q_L_log_sigma_sq", "q_L_mean_noise"], "vlmap_answer_ent": ["marginal_prob"]}


def fusion_case(model_type, seed, B=8, R=36, T=14, N=16, num_marginal=5):
    rng = np.random.default_rng(seed)
    p = O.perturb_ln_params(O.init_params(rng, model_type, **DIMS), rng)
    table, nbox = O.make_table(rng, N, R, DIMS["D"], full_boxes=False)
    batch = O.make_batch(rng, B, T, DIMS["Vq"], DIMS["A"], N)
    batch["q_intseq_len"][0] = 0                      # edge cases kept in the fixture
    batch["q_intseq"][0] = 0
    nbox[batch["image_idx"][1]] = 1
    am = O.make_answer_masks(rng, DIMS["A"], 37, exist_all=False)
    masks = O.make_dropout_masks(rng, B, R, DIMS["H"], model_type=model_type, num_marginal=num_marginal)
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    loss, report, out, mid, tape = O.forward(to64(p), to64(batch), table.astype(np.float64), nbox, to64(am),
                                             to64(masks), model_type)
    grads, dx = O.backward(to64(p), to64(batch), to64(am), to64(masks), tape, model_type)
    z = {"model_type": np.array(model_type), "B": B, "R": R, "T": T, "N": N}
    z.update({"dim_" + k: v for k, v in DIMS.items()})
    z.update({"param/" + k: v for k, v in p.items()})
    z.update({"batch/" + k: v for k, v in batch.items()})
    z.update({"amask/" + k: v for k, v in am.items()})
    z.update({"keep/" + k: v.astype(np.uint8) for k, v in masks.items() if k != "noise"})
    if "noise" in masks:                              # vlmap_answer_full: the reparameterisation's normal draws, float32
        z["noise"] = masks["noise"]
    z["table"], z["nbox"] = table, nbox
    z.update({"mid/" + k: np.asarray(mid[k]) for k in MID + ABLATION_MID.get(model_type, [])})
    z["loss"] = np.float64(loss)                      # sum of the model's losses
    z.update({"report/" + k: np.float64(v) for k, v in report.items()})
    z.update({"out/" + k: np.asarray(out[k]) for k in ("all_score", "max_train_score", "test_obj_score",
                                                       "test_attr_score", "test_obj_max_score", "test_attr_max_score")
              if k in out})
    z.update({"grad/" + k: v.astype(np.float32) for k, v in grads.items()})
    z["dx_embed"] = dx.astype(np.float32)
    if model_type == "vlmap_answer_vqa_all2":
        z["mid/logit_fixed"], z["mid/logit_tuned"] = np.asarray(tape["z1"]), np.asarray(tape["z2"])
    return z


def pretrain_case(seed=7, B=3, n=5, R=6, D=16, H=8, L=4, W=12, Vq=20, n_ws=7, A=12, ln_shared=True):
    """cfg-5 pre-training model (vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py) at toy size: inputs, masks, the
    13 report scalars, attention maps, logits and every gradient (float64 torch-autograd restatement).  ln_shared: one
    LayerNorm per shared fc_layer scope (TF 1.x, the default) or one per call site; the parameter NAMES in the fixture
    say which (oracle/pretrain_oracle.py)."""
    from oracle import pretrain_oracle as PO
    rng = np.random.default_rng(seed)
    p = PO.init_params(rng, Vq, n_ws, A, W=W, D=D, H=H, ln_shared=ln_shared)
    batch = PO.make_batch(rng, B, n, R, D, L, Vq, n_ws, A)
    masks = PO.make_masks(rng, B, n, R, H)
    to64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    total, report, mid = PO.forward(to64(p), to64(batch), to64(masks), n)
    _, _, grads, slices = PO.torch_loss_and_grads(to64(p), to64(batch), to64(masks), n)
    z = dict(B=B, n=n, R=R, D=D, H=H, L=L, W=W, Vq=Vq, n_ws=n_ws, A=A, total_loss=np.float64(total))
    z.update({"param/" + k: v for k, v in p.items()})
    z.update({"batch/" + k: v for k, v in batch.items()})
    z.update({"keep/" + k: v.astype(np.uint8) for k, v in masks.items()})
    z.update({"report/" + k: np.float64(v) for k, v in report.items()})
    for k in PO.KINDS:
        for m in ("att", "bf_logit", "ws_logit"):
            z["mid/%s/%s" % (k, m)] = np.asarray(mid[k + "/" + m])
    z.update({"grad/" + k: np.asarray(v, np.float64) for k, v in grads.items()})
    z["slice_sq"] = np.float64(sum(float((v ** 2).sum()) for v in slices.values()))
    return z


def conv_case(seed=3):
    rng = np.random.default_rng(seed)
    full = [(n, b, 1, s) for (n, b, u, s) in CO.BLOCKS_R50_B3]
    p = CO.init_resnet_params(rng, full, width_div=2)
    blocks = [(n, b // 2, u, s) for (n, b, u, s) in full]
    img = rng.uniform(0, 255, size=(1, 96, 80, 3)).astype(np.float32)
    box = CO.make_boxes(rng, 1, 6)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    v, enc = CO.model_vfeat_resnet(img.astype(np.float64), box.astype(np.float64), p64, blocks)
    z = {"image": img, "normal_box": box, "V_ft": v.astype(np.float32), "enc_I": enc.astype(np.float32),
         "blocks": np.array([[b, u, s] for (_, b, u, s) in blocks])}
    z.update({"param/" + k: v_ for k, v_ in p.items()})
    return z


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "fusion_vlmap_answer_b8.npz"), **fusion_case("vlmap_answer", 101))
    np.savez_compressed(os.path.join(HERE, "fusion_standard_b8.npz"), **fusion_case("standard", 102))
    np.savez_compressed(os.path.join(HERE, "fusion_standard_word2vec_b4.npz"), **fusion_case("standard_word2vec", 103, B=4))
    np.savez_compressed(os.path.join(HERE, "fusion_vlmap_answer_vqa_all2_b8.npz"), **fusion_case("vlmap_answer_vqa_all2", 104))
    for i, mt in enumerate(O.ABLATION_FAMILY):       # the five older ablations of model_vlmap_answer
        np.savez_compressed(os.path.join(HERE, "fusion_%s_b8.npz" % mt), **fusion_case(mt, 110 + i))
    np.savez_compressed(os.path.join(HERE, "pretrain_cfg5_toy.npz"), **pretrain_case())
    np.savez_compressed(os.path.join(HERE, "pretrain_cfg5_toy_persite.npz"), **pretrain_case(ln_shared=False))
    np.savez_compressed(os.path.join(HERE, "vfeat_resnet_narrow.npz"), **conv_case())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
