"""Shared helpers for the GPU parity tests (oracle case construction + engine)."""
import numpy as np
import torch

from oracle import vqa_oracle as O


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_case(seed, model_type, B, R, T, N, dims, dtype=np.float32, full_boxes=False, num_train=None,
              head="random", ragged=True, num_marginal=O.NUM_MARGINAL):
    rng = np.random.default_rng(seed)
    p = O.init_params(rng, model_type, dtype=dtype, head=head, **dims)
    if head == "random":
        p = O.perturb_ln_params(p, rng)
    table, nbox = O.make_table(rng, N, R, dims["D"], dtype, full_boxes=full_boxes)
    batch = O.make_batch(rng, B, T, dims["Vq"], dims["A"], N, dtype, ragged=ragged)
    am = O.make_answer_masks(rng, dims["A"], num_train or int(dims["A"] * 0.75), dtype, exist_all=False)
    masks = O.make_dropout_masks(rng, B, R, dims["H"], dtype, model_type=model_type, num_marginal=num_marginal)
    return p, table, nbox, batch, am, masks


def variant_inputs(masks):
    """the explicit extra inputs of two ablations as FusionEngine.forward keywords: vlmap_answer_full's noise [B, H],
    vlmap_answer_ent's pairing dropout keep-mask [B, M, 2H]"""
    kw = {}
    if "noise" in masks:
        kw["noise"] = dev(masks["noise"].astype(np.float32))
    if "tile_joint" in masks:
        kw["keep_tile"] = dev(masks["tile_joint"].astype(np.uint8))
    return kw


def make_engine(model_type, p, table, nbox, am, B, R, T, dims, global_batch=None, **kw):
    from vqa_transfer_externaldata_amd import fusion as F
    if model_type == "standard_word2vec":      # the constant answer-GloVe matrix travels beside the variables
        kw.setdefault("answer_glove", p[O.OUTPUT_GLOVE].astype(np.float32))
    eng = F.FusionEngine(model_type=model_type, B=B, R=R, T=T, N_img=table.shape[0],
                         params={k: v.astype(np.float32) for k, v in p.items() if not O.is_const(k)},
                         global_batch=global_batch, **dims, **kw)
    eng.bind_inputs(table=dev(table.astype(np.float32)), nbox_table=dev(nbox),
                    answer_masks={k: dev(v.astype(np.float32)) for k, v in am.items()})
    return eng


def dev_batch(batch):
    out = {k: dev(v) for k, v in batch.items()}
    out["answer_target"] = out["answer_target"].float()
    return out


def to64(d):
    return {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
