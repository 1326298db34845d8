"""Train-step time of every registry entry (vqa/importer.py:1-14) at bs 512 and the reference's dimensions, synthetic
inputs, one MI355X.  One line per model type: ms per step (forward + backward + clip + Adam, fresh dropout masks).
    python tools/variant_bench.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import bi_oracle as BO, legacy_vqa_oracle as LO, vqa_oracle as O  # noqa: E402  (random initialisers only)
from vqa_transfer_externaldata_amd import fusion as F  # noqa: E402

TYPES = ["vlmap_answer", "standard", "standard_testmask", "standard_word2vec", "vlmap_answer_vqa_all", "vlmap_answer_vqa_all2",
         "vlmap_answer_noc", "vlmap_answer2", "vlmap_answer_no_noise", "vlmap_answer_full", "vlmap_answer_adapt",
         "vlmap_answer_ent", "vlmap_finetune", "vlmap_only", "vqa"]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    B, R, T, N, Vq, A = 512, 36, 14, 2048, 16384, 3000
    rng = np.random.default_rng(0)
    batch = O.make_batch(rng, B, T, Vq, A, N, ragged=False)
    db = {k: dev(v) for k, v in batch.items()}
    am = {k: dev(v) for k, v in O.make_answer_masks(rng, A, 2250).items()}
    out = {}
    for mt in TYPES:
        kw, D, H = {}, 2048, 1024
        if mt == "vqa":
            D = H = 512
            p = LO.init_params(rng, Vq=Vq, W=300, D=512, L=512, M=512, A=A)
            kw = dict(map_dim=512, ft_vlmap=True, glove_fixed=p[LO.FIXED], answers=LO.make_answers(rng, A, Vq, 4))
            p = {k: v for k, v in p.items() if not O.is_const(k)}
        elif mt in F.BI_FAMILY:
            p = BO.init_params(rng, Vq=Vq, W=300, D=D, H=H, A=A)
        else:
            p = O.init_params(rng, mt, Vq=Vq, W=300, D=D, H=H, A=A)
            if mt == "standard_word2vec":
                kw["answer_glove"] = p[O.OUTPUT_GLOVE]
            p = {k: v for k, v in p.items() if not O.is_const(k)}
        table = dev(np.maximum(rng.standard_normal((N, R, D)), 0).astype(np.float32))
        eng = F.FusionEngine(model_type=mt, B=B, R=R, T=T, N_img=N, Vq=Vq, W=300, D=D, H=H, A=A, params=p, **kw)
        eng.bind_inputs(table=table, nbox_table=dev(np.full(N, R, np.int32)), answer_masks=am)

        def step(i):
            ka, kj = eng.make_keep_masks(11, i)
            ex = {}
            if mt in F.NOC_FAMILY:
                ex["keep_joint2"] = eng.make_keep_mask_joint2(11, i)
            if mt == "vlmap_answer_full":
                ex["noise"] = eng.make_noise(11, i)
            if mt == "vlmap_answer_ent":
                ex["keep_tile"] = eng.make_keep_mask_tile(11, i)
            if mt in F.BI_FAMILY:
                ex["keep_word"] = eng.make_keep_mask_word(11, i)
            eng.train_step(db, None if mt == "vqa" else ka, None if mt == "vqa" else kj, 1e-3, **ex)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(3 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        loss = float(eng.loss())
        assert np.isfinite(loss), mt
        out[mt] = round(ms, 3)
        print(json.dumps({"model_type": mt, "ms_per_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3), "loss": round(loss, 3),
                          "workspace_GB": round(eng.workspace.numel() / 2 ** 30, 2)}), flush=True)
        del eng, table
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
