"""Diagnostic: where do two runs of the cfg-5 trainer step differ?  plain vs plain (run-to-run noise), prefetch/deferred vs
plain, per parameter.  Not a test."""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vqa_transfer_externaldata_amd import dataset_vlmap as DV, pretrain_trainer as PTT

R, D, L, Vq, n_ws, A, B = 36, 64, 6, 40, 12, 30, 8
STEPS = int(os.environ.get("STEPS", 4))


def make(det=False):
    data = DV.synthetic_dataset(40, Vq, n_ws, A, R=R, D=D, max_len=L, seed=5)
    ds = {"train": DV.Dataset(split="train", data=data, seed=1), "val": DV.Dataset(split="val", data=data, seed=2)}
    cfg = PTT.build_parser().parse_args(["--batch_size", str(B), "--max_train_iter", "4", "--learning_rate", "0.002",
                                         "--features_on_device", "1", "--input_workers", "0", "--input_prefetch", "0"])
    cfg.data_cfg = ds["train"].get_config()
    cfg.vocab = {"vocab": ["w%d" % i for i in range(Vq)], "dict": {"w%d" % i: i for i in range(Vq)}}
    cfg.answer_dict, cfg.ws_dict = data["answer_dict"], data["ws_dict"]
    cfg.synthetic, cfg.train_dir = 1, tempfile.mkdtemp()
    t = PTT.Trainer(cfg, ds)
    t.model.engine.deterministic = det
    return t


def plain(t, grads):
    for _ in range(STEPS):
        t.model.set_batch(t._next("train")); t.model.build(); t.model.backward()
        torch.cuda.synchronize(); grads.append(t.model.engine.grad_flat.clone())
        t.model.apply_gradients(t._lr()); torch.cuda.synchronize()


def fancy(t, grads):
    for _ in range(STEPS):
        t.run_train_step(False)
        grads.append(t.model.engine.grad_flat.clone())


def cmp(name, ta, tb, ga, gb):
    print("==", name)
    eng = ta.model.engine
    for s, (x, y) in enumerate(zip(ga, gb)):
        worst = []
        for k, (o, c) in eng._tab.items():
            a, b = x[o:o + c], y[o:o + c]
            sc = float(b.abs().max()) or 1.0
            worst.append((float((a - b).abs().max()) / sc, k, sc))
        worst.sort(reverse=True)
        print("  step %d grads: worst rel diff" % s, ["%s %.2e (scale %.1e)" % (k, d, sc) for d, k, sc in worst[:4]])
    worst = []
    for k in ta.model.engine.params:
        d = (ta.model.engine.params[k] - tb.model.engine.params[k]).abs()
        worst.append((float(d.max()), float((d > 5e-6).float().mean()), k))
    worst.sort(reverse=True)
    print("  params after %d steps:" % STEPS, ["%s max %.2e frac %.3f" % (k, m, f) for m, f, k in worst[:6]])


for det in (False, True):
    print("#### deterministic =", det)
    tb, tb2, ta = make(det), make(det), make(det)
    gb, gb2, ga = [], [], []
    plain(tb, gb); plain(tb2, gb2); fancy(ta, ga)
    cmp("plain vs plain", tb, tb2, gb, gb2)
    cmp("prefetch+deferred vs plain", ta, tb, ga, gb)
