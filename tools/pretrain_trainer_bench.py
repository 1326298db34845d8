"""The cfg-5 pre-training TRAINER loop (input pipeline + step + report fetch, vlmap_memft/trainer.py:202-263) on
synthetic data: steps/s with the reference-style input side (dense feature slices, in-process assembly) and with the
tables in HBM + forked batch producers.  usage: pretrain_trainer_bench.py [mode] [steps] [n_images] [batch]
mode: dense | resident | resident-workers (default: all three, each in a fresh process)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode, steps, n_images, B):
    import numpy as np
    import tempfile
    from vqa_transfer_externaldata_amd import dataset_vlmap as DV, pretrain_trainer as PTT
    R, D, L, Vq, n_ws, A = 36, 2048, 10, 5000, 2000, 4000
    data = DV.synthetic_dataset(n_images, Vq, n_ws, A, R=R, D=D, max_len=L, seed=0)
    ds = {"train": DV.Dataset(split="train", data=data, seed=1), "val": DV.Dataset(split="val", data=data, seed=2)}
    args = ["--batch_size", str(B), "--max_train_iter", str(steps), "--learning_rate", "0.001",
            "--features_on_device", "0" if mode == "dense" else "1",
            "--input_workers", "4" if mode == "resident-workers" else "0",
            "--input_prefetch", "0" if mode == "dense" else "2"]
    cfg = PTT.build_parser().parse_args(args)
    cfg.data_cfg = ds["train"].get_config()
    cfg.vocab = {"vocab": ["w%d" % i for i in range(Vq)], "dict": {"w%d" % i: i for i in range(Vq)}}
    cfg.answer_dict, cfg.ws_dict = data["answer_dict"], data["ws_dict"]
    cfg.synthetic, cfg.train_dir = 1, tempfile.mkdtemp()
    t = PTT.Trainer(cfg, ds)                      # producers are forked in here, before the engine touches the GPU
    for _ in range(3):
        t.run_train_step(False)
    t0 = time.perf_counter()
    for _ in range(steps):
        step, summary, loss, report, dt = t.run_train_step(False)
    dt = (time.perf_counter() - t0) / steps
    print("%-17s %7.1f ms/step = %6.0f images/s  (bs %d, %d images, total_loss %.3f)" % (
        mode, dt * 1e3, B / dt, B, n_images, loss), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n_images = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    if mode == "all":
        for m in ("dense", "resident", "resident-workers"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), m, str(steps), str(n_images), str(B)])
            if r.returncode != 0:
                sys.exit(r.returncode)
    else:
        run(mode, steps, n_images, B)
