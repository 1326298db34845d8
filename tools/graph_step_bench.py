"""Eager launches against whole-step hipGraph replay of the bs-512 fusion train step (FusionEngine.train_step_graph),
for the row-chain count / delay this process was started with (VQA_HOT_GRU_CHAINS, VQA_HOT_GRU_CHAIN_DELAY_US are read
once per process).  Prints one JSON line.    python tools/graph_step_bench.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from vqa_transfer_externaldata_amd import fusion as F  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    cfg = dict(bench.CFG)
    params = bench.synth_params("vlmap_answer", cfg, seed=1234)
    table, nbox, am, batches = bench.synth_inputs(cfg, seed=1234, device=dev)
    out = {"chains": os.environ.get("VQA_HOT_GRU_CHAINS", "2"), "delay_us": os.environ.get("VQA_HOT_GRU_CHAIN_DELAY_US", "3")}
    for mode in ("eager", "graph", "eager", "graph"):
        eng = F.FusionEngine(model_type="vlmap_answer", B=cfg["B"], R=cfg["R"], D=cfg["D"], H=cfg["H"], T=cfg["T"], W=cfg["W"],
                             A=cfg["A"], Vq=cfg["Vq"], N_img=cfg["N_img"], params=params, device=dev)
        eng.bind_inputs(table=table, nbox_table=nbox, answer_masks=am)

        def step(i):
            if mode == "graph":
                return eng.train_step_graph(batches[i % len(batches)], 1e-3, 99, i)
            ka, kj = eng.make_keep_masks(99, i)
            eng.train_step(batches[i % len(batches)], ka, kj, 1e-3)
        nodes = None
        for i in range(8):
            nodes = step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(8 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        out.setdefault(mode + "_ms", []).append(round(ms, 4))
        if nodes:
            out["graph_nodes"] = nodes
        # host time to ENQUEUE one step (the GPU is idle at the start: the first calls return as fast as the host can go)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(100)
        out.setdefault(mode + "_host_enqueue_ms", []).append(round((time.perf_counter() - t0) * 1e3, 4))
        torch.cuda.synchronize()
        out.setdefault(mode + "_loss", []).append(round(float(eng.report()["answer_train_loss"]), 5))
        eng.drop_graphs()
        del eng
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
