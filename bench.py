#!/usr/bin/env python3
"""Headline benchmark: VQA train samples/sec of model_vlmap_answer at bs 512 per
GPU (BASELINE.json configs[1]) -- one pass of the hot path (forward, backward,
global-norm clip, Adam) over one synthetic batch per step, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL all-reduce of
     the flat gradient buffer; weak scaling = 512 samples per GPU)

Prints ONE JSON line (rank 0) with the `roofline` of the dominant kernel (the
v_linear_v forward GEMM, timed with HIP events on its own stream inside the timed
region) and a `cpu_baseline` (torch-CPU port of the same step on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak (= vector peak)

CFG = dict(B=512, R=36, D=2048, H=1024, T=14, W=300, A=3000, Vq=16384, N_img=8192, num_train_answer=2250)


def synth_params(model_type, cfg, seed):
    """Random-init weights of the architecture: Xavier-uniform FCs, GRU
    gate bias 1.0, embeddings U(-0.01, 0.01)."""
    from vqa_transfer_externaldata_amd import fusion as F
    g = torch.Generator().manual_seed(seed)
    shapes = F.variable_shapes(model_type, cfg["Vq"], cfg["W"], cfg["D"], cfg["H"], cfg["A"])
    p = {}
    for n, s in shapes.items():
        if n.endswith("embed_map"):
            p[n] = (torch.rand(s, generator=g) * 0.02 - 0.01)
        elif n.endswith("/weights") or n.endswith("/kernel"):
            lim = (6.0 / (s[0] + s[1])) ** 0.5
            p[n] = (torch.rand(s, generator=g) * 2 - 1) * lim
        elif n.endswith("gates/bias") or n.endswith("LayerNorm/gamma"):
            p[n] = torch.ones(s)
        else:
            p[n] = torch.zeros(s)
    return p


def synth_inputs(cfg, seed, device, n_batches=4):
    g = torch.Generator(device=device).manual_seed(seed)
    N, R, D, B, T, A = cfg["N_img"], cfg["R"], cfg["D"], cfg["B"], cfg["T"], cfg["A"]
    table = torch.randn(N, R, D, generator=g, device=device).relu_()
    nbox = torch.full((N,), R, dtype=torch.int32, device=device)
    am = {"train": (torch.arange(A, device=device) < cfg["num_train_answer"]).float()}
    is_obj = torch.rand(A, generator=g, device=device) < 0.5
    am["obj"], am["attr"] = is_obj.float(), (~is_obj).float()
    am["exist"] = torch.ones(A, device=device)
    batches = []
    scores = torch.tensor([0.3, 0.6, 0.9, 1.0], device=device)
    for _ in range(n_batches):
        tgt = torch.zeros(B, A, device=device)
        for k in range(3):
            ids = torch.randint(0, A, (B,), generator=g, device=device)
            sc = scores[torch.randint(0, 4, (B,), generator=g, device=device)]
            use = torch.rand(B, generator=g, device=device) < (1.0 if k == 0 else 0.5)
            tgt[torch.arange(B, device=device)[use], ids[use]] = sc[use]
        batches.append({
            "image_idx": torch.randint(0, N, (B,), generator=g, device=device, dtype=torch.int64),
            "q_intseq": torch.randint(0, cfg["Vq"] - 3, (B, T), generator=g, device=device, dtype=torch.int32),
            "q_intseq_len": torch.full((B,), T, dtype=torch.int32, device=device),
            "answer_target": tgt,
        })
    return table, nbox, am, batches


def cpu_baseline(params, table, nbox, am, batch, cfg, steps=4):
    """Torch-CPU fp32 port of the same train step (oracle/torch_ref.py) on the host
    cores, bounded sample: `steps` timed steps at the full bs-512 shape (about 10 s of CPU work at ~200 samples/s)."""
    from oracle import torch_ref as TR
    n = 512
    tab = table[:n].cpu().numpy()
    b = {k: v.cpu().numpy() for k, v in batch.items()}
    b["image_idx"] = b["image_idx"] % n
    amc = {k: v.cpu().numpy() for k, v in am.items()}
    pn = {k: v.numpy() for k, v in params.items()}
    stepper = TR.CpuTrainStep(pn, tab, nbox[:n].cpu().numpy(), amc, "vlmap_answer", lr=1e-3)
    rng = np.random.default_rng(0)
    masks = {"att": (rng.random((cfg["B"], cfg["R"], cfg["H"])) < 0.8).astype(np.float32),
             "joint": (rng.random((cfg["B"], 2 * cfg["H"])) < 0.5).astype(np.float32)}
    stepper(b, masks)                                   # warm-up (allocations, thread pool)
    t0 = time.time()
    for _ in range(steps):
        stepper(b, masks)
    dt = (time.time() - t0) / steps
    return {"value": cfg["B"] / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps of model_vlmap_answer at bs %d (torch-CPU fp32 restatement, "
                      "oracle/torch_ref.py; the reference TF1 path cannot run here)" % (steps, cfg["B"])}


# v_linear_v forward GEMM: 64x128 tiles, 8 waves of 32x32 (cfg 21), NN layout, plain epilogue -> 288 x 8 = 2304 workgroups
ROOFLINE_KERNEL_PREFIX = "gemm_f32_kernel<64, 128, 32, 32, 1, 32, 0, true, false, 0, false, false, 512"
ROOFLINE_KERNEL_GRID = "grid=2304"


def pmc_traffic():
    """HBM-side bytes per launch of the roofline kernel, from the committed rocprofv3 PMC passes
    (profiles/r1_pmc_traffic.json: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same bench,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; Infinity-Cache hits are included in
    the fabric-side counter).  None when the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        for k, v in d.items():
            if k.startswith(ROOFLINE_KERNEL_PREFIX) and k.endswith(ROOFLINE_KERNEL_GRID):
                return float(v["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def vfeat_bench(device, batch=128, iters=3):
    """vfeat imgs/sec (second half of BASELINE.json's metric): slim-style ResNet-101 blocks 1-4 on
    synthetic 448x448 images + 1x1 crop_and_resize of 36 boxes -> [36, 2048] per image (BASELINE
    configs[2] extractor; random-init He weights, identity-ish BN statistics)."""
    from vqa_transfer_externaldata_amd import vfeat as VF
    rng = np.random.default_rng(1234)
    params = VF.init_random_params(rng, VF.BLOCKS_R101_FULL)
    model = VF.VfeatResnetModel(params, VF.BLOCKS_R101_FULL, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    img = torch.rand(batch, 448, 448, 3, generator=g, device=device) * 255.0
    ys = torch.sort(torch.rand(batch, 36, 2, generator=g, device=device), dim=-1).values
    xs = torch.sort(torch.rand(batch, 36, 2, generator=g, device=device), dim=-1).values
    box = torch.stack([ys[..., 0], xs[..., 0], ys[..., 1], xs[..., 1]], dim=-1).contiguous()
    b = {"image": img, "normal_box": box}
    model.build(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        v = model.build(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    fl = VF.conv_flops_per_image(VF.BLOCKS_R101_FULL, 448, 448)
    return {"imgs_per_sec": batch / dt, "batch": batch, "image": "448x448x3", "net": "resnet_v1_101 blocks1-4 + "
            "1x1 crop_and_resize of 36 boxes", "gflop_per_image": fl / 1e9, "tflops": batch * fl / dt / 1e12,
            "frac_f32_mfma_peak": batch * fl / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "out_shape": list(v.shape)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vfeat", action="store_true")
    ap.add_argument("--probe", type=str, default="v_linear_v.fwd_gemm")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VQA_BENCH_BACKEND", "nccl")      # "gloo" = rehearsal on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "vqa-transfer-externaldata_amd", "libvqahot.so")):
        ge.build()
    if world > 1:
        dist.barrier()
    from vqa_transfer_externaldata_amd import _lib, fusion as F
    from vqa_transfer_externaldata_amd import dp as PAR

    cfg = dict(CFG)
    params = synth_params("vlmap_answer", cfg, seed=1234)
    table, nbox, am, batches = synth_inputs(cfg, seed=1234 + rank, device=device)
    eng = F.FusionEngine(model_type="vlmap_answer", B=cfg["B"], R=cfg["R"], D=cfg["D"], H=cfg["H"], T=cfg["T"],
                         W=cfg["W"], A=cfg["A"], Vq=cfg["Vq"], N_img=cfg["N_img"], params=params, device=device,
                         global_batch=cfg["B"] * world)
    eng.bind_inputs(table=table, nbox_table=nbox, answer_masks=am)
    reducer = PAR.BucketedAllReduce() if world > 1 else None
    lib = _lib.load()
    if os.environ.get("VQA_GRU_CFG"):
        _lib.check(lib.vqa_gemm_set_gru_config(int(os.environ["VQA_GRU_CFG"])), "vqa_gemm_set_gru_config")
    if os.environ.get("VQA_GEMM_CFG"):     # tuning only: forces ONE tile config on every plain GEMM
        _lib.check(lib.vqa_gemm_set_config(int(os.environ["VQA_GEMM_CFG"])), "vqa_gemm_set_config")

    def step(i):
        ka, kj = eng.make_keep_masks(seed=99 + rank, step=i)       # fresh dropout masks every step
        eng.train_step(batches[i % len(batches)], ka, kj, 1e-3, allreduce=reducer)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.check(lib.vqa_probe_enable(args.probe.encode(), max(args.steps, 1)), "vqa_probe_enable")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms = (C.c_float * max(args.steps, 1))()
    n = C.c_int()
    _lib.check(lib.vqa_probe_read(ms, args.steps, C.byref(n)), "vqa_probe_read")
    lib.vqa_probe_disable()
    loss = eng.report()["answer_train_loss"]
    assert np.isfinite(loss), "training diverged"

    if rank == 0:
        kern_ms = float(np.mean(ms[:n.value])) if n.value else float("nan")
        # algorithmic FLOPs of the probed kernel: v_linear_v fwd = 2*B*R*D*H
        flops = 2.0 * cfg["B"] * cfg["R"] * cfg["D"] * cfg["H"]
        achieved = flops / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "VQA train samples/sec (img+question) at bs512",
            "value": cfg["B"] * world * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "model_vlmap_answer train step (fwd+bwd+clip+Adam), bs 512 per GPU, "
                                   "36x2048 precomputed region features resident in HBM, 14-token questions, "
                                   "3000 answers (BASELINE configs[1])",
                       "global_batch": cfg["B"] * world, "Vq": cfg["Vq"], "table_images": cfg["N_img"],
                       "parallelism": "dp%d" % world if world > 1 else "single"},
            "roofline": {"kernel": "gemm_f32_kernel<64,128,32,32,1,32,0,true,false,0,false,false,512> (v_linear_v forward GEMM, "
                                   "M=18432 N=1024 K=2048, v_mfma_f32_32x32x2_f32)",
                         "bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(),
                         "traffic_unit": "bytes/launch (L2<->fabric, PMC; algorithmic minimum 235e6)",
                         "kernel_ms": kern_ms, "samples": n.value},
            "final_train_loss": loss,
        }
        if world == 1 and not args.no_vfeat:
            out["vfeat"] = vfeat_bench(device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, table, nbox, am, batches[0], cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
