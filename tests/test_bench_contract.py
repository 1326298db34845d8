"""bench.py's host-side contract (no GPU): defaults, how `--gpus N` starts its own ranks, and that the committed PMC
profile still names the kernel the roofline object quotes (a tile-config change without a re-profile would leave
`roofline.traffic` null)."""
import csv
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_are_one_gpu_and_a_run_of_minutes():
    a = bench.parse_args([])
    assert a.gpus == 1 and 1 <= a.warmup <= 10 and 10 <= a.steps <= 100
    assert not (a.no_cpu_baseline or a.no_vfeat or a.no_e2e)


def test_gpus_n_without_a_launcher_spawns_torch_distributed_run(monkeypatch):
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    rc = bench.spawn_ranks(bench.parse_args(argv), argv)
    cmd = seen["cmd"]
    assert rc == 7                                                   # the children's exit code is the parent's
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_committed_pmc_profile_names_the_roofline_kernel():
    nbytes, src = bench.pmc_traffic()
    assert src == "profiles/" + bench.PMC_TRAFFIC_FILES[0], "the newest committed profile must name the roofline kernel"
    # between the algorithmic minimum (235 MB) and an order of magnitude above it
    assert 235e6 <= nbytes <= 2.5e9
    # the same kernel, with the same grid, is in the committed kernel trace
    want = bench.ROOFLINE_KERNEL.split(" (")[0].replace(" ", "")
    with open(os.path.join(ROOT, "profiles", "r4_kernel_stats.csv")) as f:
        names = [r["Name"].replace(" ", "") for r in csv.DictReader(f)]
    assert any(want in n for n in names), want


def test_flops_and_peak_constants():
    # 2 * (512*36) * 1024 * 2048 FLOP per launch of the roofline GEMM; f32-input MFMA peak of the guide
    assert bench.F32_MFMA_PEAK_TFLOPS == pytest.approx(157.3)
    cfg = bench.CFG
    assert (cfg["B"], cfg["R"], cfg["D"], cfg["H"], cfg["T"], cfg["A"]) == (512, 36, 2048, 1024, 14, 3000)


def test_tools_and_entry_points_compile():
    """every script under tools/ (some are driven by tests and by the profile recipes) and the two root entry points
    are at least syntactically valid python"""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "dbg", "*.py")))
    paths += [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(paths) > 15
    for p in paths:
        with open(p) as f:
            compile(f.read(), p, "exec")
