"""Tile-config sweep for the fused GRU recurrence (vqa_gru_seq_fwd / _bwd) at B 512, H 1024, T 14; min of repeats."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_transfer_externaldata_amd import _lib  # noqa: E402

if os.environ.get("VQA_HOT_LIB"):      # another build of the library (same-box A/B of compile-time choices)
    _lib._LIB_PATH = os.path.abspath(os.environ["VQA_HOT_LIB"])
lib = _lib.load()
T, B, H = int(os.environ.get("GRU_T", 14)), int(os.environ.get("GRU_B", 512)), 1024
g = torch.Generator(device="cuda").manual_seed(0)
xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.1
Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.03
Wc = torch.randn(H, H, device="cuda", generator=g) * 0.03
ln = torch.randint(1, T + 1, (B,), dtype=torch.int32, device="cuda", generator=g)
hs = torch.zeros(T + 1, B, H, device="cuda")
r = torch.empty(T, B, H, device="cuda"); u = torch.empty_like(r); c = torch.empty_like(r); rh = torch.empty_like(r)
dhT0 = torch.randn(B, H, device="cuda", generator=g); dhT = dhT0.clone()
dxp = torch.empty(T, B, 3 * H, device="cuda"); dhs = torch.empty(B, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())


def fwd():
    _lib.check(lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None), "fwd")


def bwd():
    dhT.copy_(dhT0)
    _lib.check(lib.vqa_gru_seq_bwd(P(dhT), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H, None), "bwd")


def tm(f, n=30):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def tm_cold(f, n=10):
    """each call timed on its own after 1 GiB of unrelated writes (weights, tape and xp out of the caches, as in a step)"""
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(n):
        big.zero_()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


sync = torch.zeros(int(lib.vqa_gru_persistent_sync_bytes()) // 4, dtype=torch.int32, device="cuda")


def fwd_persistent():
    _lib.check(lib.vqa_gru_seq_fwd_persistent(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, P(sync),
                                              None), "fwd_persistent")


if os.environ.get("PERSIST") == "1":
    fwd(); torch.cuda.synchronize()
    ref = [x.clone() for x in (hs, r, u, c, rh)]
    for x in (r, u, c, rh):
        x.fill_(float("nan"))
    hs[1:].fill_(float("nan"))
    fwd_persistent(); torch.cuda.synchronize()
    print("persistent: sync words [0, 16, 32, 192]", sync[[0, 16, 32, 192]].tolist(),
          "(xcd mode: chain counters at 16 * chain, error word 192)" if os.environ.get("VQA_GRU_PERSIST_XCD") == "1" else "", flush=True)
    census = torch.zeros(1024 + 2 * 64 * 4 * 2, dtype=torch.int32, device="cuda")
    lib.vqa_gru_persistent_set_census(P(census)); fwd_persistent(); torch.cuda.synchronize(); lib.vqa_gru_persistent_set_census(None)
    cz = census[:512].cpu().numpy().astype("int64")
    where = {}
    for b, w in enumerate(cz):
        key = (int(w >> 16) & 15, int(w >> 13) & 7, int(w >> 12) & 1, int(w >> 8) & 15)      # xcc, se, sh, cu
        where.setdefault(key, []).append(b)
    pairs = sorted(where.values())
    print("  placement: %d distinct CUs; workgroups per CU: %s" % (len(where), sorted(set(len(v) for v in pairs))))
    print("  first CUs:", pairs[:6], " xcc of blocks 0..9:", [int(w >> 16) & 15 for w in cz[:10]])
    st = census[1024:].cpu().numpy().view("uint64").reshape(2, 64, 4).astype("float64") / 100.0     # 100 MHz -> us
    t0 = st[:, 0, 0].min()
    for ch in range(2):
        print("  chain %d workgroup: phase (start, wait, k-loop, epilogue) in us relative to launch" % ch)
        for ph in range(min(2 * T, 8)):
            a0, a1, a2, a3 = st[ch, ph]
            print("    %s t=%d  start %7.2f  wait %6.2f  kloop %6.2f  epi %6.2f" % ("G" if ph % 2 == 0 else "C", ph // 2, a0 - t0, a1 - a0, a2 - a1, a3 - a2))
        print("    ... last phase ends at %.2f us" % (st[ch, 2 * T - 1, 3] - t0))
    same = sum(1 for v in pairs if len(v) == 2 and (v[0] < 256) == (v[1] < 256))
    print("  CUs whose two workgroups are in the SAME chain (blocks both < 256 or both >= 256): %d of %d" % (same, len(pairs)), flush=True)
    for name, a, b in zip(("hs", "r", "u", "c", "rh"), ref, (hs, r, u, c, rh)):
        print("  max |%s - stepwise| = %.3e  (max |.| %.3e)" % (name, float((a - b).abs().max()), float(a.abs().max())), flush=True)
    for rep in range(3):
        print("  forward: stepwise %.1f us   persistent %.1f us" % (tm(fwd), tm(fwd_persistent)), flush=True)
    sys.exit(0)

if os.environ.get("WS") == "1":
    # weight-stationary persistent forward (csrc/gru_ws.hip) against the step kernels: results, phase stamps, time
    if lib.vqa_gru_ws_supported(T, B, H) != 1:
        print("weight-stationary recurrence does not apply here"); sys.exit(0)
    ws = torch.empty(int(lib.vqa_gru_ws_workspace_bytes(T)) // 4, dtype=torch.float32, device="cuda").fill_(float("nan"))

    def fwd_ws():
        _lib.check(lib.vqa_gru_seq_fwd_ws(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, P(ws), None), "fwd_ws")

    lib.vqa_gru_ws_set_form(int(os.environ.get("WS_FORM", 0)))      # 1: plain sub-phase order (stamps only there)
    fwd(); torch.cuda.synchronize()
    ref = [x.clone() for x in (hs, r, u, c, rh)]
    for x in (r, u, c, rh):
        x.fill_(float("nan"))
    hs[1:].fill_(float("nan"))
    stamps = torch.zeros(2048, dtype=torch.int64, device="cuda")
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); big.zero_()      # the stamped launch starts from cold caches
    lib.vqa_gru_ws_set_stamps(P(stamps)); fwd_ws(); torch.cuda.synchronize(); lib.vqa_gru_ws_set_stamps(None)
    words = ws[:1024].view(torch.int32)
    print("ws: flags (min, max per half-chain)", [(int(words[32 * i: 32 * i + 32].min()), int(words[32 * i: 32 * i + 32].max())) for i in range(16)],
          "error word", int(words[512]), "workgroups per XCD", words[576:584].tolist(), flush=True)
    for name, a, b in zip(("hs", "r", "u", "c", "rh"), ref, (hs, r, u, c, rh)):
        print("  max |%s - stepwise| = %.3e  (max |.| %.3e)" % (name, float((a - b).abs().max()), float(a.abs().max())), flush=True)
    st = stamps.cpu().numpy().astype("float64") / 100.0
    nsub = 4 if B > 256 else 2
    t0 = st[0]
    names = ["G0", "G1", "C0", "C1"] if nsub == 4 else ["G0", "C0"]
    for k in range(min(nsub * T, 12) if (st[2] > 0 and (B <= 256 or os.environ.get("WS_FORM") == "1")) else 0):
        a0, a1, a2 = st[3 * k: 3 * k + 3]
        print("    %s t=%d  start %7.2f  compute %6.2f  reduce+epilogue+arrive %6.2f" % (names[k % nsub], k // nsub, a0 - t0, a1 - a0, a2 - a1))
    if st[2] > 0 and (B <= 256 or os.environ.get("WS_FORM") == "1"):
        print("    ... last sub-phase ends at %.2f us" % (st[3 * (nsub * T - 1) + 2] - t0), flush=True)
    elif st[2] > 0:     # the spliced form stamps the start of every matrix stream (G0 G1 C0 C1 per step) and the end:
        raw = stamps.cpu().numpy().astype("float64")            # (100 MHz wall clock, shader clock) pairs
        wall, cyc = raw[0:8 * T + 2:2] / 100.0, raw[1:8 * T + 2:2]
        d, dc = wall[1:] - wall[:-1], cyc[1:] - cyc[:-1]
        for t in range(T):
            print("    t=%2d  G0 %5.2f  G1 %5.2f  C0 %5.2f  C1 %5.2f us  (step %5.2f us)   cycles %6d %6d %6d %6d   clock %.2f GHz" % (
                (t,) + tuple(d[4 * t: 4 * t + 4]) + (d[4 * t: 4 * t + 4].sum(),) + tuple(int(x) for x in dc[4 * t: 4 * t + 4]) +
                (dc[4 * t: 4 * t + 4].sum() / d[4 * t: 4 * t + 4].sum() / 1e3,)))
        print("    all streams %.2f us; a gate stream is 256 MFMAs per wave = 16384 cycles, a candidate stream 8192" % (wall[-1] - wall[0]), flush=True)
        print("    prologue (kernel entry -> first stream: placement, weights into registers and LDS, h_0 hand-off), cold caches: %.2f us" % ((raw[2047] - raw[2046]) / 100.0), flush=True)
        sl = stamps[1024:1536].cpu().numpy().astype("int64")
        if sl.any():        # a -DWS_SLOTS=n build: shader cycles between every n-th slot of step 5's streams
            for si, nm in enumerate(("G0", "G1", "C0", "C1")):
                v = sl[128 * si: 128 * si + 128]
                pos = np.nonzero(v)[0]
                print("    %s cycles between stamped slots %s: %s" % (nm, pos.tolist(), np.diff(v[pos]).tolist()))
    for rep in range(3):
        print("  forward: stepwise %.1f us   weight-stationary %.1f us" % (tm(fwd), tm(fwd_ws)), flush=True)
    print("  forward from cold caches (median of 10): stepwise %.1f us   weight-stationary %.1f us" % (tm_cold(fwd), tm_cold(fwd_ws)), flush=True)
    sys.exit(0)

if os.environ.get("WSB") == "1":
    # weight-stationary back-propagation through time (csrc/gru_ws.hip) against the step kernels
    if lib.vqa_gru_ws_bwd_supported(T, B, H) != 1:
        print("weight-stationary back-propagation does not apply here"); sys.exit(0)
    ws = torch.empty(int(lib.vqa_gru_ws_workspace_bytes(T)) // 4, dtype=torch.float32, device="cuda").fill_(float("nan"))
    dxp_ws = torch.full_like(dxp, float("nan"))

    def bwd_ws():
        _lib.check(lib.vqa_gru_seq_bwd_ws(P(dhT0), None, P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp_ws), T, B, H, P(ws), None), "bwd_ws")

    fwd(); bwd(); torch.cuda.synchronize()
    bwd_ws(); torch.cuda.synchronize()
    words = ws[:1024].view(torch.int32)
    print("ws: flags (min, max per half-chain)", [(int(words[32 * i: 32 * i + 32].min()), int(words[32 * i: 32 * i + 32].max())) for i in range(16)],
          "error word", int(words[512]), flush=True)
    for nm, lo in (("dr_pre", 0), ("du_pre", H), ("dc_pre", 2 * H)):
        a_, b_ = dxp[:, :, lo:lo + H], dxp_ws[:, :, lo:lo + H]
        print("  max |%s - stepwise| = %.3e  (max |.| %.3e)  nan %d" % (nm, float((a_ - b_).abs().max()), float(a_.abs().max()), int(torch.isnan(b_).sum())), flush=True)
    for rep in range(3):
        print("  backward: stepwise %.1f us   weight-stationary %.1f us" % (tm(bwd), tm(bwd_ws)), flush=True)
    print("  backward from cold caches (median of 10): stepwise %.1f us   weight-stationary %.1f us" % (tm_cold(bwd), tm_cold(bwd_ws)), flush=True)
    sys.exit(0)

if os.environ.get("GRAPH") == "1":
    # does replaying the 28-kernel chain from a captured graph shorten the gaps between its dependent kernels?
    def on_stream(fn_name, args):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(getattr(lib, fn_name)(*args, st), fn_name)
    fa = (P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H)
    ba = (P(dhT), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        on_stream("vqa_gru_seq_fwd", fa); on_stream("vqa_gru_seq_bwd", ba)
    torch.cuda.synchronize()
    gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf, stream=side):
        on_stream("vqa_gru_seq_fwd", fa)
    with torch.cuda.graph(gb, stream=side):
        on_stream("vqa_gru_seq_bwd", ba)
    for rep in range(3):
        print("forward: launches %.1f us  graph replay %.1f us | backward: launches %.1f us  graph replay %.1f us" % (
            tm(fwd), tm(gf.replay), tm(lambda: _lib.check(lib.vqa_gru_seq_bwd(*ba, None), "bwd")), tm(gb.replay)), flush=True)
    sys.exit(0)

if os.environ.get("SPLIT2") == "1":
    # Two half-batch chains (rows [0, B/2) and [B/2, B)) on two streams, the second one delayed: do one chain's kernel
    # boundaries (drain, launch gap, first tile) hide behind the other chain's matrix work when the chains run in
    # ANTI-phase?  (in phase -- no delay -- both compute at half rate and both wait together: no gain by construction)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    half = B // 2
    clock_hz = 2.1e9     # torch.cuda._sleep counts device cycles; only the relative delays matter

    NCH = int(os.environ.get("SPLIT_CHAINS", 2))
    streams = [torch.cuda.Stream() for _ in range(NCH)]
    tiles = (B + 31) // 32
    bounds = [min(B, 32 * ((tiles * i) // NCH)) for i in range(NCH + 1)]
    DIR = os.environ.get("SPLIT_DIR", "fwd")

    def enqueue(row0, rows, st):
        if DIR == "fwd":
            _lib.check(lib.vqa_gru_seq_fwd_rows(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, row0, rows,
                                                C.c_void_p(st.cuda_stream)), "fwd_rows")
        else:
            _lib.check(lib.vqa_gru_seq_bwd_rows(P(dhT), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(dxp), P(dhs), T, B, H,
                                                row0, rows, C.c_void_p(st.cuda_stream)), "bwd_rows")

    def run(delay_us, cfg, n=10):
        ts = []
        for _ in range(n):
            if DIR != "fwd":
                dhT.copy_(dhT0)
            torch.cuda.synchronize()
            _lib.check(lib.vqa_gemm_set_gru_config(cfg), "cfg")
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            gate = torch.cuda.Event()
            torch.cuda._sleep(int(600e-6 * clock_hz))           # hold every stream back until everything is enqueued
            gate.record()
            done = []
            for i, st in enumerate(streams):
                st.wait_event(gate)
                with torch.cuda.stream(st):
                    if i == 0:
                        t0.record()
                    elif delay_us > 0:
                        _lib.check(lib.vqa_stream_delay_us(delay_us * i, C.c_void_p(st.cuda_stream)), "delay")
                    enqueue(bounds[i], bounds[i + 1] - bounds[i], st)
                    if i > 0:
                        e = torch.cuda.Event(); e.record(); done.append(e)
            with torch.cuda.stream(streams[0]):
                for e in done:
                    streams[0].wait_event(e)
                t1.record()
            torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1) * 1e3)
        return min(ts), sorted(ts)[len(ts) // 2]

    fwd(); torch.cuda.synchronize()
    ref = hs[T].clone()
    _lib.check(lib.vqa_gemm_set_gru_config(-1), "cfg")
    print("one chain, one stream: %.1f us (%s)" % (tm(fwd if DIR == "fwd" else bwd), DIR), flush=True)
    for cfg in [int(x) for x in os.environ.get("SPLIT2_CFGS", "16,9,11,7,8").split(",")]:
        for delay in [float(x) for x in os.environ.get("SPLIT2_DELAYS", "0,4,8,12,16,24").split(",")]:
            hs[1:].zero_()
            best, med = run(delay, cfg)
            err = float((hs[T] - ref).abs().max())
            print("%d chains on %d streams (%s), cfg %2d, chain i delayed i x %4.1f us: all done after %.1f us (median %.1f)  max |h - ref| %.1e"
                  % (NCH, NCH, DIR, cfg, delay, best, med, err), flush=True)
    _lib.check(lib.vqa_gemm_set_gru_config(-1), "cfg")
    sys.exit(0)

cfgs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "11,16,17,18").split(",")]
best = {k: [1e9, 1e9] for k in cfgs}
ref = None
for rep in range(3):
    for k in cfgs:
        _lib.check(lib.vqa_gemm_set_gru_config(k), "cfg")
        tf, tb = tm(fwd), tm(bwd)
        best[k][0] = min(best[k][0], tf); best[k][1] = min(best[k][1], tb)
        if rep == 0:
            out = torch.cat([hs[T].flatten(), dxp.flatten()[::97]]).double()
            if ref is None:
                ref = out
            print("cfg %2d max |diff| vs first cfg: %.3e" % (k, float((out - ref).abs().max())), flush=True)
for k, (tf, tb) in best.items():
    print("gru cfg %2d: fwd %.1f us  bwd %.1f us  (min of 3 x 30)" % (k, tf, tb), flush=True)
_lib.check(lib.vqa_gemm_set_gru_config(-1), "cfg")
print("defaults (fwd 18 / bwd 18): fwd %.1f us  bwd %.1f us" % (tm(fwd), tm(bwd)))
