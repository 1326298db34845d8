"""In-kernel time stamps (s_memrealtime, 10 ns) of the fused GRU-step kernels: where do the per-kernel microseconds go?"""
import ctypes as C, os, numpy as np, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_dbg.so"))
T, B, H = 14, 512, 1024
g = torch.Generator(device="cuda").manual_seed(0)
xp = torch.randn(T, B, 3 * H, device="cuda", generator=g) * 0.1
Wg = torch.randn(H, 2 * H, device="cuda", generator=g) * 0.03
Wc = torch.randn(H, H, device="cuda", generator=g) * 0.03
ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
hs = torch.zeros(T + 1, B, H, device="cuda"); r = torch.empty(T, B, H, device="cuda"); u = torch.empty_like(r); c = torch.empty_like(r); rh = torch.empty_like(r)
P = lambda t: C.c_void_p(t.data_ptr())
lib.vqa_gru_seq_fwd.argtypes = [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_void_p]
def run():
    rc = lib.vqa_gru_seq_fwd(P(xp), P(Wg), P(Wc), P(ln), P(hs), P(r), P(u), P(c), P(rh), T, B, H, None); assert rc == 0, rc
lib.vqa_gemm_set_gru_config(int(os.environ.get("GRU_CFG", 11)))
for _ in range(3): run()      # first call consumes launch ids 0..27 + more; reset after warm-up
torch.cuda.synchronize()
lib.vqa_dbg_reset()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("whole recurrence: %.1f us" % (e0.elapsed_time(e1) * 1e3))
n = 28 * 1024 * 48
buf = (C.c_ulonglong * n)()
lib.vqa_dbg_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.vqa_dbg_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(28, 1024, 48).astype(np.int64)
t00 = None
prev_end = None
NB = (int(os.environ.get('NB0', 512)), int(os.environ.get('NB1', 256)))
for k in range(28):
    nb = NB[k % 2]
    a = s[k, :nb, :6] * 10.0 / 1e3          # us
    base = a[:, 0].min()
    if t00 is None: t00 = base
    rel = a - base
    gap = (base - prev_end) if prev_end is not None else 0.0
    prev_end = a[:, 5].max()
    q = lambda x: "%.1f/%.1f/%.1f" % (np.min(x), np.median(x), np.max(x))
    print("k%02d %s start+%.1f gap %.1f | entry %s | loads-issued %s | tile0-in-LDS %s | loop-done %s | wgk-red %s | end %s | total %.1f"
          % (k, "gates" if k % 2 == 0 else "cand ", base - t00, gap, q(rel[:, 0]), q(rel[:, 1]), q(rel[:, 2]), q(rel[:, 3]), q(rel[:, 4]), q(rel[:, 5]), rel[:, 5].max()))

k = 4
a = s[k, :NB[0], :] * 10.0 / 1e3
base = a[:, 0].min()
it = a[:, 8:8 + 8] - base
print("gates kernel k04: per-pair-iteration stamp (us since kernel start), median over blocks:", np.round(np.median(it, axis=0), 2))
print("  block 0:", np.round(it[0], 2), " block 300:", np.round(it[300], 2))

c = s[k, :NB[0], 20:40].reshape(NB[0], 4, 5).astype(np.float64)   # [block, wave, stamp] in shader cycles
d = np.diff(c, axis=2)
names = ["MFMA tile (16 MFMAs)", "wait vmcnt(0) for the set loaded 2 tiles ago", "6 ds_write + lgkmcnt(0)", "s_barrier"]
for i, nm in enumerate(names):
    x = d[:, :, i].reshape(-1)
    print("  %-48s cycles min/med/p90/max: %d / %d / %d / %d" % (nm, x.min(), np.median(x), np.percentile(x, 90), x.max()))
x = (c[:, :, 4] - c[:, :, 0]).reshape(-1)
print("  whole half-iteration: med %d cycles" % np.median(x))
