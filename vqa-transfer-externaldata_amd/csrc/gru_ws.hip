// Weight-stationary persistent GRU recurrence for gfx950 (H = 1024): all T steps of tf.contrib.rnn.GRUCell under
// tf.nn.dynamic_rnn(sequence_length) (vlmap/modules.py:124-140) in ONE launch, the recurrent weights loaded ONCE.
//
// Why this form.  The step kernels (gemm_f32.hip) pay ~12 us per launch that is not matrix work (profiles/
// r3_gru_loop_variants.txt); the first persistent form (gru_persistent.hip) removed the launches but every XCD then
// streamed all 12 MB of recurrent weights from the Infinity Cache in every step (its k loops ran at 0.6-0.7 of the
// matrix rate, profiles/r2_gru_persistent_xcd.txt).  Here nothing but the state moves per step:
//
//   * eight chains, one per XCD: chain x = the workgroups with blockIdx.x % 8 == x (dispatched round-robin to XCD x)
//     owns batch rows [64 x, 64 x + 64) -- rows are independent sequences, so hand-offs and barriers stay inside one
//     XCD's 32 CUs and one L2;
//   * CU j of a chain owns the 32 state columns [32 j, 32 j + 32): the matching column slabs of the gate weights
//     (1024 x 64 floats = 256 KB) live in the REGISTERS of its four waves (one wave per SIMD, 256 of its 512 VGPRs
//     each: wave w holds the k quarter [256 w, 256 w + 256) as ready-made MFMA B fragments), the candidate slab
//     (1024 x 32 = 128 KB) lives in LDS in fragment order;
//   * per step only the left operand moves: h_{t-1} and r * h_{t-1} of the chain's rows (128 KB each per 32-row
//     half) are handed from the CUs that produce their columns to all CUs of the XCD in MFMA A-fragment order
//     ([octet of k][lane][4]: one contiguous 1 KB per wave load, one dwordx4 per lane feeds 4 or 8 MFMAs);
//   * the four waves split k; their partial tiles meet in LDS (32 KB, XOR-swizzled), every thread finishes one
//     float4 of the tile (gate math in registers: h, u and the length of its row stay in registers over all steps);
//   * a chain's 64 rows are two half-chains of 32 rows in ANTI-PHASE on the same waves: G(0,t) G(1,t) C(0,t) C(1,t).
//     What a sub-phase waits for was produced one sub-phase ago on every CU, so the counter is already there when it
//     is polled and the first operand fragments are fetched under the previous sub-phase's reduction and epilogue.
//
// Hand-off protocol: as gru_persistent.hip (write-through stores, drain, workgroup barrier, one agent-scope atomic
// per workgroup; consumers poll, barrier, then read).  Every hand-off address is written once per launch (the
// fragment buffers are indexed by t), so no stale line can sit in an L1.  Spins are bounded and set an error word.
#include <stdlib.h>

#include "vqa_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WS_H = 1024;
constexpr int WS_NT = 256;                  // 4 waves, one per SIMD, 512 VGPRs each
constexpr int WS_CHAINS = 8, WS_CU = 32;    // chains (XCDs) x column slabs (CUs per XCD)
constexpr int WS_RING = 8;                  // operand fragments in flight per wave (dwordx4 each)
constexpr int WS_SC1 = 16;
// Hand-offs are PLAIN stores: a write-through (sc1) store drops its line from the XCD's L2 (MI355X_MICROARCH.md, stores
// of each flavour), and the 32 consumers of the same XCD then fetch it at the cross-XCD rate -- measured: the k loops
// ran at half the matrix rate.  A plain store keeps the line in this L2, where every consumer of the chain reads it;
// the bytes still leave for memory, so a workgroup that was ever placed on another XCD misses and reads them there.
#ifndef WS_HANDOFF_SC1
#define WS_HANDOFF_SC1 false
#endif
constexpr int WS_FRAG_T = WS_CHAINS * 2 * 128 * 256;      // floats of one time step's fragment buffer (2 MB)
constexpr size_t WS_LDS = (32768 + 8192) * sizeof(float); // candidate slab + reduction scratch = 160 KB

struct WsArgs {
    const float* xp;      // [T,B,3H]
    const float* Wg;      // [H,2H]
    const float* Wc;      // [H,H]
    const int32_t* len;   // [B]
    float *hs, *r, *u, *c, *rh;   // hs [T+1,B,H] (hs[0] given), tape [T,B,H]
    float* hF;            // [T+1] fragment-order copies of h_t      (hand-off)
    float* rhF;           // [T]   fragment-order copies of r * h    (hand-off)
    unsigned* sync;       // counters at [16 * (2 * chain + half)], error word at [256]
    int T, B;
    unsigned spin_limit;
    unsigned long long* stamps;   // optional timing study: workgroup 0 records (wait, compute, done) per sub-phase
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rs(const float* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4n ws_load(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    f32x4n o;
    o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
    return o;
}
__device__ __forceinline__ void ws_store(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4n x, bool sc1) {
    u32x4 v;
    v.x = __float_as_uint(x.x); v.y = __float_as_uint(x.y); v.z = __float_as_uint(x.z); v.w = __float_as_uint(x.w);
    if (sc1) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, WS_SC1);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
}
__device__ __forceinline__ void ws_arrive(unsigned* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ws_wait(unsigned* ctr, unsigned target, unsigned* err, unsigned limit) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

#define WS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// The gate slabs fill the wave's 256 accumulation registers (AGPRs) and the MFMA takes its B operand from there
// directly.  Written through the intrinsic, the register allocator treats those registers as spill space and copies
// every weight back with v_accvgpr_read + s_nop in front of its MFMA: 78 cycles per MFMA instead of 64.  The compiler
// does not see an MFMA in the asm, so the read-after-MFMA wait states are spelled out where the tile is first read.
#define WS_MFMA_AGPR_B(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define WS_MFMA_VGPR_B(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
constexpr int WS_AG_OCT = 28;       // octets of a wave's k quarter whose fragments sit in AGPRs (8 each); the rest in VGPRs
#define WS_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

__global__ __launch_bounds__(WS_NT, 1) void gru_ws_fwd_kernel(WsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                   // candidate slab, fragment order: [(octet * 64 + lane) * 4 + i]
    float* S = smem + 32768;            // reduction scratch (8192 floats)
    constexpr int H = WS_H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chain = blockIdx.x & 7, j = blockIdx.x >> 3;       // XCD, column slab
    const int B = a.B, T = a.T;
    const int row_base = chain * 64;
    const int halves = (B - row_base > 32) ? 2 : 1;              // 32-row half-chains with any row in the batch
    unsigned* err = a.sync + 256;
    if (row_base >= B) return;          // a chain with no rows: nobody waits for it (counters are per chain)

    // ---- recurrent weights, once: gate slabs into registers, candidate slab into LDS
    // B fragment of the MFMA for k pair i of octet m: lane l supplies W[8 m + 4 (l >> 5) + i][column l & 31]
    float wg[2][32][4];
    {
        const float* p = a.Wg + (int64_t)(256 * w + 4 * (lane >> 5)) * (2 * H) + 32 * j + (lane & 31);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int m = 0; m < 32; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) wg[cb][m][i] = p[(int64_t)(8 * m + i) * (2 * H) + cb * H];
    }
    for (int idx = tid; idx < 128 * 64; idx += WS_NT) {
        const int oct = idx >> 6, l = idx & 63;
        const float* p = a.Wc + (int64_t)(8 * oct + 4 * (l >> 5)) * H + 32 * j + (l & 31);
        f32x4n v;
        v.x = p[0]; v.y = p[H]; v.z = p[2 * H]; v.w = p[3 * H];
        *reinterpret_cast<f32x4n*>(Wl + idx * 4) = v;
    }

    // ---- this thread's float4 of every 32 x 32 tile: row lane % 32, state columns 32 j + 8 w + 4 (lane / 32) + 0..3
    const int e_row = lane & 31, e_col = 8 * w + 4 * (lane >> 5);
    const int64_t BH = (int64_t)B * H;
    const __amdgpu_buffer_rsrc_t rs_hF = ws_rs(a.hF, (int64_t)(T + 1) * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_rhF = ws_rs(a.rhF, (int64_t)T * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_xp = ws_rs(a.xp, (int64_t)T * B * 3 * H * 4);
    const __amdgpu_buffer_rsrc_t rs_hs = ws_rs(a.hs, (T + 1) * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_r = ws_rs(a.r, T * BH * 4), rs_u = ws_rs(a.u, T * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_c = ws_rs(a.c, T * BH * 4), rs_rh = ws_rs(a.rh, T * BH * 4);
    // fragment buffer offsets (bytes): operand loads of this wave's k quarter, and this thread's hand-off store
    auto frag_ld = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 32 * w) * 1024 + lane * 16); };
    auto frag_st = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j + w) * 1024 + lane * 16); };

    f32x4n h_own[2], u_own[2];
    int len_own[2];
    bool ok[2];
    unsigned o_std[2];                  // byte offset of this thread's float4 inside a [B,H] matrix
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int grow = row_base + 32 * hf + e_row;
        ok[hf] = grow < B;
        o_std[hf] = (unsigned)(((int64_t)grow * H + 32 * j + e_col) * 4);
        h_own[hf] = ok[hf] ? ws_load(rs_hs, o_std[hf]) : (f32x4n)(0.f);
        u_own[hf] = (f32x4n)(0.f);
        len_own[hf] = ok[hf] ? a.len[grow] : 0;
        if (hf < halves) ws_store(rs_hF, frag_st(0, hf), h_own[hf], WS_HANDOFF_SC1);
    }
    __syncthreads();                    // the candidate slab is in LDS
    for (int hf = 0; hf < halves; ++hf) ws_arrive(a.sync + 16 * (2 * chain + hf));

    f32x4n ring[WS_RING];
    auto fill = [&](__amdgpu_buffer_rsrc_t rs, unsigned base) {
#pragma unroll
        for (int m = 0; m < WS_RING; ++m) ring[m] = ws_load(rs, base + m * 1024);
    };
    unsigned long long* stamp = (a.stamps != nullptr && blockIdx.x == 0 && tid == 0) ? a.stamps : nullptr;

    // the first sub-phase's operand: G(0, 0) needs every CU's share of h_0
    ws_wait(a.sync + 16 * (2 * chain), WS_CU, err, a.spin_limit);
    fill(rs_hF, frag_ld(0, 0));

    for (int t = 0; t < T; ++t) {
        // ================================================================ gates: r | u = sigmoid(h W_g + xp)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf < halves) {
                unsigned* ctr = a.sync + 16 * (2 * chain + hf);
                const unsigned base = frag_ld(t, hf);
                const unsigned oxp = ok[hf] ? (unsigned)((((int64_t)t * B + row_base + 32 * hf + e_row) * 3 * H + 32 * j + e_col) * 4) : 0xFFFFFFF0u;
                const f32x4n xr = ws_load(rs_xp, oxp), xu = ws_load(rs_xp, ok[hf] ? oxp + H * 4 : 0xFFFFFFF0u);
                if (stamp) stamp[0] = wall_clock64();
                f32x16 accR, accU;
#pragma unroll
                for (int q = 0; q < 16; ++q) { accR[q] = 0.f; accU[q] = 0.f; }
#pragma unroll
                for (int m = 0; m < 32; ++m) {
                    const f32x4n av = ring[m % WS_RING];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (m < WS_AG_OCT) {
                            WS_MFMA_AGPR_B(accR, av[i], wg[0][m][i]);
                            WS_MFMA_AGPR_B(accU, av[i], wg[1][m][i]);
                        } else {
                            WS_MFMA_VGPR_B(accR, av[i], wg[0][m][i]);
                            WS_MFMA_VGPR_B(accU, av[i], wg[1][m][i]);
                        }
                    }
                    // the refill of this slot stays HERE (left alone, the scheduler sinks every load to its use,
                    // one exposed L2 round trip per octet: the loop then runs at half the matrix rate)
                    if (m + WS_RING < 32) ring[m % WS_RING] = ws_load(rs_hF, base + (m + WS_RING) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                WS_MFMA_DRAIN();
                if (stamp) stamp[1] = wall_clock64();
                // ---- what comes next on this CU: G(1,t) after G(0,t); C(0,t) after G(1,t) (or after G(0,t) alone)
                if (halves == 2) {      // its producers finished a sub-phase ago: fetch its first fragments now
                    if (hf == 0) { ws_wait(a.sync + 16 * (2 * chain + 1), WS_CU * (1 + 2 * t), err, a.spin_limit); fill(rs_hF, frag_ld(t, 1)); }
                    else { ws_wait(a.sync + 16 * (2 * chain), WS_CU * (2 + 2 * t), err, a.spin_limit); fill(rs_rhF, frag_ld(t, 0)); }
                }
                // ---- the four k quarters meet in LDS: S[wave][row][64], float4 slots XOR-swizzled by row
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), col = lane & 31;
                    float* d = S + w * 2048 + row * 64 + (col & 3);
                    d[(((col >> 2)) ^ (row & 15)) << 2] = accR[q];
                    d[(((col >> 2) + 8) ^ (row & 15)) << 2] = accU[q];
                }
                __syncthreads();
                f32x4n rv = xr, uv = xu;
                {
                    const float* s = S + e_row * 64;
                    const int sr = ((e_col >> 2) ^ (e_row & 15)) << 2, su = (((e_col >> 2) + 8) ^ (e_row & 15)) << 2;
                    f32x4n pr[4], pu[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        pr[g] = *reinterpret_cast<const f32x4n*>(s + g * 2048 + sr);
                        pu[g] = *reinterpret_cast<const f32x4n*>(s + g * 2048 + su);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) { rv += pr[g]; uv += pu[g]; }
                }
                f32x4n rhv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rv[i] = sigmoidf_stable(rv[i]);
                    uv[i] = sigmoidf_stable(uv[i]);
                    rhv[i] = rv[i] * h_own[hf][i];
                }
                u_own[hf] = uv;
                ws_store(rs_rhF, frag_st(t, hf), rhv, WS_HANDOFF_SC1);
                if (ok[hf]) {
                    const unsigned o = (unsigned)(t * BH * 4) + o_std[hf];
                    ws_store(rs_r, o, rv, false);
                    ws_store(rs_u, o, uv, false);
                    ws_store(rs_rh, o, rhv, false);
                }
                ws_arrive(ctr);         // (its barrier also frees the scratch for the next sub-phase)
                if (halves == 1) { ws_wait(ctr, WS_CU * (2 + 2 * t), err, a.spin_limit); fill(rs_rhF, frag_ld(t, 0)); }
                if (stamp) { stamp[2] = wall_clock64(); stamp += 3; }
            }
        }
        // ================================================================ candidate: c = tanh((r h) W_c + xp), h_t
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf < halves) {
                unsigned* ctr = a.sync + 16 * (2 * chain + hf);
                const unsigned base = frag_ld(t, hf);
                const f32x4n xc = ws_load(rs_xp, ok[hf] ? (unsigned)((((int64_t)t * B + row_base + 32 * hf + e_row) * 3 * H + 2 * H + 32 * j + e_col) * 4) : 0xFFFFFFF0u);
                if (stamp) stamp[0] = wall_clock64();
                f32x16 acc0, acc1;
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
                f32x4n bq[2];           // the candidate slab's fragments, two octets ahead of the MFMAs
                bq[0] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w) * 64 + lane) * 4);
                bq[1] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w + 1) * 64 + lane) * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 32; ++m) {
                    const f32x4n av = ring[m % WS_RING];
                    const f32x4n bv = bq[m & 1];
                    if (m + 2 < 32) bq[m & 1] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w + m + 2) * 64 + lane) * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (m & 1) acc1 = WS_MFMA(av[i], bv[i], acc1);
                        else acc0 = WS_MFMA(av[i], bv[i], acc0);
                    }
                    if (m + WS_RING < 32) ring[m % WS_RING] = ws_load(rs_rhF, base + (m + WS_RING) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc0 += acc1;
                if (stamp) stamp[1] = wall_clock64();
                // ---- next on this CU: C(1,t) after C(0,t); G(0,t+1) after C(1,t) (or after C(0,t) alone)
                if (halves == 2) {
                    if (hf == 0) { ws_wait(a.sync + 16 * (2 * chain + 1), WS_CU * (2 + 2 * t), err, a.spin_limit); fill(rs_rhF, frag_ld(t, 1)); }
                    else if (t + 1 < T) { ws_wait(a.sync + 16 * (2 * chain), WS_CU * (3 + 2 * t), err, a.spin_limit); fill(rs_hF, frag_ld(t + 1, 0)); }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), col = lane & 31;
                    S[w * 1024 + row * 32 + ((((col >> 2)) ^ ((row >> 1) & 7)) << 2) + (col & 3)] = acc0[q];
                }
                __syncthreads();
                f32x4n cv = xc;
                {
                    const float* s = S + e_row * 32 + (((e_col >> 2) ^ ((e_row >> 1) & 7)) << 2);
                    f32x4n pc[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) pc[g] = *reinterpret_cast<const f32x4n*>(s + g * 1024);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) cv += pc[g];
                }
                f32x4n hn;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cv[i] = tanhf(cv[i]);
                    hn[i] = (t < len_own[hf]) ? (u_own[hf][i] * h_own[hf][i] + (1.f - u_own[hf][i]) * cv[i]) : h_own[hf][i];
                }
                h_own[hf] = hn;
                ws_store(rs_hF, frag_st(t + 1, hf), hn, WS_HANDOFF_SC1);
                if (ok[hf]) {
                    ws_store(rs_c, (unsigned)(t * BH * 4) + o_std[hf], cv, false);
                    ws_store(rs_hs, (unsigned)((t + 1) * BH * 4) + o_std[hf], hn, false);
                }
                ws_arrive(ctr);
                if (halves == 1 && t + 1 < T) { ws_wait(ctr, WS_CU * (3 + 2 * t), err, a.spin_limit); fill(rs_hF, frag_ld(t + 1, 0)); }
                if (stamp) { stamp[2] = wall_clock64(); stamp += 3; }
            }
        }
    }
}

struct WsDevice { int cus = 0, blocks = -1; };
const WsDevice& ws_device() {
    static WsDevice per_dev[64];
    static bool known[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    WsDevice& d = per_dev[dev];
    if (!known[dev]) {
        known[dev] = true;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return d;
        d.cus = prop.multiProcessorCount;
        int n = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_ws_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WS_LDS) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_ws_fwd_kernel, WS_NT, WS_LDS) == hipSuccess)
            d.blocks = n;
    }
    return d;
}
int g_ws_mode = -1;                     // -1 automatic, 0 off, 1 on wherever it applies
unsigned long long* g_ws_stamps = nullptr;

}  // namespace

extern "C" int vqa_gru_ws_set_mode(int mode) {
    g_ws_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return VQA_OK;
}
extern "C" int vqa_gru_ws_set_stamps(unsigned long long* dev_words) {
    g_ws_stamps = dev_words;
    return VQA_OK;
}

// 1 when the weight-stationary recurrence applies: H = 1024, at most 512 rows (eight chains of 64), and a device of
// exactly 8 x 32 CUs on which one 160 KB / 256-thread workgroup per CU is resident.
extern "C" int vqa_gru_ws_supported(int T, int B, int H) {
    if (g_ws_mode == 0) return 0;
    if (T <= 0 || H != WS_H || B <= 0 || B > WS_CHAINS * 64) return 0;
    const WsDevice& d = ws_device();
    return (d.cus == WS_CHAINS * WS_CU && d.blocks >= 1) ? 1 : 0;
}

// device memory the call needs besides the tape: fragment-order hand-off buffers and the counters
extern "C" int64_t vqa_gru_ws_workspace_bytes(int T) {
    return T < 0 ? 0 : ((int64_t)(2 * T + 1) * WS_FRAG_T + 512) * (int64_t)sizeof(float);
}

// Whole forward recurrence in one launch; same tape contract as vqa_gru_seq_fwd (xp [T,B,3H] read only, hs [T+1,B,H]
// with hs[0] given, r,u,c,rh [T,B,H]).  `ws`: vqa_gru_ws_workspace_bytes(T) of device memory (contents irrelevant).
// After the stream has run, a non-zero word at ((unsigned*)ws)[256] reports a barrier time-out (results invalid).
extern "C" int vqa_gru_seq_fwd_ws(const float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs,
                                  float* r, float* u, float* c, float* rh, int T, int B, int H, void* ws, void* stream) {
    VQA_REQUIRE(xp && Wg_h && Wc_h && len && hs && r && u && c && rh && ws, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gru_ws_supported(T, B, H) == 1, VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE((int64_t)T * B * 3 * H * 4 < 0x7FFFFFF0ll && vqa_aligned16(ws), VQA_ERR_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    WsArgs a;
    a.sync = static_cast<unsigned*>(ws);
    if (hipMemsetAsync(a.sync, 0, 512 * sizeof(unsigned), st) != hipSuccess) return VQA_ERR_LAUNCH;
    a.hF = static_cast<float*>(ws) + 512;
    a.rhF = a.hF + (int64_t)(T + 1) * WS_FRAG_T;
    a.xp = xp; a.Wg = Wg_h; a.Wc = Wc_h; a.len = len; a.hs = hs; a.r = r; a.u = u; a.c = c; a.rh = rh;
    a.T = T; a.B = B;
    a.spin_limit = 2000000u;
    a.stamps = g_ws_stamps;
    hipLaunchKernelGGL(gru_ws_fwd_kernel, dim3(WS_CHAINS * WS_CU), dim3(WS_NT), WS_LDS, st, a);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
