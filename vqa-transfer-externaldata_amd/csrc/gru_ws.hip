// Weight-stationary persistent GRU recurrence for gfx950 (H = 1024): all T steps of tf.contrib.rnn.GRUCell under
// tf.nn.dynamic_rnn(sequence_length) (vlmap/modules.py:124-140) in ONE launch, the recurrent weights loaded ONCE.
//
// Why this form.  The step kernels (gemm_f32.hip) pay ~12 us per launch that is not matrix work (profiles/
// r3_gru_loop_variants.txt); the first persistent form (gru_persistent.hip) removed the launches but every XCD then
// streamed all 12 MB of recurrent weights from the Infinity Cache in every step (its k loops ran at 0.6-0.7 of the
// matrix rate, profiles/r2_gru_persistent_xcd.txt).  Here nothing but the state moves per step:
//
//   * eight chains, one per XCD: chain x = the workgroups that RUN on XCD x (ws_place) owns batch rows [64 x, 64 x + 64)
//     (32 x .. for 256 rows and fewer) -- rows are independent sequences, so hand-offs and flags stay inside one XCD's
//     32 CUs and one L2;
//   * CU j of a chain owns the 32 state columns [32 j, 32 j + 32): the matching column slabs of the gate weights
//     (1024 x 64 floats = 256 KB) live in the REGISTERS of its four waves (one wave per SIMD, 256 of its 512 registers
//     each: wave w holds the k quarter [256 w, 256 w + 256) as ready-made MFMA B fragments, 224 of them in AGPRs that the
//     MFMA reads directly), the candidate slab (1024 x 32 = 128 KB) lives in LDS in fragment order;
//   * per step only the left operand moves: h_{t-1} and r * h_{t-1} of the chain's rows (128 KB each per 32-row
//     half) are handed from the CUs that produce their columns to all CUs of the XCD in MFMA A-fragment order
//     ([octet of k][lane][4]: one contiguous 1 KB per wave load, one dwordx4 per lane feeds 4 or 8 MFMAs);
//   * the four waves split k; their partial tiles meet in LDS (32 KB, XOR-swizzled), every thread finishes one
//     float4 of the tile (gate math in registers: h, u and the length of its row stay in registers over all steps);
//   * a chain's 64 rows are two half-chains of 32 rows in ANTI-PHASE on the same waves: G(0,t) G(1,t) C(0,t) C(1,t).
//     What a sub-phase waits for was produced one sub-phase ago on every CU, and the tail of a sub-phase (reduction,
//     gate math, stores, arrival) runs in the issue slots of the next one's matrix stream (gru_ws_fwd2_kernel; the
//     plain order -- compute, tail, wait -- is gru_ws_fwd_kernel, used for one half-chain per XCD and as the A/B).
//   The back-propagation through time (gru_ws_bwd2_kernel) is the same frame with W^T slabs.
//
// Hand-off protocol: every hand-off address is written once per launch (the fragment buffers are indexed by t) with a
// plain store, so no stale line can sit in an L1 and the line stays in the chain's L2; every storing wave drains its
// stores, the workgroup barriers, ONE plain 4-byte store raises the workgroup's flag (a 128-byte line of 32 flags per
// half-chain); consumers load the line with one L1-bypassing request, barrier, then read.  Spins are bounded: a launch
// whose workgroups cannot all become resident (another process on the GPU) sets an error word instead of hanging.
#include <stdlib.h>

#include "vqa_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// timing-only builds (results are garbage): tail micro-steps switched off by group -- 1 LDS spill / barrier / fetch,
// 2 gate math, 4 stores, 8 arrival + peek + settle, 16 xp fetch (tools/gru_tune.py WS=1 with VQA_HOT_LIB)
#ifndef WS_DBG
#define WS_DBG 0
#endif
#ifndef WS_TOPUP_SPREAD
#define WS_TOPUP_SPREAD 1   // a gate stream's fragments 8..15: one per octet over its first eight octets instead of a burst on entry
#endif
#ifndef WS_EPI_ROWS8
#define WS_EPI_ROWS8 1
#endif
#ifndef WS_SLOTS
#define WS_SLOTS 0      // timing study: shader-clock stamp in every WS_SLOTS-th slot of step 5's four streams (build with -DWS_SLOTS=16)
#endif

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
    unsigned* sync;       // flags at [32 * (2 * chain + half) + slab], error word at [512], placement counters at [576 + xcd]
    int T, B;
    int chain_rows;       // 64: two 32-row half-chains per XCD (B > 256); 32: one (B <= 256, all eight XCDs busy)
    unsigned spin_limit;
    unsigned long long* stamps;   // optional timing study: workgroup 0 records (wait, compute, done) per sub-phase
    int handoff_sc1;      // tuning: hand-offs as write-through stores (drops the line from the XCD's L2)
    int dbg;              // timing study (results garbage): 1 tape stores dropped, 2 hand-off stores dropped, 4 xp loads dropped
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
// Arrival and wait.  A half-chain has one 128-byte line of 32 flags, one per column slab (= CU of the chain's XCD); a
// workgroup's flag holds how many of the half-chain's sub-phases it has handed off: 1 after h_0, then +1 per sub-phase.
//   arrive: every wave has drained its stores, workgroup barrier, ONE plain 4-byte store (completes at the XCD's L2).
//   wait:   wave 0 loads the 32 flags in one L1-bypassing (sc1) request, all of them must have reached the epoch.
// No atomic in the steady state: a global atomic executes at the memory side (MI355X_MICROARCH.md), and since a wave's
// vector-memory operations complete in order, a 1-2 us atomic in the queue held up every later wait for an operand
// fragment -- the sub-phase tails cost as much inside the matrix stream as outside it until the counters became flags.
__device__ __forceinline__ void ws_flag_store(unsigned* line, int slab, unsigned epoch) {
    if (threadIdx.x == 0) __hip_atomic_store(line + slab, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned ws_flag_load(const unsigned* line) {      // lanes 0..31 of a wave: one flag each
    return __hip_atomic_load(line + (threadIdx.x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ws_flags_reached(unsigned v, unsigned epoch) {
    return __builtin_amdgcn_ballot_w64(v < epoch) == 0ull;
}
// (err = word 512: this launch gave up -- zeroed by every launch; word 1023: some launch on this workspace gave up -- only
// the host clears it)
__device__ __forceinline__ void ws_spin(const unsigned* line, unsigned epoch, unsigned* err, unsigned limit) {   // wave 0
    unsigned spins = 0;
    while (!ws_flags_reached(ws_flag_load(line), epoch)) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(err + 511, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // word 1023
            break;
        }
    }
}
__device__ __forceinline__ void ws_arrive(unsigned* line, int slab, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ws_flag_store(line, slab, epoch);
}
__device__ __forceinline__ void ws_wait(const unsigned* line, unsigned epoch, unsigned* err, unsigned limit) {
    if (threadIdx.x < 64) ws_spin(line, epoch, err, limit);
    __syncthreads();
}

// Which chain and which column slab a workgroup is: chain = the XCD it RUNS on (hardware id), slab = the order in which
// the workgroups of that XCD get here (one atomic per workgroup).  All 8 x 32 workgroups are resident at once and a CU
// holds one of them (160 KB of LDS each), so every XCD hosts exactly 32: a chain's hand-offs, counters and plain
// loads / stores then stay inside one L2 by construction, whatever order the dispatcher dealt the workgroups in.
__device__ __forceinline__ void ws_place(unsigned* sync, int* place, int& chain, int& slab) {      // place: 2 words of LDS
    if (threadIdx.x == 0) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);      // HW_REG_XCC_ID[3:0]
        place[0] = xcc;
        place[1] = (int)__hip_atomic_fetch_add(sync + 576 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    chain = __builtin_amdgcn_readfirstlane(place[0]);
    slab = __builtin_amdgcn_readfirstlane(place[1]);
    __syncthreads();
}

// Gate math on the hardware's own exp2 and reciprocal (1 ulp each): a wave's vector-ALU instructions do NOT run in the
// shadow of its own MFMAs -- each costs its 5-8 issue cycles on top of the matrix stream (tools/micro/mfma_gap.hip:
// 64.0 cycles per MFMA alone, 87.3 with 8 independent v_fma per MFMA pair; LDS writes and s_nop ride for free) -- so
// the tails are priced by their VALU count: 6 instructions for a sigmoid instead of the ~30 of expf() and an IEEE
// division, 7 for a tanh.  Values within ~2e-7 of sigmoidf_stable() / tanhf() (the per-step kernels' forms).
__device__ __forceinline__ float ws_sigmoid(float x) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * __builtin_fabsf(x));      // exp(-|x|) in (0, 1]
    return (x >= 0.f ? 1.f : e) * __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ float ws_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * __builtin_fabsf(x));      // exp(-2|x|)
    return __builtin_copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), x);
}

#define WS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// The gate slabs fill the wave's 256 accumulation registers (AGPRs) and the MFMA takes its B operand from there
// directly.  Written through the intrinsic, the register allocator treats those registers as spill space and copies
// every weight back with v_accvgpr_read + s_nop in front of its MFMA: 78 cycles per MFMA instead of 64.  The compiler
// does not see an MFMA in the asm, so the read-after-MFMA wait states are spelled out where the tile is first read.
#define WS_MFMA_AGPR_B(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define WS_MFMA_VGPR_B(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define WS_MFMA_AGPR_B0(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b))   // C = 0
constexpr int WS_AG_OCT = 28;       // octets of a wave's k quarter whose fragments sit in AGPRs (8 each); the rest in VGPRs
// (all 32 would fill the 256 AGPRs to the last one; the allocator then spills to find itself a temporary)
#define WS_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

__global__ __launch_bounds__(WS_NT, 1) void gru_ws_fwd_kernel(WsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                   // candidate slab, fragment order: [(octet * 64 + lane) * 4 + i]
    float* S = smem + 32768;            // reduction scratch (8192 floats)
    constexpr int H = WS_H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int chain, j;                       // XCD, column slab
    ws_place(a.sync, reinterpret_cast<int*>(S), chain, j);
    const int B = a.B, T = a.T;
    const int row_base = chain * a.chain_rows;
    const int halves = (a.chain_rows == 64 && B - row_base > 32) ? 2 : 1;      // 32-row half-chains with any row in the batch
    unsigned* err = a.sync + 512;
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
    // (all of a thread's 32 x 4 loads in flight before the first LDS write: left as a loop of load-load-load-load-store
    // the fill is 32 dependent round trips, 30 us from cold caches)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4n tmp[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int idx = tid + (16 * half + n) * WS_NT, oct = idx >> 6, l = idx & 63;
            const float* p = a.Wc + (int64_t)(8 * oct + 4 * (l >> 5)) * H + 32 * j + (l & 31);
            tmp[n].x = p[0]; tmp[n].y = p[H]; tmp[n].z = p[2 * H]; tmp[n].w = p[3 * H];
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) *reinterpret_cast<f32x4n*>(Wl + (tid + (16 * half + n) * WS_NT) * 4) = tmp[n];
    }

    // ---- this thread's float4 of every 32 x 32 tile: row lane % 32, state columns 32 j + 8 w + 4 (lane / 32) + 0..3
#if WS_EPI_ROWS8        // (see gru_ws_fwd2_kernel: eight full cache lines per tail load / store)
    const int e_row = 8 * w + (lane >> 3), e_col = 4 * (lane & 7);
#else
    const int e_row = lane & 31, e_col = 8 * w + 4 * (lane >> 5);
#endif
    const int64_t BH = (int64_t)B * H;
    const __amdgpu_buffer_rsrc_t rs_hF = ws_rs(a.hF, (int64_t)(T + 1) * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_rhF = ws_rs(a.rhF, (int64_t)T * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_xp = ws_rs(a.xp, (int64_t)T * B * 3 * H * 4);
    const __amdgpu_buffer_rsrc_t rs_hs = ws_rs(a.hs, (T + 1) * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_r = ws_rs(a.r, T * BH * 4), rs_u = ws_rs(a.u, T * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_c = ws_rs(a.c, T * BH * 4), rs_rh = ws_rs(a.rh, T * BH * 4);
    // fragment buffer offsets (bytes): operand loads of this wave's k quarter, and this thread's hand-off store
    auto frag_ld = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 32 * w) * 1024 + lane * 16); };
#if WS_EPI_ROWS8
    auto frag_st = [&](int t, int hf) {
        return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j + (e_col >> 3)) * 1024 + (((e_col >> 2) & 1) * 32 + e_row) * 16);
    };
#else
    auto frag_st = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j + w) * 1024 + lane * 16); };
#endif

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
        if (hf < halves) ws_store(rs_hF, frag_st(0, hf), h_own[hf], a.handoff_sc1 != 0);
    }
    __syncthreads();                    // the candidate slab is in LDS
    for (int hf = 0; hf < halves; ++hf) ws_arrive(a.sync + 32 * (2 * chain + hf), j, 1u);

    f32x4n ring[WS_RING];
    auto fill = [&](__amdgpu_buffer_rsrc_t rs, unsigned base) {
#pragma unroll
        for (int m = 0; m < WS_RING; ++m) ring[m] = ws_load(rs, base + m * 1024);
    };
    unsigned long long* stamp = (a.stamps != nullptr && blockIdx.x == 0 && tid == 0) ? a.stamps : nullptr;

    // the first sub-phase's operand: G(0, 0) needs every CU's share of h_0
    ws_wait(a.sync + 32 * (2 * chain), 1u, err, a.spin_limit);
    fill(rs_hF, frag_ld(0, 0));

    for (int t = 0; t < T; ++t) {
        // ================================================================ gates: r | u = sigmoid(h W_g + xp)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf < halves) {
                unsigned* ctr = a.sync + 32 * (2 * chain + hf);
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
                    if (hf == 0) { ws_wait(a.sync + 32 * (2 * chain + 1), 1u + 2 * t, err, a.spin_limit); fill(rs_hF, frag_ld(t, 1)); }
                    else { ws_wait(a.sync + 32 * (2 * chain), 2u + 2 * t, err, a.spin_limit); fill(rs_rhF, frag_ld(t, 0)); }
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
                    rv[i] = ws_sigmoid(rv[i]);
                    uv[i] = ws_sigmoid(uv[i]);
                    rhv[i] = rv[i] * h_own[hf][i];
                }
                u_own[hf] = uv;
                ws_store(rs_rhF, frag_st(t, hf), rhv, a.handoff_sc1 != 0);
                if (ok[hf]) {
                    const unsigned o = (unsigned)(t * BH * 4) + o_std[hf];
                    ws_store(rs_r, o, rv, false);
                    ws_store(rs_u, o, uv, false);
                    ws_store(rs_rh, o, rhv, false);
                }
                ws_arrive(ctr, j, 2u + 2 * t);          // (its barrier also frees the scratch for the next sub-phase)
                if (halves == 1) { ws_wait(ctr, 2u + 2 * t, err, a.spin_limit); fill(rs_rhF, frag_ld(t, 0)); }
                if (stamp) { stamp[2] = wall_clock64(); stamp += 3; }
            }
        }
        // ================================================================ candidate: c = tanh((r h) W_c + xp), h_t
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf < halves) {
                unsigned* ctr = a.sync + 32 * (2 * chain + hf);
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
                    if (hf == 0) { ws_wait(a.sync + 32 * (2 * chain + 1), 2u + 2 * t, err, a.spin_limit); fill(rs_rhF, frag_ld(t, 1)); }
                    else if (t + 1 < T) { ws_wait(a.sync + 32 * (2 * chain), 3u + 2 * t, err, a.spin_limit); fill(rs_hF, frag_ld(t + 1, 0)); }
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
                    cv[i] = ws_tanh(cv[i]);
                    hn[i] = (t < len_own[hf]) ? (u_own[hf][i] * h_own[hf][i] + (1.f - u_own[hf][i]) * cv[i]) : h_own[hf][i];
                }
                h_own[hf] = hn;
                ws_store(rs_hF, frag_st(t + 1, hf), hn, a.handoff_sc1 != 0);
                if (ok[hf]) {
                    ws_store(rs_c, (unsigned)(t * BH * 4) + o_std[hf], cv, false);
                    ws_store(rs_hs, (unsigned)((t + 1) * BH * 4) + o_std[hf], hn, false);
                }
                ws_arrive(ctr, j, 3u + 2 * t);
                if (halves == 1 && t + 1 < T) { ws_wait(ctr, 3u + 2 * t, err, a.spin_limit); fill(rs_hF, frag_ld(t + 1, 0)); }
                if (stamp) { stamp[2] = wall_clock64(); stamp += 3; }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The two-half form with the epilogues INSIDE the matrix stream.  In the kernel above a sub-phase is
// compute -> (poll, reduce, gate math, stores, drain, barrier, atomic) and the second part -- 1.4-2.3 us of every
// 5.4-10.9 us sub-phase -- leaves the matrix cores idle.  A 32x32x2 MFMA occupies its SIMD's matrix core for 64 cycles
// and the wave's issue port for 4, so here the tail of sub-phase X runs in the issue gaps of sub-phase X+1 (the other
// half-chain), a few instructions per octet of k:
//     octet 1  partial tiles of X -> LDS          octet 3  barrier, partial tiles back (one float4 per thread)
//     octet 5  sums, gate math, stores            octet 9  drain the hand-off store, barrier, X's arrival
//     octet 12 peek at the counter X+2 waits for  octet 22 (spin if it is not there yet,) barrier
//     octets 24..31  the ring slots that X+1 has used up are refilled with the first fragments of X+2
// so the operand ring never drains and a barrier never waits on memory (plain s_barrier: __syncthreads() would add
// s_waitcnt vmcnt(0) and stall the stream for an L2 round trip each time).  X's arrival moves ~1-2 us into X+1; its
// consumers are X+2 on every CU, which looks for it at octet 22 of X+1: no wait cycle (every CU passes octet 9 of
// X+1 without waiting for anything later than X-1).
__device__ __forceinline__ unsigned ws_uni(unsigned x) { return __builtin_amdgcn_readfirstlane(x); }      // a scalar offset
__device__ __forceinline__ void ws_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void ws_bar() { asm volatile("s_barrier" ::: "memory"); }

// loads / stores with the uniform part of the address in the scalar offset (one VGPR of per-lane offset serves a whole
// stream; left in the vector offset, the loop-invariant sums are hoisted into dozens of VGPRs and the weights spill)
__device__ __forceinline__ f32x4n ws_load2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    f32x4n o;
    o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
    return o;
}
__device__ __forceinline__ void ws_store2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, f32x4n x, bool sc1) {
    u32x4 v;
    v.x = __float_as_uint(x.x); v.y = __float_as_uint(x.y); v.z = __float_as_uint(x.z); v.w = __float_as_uint(x.w);
    if (sc1) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, WS_SC1);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
}

__global__ __launch_bounds__(WS_NT, 1) void gru_ws_fwd2_kernel(WsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;
    float* S = smem + 32768;
    constexpr int H = WS_H;
    const unsigned long long t_entry = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int chain, j;
    ws_place(a.sync, reinterpret_cast<int*>(S), chain, j);
    const int B = a.B, T = a.T;
    const int row_base = chain * 64;
    unsigned* err = a.sync + 512;
    if (row_base >= B) return;
    unsigned* ctr0 = a.sync + 32 * (2 * chain);     // the two half-chains' flag lines
    unsigned* ctr1 = ctr0 + 32;

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
    // (all of a thread's 32 x 4 loads in flight before the first LDS write: left as a loop of load-load-load-load-store
    // the fill is 32 dependent round trips, 30 us from cold caches)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4n tmp[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int idx = tid + (16 * half + n) * WS_NT, oct = idx >> 6, l = idx & 63;
            const float* p = a.Wc + (int64_t)(8 * oct + 4 * (l >> 5)) * H + 32 * j + (l & 31);
            tmp[n].x = p[0]; tmp[n].y = p[H]; tmp[n].z = p[2 * H]; tmp[n].w = p[3 * H];
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) *reinterpret_cast<f32x4n*>(Wl + (tid + (16 * half + n) * WS_NT) * 4) = tmp[n];
    }

    // A thread's float4 of every 32 x 32 tile.  WS_EPI_ROWS8: row 8 w + lane / 8, columns 4 (lane % 8): a wave's tail loads and
    // stores then cover 8 rows x 128 bytes = eight FULL cache lines per instruction (tape stores, xp loads) and the hand-off
    // store eight full 128-byte pieces of the fragment buffer, instead of 32 rows x 32 bytes of 32 different lines.
#if WS_EPI_ROWS8
    const int e_row = 8 * w + (lane >> 3), e_col = 4 * (lane & 7);
#else
    const int e_row = lane & 31, e_col = 8 * w + 4 * (lane >> 5);
#endif
    const int64_t BH = (int64_t)B * H;
    const __amdgpu_buffer_rsrc_t rs_hF = ws_rs(a.hF, (int64_t)(T + 1) * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_rhF = ws_rs(a.rhF, (int64_t)T * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_xp = ws_rs(a.xp, (int64_t)T * B * 3 * H * 4);
    const __amdgpu_buffer_rsrc_t rs_hs = ws_rs(a.hs, (T + 1) * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_r = ws_rs(a.r, T * BH * 4), rs_u = ws_rs(a.u, T * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_c = ws_rs(a.c, T * BH * 4), rs_rh = ws_rs(a.rh, T * BH * 4);
    // fragment buffers: uniform (scalar) part of the offsets; the per-lane part is lane * 16
    auto frag_ld = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 32 * w) * 1024); };
#if WS_EPI_ROWS8
    // this thread's columns 32 j + e_col .. + 3 = octet 4 j + e_col / 8, k half (e_col / 4) % 2 of the fragment order
    auto frag_st = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j) * 1024); };
    const unsigned frag_v = (unsigned)((e_col >> 3) * 1024 + ((((e_col >> 2) & 1) * 32 + e_row) * 16));
#else
    auto frag_st = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j + w) * 1024); };
    const unsigned frag_v = (unsigned)lane * 16u;
#endif
    const unsigned lane16 = (unsigned)lane * 16u;

    f32x4n h_own[2], u_own[2];
    int len_own[2];
    bool ok[2];
    unsigned o_std[2], o_xp[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int grow = row_base + 32 * hf + e_row;
        ok[hf] = grow < B;
        o_std[hf] = (unsigned)(((int64_t)grow * H + 32 * j + e_col) * 4);
        o_xp[hf] = (unsigned)(((int64_t)grow * 3 * H + 32 * j + e_col) * 4);
        if (!ok[hf]) { o_std[hf] = 0xFFFFFFF0u; o_xp[hf] = 0xFFFFFFF0u; }       // past the end: loads give 0, stores vanish
        h_own[hf] = ws_load(rs_hs, o_std[hf]);
        u_own[hf] = (f32x4n)(0.f);
        len_own[hf] = ok[hf] ? a.len[grow] : 0;
        ws_store2(rs_hF, frag_v, frag_st(0, hf), h_own[hf], a.handoff_sc1 != 0);
    }
    __syncthreads();
    ws_arrive(ctr0, j, 1u);
    ws_arrive(ctr1, j, 1u);
    const unsigned lane16_st = (a.dbg & 2) ? 0xFFFFFFF0u : frag_v;
    if (a.dbg & 1) { o_std[0] = 0xFFFFFFF0u; o_std[1] = 0xFFFFFFF0u; }
    if (a.dbg & 4) { o_xp[0] = 0xFFFFFFF0u; o_xp[1] = 0xFFFFFFF0u; }

    const unsigned xp_step = (unsigned)((int64_t)B * 3 * H * 4), bh_step = (unsigned)(BH * 4);
    auto xp_soff = [&](int t, int part) { return (unsigned)t * xp_step + (unsigned)(part * H * 4); };
    // LDS addresses of the partial-tile spills: accumulator register q of lane l is row rq + 4 (l >> 5), rq = (q & 3) +
    // 8 (q >> 2), column l & 31.  The XOR swizzles below split into a per-lane part with four variants and a constant
    // part, so four base registers per tile shape serve all sixteen q (sixteen independent addresses would be hoisted
    // out of the time loop into as many VGPRs).
    //   gate tiles  S[w][row][64]: slot (col >> 2 [+ 8 for u]) ^ (row & 15);  row & 15 = (q & 3) | hl << 2 | (q >> 2 & 1) << 3
    //   cand tiles  S[w][row][32]: slot (col >> 2) ^ (row >> 1 & 7);           row >> 1 & 7 = (q >> 1 & 1) | hl << 1 | (q >> 2 & 1) << 2
    int gbase[4], cbase[4];
    {
        const int hl = lane >> 5, cs = (lane & 31) >> 2, c3 = lane & 3;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            gbase[v] = w * 2048 + (4 * hl) * 64 + (((cs ^ (hl << 2)) ^ v) << 2) + c3;
            cbase[v] = w * 1024 + (4 * hl) * 32 + (((cs ^ (hl << 1)) ^ ((v & 1) | ((v >> 1) << 2))) << 2) + c3;
        }
    }
    const int gfetch = e_row * 64 + (((e_col >> 2) ^ (e_row & 15)) << 2);           // u: the same slot ^ 8
    const int cfetch = e_row * 32 + (((e_col >> 2) ^ ((e_row >> 1) & 7)) << 2);

    // ---- the tail of a sub-phase, cut into micro-steps k = 0, 1, ... of a few instructions each; micro-step k runs in
    // slot 4 + k of the NEXT sub-phase's matrix stream (a slot = the issue gap behind one MFMA of a candidate stream
    // or behind a pair of a gate stream: 64 / 128 cycles in which the wave's other instructions are free to issue).
    // An MFMA stalls at issue until the matrix core is free and everything behind it in program order waits with it,
    // so work that is not cut up and placed between the MFMAs does not overlap with them at all.
    //    (the xp values the tail adds are fetched in slot 100 of the tail's OWN stream: from the Infinity Cache they take
    //     longer than the few slots of a candidate stream in front of the sums)
    //    1..8     partial tile registers -> LDS (swizzled scratch), two accumulator registers per step
    //    9        LDS drained, workgroup barrier
    //    10..13   the four k quarters' partial float4s back from LDS
    //    15, 16   sums (+ xp)
    //    17..31   odd steps: one sigmoid / tanh each
    //    33..39   rest of the gate math; the stores one per odd step, the hand-off store LAST
    //    54       arrival (by then >= 2 younger vector-memory operations follow the hand-off store: vmcnt(2) covers it)
    //    76       wave 0 loads the flags the sub-phase after this one waits for (the arrivals of slot 58 have landed)
    //    92       settle them (spin only if some CU lags), barrier; slots 99.. fetch that sub-phase's first fragments
    f32x4n va, vb, va2;
    f32x4n xr[2], xu[2], xc[2];       // xp values of the running step (see xp_fetch)
    f32x4n p0[4], p1[4];
    // (no branches in the micro-steps: a row outside the batch, the tail that does not exist in front of the first
    // step and the operand ring's look-ahead behind the last one all address past the end of their buffers, where a
    // buffer store is dropped and a buffer load returns zeros)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto g_tail = [&](int k, int t, int hf, const f32x16& R, const f32x16& U, unsigned* ctr) {
        if (!(WS_DBG & 1) && k >= 1 && k <= 8) {
#pragma unroll
            for (int q = 2 * (k - 1); q < 2 * k; ++q) {
                const int rq = (q & 3) + 8 * (q >> 2), hi = (q >> 2) & 1;
                S[gbase[q & 3] + rq * 64 + hi * 32] = R[q];
                S[gbase[q & 3] + rq * 64 + (hi ^ 1) * 32] = U[q];
            }
        }
        if (!(WS_DBG & 1) && k == 9) ws_lds_barrier();
        if (!(WS_DBG & 1) && k >= 10 && k <= 13) {
            const int g = k - 10;
            p0[g] = *reinterpret_cast<const f32x4n*>(S + g * 2048 + gfetch);
            p1[g] = *reinterpret_cast<const f32x4n*>(S + g * 2048 + (gfetch ^ 32));
        }
        if (!(WS_DBG & 2) && k == 15) va = xr[hf] + p0[0] + p0[1] + p0[2] + p0[3];
        if (!(WS_DBG & 2) && k == 16) vb = xu[hf] + p1[0] + p1[1] + p1[2] + p1[3];
        if (!(WS_DBG & 2) && k >= 17 && k <= 31 && (k & 1)) {
            const int i = (k - 17) >> 1;
            if (i < 4) va[i] = ws_sigmoid(va[i]);
            else vb[i - 4] = ws_sigmoid(vb[i - 4]);
        }
        // the four stores one at a time, two slots apart: issued together, 16 KB per CU and 512 KB per XCD arrive at the
        // L2 in one burst from all CUs in lockstep, the store queue fills and the wave (and its MFMAs) wait at the issue
        if (!(WS_DBG & 4) && k == 33) { va2 = va * h_own[hf]; u_own[hf] = vb; ws_store2(rs_r, o_std[hf], ws_uni((unsigned)t * bh_step), va, false); }
        if (!(WS_DBG & 4) && k == 35) ws_store2(rs_u, o_std[hf], ws_uni((unsigned)t * bh_step), vb, false);
        if (!(WS_DBG & 4) && k == 37) ws_store2(rs_rh, o_std[hf], ws_uni((unsigned)t * bh_step), va2, false);
        if (!(WS_DBG & 4) && k == 39) ws_store2(rs_rhF, lane16_st, ws_uni(frag_st(t, hf)), va2, a.handoff_sc1 != 0);      // hand-off LAST
        if (!(WS_DBG & 8) && k == 54) {
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            ws_flag_store(ctr, j, 2u + 2 * t);
        }
    };
    auto c_tail = [&](int k, int t, int hf, const f32x16& A0, unsigned* ctr, bool live) {
        if (!(WS_DBG & 1) && k >= 1 && k <= 8) {
#pragma unroll
            for (int q = 2 * (k - 1); q < 2 * k; ++q) {
                const int rq = (q & 3) + 8 * (q >> 2);
                S[cbase[((q >> 1) & 1) | (((q >> 2) & 1) << 1)] + rq * 32] = A0[q];
            }
        }
        if (!(WS_DBG & 1) && k == 9) ws_lds_barrier();
        if (!(WS_DBG & 1) && k >= 10 && k <= 13) {
            const int g = k - 10;
            p0[g] = *reinterpret_cast<const f32x4n*>(S + g * 1024 + cfetch);
        }
        if (!(WS_DBG & 2) && k == 15) va = xc[hf] + p0[0] + p0[1] + p0[2] + p0[3];
        if (!(WS_DBG & 2) && k >= 17 && k <= 23 && (k & 1)) {
            const int i = (k - 17) >> 1;
            va[i] = ws_tanh(va[i]);
        }
        if (!(WS_DBG & 4) && k == 33) {
            const bool run = live && t < len_own[hf];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                va2[i] = run ? (u_own[hf][i] * h_own[hf][i] + (1.f - u_own[hf][i]) * va[i]) : h_own[hf][i];
            h_own[hf] = va2;
            // (the tail that does not exist in front of step 0: vector offset past the end, the scalar one is then irrelevant)
            ws_store2(rs_c, live ? o_std[hf] : OOB, ws_uni((unsigned)t * bh_step), va, false);
        }
        if (!(WS_DBG & 4) && k == 35) ws_store2(rs_hs, live ? o_std[hf] : OOB, ws_uni((unsigned)(t + 1) * bh_step), va2, false);
        if (!(WS_DBG & 4) && k == 39) ws_store2(rs_hF, live ? lane16_st : OOB, ws_uni(frag_st(t + 1, hf)), va2, a.handoff_sc1 != 0);
        if (!(WS_DBG & 8) && k == 54) {
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            if (live) ws_flag_store(ctr, j, 3u + 2 * t);
        }
    };
    // xp (from the Infinity Cache / HBM) is slow and a wave's vector-memory operations return IN ORDER: a fragment load
    // issued behind an xp load cannot be seen complete before it.  So all xp a step needs is fetched at the head of the
    // step's two GATE streams, where 8 octets of look-ahead are 4096 cycles (2048 in a candidate stream: measured, one xp
    // load there cost 0.5 us of matrix time).
    auto xp_fetch = [&](int k, int t, int hf) {
        if (!(WS_DBG & 16) && k == 0) {
            xr[hf] = ws_load2(rs_xp, o_xp[hf], xp_soff(t, 0));
            xu[hf] = ws_load2(rs_xp, o_xp[hf], xp_soff(t, 1));
            xc[hf] = ws_load2(rs_xp, o_xp[hf], xp_soff(t, 2));
        }
    };
    unsigned peek = 0;
    auto way_in = [&](int k, const unsigned* line, unsigned epoch) {      // towards the sub-phase after the running one
        if (!(WS_DBG & 8) && k == 76 && w == 0) peek = ws_flag_load(line);
        if (!(WS_DBG & 8) && k == 92) {
            if (w == 0 && !ws_flags_reached(peek, epoch)) ws_spin(line, epoch, err, a.spin_limit);
            ws_bar();
        }
    };

    f32x4n ring[16];                    // the candidate streams use slots 0..7
    ws_wait(ctr0, 1u, err, a.spin_limit);
#pragma unroll
    for (int m = 0; m < 8; ++m) ring[m] = ws_load2(rs_hF, lane16, frag_ld(0, 0) + m * 1024);

    // one sub-phase's matrix stream: 32 octets of this wave's k quarter, HOOK(slot - 4) in every slot
    // (cur: this sub-phase's fragments; nxt: the next one's, fetched into the ring slots as they fall free)
#define WS_GATE_STREAM(accR, accU, rs_cur, base_cur, rs_nxt, base_nxt, voff_nxt, HOOK)                         \
    /* 16 fragments ahead inside a gate stream (xp loads -- Infinity Cache or HBM, returned in order with the  */ \
    /* fragments -- are issued at its head: 16 octets of 512 cycles are their cover); every stream hands the   */ \
    /* next one 8 fragments, so a gate stream fetches its fragments 8..15 on entry                              */ \
    if (!WS_TOPUP_SPREAD) {                                                                                    \
        _Pragma("unroll") for (int f = 8; f < 16; ++f) ring[f] = ws_load2(rs_cur, lane16, (base_cur) + f * 1024); \
    }                                                                                                          \
    _Pragma("unroll") for (int m = 0; m < 32; ++m) {                                                           \
        const f32x4n av = ring[m % 16];                                                                        \
        if (WS_TOPUP_SPREAD && m < 8) ring[8 + m] = ws_load2(rs_cur, lane16, (base_cur) + (8 + m) * 1024);     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
            if (m == 0 && i == 0) {                                                                            \
                WS_MFMA_AGPR_B0(accR, av[i], wg[0][m][i]);                                                     \
                WS_MFMA_AGPR_B0(accU, av[i], wg[1][m][i]);                                                     \
            } else if (m < WS_AG_OCT) {                                                                        \
                WS_MFMA_AGPR_B(accR, av[i], wg[0][m][i]);                                                      \
                WS_MFMA_AGPR_B(accU, av[i], wg[1][m][i]);                                                      \
            } else {                                                                                           \
                WS_MFMA_VGPR_B(accR, av[i], wg[0][m][i]);                                                      \
                WS_MFMA_VGPR_B(accU, av[i], wg[1][m][i]);                                                      \
            }                                                                                                  \
            if (i == 3) {                                                                                      \
                if (m + 16 < 32) ring[m % 16] = ws_load2(rs_cur, lane16, (base_cur) + (m + 16) * 1024);        \
                else if (m >= 24) ring[m - 24] = ws_load2(rs_nxt, (voff_nxt), (base_nxt) + (m - 24) * 1024);   \
            }                                                                                                  \
            HOOK(4 * m + i - 4);                                                                               \
            if (WS_SLOTS && ((4 * m + i) % WS_SLOTS) == WS_SLOTS - 1 && slots != nullptr && t == 5) slots[SLOT_BASE + 4 * m + i] = clock64();              \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }                                                                                                      \
    }
#define WS_CAND_STREAM(acc0, acc1, rs_cur, base_cur, rs_nxt, base_nxt, voff_nxt, HOOK)                          \
    {                                                                                                          \
        f32x4n bq[2];                                                                                          \
        bq[0] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w) * 64 + lane) * 4);                             \
        bq[1] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w + 1) * 64 + lane) * 4);                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int m = 0; m < 32; ++m) {                                                       \
            const f32x4n av = ring[m % 8];                                                                     \
            const f32x4n bv = bq[m & 1];                                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
                if (m & 1) acc1 = WS_MFMA(av[i], bv[i], acc1);                                                 \
                else acc0 = WS_MFMA(av[i], bv[i], acc0);                                                       \
                if (i == 1 && m + 2 < 32)                                                                      \
                    bq[m & 1] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w + m + 2) * 64 + lane) * 4);     \
                if (i == 3) {                                                                                  \
                    if (m + 8 < 32) ring[m % 8] = ws_load2(rs_cur, lane16, (base_cur) + (m + 8) * 1024);       \
                    else ring[m % 8] = ws_load2(rs_nxt, (voff_nxt), (base_nxt) + (m + 8 - 32) * 1024);         \
                }                                                                                              \
                HOOK(4 * m + i - 4);                                                                           \
                if (WS_SLOTS && ((4 * m + i) % WS_SLOTS) == WS_SLOTS - 1 && slots != nullptr && t == 5) slots[SLOT_BASE + 4 * m + i] = clock64();          \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                                  \
        }                                                                                                      \
    }

    f32x16 gR0, gU0, gR1, gU1, cA0, cA1;
#pragma unroll
    for (int q = 0; q < 16; ++q) cA1[q] = 0.f;
    unsigned long long* stamp = (a.stamps != nullptr && blockIdx.x == 0 && tid == 0) ? a.stamps : nullptr;
    if (stamp) { stamp[2046] = t_entry; stamp[2047] = wall_clock64(); }      // prologue: entry -> first stream
    long long* slots = (WS_SLOTS && stamp != nullptr) ? reinterpret_cast<long long*>(a.stamps) + 1024 : nullptr;      // 4 x 128 slot stamps of step 5

    for (int t = 0; t < T; ++t) {
        const unsigned n = (unsigned)(2 * t);
        // ---------------- G(0,t); in its gaps: the tail of C(1,t-1), then the way into G(1,t)
        if (stamp) { *stamp++ = wall_clock64(); *stamp++ = clock64(); }
        {
            const bool tail = t > 0;
            auto hook = [&](int k) {
                c_tail(k, t - 1, 1, cA1, ctr1, tail);
                way_in(k, ctr1, 1u + n);
                xp_fetch(k, t, 0);
            };
#define SLOT_BASE 0
            WS_GATE_STREAM(gR0, gU0, rs_hF, frag_ld(t, 0), rs_hF, frag_ld(t, 1), lane16, hook)
        }
        // ---------------- G(1,t); tail of G(0,t); way into C(0,t)
        if (stamp) { *stamp++ = wall_clock64(); *stamp++ = clock64(); }
        {
            auto hook = [&](int k) {
                g_tail(k, t, 0, gR0, gU0, ctr0);
                way_in(k, ctr0, 2u + n);
                xp_fetch(k, t, 1);
            };
#undef SLOT_BASE
#define SLOT_BASE 128
            WS_GATE_STREAM(gR1, gU1, rs_hF, frag_ld(t, 1), rs_rhF, frag_ld(t, 0), lane16, hook)
        }
        // ---------------- C(0,t); tail of G(1,t); way into C(1,t)
        if (stamp) { *stamp++ = wall_clock64(); *stamp++ = clock64(); }
        {
            f32x16 cB;
#pragma unroll
            for (int q = 0; q < 16; ++q) { cA0[q] = 0.f; cB[q] = 0.f; }
            auto hook = [&](int k) {
                g_tail(k, t, 1, gR1, gU1, ctr1);
                way_in(k, ctr1, 2u + n);
            };
#undef SLOT_BASE
#define SLOT_BASE 256
            WS_CAND_STREAM(cA0, cB, rs_rhF, frag_ld(t, 0), rs_rhF, frag_ld(t, 1), lane16, hook)
            cA0 += cB;
        }
        // ---------------- C(1,t); tail of C(0,t); way into G(0,t+1)
        if (stamp) { *stamp++ = wall_clock64(); *stamp++ = clock64(); }
        {
            f32x16 cB;
#pragma unroll
            for (int q = 0; q < 16; ++q) { cA1[q] = 0.f; cB[q] = 0.f; }
            const bool more = t + 1 < T;
            auto hook = [&](int k) {
                c_tail(k, t, 0, cA0, ctr0, true);
                way_in(k, ctr0, more ? 3u + n : 0u);      // (behind the last step: nothing to wait for)
            };
#undef SLOT_BASE
#define SLOT_BASE 384
            WS_CAND_STREAM(cA1, cB, rs_rhF, frag_ld(t, 1), rs_hF, more ? frag_ld(t + 1, 0) : 0u, more ? lane16 : OOB, hook)
            cA1 += cB;
        }
    }
    // ---------------- the tail of C(1,T-1), on its own
    if (stamp) { *stamp++ = wall_clock64(); *stamp++ = clock64(); }
#pragma unroll
    for (int k = 1; k <= 39; ++k) c_tail(k, T - 1, 1, cA1, ctr1, true);
    ws_arrive(ctr1, j, 1u + 2 * T);
#undef SLOT_BASE
#undef WS_GATE_STREAM
#undef WS_CAND_STREAM
}


// ---------------------------------------------------------------------------------------------------------------
// Back-propagation through time in the same frame (two 32-row half-chains per XCD, tails inside the streams).
// Per step t (vlmap/modules.py:124-140 differentiated; the formulas of rowops.hip gru_bwd_a / gru_bwd_b):
//     g      = dL/dh_t                                     (this thread's float4 of its row, carried in registers)
//     dc_pre = g (1 - u)(1 - c^2),  du_pre = g (h_prev - c) u (1 - u),  acc = g u          [t < len; else 0, 0, g]
//   A(hf,t)  drh = dc_pre W_c^T            K = 1024   slab W_c[32 j .. +32, :] in LDS        (as the candidate stream)
//            dr_pre = drh h_prev r (1 - r),  acc += drh r
//   B(hf,t)  dL/dh_{t-1} = (dr_pre | du_pre) W_g^T + acc   K = 2048   slab W_g[32 j .. +32, :] in registers
//            (64 octets per wave: waves 0, 1 read dr_pre's fragments, waves 2, 3 du_pre's), then the first half of
//            step t - 1 in its tail.  B(hf,0) does not exist (h_0 is not a parameter).
// dxp[t] = (dr_pre | du_pre | dc_pre) is the only output.  CU j owns the 32 state columns [32 j, 32 j + 32) of every
// product's RESULT, i.e. rows of W_c and W_g: the slabs are read transposed, once.
struct WsBwdArgs {
    const float* dh_T;    // [B,H]   dL/d(final state)
    const float* d_outs;  // [T,B,H] or null: dL/d(output of step t), added to g before step t is differentiated
    const float* Wg;      // [H,2H]
    const float* Wc;      // [H,H]
    const int32_t* len;
    const float *hs, *r, *u, *c;
    float* dxp;           // [T,B,3H]
    float *dcF, *drF, *duF;       // [T] fragment-order hand-offs
    unsigned* sync;
    int T, B;
    unsigned spin_limit;
    unsigned long long* stamps;
};

#define WS_MFMA_VGPR_B0(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
constexpr int WS_BW_AG_OCT = 56;    // of a wave's 64 octets of W_g^T: fragments in AGPRs (4 each); the rest in VGPRs

__global__ __launch_bounds__(WS_NT, 1) void gru_ws_bwd2_kernel(WsBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                   // W_c^T slab, fragment order
    float* S = smem + 32768;
    constexpr int H = WS_H;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int chain, j;
    ws_place(a.sync, reinterpret_cast<int*>(S), chain, j);
    const int B = a.B, T = a.T;
    const int row_base = chain * 64;
    unsigned* err = a.sync + 512;
    if (row_base >= B) return;
    unsigned* ctr0 = a.sync + 32 * (2 * chain);
    unsigned* ctr1 = ctr0 + 32;

    // B fragment of octet m, k pair i: lane l supplies W^T[k = 8 m + 4 (l >> 5) + i][column l & 31] = W[32 j + (l & 31)][k]
    float wq[64][4];
    {
        const float* p = a.Wg + (int64_t)(32 * j + (lane & 31)) * (2 * H) + 512 * w + 4 * (lane >> 5);
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const f32x4n v = *reinterpret_cast<const f32x4n*>(p + 8 * m);
            wq[m][0] = v.x; wq[m][1] = v.y; wq[m][2] = v.z; wq[m][3] = v.w;
        }
    }
    {
        f32x4n tmp[32];                 // all 32 loads in flight before the first LDS write
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int idx = tid + n * WS_NT, oct = idx >> 6, l = idx & 63;
            tmp[n] = *reinterpret_cast<const f32x4n*>(a.Wc + (int64_t)(32 * j + (l & 31)) * H + 8 * oct + 4 * (l >> 5));
        }
#pragma unroll
        for (int n = 0; n < 32; ++n) *reinterpret_cast<f32x4n*>(Wl + (tid + n * WS_NT) * 4) = tmp[n];
    }

#if WS_EPI_ROWS8        // (see gru_ws_fwd2_kernel: eight full cache lines per tail load / store)
    const int e_row = 8 * w + (lane >> 3), e_col = 4 * (lane & 7);
#else
    const int e_row = lane & 31, e_col = 8 * w + 4 * (lane >> 5);
#endif
    const int64_t BH = (int64_t)B * H;
    const __amdgpu_buffer_rsrc_t rs_dcF = ws_rs(a.dcF, (int64_t)T * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_drF = ws_rs(a.drF, (int64_t)T * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_duF = ws_rs(a.duF, (int64_t)T * WS_FRAG_T * 4);
    const __amdgpu_buffer_rsrc_t rs_bF = ws_rs(w < 2 ? a.drF : a.duF, (int64_t)T * WS_FRAG_T * 4);      // this wave's B-stream operand
    const __amdgpu_buffer_rsrc_t rs_dxp = ws_rs(a.dxp, (int64_t)T * B * 3 * H * 4);
    const __amdgpu_buffer_rsrc_t rs_hs = ws_rs(a.hs, (T + 1) * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_r = ws_rs(a.r, T * BH * 4), rs_u = ws_rs(a.u, T * BH * 4), rs_c = ws_rs(a.c, T * BH * 4);
    const __amdgpu_buffer_rsrc_t rs_dh = ws_rs(a.dh_T, BH * 4);
    const __amdgpu_buffer_rsrc_t rs_do = ws_rs(a.d_outs != nullptr ? a.d_outs : a.dh_T, a.d_outs != nullptr ? T * BH * 4 : 0);
    // fragment buffers: scalar parts of the offsets (the per-lane part is lane * 16)
    auto frag_a = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 32 * w) * 1024); };          // A stream: 32 octets
    auto frag_b = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 64 * (w & 1)) * 1024); };    // B stream: 64 octets
#if WS_EPI_ROWS8
    auto frag_st = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j) * 1024); };
    const unsigned frag_v = (unsigned)((e_col >> 3) * 1024 + ((((e_col >> 2) & 1) * 32 + e_row) * 16));
#else
    auto frag_st = [&](int t, int hf) { return (unsigned)((((t * 8 + chain) * 2 + hf) * 128 + 4 * j + w) * 1024); };
    const unsigned frag_v = (unsigned)lane * 16u;
#endif
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned bh_step = (unsigned)(BH * 4), xp_step = (unsigned)((int64_t)B * 3 * H * 4);

    f32x4n tr[2], tu[2], tc[2], th[2], td[2];      // tape values a half-chain's next tails need (see tape_fetch)
    f32x4n acc_own[2], hp_own[2];       // running dL/dh_{t-1} (partial), h_{t-1} of the step being differentiated
    int len_own[2];
    unsigned o_std[2], o_xp[2];
    int gbase_unused = 0; (void)gbase_unused;
    int cbase[4];
    {
        const int hl = lane >> 5, cs = (lane & 31) >> 2, c3 = lane & 3;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            cbase[v] = w * 1024 + (4 * hl) * 32 + (((cs ^ (hl << 1)) ^ ((v & 1) | ((v >> 1) << 2))) << 2) + c3;
    }
    const int cfetch = e_row * 32 + (((e_col >> 2) ^ ((e_row >> 1) & 7)) << 2);

    // first half of a step from g (= dL/dh_t of this thread's float4): dc_pre, du_pre out, acc_own / hp_own updated
    auto half_step = [&](int t, int hf, f32x4n g, f32x4n uv, f32x4n cv, f32x4n hp, f32x4n& dc, f32x4n& du) {
        const bool run = t < len_own[hf];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dc[i] = run ? g[i] * (1.f - uv[i]) * (1.f - cv[i] * cv[i]) : 0.f;
            du[i] = run ? g[i] * (hp[i] - cv[i]) * uv[i] * (1.f - uv[i]) : 0.f;
            acc_own[hf][i] = run ? g[i] * uv[i] : g[i];
        }
        hp_own[hf] = hp;
    };

    // ---- step T-1's first half from dL/d(final state)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int grow = row_base + 32 * hf + e_row;
        const bool ok = grow < B;
        o_std[hf] = ok ? (unsigned)(((int64_t)grow * H + 32 * j + e_col) * 4) : OOB;
        o_xp[hf] = ok ? (unsigned)(((int64_t)grow * 3 * H + 32 * j + e_col) * 4) : OOB;
        len_own[hf] = ok ? a.len[grow] : 0;
        const int t = T - 1;
        f32x4n g = ws_load2(rs_dh, o_std[hf], 0u);
        if (a.d_outs != nullptr) g += ws_load2(rs_do, o_std[hf], (unsigned)t * bh_step);
        const f32x4n uv = ws_load2(rs_u, o_std[hf], (unsigned)t * bh_step), cv = ws_load2(rs_c, o_std[hf], (unsigned)t * bh_step);
        const f32x4n hp = ws_load2(rs_hs, o_std[hf], (unsigned)t * bh_step);
        f32x4n dc, du;
        half_step(t, hf, g, uv, cv, hp, dc, du);
        ws_store2(rs_dxp, o_xp[hf], (unsigned)t * xp_step + 2 * H * 4, dc, false);
        ws_store2(rs_dxp, o_xp[hf], (unsigned)t * xp_step + H * 4, du, false);
        ws_store2(rs_duF, frag_v, frag_st(t, hf), du, false);
        ws_store2(rs_dcF, frag_v, frag_st(t, hf), dc, false);
        tr[hf] = ws_load2(rs_r, o_std[hf], (unsigned)t * bh_step);
    }
    __syncthreads();                    // the W_c^T slab is in LDS
    ws_arrive(ctr0, j, 1u);
    ws_arrive(ctr1, j, 1u);

    // ---- tails (micro-step schedule of the forward kernel)
    f32x4n va, vb;                      // tail temporaries
    f32x4n p0[4];
    // The tape comes from HBM (it was written a millisecond ago) and a wave's loads return in order: a tape load holds back
    // every fragment load issued behind it.  All tape values a half-chain needs next -- u, c, h_prev (d_outs) of step t-1
    // for the tail of B(hf,t), r of step t-1 for the tail of A(hf,t-1) -- are fetched at the head of the B(hf,t) stream,
    // where the operand ring is 16 deep: 16 octets of 256 cycles are the cover a 2 us load needs (the A streams keep 8).
    auto tape_fetch = [&](int k, int t, int hf) {
        if (k == -4) {
            const unsigned so = ws_uni((unsigned)(t - 1) * bh_step);
            tu[hf] = ws_load2(rs_u, o_std[hf], so);
            tc[hf] = ws_load2(rs_c, o_std[hf], so);
            th[hf] = ws_load2(rs_hs, o_std[hf], so);
            tr[hf] = ws_load2(rs_r, o_std[hf], so);
            if (a.d_outs != nullptr) td[hf] = ws_load2(rs_do, o_std[hf], so);
        }
    };
    auto spill_fetch = [&](int k, const f32x16& P) {
        if (k >= 1 && k <= 8) {
#pragma unroll
            for (int q = 2 * (k - 1); q < 2 * k; ++q) {
                const int rq = (q & 3) + 8 * (q >> 2);
                S[cbase[((q >> 1) & 1) | (((q >> 2) & 1) << 1)] + rq * 32] = P[q];
            }
        }
        if (k == 9) ws_lds_barrier();
        if (k >= 10 && k <= 13) p0[k - 10] = *reinterpret_cast<const f32x4n*>(S + (k - 10) * 1024 + cfetch);
    };
    // tail of A(hf,t): dr_pre out, acc_own += drh r
    auto a_tail = [&](int k, int t, int hf, const f32x16& P, unsigned* ctr, unsigned epoch) {
        spill_fetch(k, P);
        if (k == 15) va = p0[0] + p0[1] + p0[2] + p0[3];                       // drh
        if (k == 17) {
            acc_own[hf] += va * tr[hf];
            vb = va * hp_own[hf] * tr[hf] * ((f32x4n)(1.f) - tr[hf]);           // dr_pre
        }
        if (k == 33) ws_store2(rs_dxp, o_xp[hf], ws_uni((unsigned)t * xp_step), vb, false);
        if (k == 39) ws_store2(rs_drF, frag_v, ws_uni(frag_st(t, hf)), vb, false);
        if (k == 54) {
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            ws_flag_store(ctr, j, epoch);
        }
    };
    // tail of B(hf,t), t > 0: g = dL/dh_{t-1}, then the first half of step t-1
    auto b_tail = [&](int k, int t, int hf, const f32x16& P, unsigned* ctr, unsigned epoch, bool live) {
        spill_fetch(k, P);
        if (k == 15) {
            va = p0[0] + p0[1] + p0[2] + p0[3] + acc_own[hf];
            if (a.d_outs != nullptr) va += td[hf];
        }
        if (k == 17 && live) half_step(t - 1, hf, va, tu[hf], tc[hf], th[hf], va, vb);      // va = dc_pre, vb = du_pre afterwards
        if (k == 33) ws_store2(rs_dxp, live ? o_xp[hf] : OOB, ws_uni((unsigned)(t - 1) * xp_step + 2 * H * 4), va, false);
        if (k == 35) ws_store2(rs_dxp, live ? o_xp[hf] : OOB, ws_uni((unsigned)(t - 1) * xp_step + H * 4), vb, false);
        if (k == 37) ws_store2(rs_duF, live ? frag_v : OOB, ws_uni(frag_st(t - 1, hf)), vb, false);
        if (k == 39) ws_store2(rs_dcF, live ? frag_v : OOB, ws_uni(frag_st(t - 1, hf)), va, false);
        if (k == 54) {
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            if (live) ws_flag_store(ctr, j, epoch);
        }
    };
    unsigned peek = 0;
    auto way_in = [&](int k, const unsigned* line, unsigned epoch) {
        if (k == 76 && w == 0) peek = ws_flag_load(line);
        if (k == 92) {
            if (w == 0 && !ws_flags_reached(peek, epoch)) ws_spin(line, epoch, err, a.spin_limit);
            ws_bar();
        }
    };

    f32x4n ring[16];                    // the A streams use slots 0..7
    ws_wait(ctr0, 1u, err, a.spin_limit);
#pragma unroll
    for (int m = 0; m < 8; ++m) ring[m] = ws_load2(rs_dcF, lane16, frag_a(T - 1, 0) + m * 1024);

    // A stream: 32 octets against the LDS slab (128 slots of one MFMA); B stream: 64 octets against the register slab
    // (128 slots of two MFMAs).  HOOK(slot - 4) in every slot; the ring's last 8 refills fetch the NEXT stream's head.
#define WS_A_STREAM(acc0, base_cur, rs_nxt, base_nxt, voff_nxt, HOOK)                                    \
    {                                                                                                          \
        f32x4n bq[2];                                                                                          \
        bq[0] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w) * 64 + lane) * 4);                             \
        bq[1] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w + 1) * 64 + lane) * 4);                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int m = 0; m < 32; ++m) {                                                       \
            const f32x4n av = ring[m % 8];                                                                     \
            const f32x4n bv = bq[m & 1];                                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
                acc0 = WS_MFMA(av[i], bv[i], acc0);      /* one chain: 66 against 64.6 cycles per MFMA, 16 registers saved */ \
                if (i == 1 && m + 2 < 32)                                                                      \
                    bq[m & 1] = *reinterpret_cast<const f32x4n*>(Wl + ((32 * w + m + 2) * 64 + lane) * 4);     \
                if (i == 3) {                                                                                  \
                    if (m + 8 < 32) ring[m % 8] = ws_load2(rs_dcF, lane16, (base_cur) + (m + 8) * 1024);       \
                    else ring[m % 8] = ws_load2(rs_nxt, (voff_nxt), (base_nxt) + (m + 8 - 32) * 1024);         \
                }                                                                                              \
                HOOK(4 * m + i - 4);                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                                  \
        }                                                                                                      \
    }
    // B stream: 16 fragments ahead.  TOPUP: entered from an A stream, whose ring left fragments 0..7 -- fetch 8..15 now.
    // NEXT_A: followed by an A stream -- only its fragments 0..7 are fetched at the end (ring slots 0..7).
#define WS_B_STREAM(accA, accB, base_cur, TOPUP, NEXT_A, rs_nxt, base_nxt, voff_nxt, HOOK)                     \
    if ((TOPUP) && !WS_TOPUP_SPREAD) {                                                                         \
        _Pragma("unroll") for (int f = 8; f < 16; ++f) ring[f] = ws_load2(rs_bF, lane16, (base_cur) + f * 1024); \
    }                                                                                                          \
    _Pragma("unroll") for (int m = 0; m < 64; ++m) {                                                           \
        const f32x4n av = ring[m % 16];                                                                        \
        if ((TOPUP) && WS_TOPUP_SPREAD && m < 8) ring[8 + m] = ws_load2(rs_bF, lane16, (base_cur) + (8 + m) * 1024); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
            if (m < 2) {                                                                                       \
                if (i == 0) { if (m == 0) WS_MFMA_AGPR_B0(accA, av[i], wq[m][i]); else WS_MFMA_AGPR_B0(accB, av[i], wq[m][i]); } \
                else { if (m == 0) WS_MFMA_AGPR_B(accA, av[i], wq[m][i]); else WS_MFMA_AGPR_B(accB, av[i], wq[m][i]); } \
            } else if (m < WS_BW_AG_OCT) {                                                                     \
                if (m & 1) WS_MFMA_AGPR_B(accB, av[i], wq[m][i]); else WS_MFMA_AGPR_B(accA, av[i], wq[m][i]);   \
            } else {                                                                                           \
                if (m & 1) WS_MFMA_VGPR_B(accB, av[i], wq[m][i]); else WS_MFMA_VGPR_B(accA, av[i], wq[m][i]);   \
            }                                                                                                  \
            if (i == 3) {                                                                                      \
                if (m + 16 < 64) ring[m % 16] = ws_load2(rs_bF, lane16, (base_cur) + (m + 16) * 1024);         \
                else if (!(NEXT_A)) ring[m % 16] = ws_load2(rs_nxt, (voff_nxt), (base_nxt) + (m + 16 - 64) * 1024); \
                else if (m >= 56) ring[m - 56] = ws_load2(rs_nxt, (voff_nxt), (base_nxt) + (m - 56) * 1024);   \
            }                                                                                                  \
            if (i & 1) {                                                                                       \
                HOOK(2 * m + (i >> 1) - 4);                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                                  \
        }                                                                                                      \
    }

    f32x16 pA0, pA1, pB0, pB1;          // summed result tiles of A(0,.), A(1,.), B(0,.), B(1,.), pending their tails
#pragma unroll
    for (int q = 0; q < 16; ++q) pB1[q] = 0.f;

    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        const unsigned n = (unsigned)(2 * s);
        const bool more = t > 0;        // the B streams of this step exist
        // ---------------- A(0,t); in its gaps: the tail of B(1,t+1), then the way into A(1,t)
        {
#pragma unroll
            for (int q = 0; q < 16; ++q) pA0[q] = 0.f;
            auto hook = [&](int k) {
                b_tail(k, t + 1, 1, pB1, ctr1, 1u + n, s > 0);
                way_in(k, ctr1, 1u + n);
            };
            WS_A_STREAM(pA0, frag_a(t, 0), rs_dcF, frag_a(t, 1), lane16, hook)
        }
        // ---------------- A(1,t); tail of A(0,t); way into B(0,t)
        {
#pragma unroll
            for (int q = 0; q < 16; ++q) pA1[q] = 0.f;
            auto hook = [&](int k) {
                a_tail(k, t, 0, pA0, ctr0, 2u + n);
                way_in(k, ctr0, more ? 2u + n : 0u);
            };
            WS_A_STREAM(pA1, frag_a(t, 1), rs_bF, more ? frag_b(t, 0) : 0u, more ? lane16 : OOB, hook)
        }
        if (!more) break;
        // ---------------- B(0,t); tail of A(1,t); way into B(1,t)
        {
            f32x16 x;
            auto hook = [&](int k) {
                a_tail(k, t, 1, pA1, ctr1, 2u + n);
                way_in(k, ctr1, 2u + n);
                tape_fetch(k, t, 0);
            };
            WS_B_STREAM(pB0, x, frag_b(t, 0), true, false, rs_bF, frag_b(t, 1), lane16, hook)
            WS_MFMA_DRAIN();
            pB0 += x;
        }
        // ---------------- B(1,t); tail of B(0,t); way into A(0,t-1)
        {
            f32x16 x;
            auto hook = [&](int k) {
                b_tail(k, t, 0, pB0, ctr0, 3u + n, true);
                way_in(k, ctr0, 3u + n);
                tape_fetch(k, t, 1);
            };
            WS_B_STREAM(pB1, x, frag_b(t, 1), false, true, rs_dcF, frag_a(t - 1, 0), lane16, hook)
            WS_MFMA_DRAIN();
            pB1 += x;
        }
    }
    // ---------------- the tail of A(1,0), on its own
#pragma unroll
    for (int k = 1; k <= 39; ++k) a_tail(k, 0, 1, pA1, ctr1, 2u * T);
    ws_arrive(ctr1, j, 2u * T);
#undef WS_A_STREAM
#undef WS_B_STREAM
}

struct WsDevice { int cus = 0, blocks = -1, blocks_bwd = -1; };
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
        int n2 = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_ws_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WS_LDS) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(gru_ws_fwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)WS_LDS) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_ws_fwd_kernel, WS_NT, WS_LDS) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, gru_ws_fwd2_kernel, WS_NT, WS_LDS) == hipSuccess) {
            d.blocks = n < n2 ? n : n2;
            int n3 = 0;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_ws_bwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)WS_LDS) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&n3, gru_ws_bwd2_kernel, WS_NT, WS_LDS) == hipSuccess)
                d.blocks_bwd = n3;
        }
    }
    return d;
}
int g_ws_mode = 3;                      // bit 0: forward, bit 1: back-propagation (wherever they apply)
int g_ws_form = 0;                      // tuning: 1 = the plain sub-phase order also for two half-chains (A/B of the splice)
unsigned long long* g_ws_stamps = nullptr;

}  // namespace

namespace {
// flags, this launch's error word and the placement counters back to zero; word 1023 (the sticky error word) stays
int ws_reset(unsigned* sync, hipStream_t st) {
    return hipMemsetAsync(sync, 0, 1023 * sizeof(unsigned), st) == hipSuccess ? VQA_OK : VQA_ERR_LAUNCH;
}
}  // namespace

extern "C" int vqa_gru_ws_set_mode(int mode) {
    g_ws_mode = mode < 0 ? 3 : (mode & 3);
    return VQA_OK;
}
extern "C" int vqa_gru_ws_set_form(int form) {
    g_ws_form = form & 0x73;        // bit 0: plain sub-phase order; bit 1: write-through hand-offs; bits 4-6: timing study
    return VQA_OK;
}
extern "C" int vqa_gru_ws_set_stamps(unsigned long long* dev_words) {
    g_ws_stamps = dev_words;
    return VQA_OK;
}

// 1 when the weight-stationary recurrence applies: H = 1024, at most 512 rows (eight chains of 64), and a device of
// exactly 8 x 32 CUs on which one 160 KB / 256-thread workgroup per CU is resident.
extern "C" int vqa_gru_ws_supported(int T, int B, int H) {
    if (!(g_ws_mode & 1)) return 0;
    if (T <= 0 || H != WS_H || B <= 0 || B > WS_CHAINS * 64) return 0;
    const WsDevice& d = ws_device();
    return (d.cus == WS_CHAINS * WS_CU && d.blocks >= 1) ? 1 : 0;
}

// device memory the call needs besides the tape: fragment-order hand-off buffers and the counters
extern "C" int64_t vqa_gru_ws_workspace_bytes(int T) {
    return T < 0 ? 0 : ((int64_t)(3 * T + 1) * WS_FRAG_T + 1024) * (int64_t)sizeof(float);      // forward 2 T + 1, backward 3 T
}

// Whole forward recurrence in one launch; same tape contract as vqa_gru_seq_fwd (xp [T,B,3H] read only, hs [T+1,B,H]
// with hs[0] given, r,u,c,rh [T,B,H]).  `ws`: vqa_gru_ws_workspace_bytes(T) of device memory (contents irrelevant).
// After the stream has run, a non-zero word at ((unsigned*)ws)[512] reports a barrier time-out of THIS launch (results
// invalid); word [1023] is set with it and never cleared by a launch (the caller zeroes it once and may look later).
extern "C" int vqa_gru_seq_fwd_ws(const float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs,
                                  float* r, float* u, float* c, float* rh, int T, int B, int H, void* ws, void* stream) {
    VQA_REQUIRE(xp && Wg_h && Wc_h && len && hs && r && u && c && rh && ws, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gru_ws_supported(T, B, H) == 1, VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE((int64_t)T * B * 3 * H * 4 < 0x7FFFFFF0ll && vqa_aligned16(ws), VQA_ERR_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    WsArgs a;
    a.sync = static_cast<unsigned*>(ws);
    if (ws_reset(a.sync, st) != VQA_OK) return VQA_ERR_LAUNCH;
    a.hF = static_cast<float*>(ws) + 1024;
    a.rhF = a.hF + (int64_t)(T + 1) * WS_FRAG_T;
    a.xp = xp; a.Wg = Wg_h; a.Wc = Wc_h; a.len = len; a.hs = hs; a.r = r; a.u = u; a.c = c; a.rh = rh;
    a.T = T; a.B = B;
    a.spin_limit = 2000000u;
    a.stamps = g_ws_stamps;
    a.handoff_sc1 = (g_ws_form & 2) ? 1 : 0;
    a.dbg = (g_ws_form >> 4) & 7;
    // more than 256 rows: chains of 64 rows = two half-chains in anti-phase, tails inside the matrix stream; up to 256
    // rows: chains of 32 rows so that all eight XCDs work (one half-chain each, the plain sub-phase order)
    a.chain_rows = B > 256 ? 64 : 32;
    if (a.chain_rows == 64 && !(g_ws_form & 1))
        hipLaunchKernelGGL(gru_ws_fwd2_kernel, dim3(WS_CHAINS * WS_CU), dim3(WS_NT), WS_LDS, st, a);
    else
        hipLaunchKernelGGL(gru_ws_fwd_kernel, dim3(WS_CHAINS * WS_CU), dim3(WS_NT), WS_LDS, st, a);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

// 1 when the weight-stationary back-propagation applies: as the forward form, and more than 256 rows (it exists in the
// two-half form only)
extern "C" int vqa_gru_ws_bwd_supported(int T, int B, int H) {
    if (!(g_ws_mode & 2)) return 0;
    if (T <= 0 || H != WS_H || B <= 256 || B > WS_CHAINS * 64) return 0;
    const WsDevice& d = ws_device();
    return (d.cus == WS_CHAINS * WS_CU && d.blocks_bwd >= 1) ? 1 : 0;
}

// Whole back-propagation through time in one launch: dxp [T,B,3H] = (dr_pre | du_pre | dc_pre) from dh_T [B,H] (read
// only, unlike vqa_gru_seq_bwd) and the forward tape; d_outs [T,B,H] or NULL as in vqa_gru_seq_bwd_outs.  `ws` as in
// vqa_gru_seq_fwd_ws (the same buffer may serve both: the calls do not overlap on a stream).
extern "C" int vqa_gru_seq_bwd_ws(const float* dh_T, const float* d_outs, const float* Wg_h, const float* Wc_h,
                                  const int32_t* len, const float* hs, const float* r, const float* u, const float* c,
                                  float* dxp, int T, int B, int H, void* ws, void* stream) {
    VQA_REQUIRE(dh_T && Wg_h && Wc_h && len && hs && r && u && c && dxp && ws, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gru_ws_bwd_supported(T, B, H) == 1, VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE((int64_t)T * B * 3 * H * 4 < 0x7FFFFFF0ll && vqa_aligned16(ws) && vqa_aligned16(Wg_h) && vqa_aligned16(Wc_h),
                VQA_ERR_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    WsBwdArgs a;
    a.sync = static_cast<unsigned*>(ws);
    if (ws_reset(a.sync, st) != VQA_OK) return VQA_ERR_LAUNCH;
    a.dcF = static_cast<float*>(ws) + 1024;
    a.drF = a.dcF + (int64_t)T * WS_FRAG_T;
    a.duF = a.drF + (int64_t)T * WS_FRAG_T;
    a.dh_T = dh_T; a.d_outs = d_outs; a.Wg = Wg_h; a.Wc = Wc_h; a.len = len; a.hs = hs; a.r = r; a.u = u; a.c = c; a.dxp = dxp;
    a.T = T; a.B = B;
    a.spin_limit = 2000000u;
    a.stamps = g_ws_stamps;
    hipLaunchKernelGGL(gru_ws_bwd2_kernel, dim3(WS_CHAINS * WS_CU), dim3(WS_NT), WS_LDS, st, a);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
