// Persistent forward GRU recurrence for gfx950: all T time steps of tf.contrib.rnn.GRUCell under
// tf.nn.dynamic_rnn(sequence_length) (vlmap/modules.py:124-140) in ONE launch.
//
// Why: the per-step kernels (gemm_f32.hip, two launches per step) cost  ~7 us + MFMA time / 0.8  each
// (profiles/r2_trace_summary.txt: gates 24.3 us for 13.7 us of MFMA, candidate 15.7 us for 6.8 us); 28 launches
// make the fixed part 196 of the 553 us of a forward recurrence at B 512, H 1024, T 14.  A single launch with a
// grid barrier in place of each kernel boundary does not help by itself -- a barrier costs what a boundary costs
// (MI355X_MICROARCH.md, barrier-xcd vs boundary) -- so this kernel runs TWO independent chains (the two halves of
// the batch; samples are independent) as two co-resident 8-wave workgroups per CU: while one chain's workgroup
// waits at its barrier, writes its epilogue or ramps up its next tile, the other chain's workgroup owns the matrix
// cores.  The barrier latency is hidden, not removed.
//
// Decomposition: chain c = rows [c * chain_rows, ...); per step two phases per chain,
//   gates      G = h_{t-1} Wg_h + xp[:, :2H]   (32 x 64 tiles, 4 k-groups x 2 column sub-tiles)  -> r, u, r*h
//   candidate  C = (r*h) Wc_h + xp[:, 2H:]     (32 x 32 tiles, 8 k-groups)                        -> c, h_t
// 2H/64 * rows/32 = H/32 * rows/32 tiles per phase = one per workgroup at rows 256, H 1024.
//
// Hand-off protocol (MI355X_MICROARCH.md, visibility table, row 3): everything one phase hands to the next
// (h, r*h, u) is stored write-through (`sc1`), each 128-byte line by one store instruction of one wave; every
// storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup barriers, ONE lane adds to the chain's counter
// (agent-scope atomic); consumers poll that counter with `sc1` loads, barrier, and read the handed-off bytes with
// `sc1` loads only.  Every address is written once and read only afterwards (hs, r, u, c, rh are indexed by t), so
// no stale copy of a handed-off line can exist in any L2 / L1.  Counters are zeroed by a memset node in front of the
// launch; every spin is bounded and sets an error word instead of hanging (the launch then drains quickly).
#include <stdlib.h>

#include "vqa_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PK_NT = 512;                 // 8 waves; two workgroups per CU
constexpr int PK_BM = 32;
constexpr unsigned PK_OOB = 0xFFFFFFF0u;
constexpr int SC1 = 16;                    // buffer aux bit: system-coherence level 1 (write-through / L1 bypass)
constexpr int STG_LD = 36;
// XCD-local form (64-row tiles, 16 waves), compile-time tuning: k tile of the gate phase, SIMD-partner stagger
#ifndef PK_GATES_BK64
#define PK_GATES_BK64 128
#endif
#ifndef PK_STAGGER
#define PK_STAGGER 0
#endif

struct PkArgs {
    const float* xp;      // [T,B,3H]  x_t W_x + b for (r|u|c)
    const float* Wg;      // [H,2H]
    const float* Wc;      // [H,H]
    const int32_t* len;   // [B]
    float *hs, *r, *u, *c, *rh;   // hs [T+1,B,H] (hs[0] given), tape [T,B,H]
    unsigned* sync;       // [0], [16]: chain counters (own 64-byte lines); [32]: error word
    int T, B, H;
    int chain_rows;       // rows of chain 0 (multiple of 32); chain 1 holds the rest
    unsigned spin_limit;
    int chain1_delay;     // tuning: chain 1 starts this many 100 MHz ticks late (phase offset between the chains)
    unsigned* census;     // optional [gridDim.x]: (XCC_ID << 16) | HW_ID[15:0] of every workgroup (placement study)
    int xcd_mode;         // experiment: EIGHT chains of chain_rows rows, chain = blockIdx.x & 7 (one per XCD), one
                          // workgroup per CU; counters at sync[16 * chain]
    unsigned* err;        // error word (sync[32]; sync[192] in xcd_mode)
};

__device__ __forceinline__ f32x4n bload(__amdgpu_buffer_rsrc_t rs, unsigned off, bool sc1) {
    const u32x4 v = sc1 ? __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, SC1) : __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    f32x4n o;
    o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
    return o;
}
__device__ __forceinline__ void bstore_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4n x) {
    u32x4 v;
    v.x = __float_as_uint(x.x); v.y = __float_as_uint(x.y); v.z = __float_as_uint(x.z); v.w = __float_as_uint(x.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, SC1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrs(const float* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}

// arrive: every wave has drained its write-through stores; one lane signals for the workgroup
__device__ __forceinline__ void pk_arrive(unsigned* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: one lane polls (sc1 loads), bounded; the workgroup barrier orders every later load behind the poll
__device__ __forceinline__ void pk_wait(unsigned* ctr, unsigned target, unsigned* err, unsigned limit) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// One output tile of one phase.  GATES: BN 64, BK 64, WGK 4.  CAND: BN 32, BK 128, WGK 8.
//   A [rows, K] k-contiguous, handed off by the previous phase -> sc1 loads;  B [K, N] weights, plain loads.
// B's first tile is fetched BEFORE the barrier wait (weights do not depend on it).
// BM 32: 8 waves (the two-chain form, two workgroups per CU).  BM 64: 16 waves, two 32-row halves (wm) that share the
// weight tile -- one 64-row tile per workgroup, CU and phase in the XCD-local form.
template <bool GATES, int BM>
__device__ __forceinline__ void pk_tile(const PkArgs& a, int t, int row_lo, int row_hi, int m0, int n0, float* smem,
                                        unsigned* ctr, unsigned target) {
    constexpr int PK_BM = BM, PK_NT = BM * 16;
    constexpr int BN = GATES ? 64 : 32, BK = GATES ? (BM == 32 ? 64 : PK_GATES_BK64) : 128, WGK = GATES ? 4 : 8;
    constexpr int WAVES_N = BN / 32, WAVES_M = BM / 32;
    // Hand-offs are stored write-through (sc1) in both forms.  The two-chain form reads them back with sc1 loads (its
    // producers sit on other XCDs).  In the XCD-local form (BM 64) producer and consumer share an L2, every address is
    // written once per launch before anyone reads it and a 4 KB row never straddles two chains, so a plain load either
    // hits the producer's line in this L2 or -- should a workgroup ever be placed on another XCD -- misses and fetches
    // what the write-through store put in memory.
    constexpr bool LD_SC1 = (BM == 32);
    static_assert(WAVES_M * WAVES_N * WGK * 64 == PK_NT, "one 32x32 tile per (row half, column sub-tile, k-group) wave");
    constexpr int A_LD = BK + 4;
    constexpr int A_FL = PK_BM * A_LD, B_FL = BK * BN;
    constexpr int NVA = PK_BM * BK / 4 / PK_NT, NVB = BK * BN / 4 / PK_NT;
    constexpr int NC = BK / 8 / WGK;
    const int H = a.H, K = H, B = a.B;
    const int N = GATES ? 2 * H : H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / (WAVES_N * WGK), wrest = wave % (WAVES_N * WGK);
    const int wk = wrest / WAVES_N, wn = wrest % WAVES_N;
    const int64_t BH = (int64_t)B * H;

    const float* Aop = GATES ? a.hs + (int64_t)t * BH : a.rh + (int64_t)t * BH;      // [B, H]
    const float* Bop = GATES ? a.Wg : a.Wc;
    const int ldb = N;
    const __amdgpu_buffer_rsrc_t rsA = mkrs(Aop, BH * 4);
    const __amdgpu_buffer_rsrc_t rsB = mkrs(Bop, (int64_t)K * N * 4);

    // per-lane byte offsets of the operand tiles (rows past the chain clamp to its last row; never stored)
    unsigned offA[NVA], offB[NVB];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int idx = threadIdx.x + i * PK_NT;
        const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
        offA[i] = (unsigned)(((int64_t)min(m0 + row, row_hi - 1) * H + kq) * 4);
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int idx = threadIdx.x + i * PK_NT;
        const int k = idx / (BN / 4), nq = (idx % (BN / 4)) * 4;
        offB[i] = (unsigned)(((int64_t)k * ldb + n0 + nq) * 4);
    }
    // The left operand was just written by other CUs and comes back through the fabric (~2 us): its tiles are
    // fetched DA tiles ahead into a register queue (tile j lives in slot j % DA); the weights stay in this XCD's L2
    // and need one tile of cover.  Issue order inside an iteration is weights first, far A tile second, so waiting
    // for the weights (in-order vmcnt) leaves the far loads in flight.
    // XCD-local form (BM 64): the left operand now comes from this XCD's L2 (one tile of cover is enough) while every
    // XCD streams ALL of the weights each step (12 MB, more than its L2): the weights get the second tile of cover.
    constexpr int DA = BM == 64 ? (PK_STAGGER ? 2 : 1) : (GATES ? 4 : 2);
    constexpr int DB = BM == 64 ? 2 : 1;
    constexpr int UNR = DA > 2 ? DA : 2;
    f32x4n ra[DA][NVA], rb[DB][NVB];
    auto loadB = [&](f32x4n (&dst)[NVB], int kt) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) dst[i] = bload(rsB, offB[i] + (unsigned)(kt * BK * ldb) * 4u, false);
    };
    auto loadA = [&](f32x4n (&dst)[NVA], int kt) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) dst[i] = bload(rsA, offA[i] + (unsigned)(kt * BK) * 4u, LD_SC1);
    };
    auto stash = [&](float* L, const f32x4n (&srcA)[NVA], const f32x4n (&srcB)[NVB]) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int idx = threadIdx.x + i * PK_NT;
            *reinterpret_cast<f32x4n*>(L + (idx / (BK / 4)) * A_LD + (idx % (BK / 4)) * 4) = srcA[i];
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int idx = threadIdx.x + i * PK_NT;
            *reinterpret_cast<f32x4n*>(L + A_FL + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4) = srcB[i];
        }
    };

    unsigned long long* stamp = nullptr;                   // phase time stamps of two workgroups (timing study)
    if (a.census != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == (gridDim.x >> 1)))
        stamp = reinterpret_cast<unsigned long long*>(a.census + 1024) +
                ((blockIdx.x ? 1 : 0) * 64 + (2 * t + (GATES ? 0 : 1))) * 4;
    if (stamp) stamp[0] = wall_clock64();
    loadB(rb[0], 0);
    if (DB == 2) loadB(rb[DB - 1], 1);
    // ---- the barrier: everything below reads what the previous phase of this chain wrote
    pk_wait(ctr, target, a.err, a.spin_limit);
    if (stamp) stamp[1] = wall_clock64();
    const int nt = K / BK;                 // a multiple of UNR (H % 512 == 0 is required by the host)
#pragma unroll
    for (int j = 0; j < DA; ++j) loadA(ra[j], j);

    // epilogue side inputs of this lane's share of the tile (see the finish loop below)
    constexpr int RPG = 32 / WGK;                       // rows per k-group: 8 (gates) / 4 (candidate)
    const int e_lr = lane >> 3;                         // 0..7
    const bool e_act = e_lr < RPG;
    const int e_row = wm * 32 + wk * RPG + e_lr;        // row inside the BM-row tile
    const int grow = m0 + e_row, gcol = n0 + wn * 32 + (lane & 7) * 4;
    const bool e_ok = e_act && grow < row_hi;
    const __amdgpu_buffer_rsrc_t rsXp = mkrs(a.xp + (int64_t)t * B * 3 * H, (int64_t)B * 3 * H * 4);
    const __amdgpu_buffer_rsrc_t rsHp = mkrs(a.hs + (int64_t)t * BH, BH * 4);
    const __amdgpu_buffer_rsrc_t rsU = mkrs(a.u + (int64_t)t * BH, BH * 4);
    f32x4n e_d = (f32x4n)(0.f), e_h = (f32x4n)(0.f), e_u = (f32x4n)(0.f);
    int e_len = 0;
    {
        const unsigned oxp = e_ok ? (unsigned)(((int64_t)grow * 3 * H + (GATES ? 0 : 2 * H) + gcol) * 4) : PK_OOB;
        e_d = bload(rsXp, oxp, false);
        const unsigned ohp = (e_ok && (!GATES || gcol < H)) ? (unsigned)(((int64_t)grow * H + gcol) * 4) : PK_OOB;
        e_h = bload(rsHp, ohp, LD_SC1);
        if (!GATES) {
            e_u = bload(rsU, e_ok ? (unsigned)(((int64_t)grow * H + gcol) * 4) : PK_OOB, LD_SC1);
            e_len = e_ok ? a.len[grow] : 0;
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](const float* L) {
        const float* As = L;
        const float* Bs = L + A_FL;
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int c = cc * WGK + wk;
            const f32x4n af = *reinterpret_cast<const f32x4n*>(As + (wm * 32 + i) * A_LD + c * 8 + h * 4);
            const float* pb = Bs + (c * 8 + h * 4) * BN + wn * 32 + i;
            const float b0 = pb[0], b1 = pb[BN], b2 = pb[2 * BN], b3 = pb[3 * BN];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, b3, acc, 0, 0, 0);
        }
    };

    float* L0 = smem;
    float* L1 = smem + (A_FL + B_FL);
    stash(L0, ra[0], rb[0]);
    __syncthreads();
    for (int k0 = 0; k0 < nt; k0 += UNR) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int kt = k0 + j;
            float* cur = (j & 1) ? L1 : L0;
            float* nxt = (j & 1) ? L0 : L1;
            if (kt + DB < nt) loadB(rb[j % DB], kt + DB);       // slot j % DB: tile kt was stashed one iteration ago
            if (kt + DA < nt) loadA(ra[j % DA], kt + DA);
            __builtin_amdgcn_sched_barrier(0);
            if (BM == 64 && PK_STAGGER && ((wave >> 2) & 1)) {
                // SIMD partners (waves w and w + 4 share a SIMD): this half refills LDS first and computes second, so
                // that on every SIMD one pair is in its MFMA burst while the other writes the next tile
                if (kt + 1 < nt) stash(nxt, ra[(j + 1) % DA], rb[(j + 1) % DB]);
                __builtin_amdgcn_sched_barrier(0);
                compute(cur);
            } else {
                compute(cur);
                if (BM == 32) __builtin_amdgcn_sched_barrier(0);     // 16-wave form: the LDS writes may slide under the MFMA tail
                if (kt + 1 < nt) stash(nxt, ra[(j + 1) % DA], rb[(j + 1) % DB]);
            }
            __syncthreads();
        }
    }

    if (stamp) stamp[2] = wall_clock64();
    // ---- epilogue: the WGK partial tiles meet in LDS (transposed 32 x 36 patches), every k-group finishes RPG rows
    float* stg = smem + ((wm * WGK + wk) * WAVES_N + wn) * 32 * STG_LD;
#pragma unroll
    for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * STG_LD + (lane & 31)] = acc[r];
    __syncthreads();
    if (e_ok) {
        const float* src = smem + (wm * WGK * WAVES_N + wn) * 32 * STG_LD + (wk * RPG + e_lr) * STG_LD + (lane & 7) * 4;
        f32x4n v = *reinterpret_cast<const f32x4n*>(src);
#pragma unroll
        for (int g = 1; g < WGK; ++g) v += *reinterpret_cast<const f32x4n*>(src + g * WAVES_N * 32 * STG_LD);
        v += e_d;
        const unsigned o = (unsigned)(((int64_t)grow * H + (gcol < H ? gcol : gcol - H)) * 4);
        if (GATES) {
            f32x4n s;
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = sigmoidf_stable(v[j]);
            if (gcol < H) {
                bstore_sc1(mkrs(a.r + (int64_t)t * BH, BH * 4), o, s);            // r
                bstore_sc1(mkrs(a.rh + (int64_t)t * BH, BH * 4), o, s * e_h);     // r * h_prev
            } else {
                bstore_sc1(mkrs(a.u + (int64_t)t * BH, BH * 4), o, s);            // u
            }
        } else {
            f32x4n cc, hn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cc[j] = tanhf(v[j]);
                hn[j] = (t < e_len) ? (e_u[j] * e_h[j] + (1.f - e_u[j]) * cc[j]) : e_h[j];
            }
            bstore_sc1(mkrs(a.c + (int64_t)t * BH, BH * 4), o, cc);               // c
            bstore_sc1(mkrs(a.hs + (int64_t)(t + 1) * BH, BH * 4), o, hn);        // h_t
        }
    }
    __syncthreads();          // the patches overlap the next tile's operand buffers
    if (stamp) stamp[3] = wall_clock64();
}

template <int BM>
__global__ __launch_bounds__(BM * 16, 4) void gru_fwd_persistent_kernel(PkArgs a) {
    constexpr int PK_BM = BM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // The first `slots` workgroups are chain 0, the rest chain 1: workgroups are dealt round-robin over the XCDs and
    // (observed) over an XCD's CUs before any CU gets its second one, so a CU tends to host one workgroup of each
    // chain -- speed only, never correctness.  Within a chain, XCD x (= slot % 8) owns a contiguous run of tiles,
    // m fastest: its 32 workgroups cover all row tiles of 4 (gates) + 4 (candidate) column panels, whose weight
    // slices (1.5 MB) then stay in that XCD's L2 for all T steps and both chains.
    if (a.census != nullptr && threadIdx.x == 0)
        a.census[blockIdx.x] = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 16) |      // HW_REG_XCC_ID[3:0]
                               (__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) & 0xFFFFu);   // HW_REG_HW_ID[15:0]
    // xcd_mode (experiment): the rows of a batch are independent sequences, so chain x = the workgroups with
    // blockIdx.x % 8 == x (dealt to XCD x, see the census) owns rows [x * chain_rows, ...): its hand-offs and its
    // barrier stay among the 32 workgroups of one XCD.
    const int slots = a.xcd_mode ? (int)(gridDim.x >> 3) : (int)(gridDim.x >> 1);
    const int chain = a.xcd_mode ? (int)(blockIdx.x & 7) : (blockIdx.x >= (unsigned)slots ? 1 : 0);
    const int raw = a.xcd_mode ? (int)(blockIdx.x >> 3) : (int)blockIdx.x - chain * slots;
    const int slot = (!a.xcd_mode && slots % 8 == 0) ? (raw & 7) * (slots >> 3) + (raw >> 3) : raw;
    const int row_lo = a.xcd_mode ? min(chain * a.chain_rows, a.B) : (chain ? a.chain_rows : 0);
    const int row_hi = a.xcd_mode ? min(row_lo + a.chain_rows, a.B) : (chain ? a.B : min(a.chain_rows, a.B));
    const int tiles_m = (row_hi - row_lo + PK_BM - 1) / PK_BM;
    const int tn_g = 2 * a.H / 64, tn_c = a.H / 32;
    unsigned* ctr = a.sync + chain * 16;
    if (chain == 1 && a.chain1_delay > 0) {
        const unsigned long long t_end = wall_clock64() + (unsigned long long)a.chain1_delay;
        while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(8);
    }
    unsigned phase = 0;      // phases completed by the chain's workgroups so far
    for (int t = 0; t < a.T; ++t) {
        // gates
        bool did = false;
        for (int tile = slot; tile < tiles_m * tn_g; tile += slots) {
            pk_tile<true, BM>(a, t, row_lo, row_hi, row_lo + (tile % tiles_m) * PK_BM, (tile / tiles_m) * 64, smem, ctr,
                          phase * (unsigned)slots);
            did = true;
        }
        if (!did) pk_wait(ctr, phase * (unsigned)slots, a.err, a.spin_limit);   // keep the phase order
        pk_arrive(ctr);
        ++phase;
        // candidate
        did = false;
        for (int tile = slot; tile < tiles_m * tn_c; tile += slots) {
            pk_tile<false, BM>(a, t, row_lo, row_hi, row_lo + (tile % tiles_m) * PK_BM, (tile / tiles_m) * 32, smem, ctr,
                           phase * (unsigned)slots);
            did = true;
        }
        if (!did) pk_wait(ctr, phase * (unsigned)slots, a.err, a.spin_limit);
        pk_arrive(ctr);
        ++phase;
    }
}

constexpr size_t pk_lds_bytes(int bm = PK_BM) {
    // max over the two phases of 2 x (A tile + B tile) and of the epilogue patches (one 32 x 36 patch per wave)
    const size_t bkg = bm == 64 ? PK_GATES_BK64 : 64;
    const size_t g = 2 * (bm * (bkg + 4) + bkg * 64), c = 2 * (bm * (128 + 4) + 128 * 32);
    const size_t m = g > c ? g : c, p = (size_t)(bm / 4) * 32 * STG_LD;
    return (m > p ? m : p) * sizeof(float);
}

int g_persist = -1;      // -1 automatic (when it applies), 0 off, 1 on
unsigned* g_census = nullptr;

}  // namespace

extern "C" int vqa_gru_set_persistent(int mode) {
    g_persist = mode < 0 ? -1 : (mode ? 1 : 0);
    return VQA_OK;
}

// placement study: device array of >= 2 * #CUs words that the next launches fill with each workgroup's hardware ids
extern "C" int vqa_gru_persistent_set_census(unsigned* dev_words) {
    g_census = dev_words;
    return VQA_OK;
}

extern "C" int64_t vqa_gru_persistent_sync_bytes(void) { return 256 * sizeof(unsigned); }

// Co-residency of the variant that a launch would use, PER DEVICE (the grid barrier never completes unless every
// workgroup of the grid is resident): blocks per CU from the occupancy query for that variant's thread count and LDS.
// variant 0: <32>, two 8-wave workgroups per CU (the two-chain form and the 32-row XCD form need >= 2 resp. 1);
// variant 1: <64>, one 16-wave workgroup per CU (VQA_GRU_PERSIST_XCD=2).
namespace {
struct PkResidency { int cus = 0, blocks32 = -1, blocks64 = -1; };
const PkResidency& pk_residency() {
    static PkResidency per_dev[64];
    static bool known[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    PkResidency& r = per_dev[dev];
    if (!known[dev]) {
        known[dev] = true;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return r;
        r.cus = prop.multiProcessorCount;
        int n = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_fwd_persistent_kernel<32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk_lds_bytes()) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_fwd_persistent_kernel<32>, PK_NT, pk_lds_bytes()) == hipSuccess)
            r.blocks32 = n;
        n = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_fwd_persistent_kernel<64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk_lds_bytes(64)) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gru_fwd_persistent_kernel<64>, 1024, pk_lds_bytes(64)) == hipSuccess)
            r.blocks64 = n;
    }
    return r;
}
int pk_xcd_mode() {
    static int xcd = -1;
    if (xcd < 0) {
        const char* e = getenv("VQA_GRU_PERSIST_XCD");
        xcd = e ? atoi(e) : 0;
    }
    return xcd;
}
}  // namespace

// 1 when the persistent form applies to this shape on this device (all workgroups of the variant that would be
// launched co-resident), else 0
extern "C" int vqa_gru_fwd_persistent_supported(int T, int B, int H) {
    if (g_persist == 0) return 0;
    if (T <= 0 || B < 64 || H < 512 || H % 512 != 0) return 0;      // k tiles: a multiple of the prefetch depth
    const PkResidency& r = pk_residency();
    if (r.cus < 64) return 0;
    const int xcd = (pk_xcd_mode() && r.cus % 8 == 0) ? pk_xcd_mode() : 0;
    if (xcd == 2) return r.blocks64 >= 1 ? 1 : 0;       // grid = #CUs workgroups of 1024 threads
    if (xcd) return r.blocks32 >= 1 ? 1 : 0;            // grid = #CUs workgroups of 512 threads
    return r.blocks32 >= 2 ? 1 : 0;                     // grid = 2 x #CUs: both chains' workgroups on every CU
}

// Whole forward recurrence in one launch.  Same contract as vqa_gru_seq_fwd (xp [T,B,3H] read only, hs [T+1,B,H]
// with hs[0] given, tape r,u,c,rh [T,B,H]) plus `sync`: vqa_gru_persistent_sync_bytes() of device memory that the
// call zeroes itself; after the stream has run, a non-zero word at sync[32] reports a barrier time-out (results
// invalid).  Returns VQA_ERR_UNSUPPORTED when the shape / device does not qualify.
extern "C" int vqa_gru_seq_fwd_persistent(const float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len,
                                          float* hs, float* r, float* u, float* c, float* rh, int T, int B, int H,
                                          unsigned* sync, void* stream) {
    VQA_REQUIRE(xp && Wg_h && Wc_h && len && hs && r && u && c && rh && sync, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gru_fwd_persistent_supported(T, B, H) == 1, VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE((int64_t)(T + 1) * B * H * 4 < 0x7FFFFFF0ll && (int64_t)B * 3 * H * 4 < 0x7FFFFFF0ll, VQA_ERR_UNSUPPORTED);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(sync, 0, (size_t)vqa_gru_persistent_sync_bytes(), st) != hipSuccess) return VQA_ERR_LAUNCH;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    PkArgs a;
    a.xp = xp; a.Wg = Wg_h; a.Wc = Wc_h; a.len = len; a.hs = hs; a.r = r; a.u = u; a.c = c; a.rh = rh; a.sync = sync;
    a.T = T; a.B = B; a.H = H;
    a.chain_rows = ((B + 1) / 2 + PK_BM - 1) / PK_BM * PK_BM;
    a.xcd_mode = 0;
    a.err = sync + 32;
    {
        const int xcd = pk_xcd_mode();
        if (xcd && cus % 8 == 0) {       // experiment: eight XCD-local chains, one workgroup per CU
            const int bm = xcd == 2 ? 64 : PK_BM;      // 2: one 64-row tile per workgroup and phase (16 waves)
            a.xcd_mode = xcd == 2 ? 2 : 1;
            a.chain_rows = ((B + 7) / 8 + bm - 1) / bm * bm;
            a.err = sync + 192;
        }
    }
    a.spin_limit = 400000u;                       // ~0.5 s of polling before the launch gives up
    a.census = g_census;
    {
        static int delay = -1;
        if (delay < 0) {
            const char* e = getenv("VQA_GRU_PERSIST_DELAY");
            delay = e ? atoi(e) : 0;
        }
        a.chain1_delay = delay;
    }
    // one workgroup per chain and CU: every workgroup of the grid must be resident at the same time
    const int slots = cus;
    if (a.xcd_mode == 2)
        hipLaunchKernelGGL(gru_fwd_persistent_kernel<64>, dim3(slots), dim3(1024), pk_lds_bytes(64), st, a);
    else
        hipLaunchKernelGGL(gru_fwd_persistent_kernel<32>, dim3(a.xcd_mode ? slots : 2 * slots), dim3(PK_NT), pk_lds_bytes(), st, a);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
