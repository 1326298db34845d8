// f32 GEMM on the exact f32-input matrix core (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces tf.contrib.layers.fully_connected and its autodiff (reference
// vlmap/modules.py:635-641) and the GRUCell matmuls (vlmap/modules.py:129-135).
//
// Design (MI355X): 256-thread workgroups = 4 waves; block tile BM x BN x 16,
// wave tile WM x WN built from 32x32 MFMA tiles.  The f32 MFMA retires 64
// FLOP/clk/SIMD, i.e. one 32x32x2 instruction per 64 cycles, so LDS bandwidth is
// ~8x over-provisioned and the kernel is bound by the matrix pipe; what matters
// is (1) conflict-free fragment reads, (2) 16-byte global loads, (3) enough
// co-resident waves per SIMD to cover the global->LDS latency (LDS <= 40 KB and
// <= 128 VGPRs per block => 3-4 blocks per CU).
//
// k-permutation: one MFMA consumes 2 k's (lane half h = lane>>5 selects which).
// For an 8-wide k chunk we issue 4 MFMAs; MFMA j pairs k = 4h + j.  An operand
// whose k is contiguous in memory is staged as [row][16+4] and read with ONE
// ds_read_b128 per 4 MFMAs; an operand whose m/n is contiguous is staged as
// [k][BR] and read with conflict-free ds_read_b32.  Both sides use the same
// permutation, so every k is consumed exactly once.

#include "vqa_common.h"
#include "gemm_args.h"
#include <cstdio>
#include <cstdlib>

#ifndef VQA_GEMM_STAGGER
#define VQA_GEMM_STAGGER 1      // 0: build without the SIMD-partner stagger of the 8-wave kernels (A/B builds)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#if defined(VQA_DBG_STAMPS)   // timing experiment (tools/dbg/gru_stamps.py): thread 0 of every workgroup of a fused GRU-step
// launch records the 100 MHz real-time counter at five points; slot = epilogue kind
__device__ unsigned long long* g_dbg_stamps = nullptr;
#define DBG_STAMP(i) do { if (threadIdx.x == 0) dbg_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DBG_STAMP(i) do {} while (0)
#endif

namespace {

// A k-contiguous tile is staged as [row][BK + 4]: rows stay 16-B aligned and the 16 rows of a
// ds_read_b128 lane group land on 16 distinct 16-B bank groups for BK = 16, 32 and 64.

// XCD-aware tile order (MI355X: 8 XCDs, private L2s, workgroups dealt round-robin): workgroup ids
// that share an XCD (same id % 8) get a CONTIGUOUS range of logical tiles, n fastest, so the tiles
// that re-read one A row panel hit the same L2.  Bijective for any total.
__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

int g_force_order = -1;  // tuning override of the tile order (vqa_gemm_set_order): 0 n-fastest, 1 m-fastest

__device__ __forceinline__ float4 ld4_guard(const float* p, int n_ok, bool vec) {
    // n_ok = number of valid elements at p (<=0: none).  vec => p is 16-B aligned.
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_ok >= 4 && vec) {
        r = *reinterpret_cast<const float4*>(p);
    } else if (n_ok > 0) {
        r.x = p[0];
        if (n_ok > 1) r.y = p[1];
        if (n_ok > 2) r.z = p[2];
        if (n_ok > 3) r.w = p[3];
    }
    return r;
}

// Stage one BR x 16 (k-contiguous, KC) or 16 x BR (row-contiguous, RC) operand tile.
template <int BR, int BK, bool KC, int NT = 256>
struct Stager {
    static constexpr int KC_LD = BK + 4;
    static constexpr int KQ = BK / 4;                      // float4 per k-contiguous row
    static constexpr int NV = (BR * BK / 4 + NT - 1) / NT;  // float4 per thread
    float4 reg[NV];

    // g: operand base; ld: leading dim; r0: first row (m or n) of the tile; k0: first k;
    // RLIM: M or N; KLIM: end of this split's k range.
    __device__ __forceinline__ void load(const float* g, int ld, int r0, int k0, int RLIM, int KLIM, bool vec) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            if (KC) {
                const int row = idx / KQ, kq = (idx % KQ) * 4;
                const int gr = r0 + row, gk = k0 + kq;
                const bool in = (idx < BR * KQ) && (gr < RLIM);
                reg[i] = in ? ld4_guard(g + (int64_t)gr * ld + gk, KLIM - gk, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                constexpr int QPR = BR / 4;  // float4 per k row
                const int k = idx / QPR, rq = (idx % QPR) * 4;
                const int gk = k0 + k, gr = r0 + rq;
                const bool in = (idx < BK * QPR) && (gk < KLIM);
                reg[i] = in ? ld4_guard(g + (int64_t)gk * ld + gr, RLIM - gr, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // Fast form for the common case (16-byte aligned operands < 4 GiB, K % 4 == 0, and for a
    // row-contiguous operand M or N % 4 == 0): every float4 is wholly inside or wholly outside, so
    // the tile is fetched with BUFFER loads whose hardware range check returns zeros for the
    // out-of-range offset we substitute -- no branch and no select on the loaded value.  (Any
    // `in ? load : 0` form makes hipcc predicate the load under exec and wait vmcnt(0) right after
    // issuing it, i.e. the prefetch would not overlap the MFMA loop at all.)
    static constexpr unsigned OOB = 0xFFFFFFF0u;
    __device__ __forceinline__ static float4 bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    __device__ __forceinline__ void load_fast(__amdgpu_buffer_rsrc_t rs, int ld, int r0, int k0, int RLIM, int KLIM) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            if (KC) {
                const int row = idx / KQ, kq = (idx % KQ) * 4;
                const int gr = r0 + row, gk = k0 + kq;
                const bool in = (idx < BR * KQ) && (gr < RLIM) && (gk < KLIM);
                reg[i] = bload(rs, in ? (unsigned)(((int64_t)gr * ld + gk) * 4) : OOB);
            } else {
                constexpr int QPR = BR / 4;
                const int k = idx / QPR, rq = (idx % QPR) * 4;
                const int gk = k0 + k, gr = r0 + rq;
                const bool in = (idx < BK * QPR) && (gk < KLIM) && (gr < RLIM);
                reg[i] = bload(rs, in ? (unsigned)(((int64_t)gk * ld + gr) * 4) : OOB);
            }
        }
    }
    // Steady-state form for tiles that lie wholly inside the k range: the per-lane byte offset is
    // computed ONCE (rows past M / N are clamped to a valid row -- their accumulators are never
    // stored) and the k advance rides in the buffer load's SCALAR offset, so a tile's loads cost no
    // vector ALU work at all; what VALU the loop keeps competes with the MFMAs for the issue port.
    static constexpr int COUNT = BR * BK / 4;
    static constexpr bool EXACT = (NV * NT == COUNT);
    unsigned voff[NV];
    __device__ __forceinline__ void init_full(int ld, int r0, int kbeg, int RLIM) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = EXACT ? (int)threadIdx.x + i * NT : min((int)threadIdx.x + i * NT, COUNT - 1);
            if (KC) {
                const int row = idx / KQ, kq = (idx % KQ) * 4;
                const int gr = min(r0 + row, RLIM - 1);
                voff[i] = (unsigned)(((int64_t)gr * ld + kbeg + kq) * 4);
            } else {
                constexpr int QPR = BR / 4;
                const int k = idx / QPR, rq = (idx % QPR) * 4;
                const int gr = (r0 + rq < RLIM) ? r0 + rq : 0;
                voff[i] = (unsigned)(((int64_t)(kbeg + k) * ld + gr) * 4);
            }
        }
    }
    // row-gathered operand: every lane carries the 64-bit address of its row piece (rows of one tile come from
    // anywhere in a table that may exceed a buffer descriptor's 4 GiB), the k advance is a uniform element offset
    // (The load goes through a native vector type: assigning HIP's float4 struct from a dereferenced pointer keeps
    // the whole stager object in scratch memory -- hipcc does not scalarise that copy.)
    __device__ __forceinline__ void load_ptr(const float* const (&src)[NV], int koff) {
        typedef float f32x4n __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4n v = *reinterpret_cast<const f32x4n*>(src[i] + koff);
            reg[i] = make_float4(v.x, v.y, v.z, v.w);
        }
    }
    // by-product copy of the staged tile: buffer stores, so lanes that must not write (rows past M) carry an
    // out-of-range offset and the hardware drops them -- a per-lane `if` around a store costs branches and waits
    __device__ __forceinline__ void store_buf(__amdgpu_buffer_rsrc_t rs, const unsigned (&off)[NV], unsigned soff) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            u32x4 v;
            v.x = __float_as_uint(reg[i].x); v.y = __float_as_uint(reg[i].y);
            v.z = __float_as_uint(reg[i].z); v.w = __float_as_uint(reg[i].w);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off[i], soff, 0);
        }
    }
    __device__ __forceinline__ void load_full(__amdgpu_buffer_rsrc_t rs, unsigned soff) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0);
            reg[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    }
    // implicit im2col: one k tile lies inside ONE filter tap (Ci % BK == 0), so the tap offset is
    // wave-uniform and each lane only adds it to its pre-decoded pixel coordinates.
    __device__ __forceinline__ void load_conv(__amdgpu_buffer_rsrc_t rs, const int (&iy0)[NV], const int (&ix0)[NV],
                                              const int (&pix)[NV], int k0, int Hi, int Wi, int Ci, int kw) {
        if (Ci >= BK) {
            const int tap = k0 / Ci, c0 = k0 - tap * Ci;
            const int ky = tap / kw, kx = tap - ky * kw;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int kq = (idx % KQ) * 4;
                const int iy = iy0[i] + ky, ix = ix0[i] + kx;
                const bool in = (pix[i] >= 0) && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
                reg[i] = bload(rs, in ? (unsigned)((((int64_t)(pix[i] + iy * Wi + ix)) * Ci + c0 + kq) * 4) : OOB);
            }
        } else {
            // few channels (conv1 on 4-channel pixels): a k tile spans BK / Ci filter taps, so the tap -- and the
            // in-image test -- is per lane; Ci % 4 == 0 keeps every 16-byte load inside one tap
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int k = k0 + (idx % KQ) * 4;
                const int tap = k / Ci, c = k - tap * Ci;
                const int ky = tap / kw, kx = tap - ky * kw;
                const int iy = iy0[i] + ky, ix = ix0[i] + kx;
                const bool in = (pix[i] >= 0) && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
                reg[i] = bload(rs, in ? (unsigned)((((int64_t)(pix[i] + iy * Wi + ix)) * Ci + c) * 4) : OOB);
            }
        }
    }
    // implicit im2col, steady-state form: per-lane pixel base and a bit mask of the filter taps that
    // fall inside the image are computed once per output tile; a k tile then costs one AND, one
    // ADD and one select per load (the tap's byte offset and bit are wave-uniform scalars).
    __device__ __forceinline__ void load_conv_full(__amdgpu_buffer_rsrc_t rs, const unsigned (&cbase)[NV],
                                                   const unsigned (&cmask)[NV], unsigned tap_off, unsigned tap_bit) {
#pragma unroll
        for (int i = 0; i < NV; ++i) reg[i] = bload(rs, (cmask[i] & tap_bit) ? cbase[i] + tap_off : OOB);
    }
    __device__ __forceinline__ void store(float* s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            if (KC) {
                if (EXACT || idx < BR * KQ) {
                    const int row = idx / KQ, kq = (idx % KQ) * 4;
                    *reinterpret_cast<float4*>(s + row * KC_LD + kq) = reg[i];
                }
            } else {
                constexpr int QPR = BR / 4;
                if (EXACT || idx < BK * QPR) {
                    const int k = idx / QPR, rq = (idx % QPR) * 4;
                    *reinterpret_cast<float4*>(s + k * BR + rq) = reg[i];
                }
            }
        }
    }
};

template <int BR, int BK, bool KC>
constexpr int tile_floats() { return KC ? BR * (BK + 4) : BK * BR; }

// fragment of 4 consecutive MFMA k-steps (8-wide chunk c of the BK-deep tile) for the 32 rows at r0
template <int BR, int BK, bool KC>
__device__ __forceinline__ float4 frag4(const float* s, int r0, int c, int lane) {
    const int i = lane & 31, h = lane >> 5;
    if (KC) {
        return *reinterpret_cast<const float4*>(s + (r0 + i) * (BK + 4) + c * 8 + h * 4);
    } else {
        const float* p = s + (c * 8 + h * 4) * BR + r0 + i;
        return make_float4(p[0], p[BR], p[2 * BR], p[3 * BR]);
    }
}

template <int BM, int BN, int WM, int WN, int WGK, int BK, int DEEP, bool A_KC, bool B_KC, int EPI, bool CONV = false,
          bool EDGE = false, int NT = 256, bool GATHER = false>
// (the 4-wave implicit-GEMM form lands on 113 + 16 registers, one allocation granule above four waves per SIMD: ask for
// four, i.e. four co-resident workgroups per CU instead of three -- the compiler gets there without spilling, and the
// extractor gains 1.5 %.  The same request for the 64x64 two-tile-prefetch form (121 + 16) measured 0.5 % SLOWER
// on the pre-training step, same box, two builds: left alone.)
__global__ __launch_bounds__(NT, (CONV && NT == 256) ? 4 : 1) void gemm_f32_kernel(GemmArgs p, EpiArgs ep) {
    static_assert(!GATHER || (A_KC && DEEP == 0 && !CONV && !EDGE && EPI == EPI_PLAIN), "row-gathered A: plain NN/NT tiles");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
    static_assert(WAVES_M * WAVES_N * WGK * 64 == NT, "one 32x32-tiled wave per (m, n, k-group) slot");
    static_assert((BK / 8) % WGK == 0, "k chunks split evenly over the k wave groups");
    constexpr int A_FL = tile_floats<BM, BK, A_KC>();
    constexpr int B_FL = tile_floats<BN, BK, B_KC>();
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x (A tile | B tile)

    // The fused GRU-step kernels sit on the step's critical path and run beside a big GEMM on the
    // side stream: raise their wave priority so they win MFMA/VALU arbitration on a shared SIMD.
    // (Staggering the priorities of co-resident workgroups, to stop their waves convoying into the
    // same barrier, was measured and changes nothing: the MFMA arbiter does not follow s_setprio.)
    if (EPI != EPI_PLAIN) __builtin_amdgcn_s_setprio(3);
#if defined(VQA_DBG_STAMPS)
    unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x == 0) dbg_t[5] = __builtin_amdgcn_s_memtime();
#endif
    DBG_STAMP(0);                                            // start
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wk = wave / (WAVES_M * WAVES_N);
    const int wm = (wave / WAVES_N) % WAVES_M, wn = wave % WAVES_N;
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int total = tiles_mn * p.nsplit;
    // persistent walk: a launch limited to fewer workgroups than tiles (side-stream GEMMs keep
    // to one workgroup per CU so the latency-bound kernels beside them still find LDS/waves)
    for (int lin = blockIdx.x; lin < total; lin += gridDim.x) {
    const int tile = xcd_remap(lin, total);
    const int bz = tile / tiles_mn, bmn = tile - bz * tiles_mn;
    const int m0 = (p.m_fastest ? bmn % p.tiles_m : bmn / p.tiles_n) * BM;
    const int n0 = (p.m_fastest ? bmn / p.tiles_m : bmn % p.tiles_n) * BN;
    const int kbeg = bz * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const bool va = p.vecA, vb = p.vecB;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    // fragments are double-buffered in registers: chunk cc+1's LDS reads are issued before chunk
    // cc's MFMAs, so a wave's own reads are covered by its own MFMAs
    auto compute_tile = [&](const float* As, const float* Bs) {
        constexpr int NC = BK / 8 / WGK;
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = frag4<BM, BK, A_KC>(As, wm * WM + a * 32, wk, lane);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = frag4<BN, BK, B_KC>(Bs, wn * WN + b * 32, wk, lane);
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int cur = cc & 1, nx = cur ^ 1;
            if (cc + 1 < NC) {
                const int c = (cc + 1) * WGK + wk;   // the k wave groups interleave over the 8-wide chunks
#pragma unroll
                for (int a = 0; a < TM; ++a) af[nx][a] = frag4<BM, BK, A_KC>(As, wm * WM + a * 32, c, lane);
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[nx][b] = frag4<BN, BK, B_KC>(Bs, wn * WN + b * 32, c, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
#if defined(VQA_DBG_NOMFMA)   // timing experiment only: the loop without its matrix instructions (fragments still read and used)
                    acc[a][b][0] += af[cur][a].x * bf[cur][b].x + af[cur][a].y * bf[cur][b].y +
                                    af[cur][a].z * bf[cur][b].z + af[cur][a].w * bf[cur][b].w;
#else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a].x, bf[cur][b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a].y, bf[cur][b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a].z, bf[cur][b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a].w, bf[cur][b].w, acc[a][b], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Fused epilogues: the side inputs (x-projection addend, previous state, gates ...) do not depend on
    // this kernel's accumulators, so the k-group-0 waves fetch them BEFORE the k loop; their HBM
    // latency (~2 us for the [T,B,3H] x-projection, written long before) hides under the loop
    // instead of sitting between the last MFMA and the first store.
    static_assert(EPI == EPI_PLAIN || (TM == 1 && TN == 1), "fused epilogues: one 32x32 tile per wave");
    // ... and every k group finishes ITS share of the 32 rows of its (m, n) sub-tile (rows wk*RPG .. +RPG): the
    // partial tiles of all groups meet in LDS and each wave sums, activates and stores RPG rows, instead of group
    // 0 doing all 32 while the other waves of the workgroup have already retired.
    constexpr int RPG = (EPI != EPI_PLAIN) ? 32 / WGK : 32;   // rows per k group
    constexpr int NI = RPG >= 8 ? RPG / 8 : 1;                // 8-row slabs per wave (a lane owns 4 columns of a row)
    const int e_lr = lane >> 3;
    const bool e_act = (RPG >= 8) || (e_lr < RPG);
    const int e_row0 = wk * RPG + e_lr;                       // row inside the 32x32 sub-tile (+ 8 i)
    float4 e_d[NI], e_x0[NI], e_x1[NI], e_x2[NI];
    int e_lim[NI];
    if (EPI != EPI_PLAIN && e_act) {
        const int gcol = n0 + wn * WN + (lane & 7) * 4;
        const int grow0 = m0 + wm * WM + e_row0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int grow = grow0 + 8 * i;
            e_d[i] = e_x0[i] = e_x1[i] = e_x2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            e_lim[i] = 0;
            if (!((grow < p.M) && (gcol < p.N))) continue;
            if (p.D != nullptr) e_d[i] = *reinterpret_cast<const float4*>(p.D + (int64_t)grow * p.ldd + gcol);
            const int64_t o = (int64_t)grow * ep.H + gcol;
            if (EPI == EPI_GATES) {
                if (gcol < ep.H) e_x0[i] = *reinterpret_cast<const float4*>(ep.h_prev + o);
            } else if (EPI == EPI_CAND) {
                e_x0[i] = *reinterpret_cast<const float4*>(ep.h_prev + o);
                e_x1[i] = *reinterpret_cast<const float4*>(ep.i0 + o);
                e_lim[i] = ep.len[grow];
            } else if (EPI == EPI_BWD_RH) {
                e_x0[i] = *reinterpret_cast<const float4*>(ep.h_prev + o);
                e_x1[i] = *reinterpret_cast<const float4*>(ep.i0 + o);
                e_x2[i] = *reinterpret_cast<const float4*>(ep.o1 + o);
            } else {
                e_x0[i] = *reinterpret_cast<const float4*>(ep.h_prev + o);
                e_x1[i] = *reinterpret_cast<const float4*>(ep.i0 + o);
                e_x2[i] = *reinterpret_cast<const float4*>(ep.i1 + o);
                e_lim[i] = ep.len[grow];
            }
        }
    }
    float* L0 = smem;
    float* L1 = smem + (A_FL + B_FL);

    if (DEEP == 0) {
        // one tile of register prefetch: enough when >= 2-3 workgroups share a CU
        Stager<BM, BK, A_KC, NT> sa;
        Stager<BN, BK, B_KC, NT> sb;
        constexpr int NVA = Stager<BM, BK, A_KC, NT>::NV;
        int iy0[NVA], ix0[NVA], pix[NVA];
        if (CONV) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int m = m0 + idx / (BK / 4);
                pix[i] = -1; iy0[i] = ix0[i] = 0;
                if (idx < BM * (BK / 4) && m < p.M) {
                    const int hw = p.Ho * p.Wo;
                    const int b = m / hw, r = m - b * hw;
                    const int oy = r / p.Wo, ox = r - oy * p.Wo;
                    iy0[i] = oy * p.cstride - p.pad_t;
                    ix0[i] = ox * p.cstride - p.pad_l;
                    pix[i] = b * p.Hi * p.Wi;
                }
            }
        }
        const float* gsrc[NVA];     // dead (and optimised away) unless GATHER
        unsigned gdst[NVA];
        const bool g_store = GATHER && p.g_out != nullptr && n0 == 0;      // workgroup-uniform
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(
            p.g_out, 0, GATHER && p.g_out != nullptr ? (int)(((int64_t)(p.M - 1) * p.g_ldo + p.K) * 4) : 0, 0x00020000);
        if (GATHER) {      // host guarantees K % BK == 0 and no split-k: every tile is full
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
                const int m = m0 + row, mc = min(m, p.M - 1);
                const int smp = mc / p.g_R, reg_row = mc - smp * p.g_R;
                int64_t src = p.g_idx[smp];
                src = src < 0 ? 0 : (src >= p.g_N ? p.g_N - 1 : src);        // np.take would raise; clamp like vqa_gather_features
                gsrc[i] = p.A + (src * p.g_R + reg_row) * (int64_t)p.lda + kq;
                gdst[i] = (m < p.M) ? (unsigned)(((int64_t)m * p.g_ldo + kq) * 4) : Stager<BM, BK, A_KC, NT>::OOB;
            }
        }
        auto load_a = [&](int k0) {
            if (GATHER) sa.load_ptr(gsrc, k0 - kbeg);
            else if (CONV) sa.load_conv(rsA, iy0, ix0, pix, k0, p.Hi, p.Wi, p.Ci, p.kw);
            else if (EDGE) sa.load(p.A, p.lda, m0, k0, p.M, kend, va);
            else sa.load_fast(rsA, p.lda, m0, k0, p.M, kend);
        };
        auto load_b = [&](int k0) {
            if (EDGE) sb.load(p.B, p.ldb, n0, k0, p.N, kend, vb);
            else sb.load_fast(rsB, p.ldb, n0, k0, p.N, kend);
        };
        if (nt > 0) {
            load_a(kbeg);
            load_b(kbeg);
            sa.store(L0);
            sb.store(L0 + A_FL);
            if (g_store) sa.store_buf(rsG, gdst, 0u);
        }
        __syncthreads();
        int t = 0;
        if (CONV && p.conv_taps <= 32 && p.Ci % BK == 0) {
            // every k tile of a convolution is full (Ci % BK == 0): same two-per-trip loop as below,
            // the A tile through the tap-mask loader, the filter matrix through scalar offsets
            unsigned cbase[NVA], cmask[NVA];
            const int kh = p.conv_taps / p.kw;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int kq = (idx % (BK / 4)) * 4;
                cbase[i] = (unsigned)(((int64_t)(pix[i] + iy0[i] * p.Wi + ix0[i]) * p.Ci + kq) * 4);
                unsigned mk = 0;
                if (pix[i] >= 0) {
                    for (int ky = 0; ky < kh; ++ky)
                        for (int kx = 0; kx < p.kw; ++kx)
                            if ((unsigned)(iy0[i] + ky) < (unsigned)p.Hi && (unsigned)(ix0[i] + kx) < (unsigned)p.Wi)
                                mk |= 1u << (ky * p.kw + kx);
                }
                cmask[i] = mk;
            }
            sb.init_full(p.ldb, n0, kbeg, p.N);
            const unsigned stepB = (unsigned)(BK * p.ldb) * 4u;
            unsigned ob = stepB;
            int c0 = BK, ky = 0, kx = 0, tap = 0;     // position of tile t + 1 inside the (ky, kx, ci) k axis
            auto advance = [&]() {
                c0 += BK;
                if (c0 >= p.Ci) { c0 = 0; ++tap; if (++kx == p.kw) { kx = 0; ++ky; } }
            };
            if (c0 >= p.Ci) { c0 = 0; tap = 1; if (++kx == p.kw) { kx = 0; ++ky; } }
            for (; t + 2 < nt; t += 2) {
                sa.load_conv_full(rsA, cbase, cmask, (unsigned)(((ky * p.Wi + kx) * p.Ci + c0) * 4), 1u << tap);
                sb.load_full(rsB, ob);
                ob += stepB; advance();
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                sa.store(L1);
                sb.store(L1 + A_FL);
                __syncthreads();
                sa.load_conv_full(rsA, cbase, cmask, (unsigned)(((ky * p.Wi + kx) * p.Ci + c0) * 4), 1u << tap);
                sb.load_full(rsB, ob);
                ob += stepB; advance();
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L1, L1 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                sa.store(L0);
                sb.store(L0 + A_FL);
                __syncthreads();
            }
        }
        if (!CONV && !EDGE) {
            // full tiles, two per trip so both LDS buffers are compile-time addresses
            sa.init_full(p.lda, m0, kbeg, p.M);
            sb.init_full(p.ldb, n0, kbeg, p.N);
            const unsigned stepA = (unsigned)(A_KC ? BK : BK * p.lda) * 4u;
            const unsigned stepB = (unsigned)(B_KC ? BK : BK * p.ldb) * 4u;
            const int nfull = (kend - kbeg) / BK;
            unsigned oa = stepA, ob = stepB;   // scalar byte offsets of tile t + 1
            // Stagger (8-wave workgroups: waves w and w + 4 share a SIMD and, running the same program with one
            // barrier per k tile, reach their MFMA burst, their LDS traffic and the barrier together).  Inside one
            // barrier interval "compute tile t from one LDS buffer" and "store tile t+1 into the other" commute, so
            // waves 4..7 run the interval as store(t+1) -> fetch(t+2) -> compute(t) while waves 0..3 keep
            // fetch(t+1) -> compute(t) -> store(t+1): a SIMD's two waves of this workgroup alternate between the
            // MFMA burst and the LDS/global phase instead of colliding in both.  Same barrier count on both paths;
            // one entry fetch and one transition interval bring waves 4..7 in and out of the half-interval lead.
            // Compiled in only where it pays and costs no occupancy: the 128x64 / 64x128 tiles of 32x32 waves with BK 32
            // (95 -> 99 registers, still two workgroups per CU) and the 128x128 BK-16 weight-gradient tile; in the
            // 128x128 BK-32 kernel the second loop body costs 6 registers and with them its second workgroup per CU
            // (124 -> 130: the extractor's 1x1 layers lost 1.3 % until this was fenced off).
            constexpr bool STAG = (NT == 512) && !GATHER && EPI == EPI_PLAIN && (VQA_GEMM_STAGGER != 0) &&
                                  ((WM == 32 && WN == 32 && BK == 32 && BM + BN == 192) || (BM == 128 && BN == 128 && BK == 16));
            // Long k loops only: on the roofline GEMM (64 tiles) the k loop gets 2 % shorter (552 -> 541 us, same
            // box, two builds); on the extractor's short-k 1x1 layers (4..8 tiles) the entry fetch and the transition
            // interval cost more than the stagger returns (2064 -> 2043 imgs/s).  profiles/r2_stagger_ab.txt
            const bool stag_wg = STAG && nfull >= 16;                 // workgroup-uniform
            if (stag_wg && wave >= 4) {
                // (a static s_setprio 1 for this second-dispatched half, the guide's companion rule, made the roofline
                // GEMM 2 % SLOWER here: 563 -> 574 us, same box, two builds)
                sa.load_full(rsA, oa); sb.load_full(rsB, ob);         // tile 1 into the staging registers
                oa += stepA; ob += stepB;
                for (; t + 3 < nfull; t += 2) {
                    sa.store(L1); sb.store(L1 + A_FL);                // tile t + 1
                    sa.load_full(rsA, oa); sb.load_full(rsB, ob);     // tile t + 2
                    oa += stepA; ob += stepB;
                    __builtin_amdgcn_sched_barrier(0);
                    compute_tile(L0, L0 + A_FL);
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                    sa.store(L0); sb.store(L0 + A_FL);                // tile t + 2
                    sa.load_full(rsA, oa); sb.load_full(rsB, ob);     // tile t + 3
                    oa += stepA; ob += stepB;
                    __builtin_amdgcn_sched_barrier(0);
                    compute_tile(L1, L1 + A_FL);
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
                sa.store(L1); sb.store(L1 + A_FL);                    // transition: tile t + 1 (< nfull) is in the registers
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                ++t;
            } else if (stag_wg) {
                for (; t + 3 < nfull; t += 2) {                       // the same trip count as the waves above
                    sa.load_full(rsA, oa); sb.load_full(rsB, ob);
                    oa += stepA; ob += stepB;
                    __builtin_amdgcn_sched_barrier(0);
                    compute_tile(L0, L0 + A_FL);
                    __builtin_amdgcn_sched_barrier(0);
                    sa.store(L1); sb.store(L1 + A_FL);
                    __syncthreads();
                    sa.load_full(rsA, oa); sb.load_full(rsB, ob);
                    oa += stepA; ob += stepB;
                    __builtin_amdgcn_sched_barrier(0);
                    compute_tile(L1, L1 + A_FL);
                    __builtin_amdgcn_sched_barrier(0);
                    sa.store(L0); sb.store(L0 + A_FL);
                    __syncthreads();
                }
                sa.load_full(rsA, oa); sb.load_full(rsB, ob);         // transition interval
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                sa.store(L1); sb.store(L1 + A_FL);
                __syncthreads();
                ++t;
            }
            for (; !stag_wg && t + 2 < nfull; t += 2) {
                if (GATHER) sa.load_ptr(gsrc, (t + 1) * BK); else sa.load_full(rsA, oa);
                sb.load_full(rsB, ob);
                oa += stepA; ob += stepB;
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                sa.store(L1);
                sb.store(L1 + A_FL);
                if (g_store) sa.store_buf(rsG, gdst, (unsigned)((t + 1) * BK) * 4u);
                __syncthreads();
                if (GATHER) sa.load_ptr(gsrc, (t + 2) * BK); else sa.load_full(rsA, oa);
                sb.load_full(rsB, ob);
                oa += stepA; ob += stepB;
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L1, L1 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                sa.store(L0);
                sb.store(L0 + A_FL);
                if (g_store) sa.store_buf(rsG, gdst, (unsigned)((t + 2) * BK) * 4u);
                __syncthreads();
            }
        }
        // remaining tiles (all of them for the conv / edge loaders) without conditionals, the last
        // one peeled: loads of tile t+1 are issued before tile t's MFMA loop and consumed after it
        for (; t + 1 < nt; ++t) {
            float* cur = (t & 1) ? L1 : L0;
            float* nxt = (t & 1) ? L0 : L1;
            load_a(kbeg + (t + 1) * BK);
            load_b(kbeg + (t + 1) * BK);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMA loop (hipcc sinks it otherwise)
            compute_tile(cur, cur + A_FL);
            __builtin_amdgcn_sched_barrier(0);
            sa.store(nxt);
            sb.store(nxt + A_FL);
            if (g_store) sa.store_buf(rsG, gdst, (unsigned)((t + 1) * BK) * 4u);
            __syncthreads();
        }
        if (nt > 0) {
            float* cur = ((nt - 1) & 1) ? L1 : L0;
            compute_tile(cur, cur + A_FL);
            __syncthreads();
        }
    } else {
        // two tiles of register prefetch (two stager sets, loop unrolled by 2 so every register
        // index is static): with one workgroup per CU a tile's MFMA time (~0.5 us) is shorter than
        // the global-load latency, so the loads need two phases of cover.
        Stager<BM, BK, A_KC, NT> sa0, sa1;
        Stager<BN, BK, B_KC, NT> sb0, sb1;
        // (Measured and rejected: rotating each workgroup's k-tile order so that co-running workgroups stream
        // different k slices -- the recurrence got 4 % / 7 % SLOWER; workgroups that walk k in lockstep share the
        // weight slices of their column panel in L2.  profiles/r2_gru_krot*.txt)
        auto ld0 = [&](int tile) {
            sa0.load_fast(rsA, p.lda, m0, kbeg + tile * BK, p.M, kend);
            sb0.load_fast(rsB, p.ldb, n0, kbeg + tile * BK, p.N, kend);
        };
        auto ld1 = [&](int tile) {
            sa1.load_fast(rsA, p.lda, m0, kbeg + tile * BK, p.M, kend);
            sb1.load_fast(rsB, p.ldb, n0, kbeg + tile * BK, p.N, kend);
        };
        auto st0 = [&]() { sa0.store(L0); sb0.store(L0 + A_FL); };
        auto st1 = [&]() { sa1.store(L1); sb1.store(L1 + A_FL); };
        if (nt > 0) ld0(0);
        if (nt > 1) ld1(1);
        if (nt > 0) st0();
        __syncthreads();
        DBG_STAMP(1);                                        // first tile in LDS
        int t = 0;
        // steady state, unrolled by 2 so every register index is static and free of conditionals:
        // entering a pair, L0 holds tile t and set 1 holds tile t+1 (in flight)
        {   // full tiles: fixed per-lane offsets, the k advance in the loads' scalar offset
            sa0.init_full(p.lda, m0, kbeg, p.M); sa1.init_full(p.lda, m0, kbeg, p.M);
            sb0.init_full(p.ldb, n0, kbeg, p.N); sb1.init_full(p.ldb, n0, kbeg, p.N);
#if defined(VQA_DBG_KFREEZE)   // timing experiment only (tools/gru_tune.py, a second build): every k tile re-reads tile 0,
            // so the loop runs with its operands hot in L1 / L2 -- what is left is the tile loop's own structure
            const unsigned stepA = 0, stepB = 0;
#else
            const unsigned stepA = (unsigned)(A_KC ? BK : BK * p.lda) * 4u;
            const unsigned stepB = (unsigned)(B_KC ? BK : BK * p.ldb) * 4u;
#endif
            const int nfull = (kend - kbeg) / BK;
            unsigned oa = 2 * stepA, ob = 2 * stepB;   // scalar byte offsets of tile t + 2
            // (The SIMD-partner stagger of the one-tile-prefetch loop above was tried here as well -- upper half of the
            // waves: fetch(t+2) -> store(t+1) -> compute(t) -- and makes the recurrence SLOWER, 523 -> 556 us forward,
            // 488 -> 539 us backward: storing tile t+1 at the top of the interval leaves its load one phase of cover
            // instead of two, which is what this loop exists for.)
#if defined(VQA_DBG_NOSTAGE)   // timing experiment only: the MFMA loop alone (no global loads, no LDS refill, no barrier)
            for (; t + 3 < nfull; t += 2) {
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L1, L1 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
            }
#elif defined(VQA_DBG_NOBARRIER)   // timing experiment only: loads and LDS refill kept, the workgroup barrier dropped (races)
            for (; t + 3 < nfull; t += 2) {
                sa0.load_full(rsA, oa); sb0.load_full(rsB, ob);
                oa += stepA; ob += stepB;
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                st1();
                sa1.load_full(rsA, oa); sb1.load_full(rsB, ob);
                oa += stepA; ob += stepB;
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L1, L1 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                st0();
            }
#else
            for (; t + 3 < nfull; t += 2) {
                sa0.load_full(rsA, oa); sb0.load_full(rsB, ob);
                oa += stepA; ob += stepB;
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L0, L0 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                st1();
                __syncthreads();
                sa1.load_full(rsA, oa); sb1.load_full(rsB, ob);
                oa += stepA; ob += stepB;
                __builtin_amdgcn_sched_barrier(0);
                compute_tile(L1, L1 + A_FL);
                __builtin_amdgcn_sched_barrier(0);
                st0();
                __syncthreads();
            }
#endif
        }
        for (; t + 3 < nt; t += 2) {
            ld0(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(L0, L0 + A_FL);
            __builtin_amdgcn_sched_barrier(0);
            st1();
            __syncthreads();
            ld1(t + 3);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(L1, L1 + A_FL);
            __builtin_amdgcn_sched_barrier(0);
            st0();
            __syncthreads();
        }
        const int rem = nt - t;   // 0..3 tiles left: L0 = tile t, set 1 = tile t+1
        if (rem >= 3) {
            ld0(t + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rem >= 1) compute_tile(L0, L0 + A_FL);
        if (rem >= 2) {
            st1();
            __syncthreads();
            compute_tile(L1, L1 + A_FL);
        }
        if (rem >= 3) {
            st0();
            __syncthreads();
            compute_tile(L0, L0 + A_FL);
        }
        __syncthreads();
    }

    if (WGK > 1 && EPI == EPI_PLAIN) {
        // in-block split-k: groups 1..WGK-1 park their partial tiles in LDS (the operand tiles are
        // dead after the loop's last barrier), group 0 sums them.  Lane-contiguous => conflict-free.
        constexpr int PER_WAVE = TM * TN * 16 * 64;
        float* red = smem;
        if (wk > 0) {
            float* dst = red + ((wk - 1) * (WAVES_M * WAVES_N) + wm * WAVES_N + wn) * PER_WAVE + lane;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((a * TN + b) * 16 + r) * 64] = acc[a][b][r];
        }
        __syncthreads();
        if (wk > 0) return;   // fused-epilogue / in-block split-k launches are never persistent
#pragma unroll
        for (int g = 1; g < WGK; ++g) {
            const float* src = red + ((g - 1) * (WAVES_M * WAVES_N) + wm * WAVES_N + wn) * PER_WAVE + lane;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += src[((a * TN + b) * 16 + r) * 64];
        }
    }

    // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (EPI == EPI_PLAIN) {
        // All global loads of a tile are issued BEFORE its first store: D may alias C (in-place
        // accumulate), so interleaving them would serialise 16 round trips.
        float* Cz = p.C + (int64_t)bz * p.slab_stride;
        const bool first = (bz == 0);
        if (p.vec_epi) {
            // 16-byte form (N, the leading dimensions and the pointers allow it): the accumulator
            // tile goes through the wave's own 32 x 36 LDS patch (the operand tiles are dead after
            // the loop's last barrier) so that every lane owns 4 consecutive columns of a row --
            // addend loads and stores are line-contiguous float4 instead of 16 dword accesses with
            // their address arithmetic; short-k GEMMs (1x1 convolutions) live in this epilogue.
            constexpr int SLD = 36;
            constexpr int RED0 = (WGK - 1) * WAVES_M * WAVES_N * TM * TN * 16 * 64;
            float* stg = smem + RED0 + (wm * WAVES_N + wn) * 32 * SLD;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + (lane & 31)] = acc[a][b][r];
                    const int c4 = (lane & 7) * 4;
                    const int gcol = n0 + wn * WN + b * 32 + c4;
                    const int grow0 = m0 + wm * WM + a * 32 + (lane >> 3);
                    if (gcol < p.N) {
                        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
                        if (p.bias != nullptr && first) bv = *reinterpret_cast<const float4*>(p.bias + gcol);
                        if (p.scale != nullptr) sv = *reinterpret_cast<const float4*>(p.scale + gcol);
                        float4 dv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int grow = grow0 + 8 * i;
                            dv[i] = (p.D != nullptr && first && grow < p.M)
                                        ? *reinterpret_cast<const float4*>(p.D + (int64_t)grow * p.ldd + gcol)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int grow = grow0 + 8 * i;
                            const float4 v = *reinterpret_cast<const float4*>(stg + ((lane >> 3) + 8 * i) * SLD + c4);
                            float4 o = make_float4(v.x * sv.x + bv.x + dv[i].x, v.y * sv.y + bv.y + dv[i].y,
                                                   v.z * sv.z + bv.z + dv[i].z, v.w * sv.w + bv.w + dv[i].w);
                            if (p.relu) {
                                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                            }
                            if (grow < p.M) *reinterpret_cast<float4*>(Cz + (int64_t)grow * p.ldc + gcol) = o;
                        }
                    }
                }
            if (lin + (int)gridDim.x < total) __syncthreads();   // persistent walk: the patches overlap the next tile's operands
            continue;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn * WN + b * 32 + (lane & 31);
                const int rbase = m0 + wm * WM + a * 32 + 4 * (lane >> 5);
                if (col >= p.N) continue;
                const float bv = (p.bias != nullptr && first) ? p.bias[col] : 0.f;
                const float sv = (p.scale != nullptr) ? p.scale[col] : 1.f;
                float dv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    dv[r] = (p.D != nullptr && first && row < p.M) ? p.D[(int64_t)row * p.ldd + col] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) {
                        float v = acc[a][b][r] * sv + bv + dv[r];
                        if (p.relu) v = fmaxf(v, 0.f);
                        Cz[(int64_t)row * p.ldc + col] = v;
                    }
                }
            }
        continue;   // next tile (the loop's last barrier already fenced the LDS tiles)
    }

    DBG_STAMP(2);                                            // k loop done
    // Fused GRU epilogues: every wave parks its accumulator tile, TRANSPOSED, in its own 32 x 36 LDS patch
    // ([k group][sub-tile]; the operand tiles are dead after the loop's last barrier), so a lane owns 4 consecutive
    // columns of a row: side inputs and outputs move as 16-byte, line-contiguous accesses.  After one barrier
    // wave (wk, wm, wn) sums the WGK partial patches of ITS rows in group order (deterministic) and finishes them.
    constexpr int STG_LD = 36;
    constexpr int SUBS = WAVES_M * WAVES_N;
    const int sub = wm * WAVES_N + wn;
    float* stg = smem + (wk * SUBS + sub) * 32 * STG_LD;
    const int H = ep.H;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * STG_LD + (lane & 31)] = acc[0][0][r];
    if (WGK > 1) __syncthreads();
    DBG_STAMP(3);                                            // partial tiles exchanged
    if (e_act) {
        const int c4 = (lane & 7) * 4;
        const int gcol = n0 + wn * WN + c4;
        const int grow0 = m0 + wm * WM + e_row0;
        float4 v[NI];
        bool ok[NI];
        const float4 (&d)[NI] = e_d;
        const float4 (&x0)[NI] = e_x0;
        const float4 (&x1)[NI] = e_x1;
        const float4 (&x2)[NI] = e_x2;
        const int (&lim)[NI] = e_lim;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int grow = grow0 + 8 * i;
            ok[i] = (grow < p.M) && (gcol < p.N);
            const float* src = smem + sub * 32 * STG_LD + (e_row0 + 8 * i) * STG_LD + c4;
            float4 t = *reinterpret_cast<const float4*>(src);
#pragma unroll
            for (int g = 1; g < WGK; ++g) {
                const float4 q = *reinterpret_cast<const float4*>(src + g * SUBS * 32 * STG_LD);
                t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
            }
            v[i] = t;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (!ok[i]) continue;
            const int grow = grow0 + 8 * i;
            const int64_t o = (int64_t)grow * H + gcol;
            const float vv[4] = {v[i].x + d[i].x, v[i].y + d[i].y, v[i].z + d[i].z, v[i].w + d[i].w};
            const float a0[4] = {x0[i].x, x0[i].y, x0[i].z, x0[i].w};
            const float a1[4] = {x1[i].x, x1[i].y, x1[i].z, x1[i].w};
            const float a2[4] = {x2[i].x, x2[i].y, x2[i].z, x2[i].w};
            float r0[4], r1[4], r2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (EPI == EPI_GATES) {
                    r0[j] = sigmoidf_stable(vv[j]);
                    r1[j] = r0[j] * a0[j];
                } else if (EPI == EPI_CAND) {
                    r0[j] = tanhf(vv[j]);
                    r1[j] = (ep.t < lim[i]) ? (a1[j] * a0[j] + (1.f - a1[j]) * r0[j]) : a0[j];
                } else if (EPI == EPI_BWD_RH) {
                    r0[j] = vv[j] * a0[j] * a1[j] * (1.f - a1[j]);
                    r1[j] = a2[j] + vv[j] * a1[j];
                } else {
                    const bool live = ep.t < lim[i];
                    r0[j] = live ? vv[j] * (1.f - a1[j]) * (1.f - a2[j] * a2[j]) : 0.f;
                    r1[j] = live ? vv[j] * (a0[j] - a2[j]) * a1[j] * (1.f - a1[j]) : 0.f;
                    r2[j] = live ? vv[j] * a1[j] : vv[j];
                }
            }
            const float4 q0 = make_float4(r0[0], r0[1], r0[2], r0[3]);
            const float4 q1 = make_float4(r1[0], r1[1], r1[2], r1[3]);
            if (EPI == EPI_GATES) {
                if (gcol < H) {
                    *reinterpret_cast<float4*>(ep.o0 + o) = q0;       // r
                    *reinterpret_cast<float4*>(ep.o2 + o) = q1;       // r * h_prev
                } else {
                    *reinterpret_cast<float4*>(ep.o1 + o - H) = q0;   // u
                }
            } else if (EPI == EPI_CAND) {
                *reinterpret_cast<float4*>(ep.o0 + o) = q0;           // c
                *reinterpret_cast<float4*>(ep.o1 + o) = q1;           // h_new
            } else if (EPI == EPI_BWD_RH) {
                *reinterpret_cast<float4*>(ep.o0 + (int64_t)grow * ep.ldo + gcol) = q0;   // dr_pre
                *reinterpret_cast<float4*>(ep.o1 + o) = q1;                                // dh_acc
            } else {
                *reinterpret_cast<float4*>(ep.o0 + (int64_t)grow * ep.ldo + gcol) = q0;   // dc_pre
                *reinterpret_cast<float4*>(ep.o1 + (int64_t)grow * ep.ldo + gcol) = q1;   // du_pre
                *reinterpret_cast<float4*>(ep.o2 + o) = make_float4(r2[0], r2[1], r2[2], r2[3]);
            }
        }
    }
    }   // tile loop
#if defined(VQA_DBG_STAMPS)
    DBG_STAMP(4);                                            // epilogue issued
    if (threadIdx.x == 0 && g_dbg_stamps != nullptr && blockIdx.x < 4096) {   // (plain kernels: slot 0, start / end only)
        dbg_t[6] = __builtin_amdgcn_s_memtime();
        if (EPI == EPI_PLAIN) dbg_t[4] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = g_dbg_stamps + ((size_t)EPI * 4096 + blockIdx.x) * 8;
        for (int i = 0; i < 7; ++i) o[i] = dbg_t[i];
    }
#endif
}

// C[m, n] = sum_z slab[z][m, n]   (slabs are dense M x N with ld = N)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                                            int M, int N, int ldc, int S) {
    const int64_t n4 = (int64_t)M * N / 4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(slabs)[i];
        for (int z = 1; z < S; ++z) {
            const float4 b = reinterpret_cast<const float4*>(slabs + (int64_t)z * M * N)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int64_t e = i * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        float* c = C + (int64_t)row * ldc + col;  // N % 4 == 0 is required for split-k
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
    }
}

template <int BM, int BN, int WM, int WN, int WGK, int BK, int DEEP, bool A_KC, bool B_KC, int EPI, bool CONV = false,
          bool EDGE = false, int NT = 256, bool GATHER = false>
int launch_one(GemmArgs a, const EpiArgs& ep, int split, hipStream_t st, int max_blocks = 0) {
    constexpr size_t tiles = 2 * (tile_floats<BM, BK, A_KC>() + tile_floats<BN, BK, B_KC>()) * sizeof(float);
    constexpr size_t red = (EPI == EPI_PLAIN) ? (size_t)(WGK - 1) * (BM / WM) * (BN / WN) * (WM / 32) * (WN / 32) * 16 * 64 * sizeof(float) : 0;
    // epilogue transpose patches: one per (m, n) wave slot, and per k group as well for the fused epilogues
    constexpr size_t stage = (size_t)(EPI == EPI_PLAIN ? 1 : WGK) * (BM / WM) * (BN / WN) * 32 * 36 * sizeof(float);
    constexpr size_t lds = tiles > red + stage ? tiles : red + stage;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, WGK, BK, DEEP, A_KC, B_KC, EPI, CONV, EDGE, NT, GATHER>;
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {   // MI355X has 160 KiB of LDS per CU; > 64 KiB needs the opt-in
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return VQA_ERR_LAUNCH;
        attr_done = true;
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.nsplit = split;
    {
        // Each XCD (private 4 MiB L2) walks a contiguous run of tiles.  Pick the tile order whose
        // run touches fewer operand bytes: n fastest re-uses an A row panel across the run (tall
        // activations), m fastest keeps a B column panel resident (batch-sized M against big
        // weights: the GRU / head GEMMs, whose weights then stay partitioned over the 8 L2s).
        const double per_xcd = (double)a.tiles_m * a.tiles_n / 8.0;
        auto foot = [&](double t_fast, double t_slow, double b_fast, double b_slow) {
            const double slow_panels = std::min(t_slow, per_xcd / t_fast + 1.0);
            const double fast_panels = std::min(t_fast, per_xcd);
            return slow_panels * b_slow + fast_panels * b_fast;
        };
        const double n_fast = foot(a.tiles_n, a.tiles_m, BN, BM);   // bytes ~ panels * width (same K)
        const double m_fast = foot(a.tiles_m, a.tiles_n, BM, BN);
        a.m_fastest = (g_force_order >= 0) ? g_force_order : (m_fast < n_fast ? 1 : 0);
    }
    a.vec_epi = (a.N % 4 == 0) && (a.ldc % 4 == 0) && vqa_aligned16(a.C) && (a.slab_stride % 4 == 0) &&
                (a.D == nullptr || (a.ldd % 4 == 0 && vqa_aligned16(a.D))) &&
                (a.bias == nullptr || vqa_aligned16(a.bias)) && (a.scale == nullptr || vqa_aligned16(a.scale));
    int blocks = a.tiles_m * a.tiles_n * split;
    if (max_blocks > 0 && WGK == 1 && EPI == EPI_PLAIN && blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, st, a, ep);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

template <int BM, int BN, int WM, int WN, int WGK, int BK, int DEEP, int NT = 256>
int launch_cfg(int tA, int tB, const GemmArgs& a, int split, hipStream_t st, int mb) {
    const EpiArgs ep{};
    if (tA == 0 && tB == 0)
        return launch_one<BM, BN, WM, WN, WGK, BK, DEEP, true, false, EPI_PLAIN, false, false, NT>(a, ep, split, st, mb);
    if (tA == 0 && tB == 1)
        return launch_one<BM, BN, WM, WN, WGK, BK, DEEP, true, true, EPI_PLAIN, false, false, NT>(a, ep, split, st, mb);
    return launch_one<BM, BN, WM, WN, WGK, BK, DEEP, false, false, EPI_PLAIN, false, false, NT>(a, ep, split, st, mb);
}

struct TileCfg { int BM, BN; };
constexpr int NUM_CFG = 24;
const TileCfg kCfg[NUM_CFG] = {{128, 128}, {128, 128}, {64, 64}, {64, 64}, {64, 64}, {128, 64}, {64, 128},
                               {64, 32}, {32, 32}, {32, 64}, {64, 64}, {64, 32}, {64, 64}, {64, 32}, {128, 32},
                               {128, 64}, {128, 128}, {128, 128}, {256, 64}, {128, 128}, {128, 64}, {64, 128}, {128, 64}, {64, 32}};

// operands that allow the buffer-load fast path (see Stager::load_fast); fills the descriptor extents
bool fast_ok(int tA, int tB, GemmArgs& a) {
    if (!a.vecA || !a.vecB || a.K < 4 || (a.K % 4) != 0) return false;
    if (tA && ((a.M % 4) != 0 || a.M < 4)) return false;          // A stored [K][M]
    if (!tB && ((a.N % 4) != 0 || a.N < 4)) return false;         // B stored [K][N]
    const int64_t ab = (tA ? ((int64_t)(a.K - 1) * a.lda + a.M) : ((int64_t)(a.M - 1) * a.lda + a.K)) * 4;
    const int64_t bb = (tB ? ((int64_t)(a.N - 1) * a.ldb + a.K) : ((int64_t)(a.K - 1) * a.ldb + a.N)) * 4;
    if (ab >= 0xFFFFFF00ll || bb >= 0xFFFFFF00ll) return false;   // 32-bit buffer offsets
    a.a_bytes = (unsigned)ab;
    a.b_bytes = (unsigned)bb;
    return true;
}

int launch_edge(int tA, int tB, const GemmArgs& a, int split, hipStream_t st) {
    const EpiArgs ep{};
    if (tA == 0 && tB == 0)
        return launch_one<64, 64, 32, 32, 1, 16, false, true, false, EPI_PLAIN, false, true>(a, ep, split, st, 0);
    if (tA == 0 && tB == 1)
        return launch_one<64, 64, 32, 32, 1, 16, false, true, true, EPI_PLAIN, false, true>(a, ep, split, st, 0);
    return launch_one<64, 64, 32, 32, 1, 16, false, false, false, EPI_PLAIN, false, true>(a, ep, split, st, 0);
}

int launch_by_id(int cfg, int tA, int tB, const GemmArgs& a_in, int split, hipStream_t st, int mb = 0) {
    GemmArgs a = a_in;
    if (!fast_ok(tA, tB, a)) return launch_edge(tA, tB, a, split, st);   // ragged / unaligned shapes
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 64, 64, 1, 16, false>(tA, tB, a, split, st, mb);
        case 1: return launch_cfg<128, 128, 64, 64, 1, 32, false>(tA, tB, a, split, st, mb);
        case 2: return launch_cfg<64, 64, 32, 32, 1, 16, false>(tA, tB, a, split, st, mb);
        case 3: return launch_cfg<64, 64, 32, 32, 1, 32, false>(tA, tB, a, split, st, mb);
        case 4: return launch_cfg<64, 64, 32, 32, 1, 64, false>(tA, tB, a, split, st, mb);
        case 5: return launch_cfg<128, 64, 64, 32, 1, 32, false>(tA, tB, a, split, st, mb);
        case 6: return launch_cfg<64, 128, 32, 64, 1, 32, false>(tA, tB, a, split, st, mb);
        case 7: return launch_cfg<64, 32, 32, 32, 2, 64, false>(tA, tB, a, split, st, mb);
        case 8: return launch_cfg<32, 32, 32, 32, 4, 64, false>(tA, tB, a, split, st, mb);
        case 9: return launch_cfg<32, 64, 32, 32, 2, 64, false>(tA, tB, a, split, st, mb);
        case 10: return launch_cfg<64, 64, 32, 32, 1, 64, true>(tA, tB, a, split, st, mb);
        case 11: return launch_cfg<64, 32, 32, 32, 2, 64, true>(tA, tB, a, split, st, mb);
        case 12: return launch_cfg<64, 64, 32, 32, 1, 32, true>(tA, tB, a, split, st, mb);
        case 13: return launch_cfg<64, 32, 32, 32, 2, 32, true>(tA, tB, a, split, st, mb);
        case 14: return launch_cfg<128, 32, 32, 32, 1, 64, true>(tA, tB, a, split, st, mb);
        case 15: return launch_cfg<128, 64, 32, 64, 1, 32, true>(tA, tB, a, split, st, mb);
        case 16: return launch_cfg<128, 128, 64, 32, 1, 32, 0, 512>(tA, tB, a, split, st, mb);
        case 17: return launch_cfg<128, 128, 32, 64, 1, 32, 0, 512>(tA, tB, a, split, st, mb);
        case 18: return launch_cfg<256, 64, 64, 32, 1, 32, 0, 512>(tA, tB, a, split, st, mb);
        case 19: return launch_cfg<128, 128, 64, 32, 1, 16, 0, 512>(tA, tB, a, split, st, mb);
        case 20: return launch_cfg<128, 64, 32, 32, 1, 32, 0, 512>(tA, tB, a, split, st, mb);
        case 21: return launch_cfg<64, 128, 32, 32, 1, 32, 0, 512>(tA, tB, a, split, st, mb);
        case 22: return launch_cfg<128, 64, 32, 32, 1, 16, 0, 512>(tA, tB, a, split, st, mb);
        case 23: return launch_cfg<64, 32, 32, 32, 4, 32, 1, 512>(tA, tB, a, split, st, mb);
        default: return VQA_ERR_ARG;
    }
}

// fused GRU-step GEMMs: layout fixed by the epilogue (forward NN, backward NT)
}  // namespace
int vqa_gru_rs_launch(int epi, const GemmArgs& a, const EpiArgs& ep, hipStream_t st);   // gru_stream.hip
namespace {
template <int EPI>
int launch_gru(int cfg, const GemmArgs& a_in, const EpiArgs& ep, hipStream_t st) {
    constexpr bool BKC = (EPI == EPI_BWD_RH || EPI == EPI_BWD_DH);
    GemmArgs a = a_in;
    if (!fast_ok(0, BKC ? 1 : 0, a)) return VQA_ERR_ALIGN;
    if (cfg == 30) {   // register-streamed step kernels (gru_stream.hip); shapes they do not take fall back to the LDS-tiled form
        const int rc = vqa_gru_rs_launch(EPI, a_in, ep, st);
        if (rc != VQA_ERR_UNSUPPORTED) return rc;
        cfg = 16;
    }
    switch (cfg) {
        case 4: return launch_one<64, 64, 32, 32, 1, 64, false, true, BKC, EPI>(a, ep, 1, st);
        case 7: return launch_one<64, 32, 32, 32, 2, 64, false, true, BKC, EPI>(a, ep, 1, st);
        case 8: return launch_one<32, 32, 32, 32, 4, 64, false, true, BKC, EPI>(a, ep, 1, st);
        case 9: return launch_one<32, 64, 32, 32, 2, 64, false, true, BKC, EPI>(a, ep, 1, st);
        case 10: return launch_one<64, 64, 32, 32, 1, 64, true, true, BKC, EPI>(a, ep, 1, st);
        case 11: return launch_one<64, 32, 32, 32, 2, 64, true, true, BKC, EPI>(a, ep, 1, st);
        case 12: return launch_one<64, 64, 32, 32, 1, 32, true, true, BKC, EPI>(a, ep, 1, st);
        case 13: return launch_one<64, 32, 32, 32, 2, 32, true, true, BKC, EPI>(a, ep, 1, st);
        case 16: return launch_one<32, 32, 32, 32, 4, 32, true, true, BKC, EPI>(a, ep, 1, st);
        case 20: return launch_one<128, 64, 32, 32, 1, 32, 0, true, BKC, EPI, false, false, 512>(a, ep, 1, st);
        case 21: return launch_one<64, 128, 32, 32, 1, 32, 0, true, BKC, EPI, false, false, 512>(a, ep, 1, st);
        case 17: return launch_one<64, 32, 32, 32, 4, 32, 1, true, BKC, EPI, false, false, 512>(a, ep, 1, st);
        case 18:   // one 16-wave workgroup per CU: 64x64 tiles for the 2H-wide gate GEMM, 64x32 for the H-wide ones
            if (EPI == EPI_GATES) return launch_one<64, 64, 32, 32, 4, 64, 1, true, BKC, EPI, false, false, 1024>(a, ep, 1, st);
            return launch_one<64, 32, 32, 32, 8, 64, 1, true, BKC, EPI, false, false, 1024>(a, ep, 1, st);
        default: return VQA_ERR_ARG;
    }
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

int g_force_cfg = -1;   // tuning override (vqa_gemm_set_config)
// Tall activations x wide weights (v_linear_v forward, M 18432, N 1024, K 2048): 128x64 tiles (cfg 20) run as fast
// as 64x128 (cfg 21: 551 vs 553 us) and halve the B panel every tile streams (512 KB instead of 1 MB), i.e. the
// L2<->fabric traffic of the weight matrix, which no XCD's 4 MiB L2 can hold (8 MB).
int g_tall_cfg = 20;
int g_max_blocks = 0;   // tuning override for vqa_gemm_f32 (vqa_gemm_set_max_blocks)
int g_conv_cfg = -1;    // tile config of the implicit-GEMM convolutions (vqa_conv_set_config); -1 = by shape
int g_conv_cfg_plain = -1;
// Tile config of the fused GRU-step GEMMs: many waves with small per-wave tiles (32x32), in-block split-k and two
// tiles of register prefetch hide the per-tile barrier and load latency better than 4 waves of 64x32 per CU, and
// every k group finishes its share of the rows in the epilogue (recurrence at B 512, H 1024, T 14: 622 -> 523 us
// forward, 607 -> 487 us backward).  Default: one 16-wave workgroup per CU (cfg 18), 32x32 tiles / 4 waves
// (cfg 16) once the live prefix is down to 256 rows, plain 4-wave tiles for tall batches.
// vqa_gemm_set_gru_config(cfg) forces one config on both directions (tests, tuning); -1 = defaults.
int g_gru_cfg = -1;
// Tall batches (the pre-training model runs 2560 rows per step) fill the chip with plain 4-wave tiles.
// (2560 rows, T 10: forward 1637 us with 64x64 BK 64, 1550 with BK 32 -- profiles/r2_gru_tune_b2560.txt)
// Between 512 and 2048 rows (the live prefix of the pre-training model's 2560-row recurrence passes through all of
// them) one 16-wave workgroup per CU is no longer the best form: 32x64 / 4-wave tiles forward (768 rows: 693 -> 565 us,
// 1536 rows: 1072 -> 961 us per 10 steps) and 64x32 tiles of 8 waves backward (632 -> 559, 995 -> 869);
// profiles/r2_gru_tune_rows.txt.
inline int env_cfg(const char* name, int dflt) {      // tuning overrides, read once
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
inline int gru_cfg_fwd(int rows) {
    static const int mid = env_cfg("VQA_HOT_GRU_MID_FWD", 9), tall = env_cfg("VQA_HOT_GRU_TALL_FWD", 12);
    return g_gru_cfg >= 0 ? g_gru_cfg : (rows >= 2048 ? tall : rows > 512 ? mid : rows > 256 ? 18 : 16);
}
// The H-wide candidate kernel of a tall batch takes 32x64 tiles: at 2560 rows x 1024 columns the 64x64 tile gives 640
// tiles (2.5 per CU, a half-empty last round), 32x64 gives 1280 (5 per CU): forward recurrence 1548 -> 1489 us at
// 2560 rows, T 10.  VQA_HOT_GRU_NARROW_CFG overrides (tuning; -1 = the same config as the gate kernel).
inline int gru_narrow_override() {
    static const int v = [] { const char* e = getenv("VQA_HOT_GRU_NARROW_CFG"); return e ? atoi(e) : 9; }();
    return v;
}
inline int gru_cfg_fwd_cand(int rows) {
    return (g_gru_cfg < 0 && rows >= 2048 && gru_narrow_override() >= 0) ? gru_narrow_override() : gru_cfg_fwd(rows);
}
inline int gru_cfg_bwd(int rows) {
    static const int mid = env_cfg("VQA_HOT_GRU_MID_BWD", 17), tall = env_cfg("VQA_HOT_GRU_TALL_BWD", 13);
    return g_gru_cfg >= 0 ? g_gru_cfg : (rows >= 2048 ? tall : rows > 512 ? mid : rows > 256 ? 18 : 16);
}

// Tile / split-k choice from the gemm_tune sweep on MI355X (tools/gemm_tune.py, profiles/):
//  * dW-type (reduction over the batch rows, K huge, M x N small): 128x128 tiles, BK 16, EIGHT waves
//    (64x32 per wave) and enough split-k slabs for >= 512 workgroups (dWv 554 us = 140 TFLOP/s; the
//    4-wave form of the same tile: 600 us on the same box);
//  * tall activations (M >= 2048): EIGHT waves of 32x32 on a 64x128 tile (v_linear_v forward 565 us =
//    137 TFLOP/s; 4 waves of 64x32 on the same tile area: 573), BK 16 on a 128x64 tile when K is short;
//  * batch-sized M (512) or narrow N: 64x32 tiles, in-block split-k and two tiles of register prefetch
//    (one workgroup per CU cannot hide a global load behind a single tile's MFMAs) -- 8 waves / 4 k groups
//    (512x1024x1024: 14.4 us against 16.0 with 4 waves / 2 k groups), 4 waves for the wide answer head.
inline int tall_small_cfg() {      // VQA_HOT_TALL_SMALL_CFG: tuning override (-1 = keep the 128x64 tile)
    static const int v = [] { const char* e = getenv("VQA_HOT_TALL_SMALL_CFG"); return e ? atoi(e) : 12; }();
    return v < NUM_CFG ? v : 12;
}

inline int tune_pair(const char* env, int cfg, int split) {     // "cfg:split" from the environment, packed cfg << 8 | split
    const char* e = getenv(env);
    if (e != nullptr) {
        int c = 0, s = 0;
        if (sscanf(e, "%d:%d", &c, &s) == 2 && c >= 0 && c < NUM_CFG && s >= 0 && s <= 16) { cfg = c; split = s; }
    }
    return (cfg << 8) | split;
}

void choose(int tA, int tB, int M, int N, int K, int& cfg, int& split) {
    int64_t target;
    if (tA) {
        // weight gradients: 128x128 tiles of 16-deep k tiles for the long ones; up to k = 4096 the 128x64 tile of 32-deep k
        // tiles is 5 % (1024 x 2048 x 2560) to 23 % (the answer head's 2048 x 3000 x 512) faster (profiles/r3_tn_tune.txt)
        static const int tn_short = env_cfg("VQA_HOT_TN_SHORT_CFG", 20);
        static const int tn_mid = env_cfg("VQA_HOT_TN_MID_CFG", 20);   // (k < 20000, at most 128 tiles of 128x128: v_linear_v and recurrent dW, -5 and -9 us)
        const bool mid = K < 20000 && cdiv(M, 128) * cdiv(N, 128) <= 128;
        cfg = ((int64_t)M * N >= (1 << 20)) ? (K <= 4096 ? tn_short : (mid ? tn_mid : 19)) : 3;
        target = 512;
    }
    else if (M >= 2048) {
        cfg = (N >= 512) ? (K >= 2048 ? g_tall_cfg : 22) : 13;
        target = 256;
        // a few thousand rows only (the pre-training model's 2560 per category): fewer than four 128x64 tiles per CU
        // leave a partial last round (320 tiles = 1.25 per CU), 64x64 tiles with two tiles of prefetch fill it
        if (N >= 512 && cdiv(M, 128) * cdiv(N, 64) < 1024 && tall_small_cfg() >= 0) cfg = tall_small_cfg();
    }
    else { cfg = (!tB && N > 2048) ? 13 : 23; target = 256; }
    // The two long-k products around the packed x-projection (W = 300): dx = dxp Wx^T (tall, 300 columns, k = 3H) and
    // dWx = x^T dxp (300 rows, k = T B).  Their loss is tile quantisation (7 x 2^10 rows against 256 CUs); a k split
    // evens the rounds out: dx 129.6 -> 119.4 us (64x64 tiles, split 4), dWx 122.8 -> 114.2 us (64x128, split 4)
    // (profiles/r3_k300_tune.txt).  VQA_HOT_DX_TUNE / VQA_HOT_DWX_TUNE = "cfg:split" override (0:0 = the old choice).
    int want_split = 0;
    if (!tA && tB && M >= 4096 && N <= 320 && K >= 2048) {
        static const int t = tune_pair("VQA_HOT_DX_TUNE", 3, 4);
        if (t >> 8) { cfg = t >> 8; want_split = t & 255; }
    } else if (tA && M <= 320 && N >= 2048 && K >= 4096) {
        static const int t = tune_pair("VQA_HOT_DWX_TUNE", 21, 4);
        if (t >> 8) { cfg = t >> 8; want_split = t & 255; }
    }
    if (g_force_cfg >= 0 && g_force_cfg < NUM_CFG) { cfg = g_force_cfg; want_split = 0; }
    const int64_t blocks = cdiv(M, kCfg[cfg].BM) * cdiv(N, kCfg[cfg].BN);
    if (split <= 0) {
        split = 1;
        if ((N % 4) == 0) {
            if (want_split > 0) split = want_split;
            else
                while (blocks * split < target && K / (split * 2) >= 256 && split < 16) split *= 2;
        }
    }
}

GemmArgs make_args(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, const float* D, int ldd) {
    GemmArgs a;
    a.M = M; a.N = N; a.K = K;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.bias = bias; a.D = D; a.ldd = ldd;
    a.vecA = (lda % 4 == 0) && vqa_aligned16(A);
    a.vecB = (ldb % 4 == 0) && vqa_aligned16(B);
    a.slab_stride = 0;
    a.k_per_split = (int)cdiv(std::max(K, 1), 64) * 64;
    a.scale = nullptr; a.relu = 0;
    a.a_bytes = a.b_bytes = 0;   // filled by set_extents() once the layout is known
    a.Hi = a.Wi = a.Ci = a.Ho = a.Wo = a.cstride = a.pad_t = a.pad_l = a.kw = a.conv_taps = 0;
    a.g_idx = nullptr; a.g_R = 1; a.g_N = 0; a.g_out = nullptr; a.g_ldo = 0;
    return a;
}

}  // namespace

extern "C" int64_t vqa_gemm_workspace_floats(int transA, int transB, int M, int N, int K, int split_k) {
    int cfg, split = split_k;
    choose(transA, transB, M, N, K, cfg, split);
    return split > 1 ? (int64_t)split * M * N : 0;
}

extern "C" int vqa_gemm_set_config(int cfg) {
    VQA_REQUIRE(cfg >= -1 && cfg < NUM_CFG, VQA_ERR_ARG);
    g_force_cfg = cfg;
    return VQA_OK;
}

extern "C" int vqa_gemm_set_gru_config(int cfg) {
    VQA_REQUIRE(cfg == -1 || cfg == 4 || (cfg >= 7 && cfg <= 13) || (cfg >= 16 && cfg <= 18) || cfg == 20 || cfg == 21 || cfg == 30,
                VQA_ERR_ARG);
    g_gru_cfg = cfg;   // -1 restores the defaults
    return VQA_OK;
}

// which callers route short-K problems to gemm_shortk.hip: bit 0 = vqa_gemm_f32 (NN, K <= 304), bit 1 = the extractor's
// 1x1 convolutions (Ci <= 256).  VQA_HOT_SHORTK (default 3) or vqa_gemm_shortk_set_mode.
int g_shortk_mode = -1;
inline int shortk_mode() {
    if (g_shortk_mode < 0) {
        const char* e = getenv("VQA_HOT_SHORTK");
        g_shortk_mode = e ? (atoi(e) & 3) : 3;
    }
    return g_shortk_mode;
}
int g_bf16x3_mode = -1;
inline int bf16x3_mode() {
    if (g_bf16x3_mode < 0) {
        const char* e = getenv("VQA_HOT_BF16X3");
        g_bf16x3_mode = (e && atoi(e) != 0) ? 1 : 0;
    }
    return g_bf16x3_mode;
}
extern "C" int vqa_gemm_bf16x3_set_mode(int on) {
    g_bf16x3_mode = on < 0 ? -1 : (on ? 1 : 0);
    return VQA_OK;
}
extern "C" int vqa_gemm_shortk_set_mode(int mode) {
    g_shortk_mode = mode < 0 ? -1 : (mode & 3);
    return VQA_OK;
}

extern "C" int vqa_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* C, int ldc, const float* bias, const float* D, int ldd, int split_k,
                            float* workspace, int64_t workspace_floats, void* stream) {
    return vqa_gemm_f32_ex(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, D, ldd, split_k, workspace,
                           workspace_floats, g_max_blocks, stream);
}

#if defined(VQA_DBG_STAMPS)
extern "C" int vqa_gemm_dbg_stamps(void* buf) {    // [5 kinds][4096 workgroups][8] u64, device memory
    unsigned long long* b = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps), &b, sizeof(b)) == hipSuccess ? VQA_OK : VQA_ERR_LAUNCH;
}
#endif

extern "C" int vqa_gemm_set_order(int order) {
    g_force_order = (order == 0 || order == 1) ? order : -1;
    return VQA_OK;
}

extern "C" int vqa_gemm_set_max_blocks(int n) {
    g_max_blocks = n > 0 ? n : 0;
    return VQA_OK;
}

extern "C" int vqa_gemm_f32_ex(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                               int ldb, float* C, int ldc, const float* bias, const float* D, int ldd, int split_k,
                               float* workspace, int64_t workspace_floats, int max_blocks, void* stream) {
    VQA_REQUIRE(M >= 0 && N >= 0 && K >= 0, VQA_ERR_ARG);
    if (M == 0 || N == 0) return VQA_OK;
    VQA_REQUIRE(A && B && C, VQA_ERR_ARG);
    VQA_REQUIRE(!(transA == 1 && transB == 1), VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, VQA_ERR_ARG);
    VQA_REQUIRE(D == nullptr || ldd >= N, VQA_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);

    // EXPERIMENT, off unless asked for (VQA_HOT_BF16X3=1 / vqa_gemm_bf16x3_set_mode): the big whole-tile products -- v_linear_v
    // forward and weight gradient, the recurrent weight gradients -- on the bf16 matrix pipe through three-way operand
    // splits (gemm_bf16x3.hip); the headline path is the exact f32 MFMA below
    if (bf16x3_mode() && !transB && D == nullptr && max_blocks == 0 && g_force_cfg < 0 && vqa_gemm_bf16x3_supported(M, N, K) &&
        (int64_t)M * N * K >= (1ll << 32) && lda % 4 == 0 && ldb % 4 == 0 && vqa_aligned16(A) && vqa_aligned16(B)) {
        const int tiles = (M / 128) * (N / 128);
        int sp = 1;
        while (tiles * sp < 256 && K / (sp * 2) >= 512 && sp < 16) sp *= 2;
        while (sp > 1 && (workspace == nullptr || workspace_floats < (int64_t)sp * M * N || ldc % 4 != 0 || !vqa_aligned16(C))) sp /= 2;
        return vqa_gemm_bf16x3(transA, M, N, K, A, lda, B, ldb, C, ldc, bias, sp, workspace, workspace_floats, stream);
    }
    // K <= 304, NN: the kernel with A stationary in registers (gemm_shortk.hip) -- the GRU's packed x-projection
    if (!transA && !transB && split_k <= 1 && max_blocks == 0 && g_force_cfg < 0 && (shortk_mode() & 1) &&
        M >= 1024 && K <= 304 && vqa_gemm_shortk_supported(M, N, K, lda, ldb, ldc) && !(K > 256 && D != nullptr) && vqa_aligned16(A) &&
        vqa_aligned16(B))
        return vqa_gemm_shortk_nn(M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, D, ldd, 0, stream);

    int cfg, split = split_k;
    choose(transA, transB, M, N, K, cfg, split);
    if (split > 1 && ((N % 4) != 0 || workspace == nullptr)) split = 1;

    GemmArgs a = make_args(M, N, K, A, lda, B, ldb, C, ldc, bias, D, ldd);
    if (split > 1) {
        VQA_REQUIRE(workspace_floats >= (int64_t)split * M * N, VQA_ERR_WORKSPACE);
        int kps = (int)cdiv(cdiv(K, split), 64) * 64;
        a.k_per_split = kps;
        split = (int)cdiv(K, kps);
    }
    if (split > 1) {
        // slab 0 also carries bias / addend, so the reduce is a plain sum
        a.C = workspace; a.ldc = N; a.slab_stride = (int64_t)M * N;
        int rc = launch_by_id(cfg, transA, transB, a, split, st, max_blocks);
        if (rc != VQA_OK) return rc;
        const int64_t n4 = (int64_t)M * N / 4;
        const int grid = (int)std::min<int64_t>(cdiv(n4, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, workspace, C, M, N, ldc, split);
        VQA_CHECK_LAUNCH();
        return VQA_OK;
    }
    a.k_per_split = (int)cdiv(std::max(K, 1), 64) * 64;
    return launch_by_id(cfg, transA, transB, a, 1, st, max_blocks);
}

// C[M,N] = gather(table)[M,K] * B[K,N] (+ bias): row m of the left operand is row idx[m / R] * R + m % R of the table.
extern "C" int vqa_gemm_f32_gather(int M, int N, int K, const float* table, int lda, const int64_t* idx, int R,
                                   int64_t n_samples_in_table, const float* B, int ldb, float* C, int ldc,
                                   const float* bias, float* gathered_out, int ldg, void* stream) {
    VQA_REQUIRE(M >= 0 && N >= 0 && K > 0 && R > 0 && n_samples_in_table > 0, VQA_ERR_ARG);
    if (M == 0 || N == 0) return VQA_OK;
    VQA_REQUIRE(table && idx && B && C, VQA_ERR_ARG);
    VQA_REQUIRE(lda >= K && ldb >= N && ldc >= N && (gathered_out == nullptr || ldg >= K), VQA_ERR_ARG);
    VQA_REQUIRE(K % 32 == 0 && N % 4 == 0, VQA_ERR_UNSUPPORTED);          // every k tile full, 16-byte rows
    VQA_REQUIRE(lda % 4 == 0 && vqa_aligned16(table) && ldb % 4 == 0 && vqa_aligned16(B) &&
                    (gathered_out == nullptr || (ldg % 4 == 0 && vqa_aligned16(gathered_out))),
                VQA_ERR_ALIGN);
    GemmArgs a = make_args(M, N, K, table, lda, B, ldb, C, ldc, bias, nullptr, 0);
    const int64_t bb = ((int64_t)(K - 1) * ldb + N) * 4;
    VQA_REQUIRE(bb < 0xFFFFFF00ll, VQA_ERR_UNSUPPORTED);
    a.a_bytes = 0;                     // A is addressed through per-lane 64-bit pointers (the table may exceed 4 GiB)
    a.b_bytes = (unsigned)bb;
    VQA_REQUIRE(gathered_out == nullptr || ((int64_t)(M - 1) * ldg + K) * 4 < 0xFFFFFF00ll, VQA_ERR_UNSUPPORTED);
    a.g_idx = idx; a.g_R = R; a.g_N = n_samples_in_table; a.g_out = gathered_out; a.g_ldo = ldg;
    a.k_per_split = (int)cdiv(K, 64) * 64;
    const EpiArgs ep{};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_tall_cfg == 21)
        return launch_one<64, 128, 32, 32, 1, 32, 0, true, false, EPI_PLAIN, false, false, 512, true>(a, ep, 1, st, 0);
    return launch_one<128, 64, 32, 32, 1, 32, 0, true, false, EPI_PLAIN, false, false, 512, true>(a, ep, 1, st, 0);
}

extern "C" int vqa_gemm_set_tall_config(int cfg) {
    VQA_REQUIRE(cfg == 20 || cfg == 21, VQA_ERR_ARG);
    g_tall_cfg = cfg;
    return VQA_OK;
}

// ---------------------------------------------------------------------------- fused GRU recurrence
// tf.contrib.rnn.GRUCell + tf.nn.dynamic_rnn(sequence_length) (vlmap/modules.py:124-140): two
// GEMM launches per time step with the gate math in their epilogues.
extern "C" int vqa_gru_seq_fwd(float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs,
                               float* r, float* u, float* c, float* rh, int T, int B, int H, void* stream) {
    return vqa_gru_seq_fwd_rows(xp, Wg_h, Wc_h, len, hs, r, u, c, rh, T, B, H, 0, B, stream);
}

// rows [row0, row0 + rows) of the batch only (independent chains: one per stream)
extern "C" int vqa_gru_seq_fwd_rows(float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs,
                                    float* r, float* u, float* c, float* rh, int T, int B, int H, int row0, int rows,
                                    void* stream) {
    VQA_REQUIRE(xp && Wg_h && Wc_h && len && hs && r && u && c && rh && T >= 0 && B > 0 && H > 0, VQA_ERR_ARG);
    VQA_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= B, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0, VQA_ERR_ALIGN);
    if (rows == 0) return VQA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t BH = (int64_t)B * H, o = (int64_t)row0 * H;
    for (int t = 0; t < T; ++t) {
        float* xpt = xp + ((int64_t)t * B + row0) * 3 * H;
        const float* hp = hs + t * BH + o;
        EpiArgs eg{};
        eg.H = H; eg.h_prev = hp; eg.o0 = r + t * BH + o; eg.o1 = u + t * BH + o; eg.o2 = rh + t * BH + o;
        GemmArgs ag = make_args(rows, 2 * H, H, hp, H, Wg_h, 2 * H, nullptr, 0, nullptr, xpt, 3 * H);
        int rc = launch_gru<EPI_GATES>(gru_cfg_fwd(rows), ag, eg, st);
        if (rc != VQA_OK) return rc;
        EpiArgs ec{};
        ec.H = H; ec.t = t; ec.len = len + row0; ec.h_prev = hp; ec.i0 = u + t * BH + o; ec.o0 = c + t * BH + o;
        ec.o1 = hs + (t + 1) * BH + o;
        GemmArgs ac = make_args(rows, H, H, rh + t * BH + o, H, Wc_h, H, nullptr, 0, nullptr, xpt + 2 * H, 3 * H);
        rc = launch_gru<EPI_CAND>(gru_cfg_fwd_cand(rows), ac, ec, st);
        if (rc != VQA_OK) return rc;
    }
    return VQA_OK;
}

// Back-propagation through time.  dh_T [B,H] is the gradient wrt the final state (consumed:
// used as scratch); dxp [T,B,3H] receives (dr_pre | du_pre | dc_pre) per step; dh0 [B,H]
// scratch/returns the gradient wrt the initial state.
// Recurrence over the LIVE prefix only.  Contract: the batch rows are sorted by length, longest first, and
// live_rows[t] (HOST array of T ints) = number of rows with len > t.  Step t then runs on rows [0, live_rows[t])
// -- the gate / candidate GEMMs shrink with the sequences that are still running (real questions average ~6 of
// 14 tokens) -- and finished rows are filled in afterwards exactly as the masked recurrence leaves them.
extern "C" int vqa_gru_seq_fwd_live(float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len,
                                    const int32_t* live_rows, float* hs, float* r, float* u, float* c, float* rh,
                                    int T, int B, int H, void* stream) {
    VQA_REQUIRE(xp && Wg_h && Wc_h && len && live_rows && hs && r && u && c && rh && T >= 0 && B > 0 && H > 0, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0, VQA_ERR_ALIGN);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t BH = (int64_t)B * H;
    int prev = B;
    for (int t = 0; t < T; ++t) {
        const int rows = live_rows[t];
        VQA_REQUIRE(rows >= 0 && rows <= prev, VQA_ERR_ARG);      // non-increasing
        prev = rows;
        if (rows == 0) break;
        float* xpt = xp + (int64_t)t * B * 3 * H;
        const float* hp = hs + t * BH;
        EpiArgs eg{};
        eg.H = H; eg.h_prev = hp; eg.o0 = r + t * BH; eg.o1 = u + t * BH; eg.o2 = rh + t * BH;
        GemmArgs ag = make_args(rows, 2 * H, H, hp, H, Wg_h, 2 * H, nullptr, 0, nullptr, xpt, 3 * H);
        int rc = launch_gru<EPI_GATES>(gru_cfg_fwd(rows), ag, eg, st);
        if (rc != VQA_OK) return rc;
        EpiArgs ec{};
        ec.H = H; ec.t = t; ec.len = len; ec.h_prev = hp; ec.i0 = u + t * BH; ec.o0 = c + t * BH;
        ec.o1 = hs + (t + 1) * BH;
        GemmArgs ac = make_args(rows, H, H, rh + t * BH, H, Wc_h, H, nullptr, 0, nullptr, xpt + 2 * H, 3 * H);
        rc = launch_gru<EPI_CAND>(gru_cfg_fwd_cand(rows), ac, ec, st);
        if (rc != VQA_OK) return rc;
    }
    return vqa_gru_fill_finished(hs, rh, len, T, B, H, stream);
}

extern "C" int vqa_gru_seq_bwd_live(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len,
                                    const int32_t* live_rows, const float* hs, const float* r, const float* u,
                                    const float* c, float* dxp, float* dh_scratch, int T, int B, int H, void* stream) {
    VQA_REQUIRE(dh_T && Wg_h && Wc_h && len && live_rows && hs && r && u && c && dxp && dh_scratch && T >= 0 && B > 0 &&
                    H > 0,
                VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0, VQA_ERR_ALIGN);
    if (T == 0) return VQA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t BH = (int64_t)B * H;
    const int ld = 3 * H;
    // both state-gradient buffers start as dL/dh_final: a row is first touched at its own last step
    if (hipMemcpyAsync(dh_scratch, dh_T, (size_t)BH * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return VQA_ERR_LAUNCH;
    int rc = vqa_gru_zero_finished(dxp, len, T, B, H, stream);
    if (rc != VQA_OK) return rc;
    float* cur = dh_T;
    float* other = dh_scratch;
    int entered = 0;   // rows [0, entered) already carry a running dh_acc in `cur`
    for (int t = T - 1; t >= 0; --t) {
        const int rows = live_rows[t];
        VQA_REQUIRE(rows >= entered && rows <= B, VQA_ERR_ARG);
        float* dxpt = dxp + (int64_t)t * B * ld;
        if (rows > entered) {   // rows whose LAST step is t: first half of the step from dL/dh_final, in place
            const int64_t o = (int64_t)entered * H;
            rc = vqa_gru_bwd_a(cur + o, hs + t * BH + o, u + t * BH + o, c + t * BH + o, len + entered, t,
                               dxpt + (int64_t)entered * ld + 2 * H, ld, dxpt + (int64_t)entered * ld + H, ld, cur + o,
                               rows - entered, H, stream);
            if (rc != VQA_OK) return rc;
            entered = rows;
        }
        if (rows == 0) continue;
        EpiArgs e1{};
        e1.H = H; e1.ldo = ld; e1.h_prev = hs + t * BH; e1.i0 = r + t * BH; e1.o0 = dxpt; e1.o1 = cur;
        GemmArgs a1 = make_args(rows, H, H, dxpt + 2 * H, ld, Wc_h, H, nullptr, 0, nullptr, nullptr, 0);
        rc = launch_gru<EPI_BWD_RH>(gru_cfg_bwd(rows), a1, e1, st);
        if (rc != VQA_OK) return rc;
        if (t > 0) {
            float* dxpp = dxp + (int64_t)(t - 1) * B * ld;
            EpiArgs e2{};
            e2.H = H; e2.t = t - 1; e2.ldo = ld; e2.len = len; e2.h_prev = hs + (t - 1) * BH;
            e2.i0 = u + (t - 1) * BH; e2.i1 = c + (t - 1) * BH; e2.o0 = dxpp + 2 * H; e2.o1 = dxpp + H;
            e2.o2 = other;
            GemmArgs a2 = make_args(rows, H, 2 * H, dxpt, ld, Wg_h, 2 * H, nullptr, 0, nullptr, cur, H);
            rc = launch_gru<EPI_BWD_DH>(gru_cfg_bwd(rows), a2, e2, st);
            if (rc != VQA_OK) return rc;
            float* x = cur; cur = other; other = x;
        }
    }
    return VQA_OK;
}

extern "C" int vqa_gru_seq_bwd(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len,
                               const float* hs, const float* r, const float* u, const float* c, float* dxp,
                               float* dh_scratch, int T, int B, int H, void* stream) {
    return vqa_gru_seq_bwd_rows(dh_T, Wg_h, Wc_h, len, hs, r, u, c, dxp, dh_scratch, T, B, H, 0, B, stream);
}

extern "C" int vqa_gru_seq_bwd_rows(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len,
                                    const float* hs, const float* r, const float* u, const float* c, float* dxp,
                                    float* dh_scratch, int T, int B, int H, int row0, int rows, void* stream) {
    VQA_REQUIRE(dh_T && Wg_h && Wc_h && len && hs && r && u && c && dxp && dh_scratch && T >= 0 && B > 0 && H > 0,
                VQA_ERR_ARG);
    VQA_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= B, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0, VQA_ERR_ALIGN);
    if (T == 0 || rows == 0) return VQA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t BH = (int64_t)B * H, o = (int64_t)row0 * H;
    const int ld = 3 * H;
    len += row0;
    float* dhT = dh_T + o;
    float* dhS = dh_scratch + o;
    float* dh_acc = dhS;   // running dL/dh_{t-1} (partial)
    // first half of step T-1 from dh_T
    {
        const int t = T - 1;
        float* dxpt = dxp + ((int64_t)t * B + row0) * ld;
        int rc = vqa_gru_bwd_a(dhT, hs + t * BH + o, u + t * BH + o, c + t * BH + o, len, t, dxpt + 2 * H, ld,
                               dxpt + H, ld, dh_acc, rows, H, stream);
        if (rc != VQA_OK) return rc;
    }
    for (int t = T - 1; t >= 0; --t) {
        float* dxpt = dxp + ((int64_t)t * B + row0) * ld;
        // drh = dc_pre * Wc_h^T ; epilogue: dr_pre, dh_acc += drh*r
        EpiArgs e1{};
        e1.H = H; e1.ldo = ld; e1.h_prev = hs + t * BH + o; e1.i0 = r + t * BH + o; e1.o0 = dxpt; e1.o1 = dh_acc;
        GemmArgs a1 = make_args(rows, H, H, dxpt + 2 * H, ld, Wc_h, H, nullptr, 0, nullptr, nullptr, 0);
        int rc = launch_gru<EPI_BWD_RH>(gru_cfg_bwd(rows), a1, e1, st);
        if (rc != VQA_OK) return rc;
        // dh_{t-1} = (dr_pre|du_pre) * Wg_h^T + dh_acc ; epilogue: first half of step t-1
        if (t > 0) {
            float* dxpp = dxp + ((int64_t)(t - 1) * B + row0) * ld;
            float* dh_next = (dh_acc == dhS) ? dhT : dhS;
            EpiArgs e2{};
            e2.H = H; e2.t = t - 1; e2.ldo = ld; e2.len = len; e2.h_prev = hs + (t - 1) * BH + o;
            e2.i0 = u + (t - 1) * BH + o; e2.i1 = c + (t - 1) * BH + o; e2.o0 = dxpp + 2 * H; e2.o1 = dxpp + H;
            e2.o2 = dh_next;
            GemmArgs a2 = make_args(rows, H, 2 * H, dxpt, ld, Wg_h, 2 * H, nullptr, 0, nullptr, dh_acc, H);
            rc = launch_gru<EPI_BWD_DH>(gru_cfg_bwd(rows), a2, e2, st);
            if (rc != VQA_OK) return rc;
            dh_acc = dh_next;
        }
    }
    return VQA_OK;
}

// BPTT of a recurrence whose per-step OUTPUTS are consumed too (the bi-directional encoder of vqa/model_vlmap_finetune.py:
// q_L_map = every step's state): d_outs [T,B,H], the gradient wrt the output of step t (zero where t >= len, as
// dynamic_rnn zeroes those outputs), joins the running state gradient before step t is differentiated -- one small add
// per step in front of the same two fused launches as vqa_gru_seq_bwd.
extern "C" int vqa_gru_seq_bwd_outs(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len, const float* hs,
                                    const float* r, const float* u, const float* c, const float* d_outs, float* dxp,
                                    float* dh_scratch, int T, int B, int H, void* stream) {
    VQA_REQUIRE(dh_T && Wg_h && Wc_h && len && hs && r && u && c && d_outs && dxp && dh_scratch && T >= 0 && B > 0 && H > 0,
                VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0, VQA_ERR_ALIGN);
    if (T == 0) return VQA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t BH = (int64_t)B * H;
    const int ld = 3 * H;
    float* dh_acc = dh_scratch;
    {
        const int t = T - 1;
        float* dxpt = dxp + (int64_t)t * B * ld;
        int rc = vqa_add_inplace(dh_T, d_outs + t * BH, BH, stream);
        if (rc != VQA_OK) return rc;
        rc = vqa_gru_bwd_a(dh_T, hs + t * BH, u + t * BH, c + t * BH, len, t, dxpt + 2 * H, ld, dxpt + H, ld, dh_acc, B, H,
                           stream);
        if (rc != VQA_OK) return rc;
    }
    for (int t = T - 1; t >= 0; --t) {
        float* dxpt = dxp + (int64_t)t * B * ld;
        EpiArgs e1{};
        e1.H = H; e1.ldo = ld; e1.h_prev = hs + t * BH; e1.i0 = r + t * BH; e1.o0 = dxpt; e1.o1 = dh_acc;
        GemmArgs a1 = make_args(B, H, H, dxpt + 2 * H, ld, Wc_h, H, nullptr, 0, nullptr, nullptr, 0);
        int rc = launch_gru<EPI_BWD_RH>(gru_cfg_bwd(B), a1, e1, st);
        if (rc != VQA_OK) return rc;
        if (t > 0) {
            rc = vqa_add_inplace(dh_acc, d_outs + (t - 1) * BH, BH, stream);     // + dL/d(output of step t-1)
            if (rc != VQA_OK) return rc;
            float* dxpp = dxp + (int64_t)(t - 1) * B * ld;
            float* dh_next = (dh_acc == dh_scratch) ? dh_T : dh_scratch;
            EpiArgs e2{};
            e2.H = H; e2.t = t - 1; e2.ldo = ld; e2.len = len; e2.h_prev = hs + (t - 1) * BH;
            e2.i0 = u + (t - 1) * BH; e2.i1 = c + (t - 1) * BH; e2.o0 = dxpp + 2 * H; e2.o1 = dxpp + H;
            e2.o2 = dh_next;
            GemmArgs a2 = make_args(B, H, 2 * H, dxpt, ld, Wg_h, 2 * H, nullptr, 0, nullptr, dh_acc, H);
            rc = launch_gru<EPI_BWD_DH>(gru_cfg_bwd(B), a2, e2, st);
            if (rc != VQA_OK) return rc;
            dh_acc = dh_next;
        }
    }
    return VQA_OK;
}

// ---------------------------------------------------------------------------- convolution (NHWC)
// slim resnet_v1 inference conv + folded BatchNorm (+ residual) (+ ReLU): vlmap/modules.py:143-191,
// 219-239, 552-572.  1x1/stride-1 convs are plain GEMMs over [B*H*W, Ci]; every other filter is an
// implicit GEMM whose A tiles are gathered from the NHWC activations (no im2col buffer).
extern "C" int vqa_conv2d_nhwc(const float* x, int B, int Hi, int Wi, int Ci, const float* w, int kh, int kw, int Co,
                               int stride, int pad_t, int pad_l, int Ho, int Wo, const float* scale,
                               const float* shift, const float* residual, int relu, float* y, void* stream) {
    VQA_REQUIRE(x && w && y && B > 0 && Hi > 0 && Wi > 0 && Ci > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 &&
                    Ho > 0 && Wo > 0,
                VQA_ERR_ARG);
    VQA_REQUIRE((int64_t)B * Ho * Wo < (1ll << 31) && (int64_t)B * Hi * Wi < (1ll << 31), VQA_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int M = B * Ho * Wo, K = kh * kw * Ci;
    const bool plain = (kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == Hi && Wo == Wi);
    GemmArgs a = make_args(M, Co, K, x, Ci, w, Co, y, Co, shift, residual, Co);
    a.scale = scale;
    a.relu = relu;
    if (plain && (shortk_mode() & 2) && Ci >= 128 && Ci <= 256 && Co >= 2 * Ci && g_conv_cfg_plain < 0 && g_force_cfg < 0 &&
        vqa_gemm_shortk_supported(M, Co, K, Ci, Co, Co) && vqa_aligned16(x) && vqa_aligned16(w) &&
        (residual == nullptr || (int64_t)M * Co * 4 < 0xFFFFFFF0ll))
        // 1x1 expansions with Ci 128..256 (the K = 512 instance ties with the 64x64 tile on the conv5 expansions: 456 against
        // 442 us, not routed; Co >= 2 Ci: enough column tiles per row panel to pay for loading the panel into
        // registers -- the 256 -> 64 reduction has two and loses, 595 against 481 us): the whole k extent stays in registers
        return vqa_gemm_shortk_nn(M, Co, K, x, Ci, w, Co, y, Co, shift, scale, residual, Co, relu, stream);
    if (plain) {
        // 64x64 tiles: the bottleneck 1x1 convolutions have short K (64..1024) and modest M.  The block-1/2 expansions
        // (64 -> 256, 128 -> 512: a k loop of 2..4 tiles, then an epilogue that reads the residual and writes 4x the
        // input) amortise that epilogue better on 128x128 tiles of 8 waves: in the network 809 -> 738 us and
        // 689 -> 658 us per launch; from 256 input channels on the 64x64 tile is faster again (456 against 467 us,
        // 412 against 442: profiles/r2_vfeat_trace_summary.txt and its history).  The 256 -> 64 reduction at 112x112
        // likes 128x64.
        int cfg = 3;
        if (Ci <= 128 && Co >= 256 && Co >= 2 * Ci && (int64_t)M * Co >= (1ll << 25)) cfg = 16;
        else if (Co <= 64 && Ci >= 256) cfg = 20;
        if (g_conv_cfg_plain >= 0) cfg = g_conv_cfg_plain;
        if (g_force_cfg >= 0 && g_force_cfg < NUM_CFG) cfg = g_force_cfg;
        return launch_by_id(cfg, 0, 0, a, 1, st, 0);
    }
    // 64x64 / 4-wave tiles for every implicit-GEMM layer: at four workgroups per CU (see the kernel's launch bounds)
    // they beat the 8-wave 128x64 and 64x128 tiles in the network (profiles/r2_conv_cfg_ab.txt); in isolation on
    // uniform random inputs the 128x64 tile wins the stride-1 layers (profiles/r2_conv_tune.txt) -- not in situ
    const int ccfg = g_conv_cfg >= 0 ? g_conv_cfg : 0;
    // one 32-deep k tile per filter tap, or 4-channel pixels with a K that is a whole number of tiles (conv1)
    VQA_REQUIRE((Ci % 32 == 0 || (Ci == 4 && K % 32 == 0)) && vqa_aligned16(x), VQA_ERR_ALIGN);
    a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.cstride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.kw = kw;
    a.conv_taps = kh * kw;
    const int64_t xb = (int64_t)B * Hi * Wi * Ci * 4, wb = (int64_t)K * Co * 4;
    VQA_REQUIRE(xb < 0xFFFFFF00ll && wb < 0xFFFFFF00ll && Co % 4 == 0 && vqa_aligned16(w), VQA_ERR_UNSUPPORTED);
    a.a_bytes = (unsigned)xb;
    a.b_bytes = (unsigned)wb;
    const EpiArgs ep{};
    switch (ccfg) {
        case 1: return launch_one<128, 64, 32, 32, 1, 32, 0, true, false, EPI_PLAIN, true, false, 512>(a, ep, 1, st, 0);
        case 2: return launch_one<64, 128, 32, 32, 1, 32, 0, true, false, EPI_PLAIN, true, false, 512>(a, ep, 1, st, 0);
        case 3: return launch_one<128, 128, 64, 32, 1, 32, 0, true, false, EPI_PLAIN, true, false, 512>(a, ep, 1, st, 0);
        default: return launch_one<64, 64, 32, 32, 1, 32, false, true, false, EPI_PLAIN, true>(a, ep, 1, st, 0);
    }
}

extern "C" int vqa_conv_set_config(int cfg) {
    VQA_REQUIRE(cfg >= -1 && cfg <= 3, VQA_ERR_ARG);
    g_conv_cfg = cfg;
    g_conv_cfg_plain = cfg < 0 ? -1 : 3;     // a forced config also pins the 1x1 path to its 64x64 tile
    return VQA_OK;
}
