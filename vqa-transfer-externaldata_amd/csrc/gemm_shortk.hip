// Short-K GEMM for gfx950 with the LEFT operand stationary in registers:  C[M, N] = A[M, K] B[K, N] (+ epilogue), K <= 512.
//
// Who calls it: the packed x-projection of the GRU (x_tm [T*B, 300] times wx_cat [300, 3H]; vlmap/modules.py:124-140 feeds
// the cell one embedded token per step, the projection of all steps is one GEMM here) and the extractor's 1x1 expansion
// convolutions (Ci 64 / 128 / 256 -> 4 Ci; ResNet-101 bottlenecks, vlmap/model_vlmap.py:675-690 and the frozen extractor).
//
// Why a second GEMM kernel: with K <= 304 the general kernel's tile (gemm_f32.hip) runs a k loop of 2-10 tiles between a
// prologue that waits for its first operands and an epilogue that writes 32 KB; those two ends, not the loop, set its
// 0.65-0.72 of the f32 MFMA peak on these shapes (profiles/r2_gemm_tune_k300.txt: "whatever the tile").  Here
//   * a wave keeps its 32 rows of A -- ALL of K -- in registers as ready MFMA fragments (4 * KC VGPRs, 152 for K = 304;
//     the 512-register file is the stationary store, A never passes through LDS and is read from memory once per
//     128-row panel);
//   * the workgroup (4 waves = 128 rows) walks the 32-column tiles of B: a whole [K, 32] tile sits in LDS (<= 38 KB,
//     two buffers), every wave reads its fragments from it -- one LDS dword per MFMA -- and the NEXT tile is on its way
//     into the other buffer by LDS-DMA (global_load_lds_dwordx4: no staging registers, the file belongs to A) while the
//     MFMA run of the current one goes on: one workgroup barrier per 4 KC MFMAs, no exposed prologue after the first
//     unit;
//   * the (panel, column tile) units are dealt to the workgroups in EQUAL contiguous shares (two workgroups per CU,
//     all resident), so no CU runs a tile round more than another: tile-count quantisation was the other half of the
//     loss (2688 128x64 tiles on 256 CUs x 2 = 5.25 rounds -> 6);
//   * while one workgroup writes its 32 x 32 results (plain 128-byte row segments) the CU's other workgroup owns the
//     matrix cores.
// LDS layout of a B tile: row k of the tile at row pi(k) = k ^ ((k >> 2) & 1), 32 floats per row, no padding.  The
// MFMA's two k lanes-halves (lane >> 5 = 0 / 1) read rows 8c + j and 8c + 4 + j of a chunk; pi swaps 4<->5 and 6<->7 so
// that the two rows of every step have opposite parity, i.e. sit in opposite halves of the 64 banks: conflict-free.
#include <stdlib.h>

#include "vqa_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

namespace {

// WV waves per workgroup, 32 rows each (template parameter): 4 = two workgroups per CU (default), 8 = ONE workgroup per CU
// whose two waves per SIMD share the B tile (half the DMA traffic) and start together -- written because in the
// two-workgroup form the second workgroup's A loads starve behind the first one's MFMA chains for ~40 k cycles
// (profiles/r3_shortk_stamps.txt).  In situ the two forms tie (x-projection 119.6 against 121.5 us, extractor 2134 against
// 2139 imgs/s, profiles/r3_shortk_ab.txt); VQA_HOT_SHORTK_WAVES / vqa_gemm_shortk_set_waves select.
constexpr unsigned SK_OOB = 0xFFFFFFF0u;


struct SkArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* scale; const float* D;
    int M, N, K, lda, ldb, ldc, ldd, relu;
    int n_tiles;        // N / 32
    int units;          // row panels * n_tiles
    unsigned a_bytes, b_bytes, c_bytes, d_bytes;
    unsigned* stamps;   // SK_DBG_STAMPS builds only: [gridDim.x][64] cycle stamps of wave 0 (tools/dbg/shortk_stamps.py)
};

#if defined(SK_DBG_STAMPS)   // timing experiment: wave 0 keeps the low word of s_memtime at up to 64 points in the lanes of one VGPR
#define SK_STAMP()                                                                                                   \
    do {                                                                                                             \
        { unsigned long long t__; __builtin_amdgcn_sched_barrier(0);                                                  \
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory");                               \
          __builtin_amdgcn_sched_barrier(0); stampv = (lane == sidx) ? (unsigned)t__ : stampv; }                        \
        ++sidx;                                                                                                      \
    } while (0)
#else
#define SK_STAMP() do {} while (0)
#endif

// same XCD-contiguous order as gemm_f32.hip: workgroups that share an XCD (id % 8) take neighbouring unit ranges, so the
// A panels they share are read into ONE L2
__device__ __forceinline__ int sk_xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int KC, bool RES, int WV>
__global__ __launch_bounds__(64 * WV, (WV == 4 && KC <= 38) ? 2 : 1) void shortk_nn_kernel(SkArgs p) {
    constexpr int SK_NT = 64 * WV, SK_BM = 32 * WV, RPR = 8 * WV;       // threads, rows of a panel, B rows per DMA round
    constexpr int KP = 8 * KC, TILE = KP * 32;
    constexpr int NSTG = (KP + RPR - 1) / RPR;              // DMA wave-instructions per B tile and wave
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [KP][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, s = lane >> 5;
    const int g = sk_xcd_remap(blockIdx.x, gridDim.x), G = gridDim.x;
    const int u_lo = (int)((int64_t)g * p.units / G), u_hi = (int)((int64_t)(g + 1) * p.units / G);
    if (u_lo >= u_hi) return;
    // Every operand goes through a buffer descriptor sized to its extent: rows >= M of A / C / D and rows >= K of B fall
    // outside (loads give 0, stores are dropped) with no per-lane test, and the per-unit part of every address is a
    // SCALAR offset -- the unit loop carries next to no vector address arithmetic, which matters more than its count
    // suggests: a wave that runs its MFMA chain keeps the SIMD's vector issue port, and the co-resident workgroup's
    // vector instructions wait for the gaps (tools/dbg/shortk_stamps.py).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.D : p.A), 0, (int)p.d_bytes, 0x00020000);

    // ---- A: 32 rows x KP of this wave as MFMA fragments (lane (m, s), chunk c: A[row][8c + 4s .. + 3])
    float4 areg[KC];
    const unsigned a_voff = ((unsigned)(wave * 32 + m) * (unsigned)p.lda + 4u * (unsigned)s) * 4u;
    // columns K .. lda-1 of a row may hold anything (padding): only whole float4 below K are read.  Chunks below
    // kf = K / 8 are whole, chunk kf (K % 8 == 4) has its lower half only, the rest are zero -- uniform branches, so that no
    // per-chunk offset register lives through the unit loop.
    const int kf = p.K >> 3;
    const unsigned a_half = ((p.K & 4) && s == 0) ? a_voff : SK_OOB;
    auto load_a = [&](int panel) {
        const unsigned soff = (unsigned)panel * (unsigned)(SK_BM * 4) * (unsigned)p.lda;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            u32x4s q = {0u, 0u, 0u, 0u};
            if (c < kf) q = __builtin_amdgcn_raw_buffer_load_b128(rsA, a_voff, soff + (unsigned)c * 32u, 0);
            else if (c == kf) q = __builtin_amdgcn_raw_buffer_load_b128(rsA, a_half, soff + (unsigned)c * 32u, 0);
            areg[c] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
        }
    };
    // ---- B tile by LDS-DMA: one wave-instruction fills 8 LDS rows (1 KB, lane-linear: lane -> row lane / 8, 16 bytes at
    // column 4 (lane % 8)); the row permutation pi is applied on the SOURCE side -- LDS row q holds global row pi(q),
    // and pi(q + 32 i) = pi(q) + 32 i, so round i differs from round 0 by a scalar offset.  Rows K .. KP-1: outside B's
    // descriptor (nothing is fetched for them); they are zeroed here once as well.
    // A last round of fewer than 32 rows (KP % 32 != 0) is fetched by ALL waves, the spare ones doubling the others'
    // rows (same bytes to the same place): no branch, the unit loop stays one basic block.
    constexpr int LASTW = (KP % RPR == 0) ? WV : (KP % RPR) / 8;        // waves the last round needs
    const int wl = wave % LASTW;
    const int brow = 8 * wave + (lane >> 3), brow_l = 8 * wl + (lane >> 3);
    const unsigned b_voff = ((unsigned)(brow ^ ((brow >> 2) & 1)) * (unsigned)p.ldb + (unsigned)(lane & 7) * 4u) * 4u;
    const unsigned b_voff_l = ((unsigned)(brow_l ^ ((brow_l >> 2) & 1)) * (unsigned)p.ldb + (unsigned)(lane & 7) * 4u) * 4u;
    const unsigned b_step = (unsigned)RPR * (unsigned)p.ldb * 4u;
    auto dma_round = [&](int i, int nt, float* buf) {
        const bool last = (KP % RPR != 0) && (i + 1 == NSTG);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsB, (__attribute__((address_space(3))) void*)(buf + (8 * (last ? wl : wave) + RPR * i) * 32), 16,
            last ? b_voff_l : b_voff, (unsigned)nt * 128u + (unsigned)i * b_step, 0, 0);
    };
    auto fetch_b = [&](int nt, float* buf) {
#pragma unroll
        for (int i = 0; i < NSTG; ++i) dma_round(i, nt, buf);
    };
    for (int i = p.K * 32 + tid; i < KP * 32; i += SK_NT) smem[i] = smem[TILE + i] = 0.f;

#if defined(SK_DBG_STAMPS)
    unsigned stampv = 0;
    int sidx = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#endif
    SK_STAMP();                                              // 0: start
    int panel = u_lo / p.n_tiles, nt = u_lo - panel * p.n_tiles;
    load_a(panel);
    fetch_b(nt, smem);
    __syncthreads();                                         // (drains the DMA: vmcnt(0))
    SK_STAMP();                                              // 1: prologue done
    // per-lane fragment bases: even steps j = 0, 2 read row 8c + j (+ 5 for the upper k half), odd steps row 8c + j (+ 3)
    const int off_e = (s ? 5 * 32 : 0) + m, off_o = (s ? 3 * 32 : 0) + m;
    // C / D map of the 32 x 32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): every store
    // instruction writes two full 128-byte row segments
    const unsigned c_voff = ((unsigned)(wave * 32 + 4 * s) * (unsigned)p.ldc + (unsigned)m) * 4u;
    const unsigned d_voff = ((unsigned)(wave * 32 + 4 * s) * (unsigned)p.ldd + (unsigned)m) * 4u;
    const float lo = p.relu ? 0.f : -__builtin_inff();

    // The unit loop is ONE basic block whose only long-latency resource is the wave's own MFMA chain; everything else
    // of a unit rides inside that chain, one vector-memory instruction per MFMA or two (in-kernel stamps,
    // tools/dbg/shortk_stamps.py: a co-resident workgroup does NOT fill the gaps -- while one wave runs its chain its
    // SIMD partner issues next to nothing, so phases in front of or behind the chain are simply added to it: 1.1k
    // cycles for the DMA issues and 0.9k for the stores on a 9.8k-cycle chain):
    //   * the 16 stores of the PREVIOUS unit's finished values (outv; dropped through an out-of-range offset on the
    //     first unit),
    //   * the DMA of the NEXT tile into the other buffer (fetched even behind the last unit: a harmless tile),
    //   * this unit's residual loads.
    float outv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) outv[r] = 0.f;
    unsigned st_voff = SK_OOB, st_soff = 0u;

    for (int u = u_lo; u < u_hi; ++u) {
        const float* cur = smem + ((u - u_lo) & 1) * TILE;
        float* oth = smem + (((u - u_lo) & 1) ^ 1) * TILE;
        int np = panel, nnt = nt + 1;
        if (nnt == p.n_tiles) { nnt = 0; ++np; }
        const bool has_next = u + 1 < u_hi;
        SK_STAMP();                                          // 2 + 4 i: unit start

        const int col = nt * 32 + m;
        const float bv = p.bias != nullptr ? p.bias[col] : 0.f;
        const float sv = p.scale != nullptr ? p.scale[col] : 1.f;
        const unsigned c_soff = ((unsigned)panel * (unsigned)SK_BM * (unsigned)p.ldc + (unsigned)nt * 32u) * 4u;
        const unsigned d_soff = ((unsigned)panel * (unsigned)SK_BM * (unsigned)p.ldd + (unsigned)nt * 32u) * 4u;

        __builtin_amdgcn_sched_barrier(0);
        // ---- the unit's vector-memory work, in the order it is issued inside the chain: this unit's residual loads
        // (needed at the end of the chain), the next tile's DMA (must have landed by then), the previous unit's 16
        // stores (youngest: the counted wait at the end of the unit leaves exactly them in flight)
        constexpr int NRES = RES ? 16 : 0, NVM = NRES + NSTG + 16;
        constexpr int VPC = (NVM + KC - 1) / KC;             // per chunk of four MFMAs
        float dv[RES ? 16 : 1];
        auto vmem_op = [&](int j) {
            if (j < NRES) {
                const int r = j;
                dv[RES ? r : 0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    rsD, d_voff, d_soff + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.ldd * 4u, 0));
            } else if (j < NRES + NSTG) {
#if !defined(SK_DBG_NODMA)     // timing experiments (tools/dbg): results garbage by construction
                dma_round(j - NRES, nnt, oth);               // `oth` was released by the barrier that ended the previous unit
#endif
            } else if (j < NVM) {
                const int r = j - NRES - NSTG;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[r]), rsC, st_voff,
                                                      st_soff + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.ldc * 4u, 0);
            }
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* be = cur + off_e;
        const float* bo = cur + off_o;
        float bf[2][4];
        bf[0][0] = be[0]; bf[0][2] = be[64]; bf[0][1] = bo[32]; bf[0][3] = bo[96];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // (chunk 0's own two reads lead the pattern below)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int cu = c & 1, nx = cu ^ 1;
            if (c + 1 < KC) {
                bf[nx][0] = be[(c + 1) * 256]; bf[nx][2] = be[(c + 1) * 256 + 64];
                bf[nx][1] = bo[(c + 1) * 256 + 32]; bf[nx][3] = bo[(c + 1) * 256 + 96];
            }
#pragma unroll
            for (int t = 0; t < VPC; ++t) vmem_op(c * VPC + t);
#if defined(SK_DBG_NOMFMA)
            acc[c & 15] += areg[c].x * bf[cu][0] + areg[c].y * bf[cu][1] + areg[c].z * bf[cu][2] + areg[c].w * bf[cu][3];
#else
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c].x, bf[cu][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c].y, bf[cu][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c].z, bf[cu][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c].w, bf[cu][3], acc, 0, 0, 0);
#endif
            // issue order of a chunk: its first MFMA, THEN the two fragment reads of the next chunk (the compiler's LDS
            // wait in front of the next chunk may be a full lgkmcnt(0): this way it finds the reads three MFMAs old
            // instead of just issued), and the chunk's vector-memory instructions behind the next two MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, (VPC + 1) / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, VPC / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        SK_STAMP();                                          // 3 + 4 i: chain issued
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] * sv + bv;
            if (RES) v += dv[r];
            outv[r] = fmaxf(v, lo);
        }
        st_voff = c_voff;
        st_soff = c_soff;
        SK_STAMP();                                          // 4 + 4 i: values finished
        __builtin_amdgcn_s_waitcnt(0x4070);                  // vmcnt(16) lgkmcnt(0): all but the 16 stores -- the DMA has landed
#if !defined(SK_DBG_NOBARRIER)
        asm volatile("s_barrier" ::: "memory");              // ... and everybody's; `cur` is free
#endif
        SK_STAMP();                                          // 5 + 4 i: through the barrier
        if (np != panel && has_next) {                       // next panel (rare): its A, fully waited for here so that the
            load_a(np);                                      // steady-state loop carries no wait on A
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
        }
        panel = np; nt = nnt;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)                             // the last unit's values
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[r]), rsC, st_voff,
                                              st_soff + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.ldc * 4u, 0);
#if defined(SK_DBG_STAMPS)
    {   // lanes 61..63: whole-kernel core cycles and 100 MHz ticks (the clock the kernel ran at = cycles / ticks * 100 MHz)
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime(), ct1 = __builtin_amdgcn_s_memtime();
        if (lane == 61) stampv = (unsigned)(ct1 - ct0);
        if (lane == 62) stampv = (unsigned)(rt1 - rt0);
        if (lane == 63) stampv = (unsigned)rt0;
    }
    if (p.stamps != nullptr && wave == 0) p.stamps[blockIdx.x * 64 + lane] = stampv;
#endif
}

template <int KC, bool RES, int WV>
int sk_launch(const SkArgs& a, int grid, hipStream_t st) {
    constexpr int lds = 2 * 8 * KC * 32 * (int)sizeof(float);
    static bool raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VQA_ERR_LAUNCH;
    if (lds > 65536 && !raised[dev & 63]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(shortk_nn_kernel<KC, RES, WV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return VQA_ERR_LAUNCH;
        raised[dev & 63] = true;
    }
    hipLaunchKernelGGL((shortk_nn_kernel<KC, RES, WV>), dim3(grid), dim3(64 * WV), lds, st, a);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

inline int sk_chunks(int K) {   // template instance (k chunks of 8) that holds K, 0 = none
    if (K > 0 && K <= 64) return 8;
    if (K <= 128) return 16;
    if (K <= 256) return 32;
    if (K <= 304) return 38;
    if (K <= 512) return 64;       // one workgroup per CU (256 A registers per wave, 128 KB of LDS); four waves only
    return 0;
}

unsigned* g_sk_stamps = nullptr;
int g_sk_grid = 0;   // tuning override of the number of workgroups (vqa_gemm_shortk_set_grid); 0 = all resident at once
int g_sk_waves = 0;  // tuning override of the waves per workgroup (vqa_gemm_shortk_set_waves): 4 or 8
constexpr int SK_DEFAULT_WAVES = 4;

}  // namespace

// 1 when vqa_gemm_shortk_nn takes this problem (NN, row-major, 16-byte aligned rows, whole 32-column tiles)
extern "C" int vqa_gemm_shortk_supported(int M, int N, int K, int lda, int ldb, int ldc) {
    if (M <= 0 || N <= 0 || K <= 0 || sk_chunks(K) == 0) return 0;
    if (N % 32 != 0 || K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0 || lda < K || ldb < N || ldc < N) return 0;
    if ((int64_t)M * lda * 4 >= (int64_t)0xFFFFFFF0u || (int64_t)M * ldc * 4 >= (int64_t)0xFFFFFFF0u ||
        (int64_t)K * ldb * 4 >= (int64_t)0xFFFFFFF0u)
        return 0;
    return 1;
}

extern "C" int vqa_gemm_shortk_set_waves(int n) {
    g_sk_waves = (n == 4 || n == 8) ? n : 0;
    return VQA_OK;
}

extern "C" int vqa_gemm_shortk_set_grid(int n) {
    g_sk_grid = n > 0 ? n : 0;
    return VQA_OK;
}

#if defined(SK_DBG_STAMPS)
extern "C" int vqa_gemm_shortk_dbg_stamps(unsigned* buf) {
    g_sk_stamps = buf;
    return VQA_OK;
}
#endif

// C = [relu]( (A B) * scale[n] + bias[n] + D )   -- bias, scale, D optional (NULL)
extern "C" int vqa_gemm_shortk_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                  const float* bias, const float* scale, const float* D, int ldd, int relu, void* stream) {
    VQA_REQUIRE(A && B && C, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gemm_shortk_supported(M, N, K, lda, ldb, ldc), VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE(vqa_aligned16(A) && vqa_aligned16(B), VQA_ERR_ALIGN);
    VQA_REQUIRE(D == nullptr || (ldd >= N && (int64_t)M * ldd * 4 < (int64_t)0xFFFFFFF0u), VQA_ERR_ARG);
    SkArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.scale = scale; a.D = D;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldd = ldd; a.relu = relu;
    a.n_tiles = N / 32;
    static const int env_waves = [] { const char* e = getenv("VQA_HOT_SHORTK_WAVES"); return e ? atoi(e) : 0; }();   // tuning: 4 / 8
    const int waves = g_sk_waves > 0 ? g_sk_waves : (env_waves == 4 || env_waves == 8 ? env_waves : SK_DEFAULT_WAVES);
    const int bm = 32 * (K > 304 ? 4 : waves);
    const int64_t units = (int64_t)((M + bm - 1) / bm) * a.n_tiles;
    VQA_REQUIRE(units < (1ll << 30), VQA_ERR_ARG);
    a.units = (int)units;
    a.a_bytes = (unsigned)((int64_t)M * lda * 4);
    a.b_bytes = (unsigned)((int64_t)K * ldb * 4);
    a.c_bytes = (unsigned)((int64_t)M * ldc * 4);
    a.d_bytes = D != nullptr ? (unsigned)((int64_t)M * ldd * 4) : 0u;
    a.stamps = g_sk_stamps;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached[64] = {};
        if (hipGetDevice(&dev) == hipSuccess) {
            if (cached[dev & 63] == 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached[dev & 63] = prop.multiProcessorCount;
            if (cached[dev & 63] > 0) cus = cached[dev & 63];
        }
    }
    static const int env_grid = [] { const char* e = getenv("VQA_HOT_SHORTK_GRID"); return e ? atoi(e) : 0; }();   // tuning
    int grid = g_sk_grid > 0 ? g_sk_grid : (env_grid > 0 ? env_grid : ((waves == 8 || K > 304) ? cus : 2 * cus));
    if (grid > a.units) grid = a.units;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int kc = sk_chunks(K);
    const bool res = D != nullptr;
#define SK_GO(KC_)                                                                                            \
    case KC_:                                                                                                 \
        if (waves == 8 && K <= 304) return res ? sk_launch<KC_, true, 8>(a, grid, st) : sk_launch<KC_, false, 8>(a, grid, st); \
        return res ? sk_launch<KC_, true, 4>(a, grid, st) : sk_launch<KC_, false, 4>(a, grid, st);
    switch (kc) {
        SK_GO(8)
        SK_GO(16)
        SK_GO(32)
        SK_GO(38)
        case 64: return res ? sk_launch<64, true, 4>(a, grid, st) : sk_launch<64, false, 4>(a, grid, st);
    }
#undef SK_GO
    return VQA_ERR_UNSUPPORTED;
}
