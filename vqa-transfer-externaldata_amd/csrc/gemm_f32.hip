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

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int KC_LD = BK + 4;  // padded row of a k-contiguous tile (80 B: 16-B aligned, conflict-free b128)

struct GemmArgs {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* D; int ldd;
    int k_per_split;      // multiple of BK
    int64_t slab_stride;  // floats between split-k slabs of C (0 when split_k == 1)
    int vecA, vecB;       // 16-byte global loads allowed for A / B
};

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
template <int BR, bool KC>
struct Stager {
    static constexpr int NV = (BR * BK / 4 + 255) / 256;  // float4 per thread
    float4 reg[NV];

    // g: operand base; ld: leading dim; r0: first row (m or n) of the tile; k0: first k;
    // RLIM: M or N; KLIM: end of this split's k range.
    __device__ __forceinline__ void load(const float* g, int ld, int r0, int k0, int RLIM, int KLIM, bool vec) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (KC) {
                const int row = idx >> 2, kq = (idx & 3) * 4;
                const int gr = r0 + row, gk = k0 + kq;
                const bool in = (idx < BR * 4) && (gr < RLIM);
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
    __device__ __forceinline__ void store(float* s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (KC) {
                if (idx < BR * 4) {
                    const int row = idx >> 2, kq = (idx & 3) * 4;
                    *reinterpret_cast<float4*>(s + row * KC_LD + kq) = reg[i];
                }
            } else {
                constexpr int QPR = BR / 4;
                if (idx < BK * QPR) {
                    const int k = idx / QPR, rq = (idx % QPR) * 4;
                    *reinterpret_cast<float4*>(s + k * BR + rq) = reg[i];
                }
            }
        }
    }
};

template <int BR, bool KC>
constexpr int tile_floats() { return KC ? BR * KC_LD : BK * BR; }

// fragment of 4 consecutive MFMA k-steps (chunk c of the 16-deep tile) for the 32 rows at r0
template <int BR, bool KC>
__device__ __forceinline__ float4 frag4(const float* s, int r0, int c, int lane) {
    const int i = lane & 31, h = lane >> 5;
    if (KC) {
        return *reinterpret_cast<const float4*>(s + (r0 + i) * KC_LD + c * 8 + h * 4);
    } else {
        const float* p = s + (c * 8 + h * 4) * BR + r0 + i;
        return make_float4(p[0], p[BR], p[2 * BR], p[3 * BR]);
    }
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    constexpr int A_FL = tile_floats<BM, A_KC>();
    constexpr int B_FL = tile_floats<BN, B_KC>();
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_FL + B_FL)];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    Stager<BM, A_KC> sa;
    Stager<BN, B_KC> sb;
    const bool va = p.vecA, vb = p.vecB;

    if (nt > 0) {
        sa.load(p.A, p.lda, m0, kbeg, p.M, kend, va);
        sb.load(p.B, p.ldb, n0, kbeg, p.N, kend, vb);
        sa.store(smem);
        sb.store(smem + A_FL);
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const float* As = smem + (t & 1) * (A_FL + B_FL);
        const float* Bs = As + A_FL;
        if (t + 1 < nt) {
            sa.load(p.A, p.lda, m0, kbeg + (t + 1) * BK, p.M, kend, va);
            sb.load(p.B, p.ldb, n0, kbeg + (t + 1) * BK, p.N, kend, vb);
        }
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = frag4<BM, A_KC>(As, wm * WM + a * 32, c, lane);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = frag4<BN, B_KC>(Bs, wn * WN + b * 32, c, lane);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
                }
        }
        if (t + 1 < nt) {
            float* An = smem + ((t + 1) & 1) * (A_FL + B_FL);
            sa.store(An);
            sb.store(An + A_FL);
        }
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* Cz = p.C + (int64_t)blockIdx.z * p.slab_stride;
    const bool first = (blockIdx.z == 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn * WN + b * 32 + (lane & 31);
            const int rbase = m0 + wm * WM + a * 32 + 4 * (lane >> 5);
            if (col < p.N) {
                const float bv = (p.bias != nullptr && first) ? p.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) {
                        float v = acc[a][b][r] + bv;
                        if (p.D != nullptr && first) v += p.D[(int64_t)row * p.ldd + col];
                        Cz[(int64_t)row * p.ldc + col] = v;
                    }
                }
            }
        }
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

template <int BM, int BN, int WM, int WN>
int launch_cfg(int tA, int tB, const GemmArgs& a, int split, hipStream_t st) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, split);
    if (tA == 0 && tB == 0)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, a);
    else if (tA == 0 && tB == 1)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, a);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// tile / split-k choice: fill 256 CUs (>= ~2 blocks per CU when the problem allows it)
void choose(int M, int N, int K, int& big, int& split) {
    const int64_t blocks_big = cdiv(M, 128) * cdiv(N, 128);
    const int64_t blocks_small = cdiv(M, 64) * cdiv(N, 64);
    big = (blocks_big >= 384) ? 1 : 0;
    if (split <= 0) {
        const int64_t blocks = big ? blocks_big : blocks_small;
        split = 1;
        if ((N % 4) == 0) {
            while (blocks * split < 256 && K / (split * 2) >= 256 && split < 16) split *= 2;
        }
    }
}

}  // namespace

extern "C" int64_t vqa_gemm_workspace_floats(int M, int N, int K, int split_k) {
    int big, split = split_k;
    choose(M, N, K, big, split);
    return split > 1 ? (int64_t)split * M * N : 0;
}

extern "C" int vqa_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* C, int ldc, const float* bias, const float* D, int ldd, int split_k,
                            float* workspace, int64_t workspace_floats, void* stream) {
    VQA_REQUIRE(M >= 0 && N >= 0 && K >= 0, VQA_ERR_ARG);
    if (M == 0 || N == 0) return VQA_OK;
    VQA_REQUIRE(A && B && C, VQA_ERR_ARG);
    VQA_REQUIRE(!(transA == 1 && transB == 1), VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, VQA_ERR_ARG);
    VQA_REQUIRE(D == nullptr || ldd >= N, VQA_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);

    int big, split = split_k;
    choose(M, N, K, big, split);
    if (split > 1 && ((N % 4) != 0 || workspace == nullptr)) split = 1;

    GemmArgs a;
    a.M = M; a.N = N; a.K = K;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb;
    a.bias = bias; a.D = D; a.ldd = ldd;
    a.vecA = (lda % 4 == 0) && vqa_aligned16(A);
    a.vecB = (ldb % 4 == 0) && vqa_aligned16(B);
    if (split > 1) {
        VQA_REQUIRE(workspace_floats >= (int64_t)split * M * N, VQA_ERR_WORKSPACE);
        int kps = (int)cdiv(cdiv(K, split), BK) * BK;
        a.k_per_split = kps;
        split = (int)cdiv(K, kps);
    }
    int rc;
    if (split > 1) {
        // slab 0 also carries bias / addend, so the reduce is a plain sum
        a.C = workspace; a.ldc = N; a.slab_stride = (int64_t)M * N;
        rc = big ? launch_cfg<128, 128, 64, 64>(transA, transB, a, split, st)
                 : launch_cfg<64, 64, 32, 32>(transA, transB, a, split, st);
        if (rc != VQA_OK) return rc;
        const int64_t n4 = (int64_t)M * N / 4;
        const int grid = (int)std::min<int64_t>(cdiv(n4, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, workspace, C, M, N, ldc, split);
        VQA_CHECK_LAUNCH();
        return VQA_OK;
    }
    a.C = C; a.ldc = ldc; a.slab_stride = 0;
    a.k_per_split = (int)cdiv(std::max(K, 1), BK) * BK;
    return big ? launch_cfg<128, 128, 64, 64>(transA, transB, a, 1, st)
               : launch_cfg<64, 64, 32, 32>(transA, transB, a, 1, st);
}
