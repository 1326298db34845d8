// EXPERIMENT (not on the default path): the NN GEMM of v_linear_v on the bf16 matrix pipe with
// f32-equivalent products -- every f32 operand is split into three bf16 pieces x = x1 + x2 + x3 (8 + 8 + 8 mantissa bits:
// x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2), each residual exact in f32) and a*b is accumulated in f32 from the
// six products whose weight is >= 2^-24 of a1*b1: a3*b1, a2*b2, a1*b3, a2*b1, a1*b2, a1*b1 (the three dropped ones are
// below f32 rounding).  v_mfma_f32_32x32x16_bf16 runs 16x the MACs per cycle of v_mfma_f32_32x32x2_f32, so six products
// cost 6/16 of the f32 instruction's matrix time.  SURVEY section 7 "Hard parts" names split-bf16 x 3 as an acceptable parity
// mode; the headline path stays on the exact f32 MFMA.
//
// C[M,N] = A[M,K] * B[K,N] (+ bias[N]), all f32 row-major in HBM; the split happens on the way into LDS (A: 151 MB, never
// written back in split form).  Tile 128 x 128, BK 32, 256 threads = 4 waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32);
// operands live in LDS as three bf16 planes [row or column][k] with 80-byte rows (conflict-free ds_read_b128 of the 8
// consecutive k a lane feeds to one MFMA).  TWO LDS buffers (120 KB, one workgroup per CU): while the matrix pipe works
// through tile t, the same waves split tile t + 1 (already in registers) into the other buffer -- the split's vector ALU
// work (~260 instructions per wave and k tile) rides in the issue slots between the 48 MFMAs of a k tile
// (sched_group_barrier pattern) -- and fetch tile t + 2; one barrier per k tile.  (The first version, one buffer and two
// workgroups per CU, kept the matrix pipe 45 % busy: profiles/r3_pmc_bf16x3.txt.)
#include <algorithm>

#include "vqa_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int RS = 40;                       // bf16 per LDS row: 32 k + 8 pad = 80 bytes
constexpr int PLANE = 128 * RS;              // bf16 per plane of one operand
constexpr int OPER = 3 * PLANE;              // bf16 per operand (3 planes)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    float r = x - (float)h;                  // exact: x and h agree in the leading 8 bits
    m = (__bf16)r;
    r = r - (float)m;                        // exact again
    l = (__bf16)r;
}

// A_KM: A is stored [K][M] (the transposed left operand of a weight gradient, dW = X^T dY); then both operands are k-major
// and A's tile goes through the same 4 x 4 transposing loads as B's.  Split k: blockIdx.x = z * tiles + tile, slab z of
// `C` (stride slab floats, leading dimension ldc) receives the partial product of k range z; bias rides in slab 0.
template <bool A_KM>
__global__ __launch_bounds__(NT, 1) void gemm_bf16x3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                            int ldb, float* __restrict__ C, int ldc,
                                                            const float* __restrict__ bias, int M, int N, int K, int tiles_n,
                                                            int tiles, int k_per_split, int64_t slab) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];      // [A planes | B planes]
    __bf16* sA = lds;
    __bf16* sB = lds + OPER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.x / tiles, tile = blockIdx.x - z * tiles;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;           // the wave's 64 x 64 corner of the tile
    const int kbeg = z * k_per_split, kend = min(K, kbeg + k_per_split);
    C += z * slab;
    if (z > 0) bias = nullptr;

    // A tile [128 rows][32 k], A row-major: thread -> (row = idx / 8, 4 consecutive k): one 16-byte load, 8 lanes cover a
    //                          128-byte row
    // B tile [32 k][128 n] (and A's when A_KM): thread -> a 4 (k) x 4 (n) block: four 16-byte loads from consecutive k rows
    //                          (8 lanes cover 128 contiguous bytes of a row), transposed in registers into 4 k-consecutive
    //                          bf16 (8 bytes) for each of its 4 columns
    f32x4n ra[4];
    f32x4n rb[4];
    const float* pa[4];
    const float* pb;
    const int bk4 = (tid % 8) * 4, bn4 = (tid / 8) * 4;            // the block's first k and first column inside the tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * NT;
        pa[i] = A_KM ? A + (int64_t)(kbeg + bk4 + i) * lda + m0 + bn4 : A + (int64_t)(m0 + idx / 8) * lda + kbeg + (idx % 8) * 4;
    }
    pb = B + (int64_t)(kbeg + bk4) * ldb + n0 + bn4;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4n*>(A_KM ? pa[i] + (int64_t)k0 * lda : pa[i] + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4n*>(pb + (int64_t)(k0 + i) * ldb);
    };
    auto stage = [&]() {
        if (A_KM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // row (of the tile) bn4 + j: its four consecutive k
                bf16x4 h, m, l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __bf16 a, b, c;
                    split3(ra[i][j], a, b, c);
                    h[i] = a; m[i] = b; l[i] = c;
                }
                *reinterpret_cast<bf16x4*>(sA + (bn4 + j) * RS + bk4) = h;
                *reinterpret_cast<bf16x4*>(sA + PLANE + (bn4 + j) * RS + bk4) = m;
                *reinterpret_cast<bf16x4*>(sA + 2 * PLANE + (bn4 + j) * RS + bk4) = l;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * NT;
                const int row = idx / 8, kq = (idx % 8) * 4;
                bf16x4 h, m, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __bf16 a, b, c;
                    split3(ra[i][j], a, b, c);
                    h[j] = a; m[j] = b; l[j] = c;
                }
                *reinterpret_cast<bf16x4*>(sA + row * RS + kq) = h;
                *reinterpret_cast<bf16x4*>(sA + PLANE + row * RS + kq) = m;
                *reinterpret_cast<bf16x4*>(sA + 2 * PLANE + row * RS + kq) = l;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // column bn4 + j: its four consecutive k
            bf16x4 h, m, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __bf16 a, b, c;
                split3(rb[i][j], a, b, c);
                h[i] = a; m[i] = b; l[i] = c;
            }
            *reinterpret_cast<bf16x4*>(sB + (bn4 + j) * RS + bk4) = h;
            *reinterpret_cast<bf16x4*>(sB + PLANE + (bn4 + j) * RS + bk4) = m;
            *reinterpret_cast<bf16x4*>(sB + 2 * PLANE + (bn4 + j) * RS + bk4) = l;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fr = lane & 31, fk = (lane >> 5) * 8;      // the lane's row / column inside a 32-wide MFMA tile, its 8 k
    const int nk = (kend - kbeg) / BK;
    constexpr int BUF = 2 * OPER;                         // bf16 per LDS buffer (A planes | B planes)
    auto stage_into = [&](int buf) {
        sA = lds + buf * BUF;
        sB = sA + OPER;
        stage();
    };
    fetch(0);
    stage_into(0);
    if (nk > 1) fetch(BK);                                // tile 1 waits in the registers
    __syncthreads();
    bf16x8 fa[2][2][3], fb[2][2][3];                      // [k step][tile][plane]: all 24 fragment reads of a tile up front
    auto read_frags = [&](int t) {
        const __bf16* cA = lds + (t & 1) * BUF;
        const __bf16* cB = cA + OPER;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[ks][x][p] = *reinterpret_cast<const bf16x8*>(cA + p * PLANE + (wm + x * 32 + fr) * RS + ks * 16 + fk);
                    fb[ks][x][p] = *reinterpret_cast<const bf16x8*>(cB + p * PLANE + (wn + x * 32 + fr) * RS + ks * 16 + fk);
                }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // smallest products first, the leading one last
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][2], fb[ks][b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][1], fb[ks][b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][0], fb[ks][b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][1], fb[ks][b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][0], fb[ks][b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a][0], fb[ks][b][0], acc[a][b], 0, 0, 0);
                }
    };
    int t = 0;
    for (; t + 1 < nk; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        read_frags(t);
        stage_into((t + 1) & 1);       // tile t + 1: split and store into the other buffer -- ONE basic block with the MFMAs
        mfmas();
        // issue pattern of the block: 24 LDS reads first, then per MFMA five or six vector ALU instructions and, every
        // second MFMA, one LDS store (masks: 0x008 MFMA, 0x002 VALU, 0x100 DS read, 0x200 DS write)
        __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nk) fetch((t + 2) * BK);              // its latency hides behind the next tile's MFMAs
        __syncthreads();                                  // tile t + 1 is in LDS; tile t's buffer is free
    }
    read_frags(t);                                        // last tile: nothing left to stage
    mfmas();
    // C / D map of a 32 x 32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn + b * 32 + (lane & 31);
            const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(int64_t)row * ldc + col] = acc[a][b][r] + bv;
            }
        }
}

}  // namespace

// 1 when the experimental kernel applies to the shape (whole tiles only), else 0
extern "C" int vqa_gemm_bf16x3_supported(int M, int N, int K) {
    return (M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0) ? 1 : 0;
}

namespace {
// C[m, n] = sum_z slab[z][m, n]   (slabs dense M x N)
__global__ __launch_bounds__(256) void bf16x3_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N,
                                                            int ldc, int S) {
    const int64_t n4 = (int64_t)M * N / 4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4n v = reinterpret_cast<const f32x4n*>(slabs)[i];
        for (int z = 1; z < S; ++z) v += reinterpret_cast<const f32x4n*>(slabs + (int64_t)z * M * N)[i];
        const int64_t e = i * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        *reinterpret_cast<f32x4n*>(C + (int64_t)row * ldc + col) = v;
    }
}

template <bool A_KM>
int bf16x3_go(const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, int M, int N, int K,
              int split, int kps, int64_t slab, hipStream_t st) {
    constexpr size_t lds_bytes = (size_t)2 * 2 * OPER * sizeof(__bf16);      // two buffers of (A planes | B planes)
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel<A_KM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return VQA_ERR_LAUNCH;
        attr = true;
    }
    const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
    hipLaunchKernelGGL(gemm_bf16x3_kernel<A_KM>, dim3((unsigned)(tiles * split)), dim3(NT), lds_bytes, st, A, lda, B, ldb, C, ldc,
                       bias, M, N, K, tiles_n, tiles, kps, slab);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
}  // namespace

// floats of workspace vqa_gemm_bf16x3 needs for its split-k slabs (0: none)
extern "C" int64_t vqa_gemm_bf16x3_workspace_floats(int M, int N, int K, int split_k) {
    return split_k > 1 ? (int64_t)split_k * M * N : 0;
}

// C[M,N] = op(A) B (+ bias): transA = 0: A [M,K]; 1: A stored [K,M].  split_k <= 1: one pass; else k is cut into
// split_k ranges (multiples of 32) whose partial products meet in `workspace` and are summed in slab order.
extern "C" int vqa_gemm_bf16x3(int transA, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                               int ldc, const float* bias, int split_k, float* workspace, int64_t workspace_floats,
                               void* stream) {
    VQA_REQUIRE(A && B && C, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gemm_bf16x3_supported(M, N, K) == 1, VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE(lda >= (transA ? M : K) && ldb >= N && ldc >= N, VQA_ERR_ARG);
    VQA_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && vqa_aligned16(A) && vqa_aligned16(B), VQA_ERR_ALIGN);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int split = split_k > 1 ? split_k : 1;
    int kps = ((K / split + BK - 1) / BK) * BK;
    if (kps < BK) kps = BK;
    split = (K + kps - 1) / kps;
    if (split <= 1)
        return transA ? bf16x3_go<true>(A, lda, B, ldb, C, ldc, bias, M, N, K, 1, K, 0, st)
                      : bf16x3_go<false>(A, lda, B, ldb, C, ldc, bias, M, N, K, 1, K, 0, st);
    VQA_REQUIRE(workspace != nullptr && workspace_floats >= (int64_t)split * M * N && ldc % 4 == 0 && vqa_aligned16(C) &&
                    vqa_aligned16(workspace),
                VQA_ERR_WORKSPACE);
    const int rc = transA ? bf16x3_go<true>(A, lda, B, ldb, workspace, N, bias, M, N, K, split, kps, (int64_t)M * N, st)
                          : bf16x3_go<false>(A, lda, B, ldb, workspace, N, bias, M, N, K, split, kps, (int64_t)M * N, st);
    if (rc != VQA_OK) return rc;
    const int64_t n4 = (int64_t)M * N / 4;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(bf16x3_reduce_kernel, dim3(grid), dim3(256), 0, st, workspace, C, M, N, ldc, split);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gemm_bf16x3_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                  const float* bias, void* stream) {
    return vqa_gemm_bf16x3(0, M, N, K, A, lda, B, ldb, C, ldc, bias, 1, nullptr, 0, stream);
}
