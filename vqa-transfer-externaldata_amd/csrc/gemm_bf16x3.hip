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
// consecutive k a lane feeds to one MFMA); one LDS buffer, register prefetch of the next tile, two workgroups per CU.
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

__global__ __launch_bounds__(NT, 2) void gemm_bf16x3_nn_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                               int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                               int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];      // [A planes | B planes]
    __bf16* sA = lds;
    __bf16* sB = lds + OPER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;           // the wave's 64 x 64 corner of the tile

    // A tile [128 rows][32 k]: thread -> (row = idx / 8, 4 consecutive k): one 16-byte load, 8 lanes cover a 128-byte row
    // B tile [32 k][128 n]:    thread -> a 4 (k) x 4 (n) block: four 16-byte loads from consecutive k rows (8 lanes cover
    //                          128 contiguous bytes of a row), transposed in registers into 4 k-consecutive bf16 (8 bytes)
    //                          for each of its 4 columns
    f32x4n ra[4];
    f32x4n rb[4];
    const float* pa[4];
    const float* pb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * NT;
        pa[i] = A + (int64_t)(m0 + idx / 8) * lda + (idx % 8) * 4;
    }
    const int bk4 = (tid % 8) * 4, bn4 = (tid / 8) * 4;            // the block's first k and first column inside the tile
    pb = B + (int64_t)bk4 * ldb + n0 + bn4;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4n*>(pa[i] + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4n*>(pb + (int64_t)(k0 + i) * ldb);
    };
    auto stage = [&]() {
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
    const int nk = K / BK;
    fetch(0);
    stage();
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) fetch((t + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[x][p] = *reinterpret_cast<const bf16x8*>(sA + p * PLANE + (wm + x * 32 + fr) * RS + ks * 16 + fk);
                    fb[x][p] = *reinterpret_cast<const bf16x8*>(sB + p * PLANE + (wn + x * 32 + fr) * RS + ks * 16 + fk);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // smallest products first, the leading one last
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) {
            __syncthreads();          // every wave is done reading tile t
            stage();                  // tile t + 1 (fetched above, behind the MFMAs)
            __syncthreads();
        }
    }
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

extern "C" int vqa_gemm_bf16x3_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                  const float* bias, void* stream) {
    VQA_REQUIRE(A && B && C, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_gemm_bf16x3_supported(M, N, K) == 1, VQA_ERR_UNSUPPORTED);
    VQA_REQUIRE(lda >= K && ldb >= N && ldc >= N, VQA_ERR_ARG);
    VQA_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && vqa_aligned16(A) && vqa_aligned16(B), VQA_ERR_ALIGN);
    constexpr size_t lds_bytes = (size_t)2 * OPER * sizeof(__bf16);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_nn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return VQA_ERR_LAUNCH;
        attr = true;
    }
    const int tiles_n = N / BN;
    hipLaunchKernelGGL(gemm_bf16x3_nn_kernel, dim3((unsigned)((M / BM) * tiles_n)), dim3(NT), lds_bytes,
                       static_cast<hipStream_t>(stream), A, lda, B, ldb, C, ldc, bias, M, N, K, tiles_n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
