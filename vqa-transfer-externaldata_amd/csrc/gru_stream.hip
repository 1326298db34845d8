// Register-streamed GRU step kernels for gfx950 (tf.contrib.rnn.GRUCell under tf.nn.dynamic_rnn, vlmap/modules.py:124-140):
// the four fused step GEMMs of the recurrence -- gates, candidate, and the two of back-propagation through time -- for the
// latency-bound regime (a few hundred batch rows: the whole launch is ONE tile per wave).
//
// Why another form: the LDS-tiled step kernels (gemm_f32.hip, 32x32 tiles, k split over 4 waves) pay, per launch, a
// first-tile round trip through LDS, a barrier per k tile, a partial-sum exchange and an epilogue that all sit OUTSIDE the
// MFMA chain, and co-resident workgroups do not hide them: a wave that runs an MFMA chain keeps its SIMD's issue port, its
// partner issues next to nothing meanwhile (in-kernel stamps of gemm_shortk.hip, tools/dbg/shortk_stamps.py).  Here
//   * one wave owns one 32 x 32 output tile and ALL of k (or half of it, see WGK): a single dependent chain of
//     v_mfma_f32_32x32x2_f32 (66 cycles per instruction measured for such a chain, tools/micro/mfma_chain.hip), no LDS
//     staging, no barrier in the loop, no k exchange;
//   * both operands stream from L2 straight into MFMA fragment registers through a ring of 8 chunks (8 k each: one
//     16-byte load of the row operand, four 4-byte -- or one 16-byte -- loads of the weights), refilled in the issue slots
//     behind the MFMAs of the chunk that was just consumed: the wave's own loads ride inside its own chain;
//   * the epilogue's side inputs (x-projection addend, previous state, gates) are requested before the chain starts and
//     its outputs leave in the accumulator's own layout (every store instruction writes two 128-byte row segments);
//   * one workgroup (4 waves) per CU and launch at B = 512: 32 rows x 128 columns for the 2H-wide gate GEMM, 32 x 64 with
//     k split over wave pairs (WGK = 2: one exchange through LDS at the very end) for the H-wide ones; workgroups of an
//     XCD take neighbouring column tiles, so each L2 holds an eighth of the weights.
#include "vqa_common.h"
#include "gemm_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));

namespace {

constexpr int RS_PD = 8;                 // chunks (8 k each) in flight per wave

__device__ __forceinline__ int rs_xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float4 rs_ld4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float rs_ld1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

// C[m, n] = sum_k A[m, k] B[k, n]   (B_KC: B[n, k]) with the fused epilogue EPI of gemm_args.h
// CH = chunks of 8 k per wave (K / WGK / 8), a template parameter: the chain is straight-line code, so that the compiler's
// memory waits are exact counts (with a loop it drains the whole ring at every back edge)
template <int EPI, bool B_KC, int WGK, int CH>
__global__ __launch_bounds__(256, 1) void gru_rs_kernel(GemmArgs p, EpiArgs ep) {
    constexpr int WN = 4 / WGK;                       // 32-column tiles of a workgroup
    __shared__ float red[WGK > 1 ? WN * 16 * 64 : 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, s = lane >> 5;
    const int wn = wave % WN, wk = wave / WN;
    const int tiles_m = (p.M + 31) / 32, tiles_n = p.N / (32 * WN);
    const int id = rs_xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int bn = id / tiles_m, bm = id - bn * tiles_m;      // an XCD's workgroups: all row tiles of a few column tiles
    const int m0 = bm * 32, n0 = (bn * WN + wn) * 32;
    const int kw = p.K / WGK, kb = wk * kw;                   // this wave's k range

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    // lane (m, s) feeds MFMA step j of chunk c with k = kb + 8c + 4s + j
    const unsigned a_voff = ((unsigned)(m0 + m) * (unsigned)p.lda + (unsigned)(kb + 4 * s)) * 4u;
    const unsigned b_voff = B_KC ? ((unsigned)(n0 + m) * (unsigned)p.ldb + (unsigned)(kb + 4 * s)) * 4u
                                 : ((unsigned)(kb + 4 * s) * (unsigned)p.ldb + (unsigned)(n0 + m)) * 4u;
    const unsigned b_row = (unsigned)p.ldb * 4u;              // NN: bytes between consecutive k rows of B

    // ---- the epilogue's side inputs, requested now (k group 0 only: it finishes the tile)
    const int col = n0 + m;
    const int H = ep.H;
    float e_d[16], e_x0[16], e_x1[16], e_x2[16];
    int e_lim[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        e_d[r] = e_x0[r] = e_x1[r] = e_x2[r] = 0.f;
        e_lim[r] = 0;
    }
    if (wk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * s;
            if (row >= p.M) continue;
            const int64_t o = (int64_t)row * H + col;
            if (p.D != nullptr) e_d[r] = p.D[(int64_t)row * p.ldd + col];
            if (EPI == EPI_GATES) {
                if (col < H) e_x0[r] = ep.h_prev[o];
            } else if (EPI == EPI_CAND) {
                e_x0[r] = ep.h_prev[o]; e_x1[r] = ep.i0[o]; e_lim[r] = ep.len[row];
            } else if (EPI == EPI_BWD_RH) {
                e_x0[r] = ep.h_prev[o]; e_x1[r] = ep.i0[o]; e_x2[r] = ep.o1[o];
            } else {
                e_x0[r] = ep.h_prev[o]; e_x1[r] = ep.i0[o]; e_x2[r] = ep.i1[o]; e_lim[r] = ep.len[row];
            }
        }
    }

    // ---- the chain
    float4 ra[RS_PD], rb[RS_PD];
    auto fetch = [&](int slot, int c) {
        ra[slot] = rs_ld4(rsA, a_voff, (unsigned)c * 32u);
        if (B_KC) {
            rb[slot] = rs_ld4(rsB, b_voff, (unsigned)c * 32u);
        } else {
            const unsigned so = (unsigned)c * 8u * b_row;
            rb[slot].x = rs_ld1(rsB, b_voff, so);
            rb[slot].y = rs_ld1(rsB, b_voff, so + b_row);
            rb[slot].z = rs_ld1(rsB, b_voff, so + 2u * b_row);
            rb[slot].w = rs_ld1(rsB, b_voff, so + 3u * b_row);
        }
    };
#pragma unroll
    for (int i = 0; i < RS_PD; ++i) fetch(i, i);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = c % RS_PD;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i].x, rb[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i].y, rb[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i].z, rb[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i].w, rb[i].w, acc, 0, 0, 0);
        if (c + RS_PD < CH) fetch(i, c + RS_PD);             // the refill of this slot, behind the chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);                   // (the scheduler would sink the refills towards their use)
    }

    // ---- k exchange (WGK = 2): the upper k group parks its tile, lane-contiguous, the lower one adds it
    if (WGK > 1) {
        if (wk > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wn * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(wn * 16 + r) * 64 + lane];
    }

    // ---- epilogue: C / D map of the 32 x 32 tile, col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * s;
        if (row >= p.M) continue;
        const int64_t o = (int64_t)row * H + col;
        const float v = acc[r] + e_d[r];
        if (EPI == EPI_GATES) {
            const float g = sigmoidf_stable(v);
            if (col < H) {
                ep.o0[o] = g;                    // r
                ep.o2[o] = g * e_x0[r];          // r * h_prev
            } else {
                ep.o1[o - H] = g;                // u
            }
        } else if (EPI == EPI_CAND) {
            const float cc = tanhf(v);
            ep.o0[o] = cc;
            ep.o1[o] = (ep.t < e_lim[r]) ? (e_x1[r] * e_x0[r] + (1.f - e_x1[r]) * cc) : e_x0[r];
        } else if (EPI == EPI_BWD_RH) {
            ep.o0[(int64_t)row * ep.ldo + col] = v * e_x0[r] * e_x1[r] * (1.f - e_x1[r]);   // dr_pre
            ep.o1[o] = e_x2[r] + v * e_x1[r];                                                // dh_acc
        } else {
            const bool live = ep.t < e_lim[r];
            ep.o0[(int64_t)row * ep.ldo + col] = live ? v * (1.f - e_x1[r]) * (1.f - e_x2[r] * e_x2[r]) : 0.f;      // dc_pre
            ep.o1[(int64_t)row * ep.ldo + col] = live ? v * (e_x0[r] - e_x2[r]) * e_x1[r] * (1.f - e_x1[r]) : 0.f; // du_pre
            ep.o2[o] = live ? v * e_x1[r] : v;                                                                     // dh_acc
        }
    }
}

template <int EPI, bool B_KC>
int rs_launch(const GemmArgs& a, const EpiArgs& ep, hipStream_t st) {
    // k split over wave pairs when the plain form leaves CUs without a workgroup
    const int tiles_m = (a.M + 31) / 32;
    int cus = 256;
    {
        static int cached[64] = {};
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess) {
            if (cached[dev & 63] == 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached[dev & 63] = prop.multiProcessorCount;
            if (cached[dev & 63] > 0) cus = cached[dev & 63];
        }
    }
    const bool split = (int64_t)tiles_m * (a.N / 128) < cus && a.N % 64 == 0;
    const int ch = a.K / (split ? 2 : 1) / 8;
    if (!split && a.N % 128 != 0) return VQA_ERR_UNSUPPORTED;
#define RS_GO(WGK_, CH_)                                                                                              \
    hipLaunchKernelGGL((gru_rs_kernel<EPI, B_KC, WGK_, CH_>), dim3(tiles_m * (a.N / (128 / WGK_))), dim3(256), 0, st, a, ep)
    if (split && ch == 64) RS_GO(2, 64);
    else if (split && ch == 128) RS_GO(2, 128);
    else if (!split && ch == 128) RS_GO(1, 128);
    else if (!split && ch == 256) RS_GO(1, 256);
    else return VQA_ERR_UNSUPPORTED;
#undef RS_GO
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

}  // namespace

// 1 when the register-streamed form takes this step GEMM (whole 64-wide k rings, whole column tiles, 32-bit offsets)
int vqa_gru_rs_supported(const GemmArgs& a, bool b_kc) {
    if (a.M <= 0 || a.N % 64 != 0 || (a.K != 1024 && a.K != 2048) || a.lda % 4 != 0 || a.ldb % 4 != 0) return 0;
    if (!vqa_aligned16(a.A) || !vqa_aligned16(a.B)) return 0;
    const int64_t ab = ((int64_t)(a.M - 1) * a.lda + a.K) * 4;
    const int64_t bb = (b_kc ? ((int64_t)(a.N - 1) * a.ldb + a.K) : ((int64_t)(a.K - 1) * a.ldb + a.N)) * 4;
    return ab < 0xFFFFFF00ll && bb < 0xFFFFFF00ll;
}

// launches one fused step GEMM (epi = EPI_GATES / EPI_CAND: B [K, N]; EPI_BWD_RH / EPI_BWD_DH: B [N, K])
int vqa_gru_rs_launch(int epi, const GemmArgs& a_in, const EpiArgs& ep, hipStream_t st) {
    GemmArgs a = a_in;
    const bool kc = (epi == EPI_BWD_RH || epi == EPI_BWD_DH);
    if (!vqa_gru_rs_supported(a, kc)) return VQA_ERR_UNSUPPORTED;
    a.a_bytes = (unsigned)(((int64_t)(a.M - 1) * a.lda + a.K) * 4);
    a.b_bytes = (unsigned)((kc ? ((int64_t)(a.N - 1) * a.ldb + a.K) : ((int64_t)(a.K - 1) * a.ldb + a.N)) * 4);
    switch (epi) {
        case EPI_GATES: return rs_launch<EPI_GATES, false>(a, ep, st);
        case EPI_CAND: return rs_launch<EPI_CAND, false>(a, ep, st);
        case EPI_BWD_RH: return rs_launch<EPI_BWD_RH, true>(a, ep, st);
        case EPI_BWD_DH: return rs_launch<EPI_BWD_DH, true>(a, ep, st);
    }
    return VQA_ERR_ARG;
}
