// Fused Hadamard attention + attention pooling for gfx950 (SURVEY rows a6+a7).
//
// Reference: modules.hadamard_attention (vlmap/modules.py:67-97) followed by
// modules.attention_pooling (vlmap/modules.py:23-39):
//   s[b,r]   = sum_h (v[b,r,h] * qv[b,h]) * keep[b,r,h]/keep_prob * w[h] + bias
//   s[r>=nb] = -inf ; att = softmax_R(s) ; pooled[b,:] = sum_r att[b,r] * V[b,r,:]
//
// `rep` queries may share one memory (the pre-training model attends 5 key boxes per image over the
// same 36 regions: vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:323-364 tiles V_ft x5 -- here
// the tile is never materialised): v, V, nb are indexed by the memory m, qv / keepmask / att / pooled
// by the query q = m*rep + j.
//
// HBM-bound.  One workgroup (4 waves) per sample keeps s/att for the <= 36
// regions in LDS, so v (147 KB), the keep mask (37 KB) and the raw features V
// (295 KB) are each read exactly once with 16-byte-per-lane coalesced loads and
// the scores never leave the CU.  Row dot products use wavefront (64-lane)
// shuffles; the softmax over R runs in one wave.
#include "vqa_common.h"

namespace {

constexpr int MAX_R = 1024;

// 512 threads: twice the waves (and loads in flight) per sample of the 256-thread form; the kernel is a
// pure stream over v, the keep mask and V (~480 KB per sample)
constexpr int FWD_THREADS = 512;
__global__ __launch_bounds__(FWD_THREADS) void attn_pool_fwd_kernel(
    const float* __restrict__ v, const float* __restrict__ qv, const float* __restrict__ V,
    const int32_t* __restrict__ nb, const float* __restrict__ w, const float* __restrict__ bias,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ att_out, float* __restrict__ pooled, int R,
    int H, int D, int rep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // qw[H] | s[R]
    float* qw = lds;
    float* s = lds + H;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // b = query index
    const int mem = b / rep;
    const float* vb = v + (int64_t)mem * R * H;
    const uint8_t* mb = keepmask ? keepmask + (int64_t)b * R * H : nullptr;
    const float* Vb = V + (int64_t)mem * R * D;

    for (int h = threadIdx.x; h < H; h += FWD_THREADS) qw[h] = qv[(int64_t)b * H + h] * w[h];
    __syncthreads();

    const int H4 = H / 4;
    for (int r = wave; r < R; r += FWD_THREADS / 64) {
        float acc = 0.f;
        const float* vr = vb + (int64_t)r * H;
#pragma unroll 4
        for (int hu = lane; hu < H4; hu += 64) {
            const float4 x = reinterpret_cast<const float4*>(vr)[hu];
            const float4 q = reinterpret_cast<const float4*>(qw)[hu];
            if (mb != nullptr) {
                const uchar4 m = reinterpret_cast<const uchar4*>(mb + (int64_t)r * H)[hu];
                acc += (x.x * q.x * m.x + x.y * q.y * m.y + x.z * q.z * m.z + x.w * q.w * m.w) * inv_keep;
            } else {
                acc += x.x * q.x + x.y * q.y + x.z * q.z + x.w * q.w;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) s[r] = acc + bias[0];
    }
    __syncthreads();

    if (wave == 0) {
        const int n_valid = nb[mem];
        float mx = -INFINITY;
        for (int r = lane; r < R; r += 64) {
            const float x = (r < n_valid) ? s[r] : -INFINITY;
            s[r] = x;
            mx = fmaxf(mx, x);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int r = lane; r < R; r += 64) {
            const float e = expf(s[r] - mx);  // all -inf (nb == 0) -> NaN, like TF
            s[r] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int r = lane; r < R; r += 64) {
            const float a = s[r] / sum;
            s[r] = a;
            att_out[(int64_t)b * R + r] = a;
        }
    }
    __syncthreads();

    const int D4 = D / 4;
    for (int du = threadIdx.x; du < D4; du += FWD_THREADS) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 6
        for (int r = 0; r < R; ++r) {
            const float a = s[r];
            const float4 x = reinterpret_cast<const float4*>(Vb + (int64_t)r * D)[du];
            acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w;
        }
        reinterpret_cast<float4*>(pooled + (int64_t)b * D)[du] = acc;
    }
}

// The same for the shapes of the models (H a multiple of 256 up to 1024, D a multiple of 2048 up to 4096, R <= 40),
// restructured around loads in flight.  The generic kernel above streams 480 KB per sample in three dependent phases
// whose workgroups run in lockstep (all 512 are resident at once), so the memory system idles at every phase
// boundary: 251.9 MB in 64.5 us = 3.9 TB/s at bs 512.  Here (a) each thread fetches its first PF rows of V before
// anything else -- they do not depend on the scores and land during the score phase, (b) a wave issues the loads of
// all rows of a score batch (3 + 2 of its <= 5 rows) before the first use, so the batch costs one memory latency
// instead of one per row, (c) the pooling loop keeps 7 rows per thread in flight.
// (hipcc note: the loads are written on a native 4-float vector type and are UNCONDITIONAL -- row indices are clamped
// and rows past R get weight zero.  With HIP's float4 struct a guarded `in ? load : zero` becomes four scalar loads
// in four branches, each waited for before the next is issued.)
#ifndef VQA_ATTN_PF
#define VQA_ATTN_PF 8
#endif
#ifndef VQA_ATTN_POOL_BATCH
#define VQA_ATTN_POOL_BATCH 7
#endif
constexpr int FAST_PF = VQA_ATTN_PF, FAST_POOL_BATCH = VQA_ATTN_POOL_BATCH;
typedef float f32x4v __attribute__((ext_vector_type(4)));

// scores of CNT rows (row0, row0 + 8, ...) of one wave: every load of the batch is issued before the first use
template <int H4L, int CNT, bool MASK>
__device__ __forceinline__ void attn_score_rows(const f32x4v* __restrict__ vb4, const unsigned* __restrict__ mb4,
                                                const f32x4v* qw4, float* s, int R, int row0, float inv_keep, float bias0,
                                                int lane, bool sync_first) {
    constexpr int H4 = H4L * 64;
    f32x4v x[CNT][H4L];
    unsigned m[CNT][H4L];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int r = min(row0 + 8 * i, R - 1);
#pragma unroll
        for (int k = 0; k < H4L; ++k) {
            x[i][k] = vb4[(int64_t)r * H4 + lane + 64 * k];
            if (MASK) m[i][k] = mb4[(int64_t)r * H4 + lane + 64 * k];
        }
    }
    if (sync_first) __syncthreads();                 // qw is complete
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < H4L; ++k) {
            const f32x4v q = qw4[lane + 64 * k];
            if (MASK) {
                const unsigned mm = m[i][k];
                acc += (x[i][k].x * q.x * (float)(mm & 0xFFu) + x[i][k].y * q.y * (float)((mm >> 8) & 0xFFu) +
                        x[i][k].z * q.z * (float)((mm >> 16) & 0xFFu) + x[i][k].w * q.w * (float)(mm >> 24)) * inv_keep;
            } else {
                acc += x[i][k].x * q.x + x[i][k].y * q.y + x[i][k].z * q.z + x[i][k].w * q.w;
            }
        }
        acc = wave_sum(acc);
        if (lane == 0 && row0 + 8 * i < R) s[row0 + 8 * i] = acc + bias0;
    }
}

template <int H4L, int D4T, bool MASK>
__global__ __launch_bounds__(FWD_THREADS, 4) void attn_pool_fwd_fast_kernel(
    const float* __restrict__ v, const float* __restrict__ qv, const float* __restrict__ V,
    const int32_t* __restrict__ nb, const float* __restrict__ w, const float* __restrict__ bias,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ att_out, float* __restrict__ pooled, int R,
    int rep) {
    constexpr int H = H4L * 256, D = D4T * 2048, D4 = D / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // qw[H] | s[R]
    float* qw = lds;
    float* s = lds + H;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // b = query index
    const int mem = b / rep;
    const f32x4v* vb4 = reinterpret_cast<const f32x4v*>(v + (int64_t)mem * R * H);
    const unsigned* mb4 = MASK ? reinterpret_cast<const unsigned*>(keepmask + (int64_t)b * R * H) : nullptr;
    const f32x4v* Vb4 = reinterpret_cast<const f32x4v*>(V + (int64_t)mem * R * D);

    // (a) first rows of this thread's V columns
    f32x4v xv[FAST_PF][D4T];
#pragma unroll
    for (int j = 0; j < FAST_PF; ++j)
#pragma unroll
        for (int c = 0; c < D4T; ++c) xv[j][c] = Vb4[(int64_t)min(j, R - 1) * D4 + threadIdx.x + c * FWD_THREADS];

    for (int h = threadIdx.x; h < H; h += FWD_THREADS) qw[h] = qv[(int64_t)b * H + h] * w[h];
    const float bias0 = bias[0];

    // (b) scores: wave `wave` owns rows wave + 8 i, i < 5 (R <= 40)
    const f32x4v* qw4 = reinterpret_cast<const f32x4v*>(qw);
    attn_score_rows<H4L, 3, MASK>(vb4, mb4, qw4, s, R, wave, inv_keep, bias0, lane, true);
    attn_score_rows<H4L, 2, MASK>(vb4, mb4, qw4, s, R, wave + 24, inv_keep, bias0, lane, false);
    __syncthreads();

    if (wave == 0) {
        const int n_valid = nb[mem];
        float mx = -INFINITY;
        for (int r = lane; r < R; r += 64) {
            const float x = (r < n_valid) ? s[r] : -INFINITY;
            s[r] = x;
            mx = fmaxf(mx, x);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int r = lane; r < R; r += 64) {
            const float e = expf(s[r] - mx);  // all -inf (nb == 0) -> NaN, like TF
            s[r] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int r = lane; r < R; r += 64) {
            const float a = s[r] / sum;
            s[r] = a;
            att_out[(int64_t)b * R + r] = a;
        }
    }
    __syncthreads();

    // (c) pooling, rows in order
    f32x4v acc[D4T];
#pragma unroll
    for (int c = 0; c < D4T; ++c) acc[c] = (f32x4v)(0.f);
#pragma unroll
    for (int j = 0; j < FAST_PF; ++j) {
        const float a = (j < R) ? s[min(j, R - 1)] : 0.f;
#pragma unroll
        for (int c = 0; c < D4T; ++c) acc[c] += a * xv[j][c];
    }
    for (int r0 = FAST_PF; r0 < R; r0 += FAST_POOL_BATCH) {
        f32x4v y[FAST_POOL_BATCH][D4T];
#pragma unroll
        for (int j = 0; j < FAST_POOL_BATCH; ++j)
#pragma unroll
            for (int c = 0; c < D4T; ++c) y[j][c] = Vb4[(int64_t)min(r0 + j, R - 1) * D4 + threadIdx.x + c * FWD_THREADS];
#pragma unroll
        for (int j = 0; j < FAST_POOL_BATCH; ++j) {
            const float a = (r0 + j < R) ? s[min(r0 + j, R - 1)] : 0.f;
#pragma unroll
            for (int c = 0; c < D4T; ++c) acc[c] += a * y[j][c];
        }
    }
#pragma unroll
    for (int c = 0; c < D4T; ++c)
        reinterpret_cast<f32x4v*>(pooled + (int64_t)b * D)[threadIdx.x + c * FWD_THREADS] = acc[c];
}

// `REP` queries per memory (the pre-training model: 5 key boxes per image), one workgroup per MEMORY: v and V are read
// once per image instead of once per query (the per-query form re-reads them REP times through L2: 1.2 GB of requests
// for 0.32 GB of distinct bytes at 512 images), only the keep masks are per query.  Same three phases as above; a
// wave scores its rows for all REP queries from one load of the row, waves 0..REP-1 run the REP softmaxes, and the
// pooling loop keeps REP accumulators per thread over one stream of V.
constexpr int REP_PF = 4, REP_POOL_BATCH = 6;
template <int H4L, int CNT, bool MASK, int REP>
__device__ __forceinline__ void attn_score_rows_rep(const f32x4v* __restrict__ vb4, const unsigned* __restrict__ mb4,
                                                    const f32x4v* qw4, float* s, int R, int row0, float inv_keep,
                                                    float bias0, int lane, bool sync_first) {
    constexpr int H4 = H4L * 64;
    f32x4v x[CNT][H4L];
    int rr[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        rr[i] = min(row0 + 8 * i, R - 1);
#pragma unroll
        for (int k = 0; k < H4L; ++k) x[i][k] = vb4[(int64_t)rr[i] * H4 + lane + 64 * k];
    }
    if (sync_first) __syncthreads();                 // qw is complete
#pragma unroll
    for (int j = 0; j < REP; ++j) {
        unsigned m[CNT][H4L];
        if (MASK) {
#pragma unroll
            for (int i = 0; i < CNT; ++i)
#pragma unroll
                for (int k = 0; k < H4L; ++k) m[i][k] = mb4[((int64_t)j * R + rr[i]) * H4 + lane + 64 * k];
        }
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < H4L; ++k) {
                const f32x4v q = qw4[j * H4 + lane + 64 * k];
                if (MASK) {
                    const unsigned mm = m[i][k];
                    acc += (x[i][k].x * q.x * (float)(mm & 0xFFu) + x[i][k].y * q.y * (float)((mm >> 8) & 0xFFu) +
                            x[i][k].z * q.z * (float)((mm >> 16) & 0xFFu) + x[i][k].w * q.w * (float)(mm >> 24)) * inv_keep;
                } else {
                    acc += x[i][k].x * q.x + x[i][k].y * q.y + x[i][k].z * q.z + x[i][k].w * q.w;
                }
            }
            acc = wave_sum(acc);
            if (lane == 0 && row0 + 8 * i < R) s[j * 40 + row0 + 8 * i] = acc + bias0;
        }
    }
}

template <int H4L, int D4T, bool MASK, int REP>
__global__ __launch_bounds__(FWD_THREADS, 2) void attn_pool_fwd_rep_kernel(
    const float* __restrict__ v, const float* __restrict__ qv, const float* __restrict__ V,
    const int32_t* __restrict__ nb, const float* __restrict__ w, const float* __restrict__ bias,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ att_out, float* __restrict__ pooled, int R) {
    static_assert(REP <= FWD_THREADS / 64, "one softmax wave per query");
    constexpr int H = H4L * 256, D = D4T * 2048, D4 = D / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // qw[REP][H] | s[REP][40]
    float* qw = lds;
    float* s = lds + REP * H;
    const int mem = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q0 = (int64_t)mem * REP;                         // first query of this memory
    const f32x4v* vb4 = reinterpret_cast<const f32x4v*>(v + (int64_t)mem * R * H);
    const unsigned* mb4 = MASK ? reinterpret_cast<const unsigned*>(keepmask + q0 * R * H) : nullptr;
    const f32x4v* Vb4 = reinterpret_cast<const f32x4v*>(V + (int64_t)mem * R * D);

    f32x4v xv[REP_PF][D4T];
#pragma unroll
    for (int j = 0; j < REP_PF; ++j)
#pragma unroll
        for (int c = 0; c < D4T; ++c) xv[j][c] = Vb4[(int64_t)min(j, R - 1) * D4 + threadIdx.x + c * FWD_THREADS];

    for (int i = threadIdx.x; i < REP * H; i += FWD_THREADS) qw[i] = qv[q0 * H + i] * w[i % H];
    const float bias0 = bias[0];

    const f32x4v* qw4 = reinterpret_cast<const f32x4v*>(qw);
    attn_score_rows_rep<H4L, 2, MASK, REP>(vb4, mb4, qw4, s, R, wave, inv_keep, bias0, lane, true);
    attn_score_rows_rep<H4L, 2, MASK, REP>(vb4, mb4, qw4, s, R, wave + 16, inv_keep, bias0, lane, false);
    attn_score_rows_rep<H4L, 1, MASK, REP>(vb4, mb4, qw4, s, R, wave + 32, inv_keep, bias0, lane, false);
    __syncthreads();

    if (wave < REP) {
        float* sj = s + wave * 40;
        const int n_valid = nb[mem];
        const float x = (lane < R && lane < n_valid) ? sj[min(lane, R - 1)] : -INFINITY;      // R <= 40 < 64: one lane per row
        const float mx = wave_max(x);
        const float e = (lane < R) ? expf(x - mx) : 0.f;  // all -inf (nb == 0) -> NaN, like TF
        const float sum = wave_sum(e);
        if (lane < R) {
            const float a = e / sum;
            sj[lane] = a;
            att_out[(q0 + wave) * R + lane] = a;
        }
    }
    __syncthreads();

    f32x4v acc[REP][D4T];
#pragma unroll
    for (int j = 0; j < REP; ++j)
#pragma unroll
        for (int c = 0; c < D4T; ++c) acc[j][c] = (f32x4v)(0.f);
#pragma unroll
    for (int r = 0; r < REP_PF; ++r) {
#pragma unroll
        for (int j = 0; j < REP; ++j) {
            const float a = (r < R) ? s[j * 40 + min(r, R - 1)] : 0.f;
#pragma unroll
            for (int c = 0; c < D4T; ++c) acc[j][c] += a * xv[r][c];
        }
    }
    for (int r0 = REP_PF; r0 < R; r0 += REP_POOL_BATCH) {
        f32x4v y[REP_POOL_BATCH][D4T];
#pragma unroll
        for (int i = 0; i < REP_POOL_BATCH; ++i)
#pragma unroll
            for (int c = 0; c < D4T; ++c) y[i][c] = Vb4[(int64_t)min(r0 + i, R - 1) * D4 + threadIdx.x + c * FWD_THREADS];
#pragma unroll
        for (int i = 0; i < REP_POOL_BATCH; ++i) {
#pragma unroll
            for (int j = 0; j < REP; ++j) {
                const float a = (r0 + i < R) ? s[j * 40 + min(r0 + i, R - 1)] : 0.f;
#pragma unroll
                for (int c = 0; c < D4T; ++c) acc[j][c] += a * y[i][c];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < REP; ++j)
#pragma unroll
        for (int c = 0; c < D4T; ++c)
            reinterpret_cast<f32x4v*>(pooled + (q0 + j) * D)[threadIdx.x + c * FWD_THREADS] = acc[j][c];
}

// Backward.  One workgroup per MEMORY walks its `rep` queries, so dv (the gradient of the shared
// v block) is accumulated over the queries in registers and written once.  512 threads: the last
// phase gives every float4 column of v to TWO threads that take alternate rows (dv rows are
// independent; the per-query column sums are combined through LDS).
constexpr int BWD_THREADS = 512;
template <int REP>
__global__ __launch_bounds__(BWD_THREADS) void attn_pool_bwd_kernel(
    const float* __restrict__ dpooled, const float* __restrict__ v, const float* __restrict__ qv,
    const float* __restrict__ V, const float* __restrict__ att, const float* __restrict__ w,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ dv, float* __restrict__ dqv,
    float* __restrict__ part_dw, float* __restrict__ part_db, int R, int H, int D, int rep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // dp[D] | ds[REP][R] | comb[REP][H]
    float* dp = lds;
    float* ds = lds + D;
    float* comb = ds + ((REP * R + 3) / 4) * 4;
    const int mem = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = BWD_THREADS / 64;
    const float* vb = v + (int64_t)mem * R * H;
    const float* Vb = V + (int64_t)mem * R * D;
    const int D4 = D / 4;

    for (int j = 0; j < rep; ++j) {
        const int q = mem * rep + j;
        __syncthreads();
        for (int d = threadIdx.x; d < D; d += BWD_THREADS) dp[d] = dpooled[(int64_t)q * D + d];
        __syncthreads();
        // datt[r] = <dpooled[q], V[mem,r]>
        float* dsj = ds + j * R;
        for (int r = wave; r < R; r += NW) {
            float acc = 0.f;
#pragma unroll 8
            for (int du = lane; du < D4; du += 64) {
                const float4 x = reinterpret_cast<const float4*>(Vb + (int64_t)r * D)[du];
                const float4 g = reinterpret_cast<const float4*>(dp)[du];
                acc += x.x * g.x + x.y * g.y + x.z * g.z + x.w * g.w;
            }
            acc = wave_sum(acc);
            if (lane == 0) dsj[r] = acc;
        }
        __syncthreads();
        // softmax backward: ds = att * (datt - sum(att*datt))
        if (wave == 0) {
            float dot = 0.f;
            for (int r = lane; r < R; r += 64) dot += att[(int64_t)q * R + r] * dsj[r];
            dot = wave_sum(dot);
            float tot = 0.f;
            for (int r = lane; r < R; r += 64) {
                const float g = att[(int64_t)q * R + r] * (dsj[r] - dot);
                dsj[r] = g;
                tot += g;
            }
            tot = wave_sum(tot);
            if (lane == 0) part_db[q] = tot;
        }
    }
    __syncthreads();

    const int H4 = H / 4;
    const int half = threadIdx.x / (BWD_THREADS / 2), tcol = threadIdx.x % (BWD_THREADS / 2);
    for (int hu0 = 0; hu0 < H4; hu0 += BWD_THREADS / 2) {
        const int hu = hu0 + tcol;
        const bool live = hu < H4;
        float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 qj[REP], S[REP];
        if (live) ww = reinterpret_cast<const float4*>(w)[hu];
#pragma unroll
        for (int j = 0; j < REP; ++j) {
            S[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            qj[j] = (live && j < rep) ? reinterpret_cast<const float4*>(qv + (int64_t)(mem * rep + j) * H)[hu]
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (live) {
#pragma unroll 2
            for (int r = half; r < R; r += 2) {
                const float4 x = reinterpret_cast<const float4*>(vb + (int64_t)r * H)[hu];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < REP; ++j) {
                    if (j < rep) {
                        const float d = ds[j * R + r];
                        float4 g = make_float4(d, d, d, d);
                        if (keepmask != nullptr) {
                            const uchar4 m =
                                reinterpret_cast<const uchar4*>(keepmask + ((int64_t)(mem * rep + j) * R + r) * H)[hu];
                            g.x *= m.x * inv_keep; g.y *= m.y * inv_keep; g.z *= m.z * inv_keep; g.w *= m.w * inv_keep;
                        }
                        S[j].x += g.x * x.x; S[j].y += g.y * x.y; S[j].z += g.z * x.z; S[j].w += g.w * x.w;
                        acc.x += g.x * qj[j].x * ww.x; acc.y += g.y * qj[j].y * ww.y;
                        acc.z += g.z * qj[j].z * ww.z; acc.w += g.w * qj[j].w * ww.w;
                    }
                }
                reinterpret_cast<float4*>(dv + ((int64_t)mem * R + r) * H)[hu] = acc;
            }
        }
        // odd-row partial sums -> LDS, the even-row thread of the same column finishes (fixed order)
        __syncthreads();
        if (live && half == 1) {
#pragma unroll
            for (int j = 0; j < REP; ++j)
                if (j < rep) reinterpret_cast<float4*>(comb + (size_t)j * H)[hu] = S[j];
        }
        __syncthreads();
        if (live && half == 0) {
#pragma unroll
            for (int j = 0; j < REP; ++j) {
                if (j < rep) {
                    const float4 o = reinterpret_cast<const float4*>(comb + (size_t)j * H)[hu];
                    const float4 t = make_float4(S[j].x + o.x, S[j].y + o.y, S[j].z + o.z, S[j].w + o.w);
                    const int64_t q = (int64_t)mem * rep + j;
                    reinterpret_cast<float4*>(dqv + q * H)[hu] = make_float4(t.x * ww.x, t.y * ww.y, t.z * ww.z, t.w * ww.w);
                    reinterpret_cast<float4*>(part_dw + q * H)[hu] =
                        make_float4(t.x * qj[j].x, t.y * qj[j].y, t.z * qj[j].z, t.w * qj[j].w);
                }
            }
        }
    }
}

// The same backward for the models' shapes (D == 2048, H == 1024, R <= 40; 1 or 5 queries per memory), restructured around loads in
// flight like the forward: the generic kernel walks its queries one after the other (load dpooled[q], then one V row
// per wave at a time: REP x ~6 dependent memory latencies with two workgroups per CU), and its last phase keeps two
// rows per thread in flight.  Here dpooled of ALL queries of the memory sits in LDS, a wave reads each of its V rows
// ONCE (two rows in flight) and scores it against the REP queries, the REP softmax backwards run in REP waves, and
// the dv phase keeps four rows (+ their REP mask words) per thread in flight.  Same per-lane summation order.
template <int REP, bool MASK>
__global__ __launch_bounds__(BWD_THREADS) void attn_pool_bwd_fast_kernel(
    const float* __restrict__ dpooled, const float* __restrict__ v, const float* __restrict__ qv,
    const float* __restrict__ V, const float* __restrict__ att, const float* __restrict__ w,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ dv, float* __restrict__ dqv,
    float* __restrict__ part_dw, float* __restrict__ part_db, int R, int H) {
    constexpr int D = 2048, D4 = D / 4, DL = D4 / 64, NW = BWD_THREADS / 64, RB = 4;
    static_assert(REP <= NW, "one softmax wave per query");
    extern __shared__ __attribute__((aligned(16))) float lds[];  // dp[REP][D] | ds[REP][40] | comb[REP][H]
    float* ds = lds + REP * D;
    float* comb = ds + REP * 40;
    const int mem = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q0 = (int64_t)mem * REP;
    f32x4v* dp4 = reinterpret_cast<f32x4v*>(lds);
    {
        const f32x4v* g4 = reinterpret_cast<const f32x4v*>(dpooled + q0 * D);
        f32x4v t[REP];
#pragma unroll
        for (int j = 0; j < REP; ++j) t[j] = g4[j * D4 + threadIdx.x];       // D4 == BWD_THREADS
#pragma unroll
        for (int j = 0; j < REP; ++j) dp4[j * D4 + threadIdx.x] = t[j];
    }
    __syncthreads();
    // datt[j][r] = <dpooled[q0 + j], V[mem, r]>
    const f32x4v* Vb4 = reinterpret_cast<const f32x4v*>(V + (int64_t)mem * R * D);
    for (int r0 = wave; r0 < R; r0 += 2 * NW) {
        const int rb = min(r0 + NW, R - 1);
        const bool okb = r0 + NW < R;
        f32x4v xa[DL], xb[DL];
#pragma unroll
        for (int k = 0; k < DL; ++k) {
            xa[k] = Vb4[(int64_t)r0 * D4 + lane + 64 * k];
            xb[k] = Vb4[(int64_t)rb * D4 + lane + 64 * k];
        }
#pragma unroll
        for (int j = 0; j < REP; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < DL; ++k) {
                const f32x4v g = dp4[j * D4 + lane + 64 * k];
                a += xa[k].x * g.x + xa[k].y * g.y + xa[k].z * g.z + xa[k].w * g.w;
                b += xb[k].x * g.x + xb[k].y * g.y + xb[k].z * g.z + xb[k].w * g.w;
            }
            a = wave_sum(a);
            b = wave_sum(b);
            if (lane == 0) {
                ds[j * 40 + r0] = a;
                if (okb) ds[j * 40 + rb] = b;
            }
        }
    }
    __syncthreads();
    // softmax backward: ds = att * (datt - sum(att*datt)); R <= 40 < 64: one lane per region
    if (wave < REP) {
        const int64_t q = q0 + wave;
        const float a = lane < R ? att[q * R + lane] : 0.f;
        const float d = lane < R ? ds[wave * 40 + lane] : 0.f;
        const float dot = wave_sum(a * d);
        const float g = a * (d - dot);
        if (lane < R) ds[wave * 40 + lane] = g;
        const float tot = wave_sum(g);
        if (lane == 0) part_db[q] = tot;
    }
    __syncthreads();

    const int H4 = H / 4;
    const int half = threadIdx.x / (BWD_THREADS / 2), tcol = threadIdx.x % (BWD_THREADS / 2);
    const f32x4v* vb4 = reinterpret_cast<const f32x4v*>(v + (int64_t)mem * R * H);
    const unsigned* mk4 = MASK ? reinterpret_cast<const unsigned*>(keepmask + q0 * R * H) : nullptr;
    f32x4v* dv4 = reinterpret_cast<f32x4v*>(dv + (int64_t)mem * R * H);
    for (int hu0 = 0; hu0 < H4; hu0 += BWD_THREADS / 2) {       // H == 1024: one pass, every thread has a column
        const int hu = hu0 + tcol;
        const f32x4v ww = reinterpret_cast<const f32x4v*>(w)[hu];
        f32x4v qj[REP], S[REP];
#pragma unroll
        for (int j = 0; j < REP; ++j) {
            S[j] = (f32x4v)(0.f);
            qj[j] = reinterpret_cast<const f32x4v*>(qv + (q0 + j) * H)[hu];
        }
        for (int rb = half; rb < R; rb += 2 * RB) {
            f32x4v x[RB];
            unsigned m[RB][REP];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = min(rb + 2 * i, R - 1);
                x[i] = vb4[(int64_t)r * H4 + hu];
                if (MASK) {
#pragma unroll
                    for (int j = 0; j < REP; ++j) m[i][j] = mk4[((int64_t)j * R + r) * H4 + hu];
                }
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = rb + 2 * i;
                if (r >= R) break;                                      // workgroup-uniform per half; no barrier inside
                f32x4v acc = (f32x4v)(0.f);
#pragma unroll
                for (int j = 0; j < REP; ++j) {
                    const float d = ds[j * 40 + r];
                    f32x4v g = (f32x4v)(d);
                    if (MASK) {
                        const unsigned mm = m[i][j];
                        g.x *= (float)(mm & 0xFFu) * inv_keep; g.y *= (float)((mm >> 8) & 0xFFu) * inv_keep;
                        g.z *= (float)((mm >> 16) & 0xFFu) * inv_keep; g.w *= (float)(mm >> 24) * inv_keep;
                    }
                    S[j].x += g.x * x[i].x; S[j].y += g.y * x[i].y; S[j].z += g.z * x[i].z; S[j].w += g.w * x[i].w;
                    acc.x += g.x * qj[j].x * ww.x; acc.y += g.y * qj[j].y * ww.y;
                    acc.z += g.z * qj[j].z * ww.z; acc.w += g.w * qj[j].w * ww.w;
                }
                dv4[(int64_t)r * H4 + hu] = acc;
            }
        }
        // odd-row partial sums -> LDS, the even-row thread of the same column finishes (fixed order)
        __syncthreads();
        if (half == 1) {
#pragma unroll
            for (int j = 0; j < REP; ++j) reinterpret_cast<f32x4v*>(comb + (size_t)j * H)[hu] = S[j];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < REP; ++j) {
                const f32x4v o = reinterpret_cast<const f32x4v*>(comb + (size_t)j * H)[hu];
                const f32x4v t = S[j] + o;
                reinterpret_cast<f32x4v*>(dqv + (q0 + j) * H)[hu] = t * ww;
                reinterpret_cast<f32x4v*>(part_dw + (q0 + j) * H)[hu] = t * qj[j];
            }
        }
    }
}

int g_attn_fast = 1;   // tuning / A-B switch (vqa_attn_set_fast)

}  // namespace

extern "C" int vqa_attn_set_fast(int on) {
    // 0 generic; 1 (default) fast kernel for one query per memory and the per-memory kernel for rep 5; 2 per-query fast
    // kernel for the other reps as well; 3 per-query fast kernel for every rep (A/B of the per-memory kernel)
    g_attn_fast = on;
    return VQA_OK;
}

extern "C" int vqa_attn_pool_fwd(const float* v, const float* qv, const float* V, const int32_t* nb, const float* w,
                                 const float* bias, const uint8_t* keepmask, float keep_prob, float* att,
                                 float* pooled, int B, int R, int H, int D, void* stream) {
    return vqa_attn_pool_fwd_rep(v, qv, V, nb, w, bias, keepmask, keep_prob, att, pooled, B, 1, R, H, D, stream);
}

extern "C" int vqa_attn_pool_fwd_rep(const float* v, const float* qv, const float* V, const int32_t* nb,
                                     const float* w, const float* bias, const uint8_t* keepmask, float keep_prob,
                                     float* att, float* pooled, int B, int rep, int R, int H, int D, void* stream) {
    VQA_REQUIRE(rep >= 1 && rep <= 8, VQA_ERR_ARG);
    VQA_REQUIRE(v && qv && V && nb && w && bias && att && pooled, VQA_ERR_ARG);
    VQA_REQUIRE(B >= 0 && R > 0 && R <= MAX_R && H > 0 && D > 0, VQA_ERR_ARG);
    VQA_REQUIRE(keepmask == nullptr || keep_prob > 0.f, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0 && D % 4 == 0, VQA_ERR_ALIGN);
    VQA_REQUIRE(vqa_aligned16(v) && vqa_aligned16(V) && vqa_aligned16(pooled), VQA_ERR_ALIGN);
    VQA_REQUIRE(keepmask == nullptr || (reinterpret_cast<uintptr_t>(keepmask) & 3u) == 0, VQA_ERR_ALIGN);
    if (B == 0) return VQA_OK;
    const size_t lds = (size_t)(H + R) * sizeof(float);
    const float ik = keepmask ? 1.f / keep_prob : 1.f;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = g_attn_fast && (rep == 1 || rep == 5 || g_attn_fast > 1) && R <= 40 && H % 256 == 0 && H <= 1024 && D % 2048 == 0 && D <= 4096 &&
                      vqa_aligned16(qv) && vqa_aligned16(w);
    if (fast && rep == 5 && g_attn_fast != 3) {
        // one workgroup per memory for the pre-training model's 5 queries per image
        const size_t lds5 = (size_t)(5 * H + 5 * 40) * sizeof(float);
#define VQA_ATTN_REP5(h4l, d4t)                                                                                        \
    do {                                                                                                                \
        if (keepmask != nullptr)                                                                                        \
            hipLaunchKernelGGL((attn_pool_fwd_rep_kernel<h4l, d4t, true, 5>), dim3(B), dim3(FWD_THREADS), lds5, st, v,  \
                               qv, V, nb, w, bias, keepmask, ik, att, pooled, R);                                       \
        else                                                                                                            \
            hipLaunchKernelGGL((attn_pool_fwd_rep_kernel<h4l, d4t, false, 5>), dim3(B), dim3(FWD_THREADS), lds5, st, v, \
                               qv, V, nb, w, bias, keepmask, ik, att, pooled, R);                                       \
    } while (0)
        const int h4l = H / 256, d4t = D / 2048;
        if (d4t == 1) {
            if (h4l == 1) VQA_ATTN_REP5(1, 1); else if (h4l == 2) VQA_ATTN_REP5(2, 1);
            else if (h4l == 3) VQA_ATTN_REP5(3, 1); else VQA_ATTN_REP5(4, 1);
        } else {
            if (h4l == 1) VQA_ATTN_REP5(1, 2); else if (h4l == 2) VQA_ATTN_REP5(2, 2);
            else if (h4l == 3) VQA_ATTN_REP5(3, 2); else VQA_ATTN_REP5(4, 2);
        }
#undef VQA_ATTN_REP5
    } else if (fast) {
#define VQA_ATTN_FAST(h4l, d4t)                                                                                         \
    do {                                                                                                                \
        if (keepmask != nullptr)                                                                                        \
            hipLaunchKernelGGL((attn_pool_fwd_fast_kernel<h4l, d4t, true>), dim3(B * rep), dim3(FWD_THREADS), lds, st, v, \
                               qv, V, nb, w, bias, keepmask, ik, att, pooled, R, rep);                                  \
        else                                                                                                            \
            hipLaunchKernelGGL((attn_pool_fwd_fast_kernel<h4l, d4t, false>), dim3(B * rep), dim3(FWD_THREADS), lds, st, v, \
                               qv, V, nb, w, bias, keepmask, ik, att, pooled, R, rep);                                  \
    } while (0)
        const int h4l = H / 256, d4t = D / 2048;
        if (d4t == 1) {
            if (h4l == 1) VQA_ATTN_FAST(1, 1); else if (h4l == 2) VQA_ATTN_FAST(2, 1);
            else if (h4l == 3) VQA_ATTN_FAST(3, 1); else VQA_ATTN_FAST(4, 1);
        } else {
            if (h4l == 1) VQA_ATTN_FAST(1, 2); else if (h4l == 2) VQA_ATTN_FAST(2, 2);
            else if (h4l == 3) VQA_ATTN_FAST(3, 2); else VQA_ATTN_FAST(4, 2);
        }
#undef VQA_ATTN_FAST
    } else {
        hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3(B * rep), dim3(FWD_THREADS), lds, st, v, qv, V, nb, w, bias,
                           keepmask, ik, att, pooled, R, H, D, rep);
    }
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_attn_pool_bwd(const float* dpooled, const float* v, const float* qv, const float* V,
                                 const float* att, const float* w, const uint8_t* keepmask, float keep_prob, float* dv,
                                 float* dqv, float* part_dw, float* part_db, int B, int R, int H, int D, void* stream) {
    return vqa_attn_pool_bwd_rep(dpooled, v, qv, V, att, w, keepmask, keep_prob, dv, dqv, part_dw, part_db, B, 1, R, H,
                                 D, stream);
}

extern "C" int vqa_attn_pool_bwd_rep(const float* dpooled, const float* v, const float* qv, const float* V,
                                     const float* att, const float* w, const uint8_t* keepmask, float keep_prob,
                                     float* dv, float* dqv, float* part_dw, float* part_db, int B, int rep, int R,
                                     int H, int D, void* stream) {
    VQA_REQUIRE(rep >= 1 && rep <= 8, VQA_ERR_ARG);
    VQA_REQUIRE(dpooled && v && qv && V && att && w && dv && dqv && part_dw && part_db, VQA_ERR_ARG);
    VQA_REQUIRE(B >= 0 && R > 0 && R <= MAX_R && H > 0 && D > 0, VQA_ERR_ARG);
    VQA_REQUIRE(keepmask == nullptr || keep_prob > 0.f, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0 && D % 4 == 0, VQA_ERR_ALIGN);
    VQA_REQUIRE(vqa_aligned16(v) && vqa_aligned16(V) && vqa_aligned16(dv) && vqa_aligned16(dqv) &&
                    vqa_aligned16(part_dw) && vqa_aligned16(qv) && vqa_aligned16(w),
                VQA_ERR_ALIGN);
    VQA_REQUIRE(keepmask == nullptr || (reinterpret_cast<uintptr_t>(keepmask) & 3u) == 0, VQA_ERR_ALIGN);
    if (B == 0) return VQA_OK;
    const float ik = keepmask ? 1.f / keep_prob : 1.f;
    hipStream_t st = (hipStream_t)stream;
    auto lds_for = [&](int REPt) { return (size_t)(D + ((REPt * R + 3) / 4) * 4 + REPt * H) * sizeof(float); };
    VQA_REQUIRE(lds_for(rep == 1 ? 1 : rep <= 5 ? 5 : 8) <= 64 * 1024, VQA_ERR_UNSUPPORTED);
    if (g_attn_fast && (rep == 1 || rep == 5) && D == 2048 && H == 1024 && R <= 40 && vqa_aligned16(dpooled)) {
        const size_t l = (size_t)(rep * D + rep * 40 + rep * H) * sizeof(float);
#define VQA_ATTN_BWD_FAST(r, mk)                                                                                       \
    hipLaunchKernelGGL((attn_pool_bwd_fast_kernel<r, mk>), dim3(B), dim3(BWD_THREADS), l, st, dpooled, v, qv, V, att, w, \
                       keepmask, ik, dv, dqv, part_dw, part_db, R, H)
        if (rep == 1) { if (keepmask) VQA_ATTN_BWD_FAST(1, true); else VQA_ATTN_BWD_FAST(1, false); }
        else { if (keepmask) VQA_ATTN_BWD_FAST(5, true); else VQA_ATTN_BWD_FAST(5, false); }
#undef VQA_ATTN_BWD_FAST
        VQA_CHECK_LAUNCH();
        return VQA_OK;
    }
    if (rep == 1)
        hipLaunchKernelGGL(attn_pool_bwd_kernel<1>, dim3(B), dim3(BWD_THREADS), lds_for(1), st, dpooled, v, qv, V, att, w,
                           keepmask, ik, dv, dqv, part_dw, part_db, R, H, D, rep);
    else if (rep <= 5)
        hipLaunchKernelGGL(attn_pool_bwd_kernel<5>, dim3(B), dim3(BWD_THREADS), lds_for(5), st, dpooled, v, qv, V, att, w,
                           keepmask, ik, dv, dqv, part_dw, part_db, R, H, D, rep);
    else
        hipLaunchKernelGGL(attn_pool_bwd_kernel<8>, dim3(B), dim3(BWD_THREADS), lds_for(8), st, dpooled, v, qv, V, att, w,
                           keepmask, ik, dv, dqv, part_dw, part_db, R, H, D, rep);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
