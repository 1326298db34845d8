// Kernels of the five older model_vlmap_answer ablations: everything the base model's kernels do
// not already cover.  All HBM-bound row kernels, fp32, one workgroup per sample.
//   vlmap_answer_full   vqa/model_vlmap_answer_full.py:124-134, 272-276   VAE reparameterisation + KL row sums
//   vlmap_answer_adapt  vqa/model_vlmap_answer_adapt.py:132-142            d v_adapt = att (x) d pooled
//   vlmap_answer_ent    vqa/model_vlmap_answer_ent.py:191-211, 281-292     pairing product, softmax marginal, entropy
#include <algorithm>

#include "vqa_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// x = mean + noise * sqrt(exp(ls));  kl_row[b] = -0.5 * sum_h (1 + ls - mean^2 - exp(ls))
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ ls,
                                                          const float* __restrict__ noise, float* __restrict__ x,
                                                          float* __restrict__ kl_row, int H) {
    __shared__ float red[16];
    const int64_t o = (int64_t)blockIdx.x * H;
    float acc = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) {
        const float m = mean[o + h], l = ls[o + h];
        const float e = expf(l);
        x[o + h] = m + noise[o + h] * sqrtf(e);
        acc += 1.f + l - m * m - e;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) kl_row[blockIdx.x] = -0.5f * acc;
}

// d mean = dx + c * mean;  d ls = dx * noise * 0.5 * sqrt(exp(ls)) + c * 0.5 * (exp(ls) - 1),  c = weight / global batch
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ mean,
                                                          const float* __restrict__ ls, const float* __restrict__ noise,
                                                          float c, float* __restrict__ dmean, float* __restrict__ dls,
                                                          int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = dx[i], e = expf(ls[i]);
        dmean[i] = g + c * mean[i];
        dls[i] = g * noise[i] * 0.5f * sqrtf(e) + c * 0.5f * (e - 1.f);
    }
}

// out[b, r, :] = att[b, r] * dp[b, :]      (gradient of pooled = sum_r att * v_adapt wrt v_adapt)
__global__ __launch_bounds__(256) void outer_rows_kernel(const float* __restrict__ att, const float* __restrict__ dp,
                                                         float* __restrict__ out, int R, int H) {
    const int b = blockIdx.x;
    const float* a = att + (int64_t)b * R;
    const float* d = dp + (int64_t)b * H;
    float* o = out + (int64_t)b * R * H;
    if ((H & 3) == 0) {
        const int H4 = H >> 2;
        for (int i = threadIdx.x; i < R * H4; i += 256) {
            const int r = i / H4, h4 = i - r * H4;
            const f4 g = reinterpret_cast<const f4*>(d)[h4];
            reinterpret_cast<f4*>(o)[i] = g * a[r];
        }
    } else {
        for (int i = threadIdx.x; i < R * H; i += 256) o[i] = a[i / H] * d[i % H];
    }
}

// x[(b, m), :] = pl[(b * M + m) % B, :] * ll[b, :]     (tf.tile([M, 1]) + reshape of vqa/model_vlmap_answer_ent.py:196-199)
__global__ __launch_bounds__(256) void tile_mul_fwd_kernel(const float* __restrict__ pl, const float* __restrict__ ll,
                                                           float* __restrict__ x, int B, int M, int H) {
    const int64_t row = blockIdx.x;                       // b * M + m
    const int b = (int)(row / M), src = (int)(row % B);
    const float* p = pl + (int64_t)src * H;
    const float* l = ll + (int64_t)b * H;
    float* o = x + row * H;
    if ((H & 3) == 0) {
        for (int h4 = threadIdx.x; h4 < (H >> 2); h4 += 256)
            reinterpret_cast<f4*>(o)[h4] = reinterpret_cast<const f4*>(p)[h4] * reinterpret_cast<const f4*>(l)[h4];
    } else {
        for (int h = threadIdx.x; h < H; h += 256) o[h] = p[h] * l[h];
    }
}

// dll[b, h] (+)= sum_m dx[(b, m), h] * pl[(b * M + m) % B, h]     (pl is behind tf.stop_gradient: no d pl)
__global__ __launch_bounds__(256) void tile_mul_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ pl,
                                                           float* __restrict__ dll, int B, int M, int H, int accumulate) {
    const int b = blockIdx.x;
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) {
            const int64_t row = (int64_t)b * M + m;
            acc += dx[row * H + h] * pl[(int64_t)(row % B) * H + h];
        }
        const int64_t o = (int64_t)b * H + h;
        dll[o] = accumulate ? dll[o] + acc : acc;
    }
}

// one block-wide reduction behind ONE barrier: per-wave partials into one of two alternating LDS slots
template <bool MAX>
__device__ __forceinline__ float block_red(float v, float (*red)[4], int& slot) {
    v = MAX ? wave_max(v) : wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[slot][threadIdx.x >> 6] = v;
    __syncthreads();
    const float* r = red[slot];
    slot ^= 1;
    return MAX ? fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])) : ((r[0] + r[1]) + r[2]) + r[3];
}

// The marginal-entropy regulariser on the pairings' logits tz [B * M, ldz] (in place):
//   prob = softmax over the selected answers (sel[a] = train[a] * exist[a] > 0.5, a < cols) of every pairing
//   marginal[b, a] = mean_m prob[(b, m), a];  ent_row[b] = sum_a marginal * log(marginal + 1e-8)
//   want_dz: tz <- d loss / d logit = prob * (dprob - sum_a prob * dprob),
//            dprob[a] = coef * (log(marginal + 1e-8) + marginal / (marginal + 1e-8)) / M;   otherwise tz <- prob
// One workgroup per question; thread t owns columns t, t + 256, ... of every row (CPT of them in registers).
template <int CPT>
__global__ __launch_bounds__(256) void marginal_entropy_kernel(float* __restrict__ tz, const float* __restrict__ train,
                                                               const float* __restrict__ exist, float coef,
                                                               float* __restrict__ marginal, float* __restrict__ ent_row,
                                                               int M, int cols, int ldz, int want_dz) {
    __shared__ float red[2][4];
    int slot = 0;
    const int b = blockIdx.x, t = threadIdx.x;
    bool sel[CPT];
    float marg[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = t + 256 * i;
        sel[i] = c < cols && train[c < cols ? c : 0] * exist[c < cols ? c : 0] > 0.5f;
        marg[i] = 0.f;
    }
    float* base = tz + (int64_t)b * M * ldz;
    for (int m = 0; m < M; ++m) {
        float* row = base + (int64_t)m * ldz;
        float v[CPT];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = t + 256 * i;
            v[i] = sel[i] ? row[c] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = block_red<true>(mx, red, slot);
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            v[i] = sel[i] ? expf(v[i] - mx) : 0.f;
            sm += v[i];
        }
        sm = block_red<false>(sm, red, slot);
        const float inv = 1.f / sm;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = t + 256 * i;
            const float p = v[i] * inv;
            marg[i] += p;
            if (c < cols) row[c] = p;
        }
    }
    const float inv_m = 1.f / (float)M;
    float ent = 0.f;
    float dprob[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = t + 256 * i;
        const float p = marg[i] * inv_m;
        const float lg = logf(p + 1e-8f);
        if (sel[i]) ent += p * lg;
        dprob[i] = sel[i] ? coef * inv_m * (lg + p / (p + 1e-8f)) : 0.f;
        if (c < cols) marginal[(int64_t)b * cols + c] = p;
    }
    ent = block_red<false>(ent, red, slot);
    if (t == 0) ent_row[b] = ent;
    if (!want_dz) return;
    for (int m = 0; m < M; ++m) {
        float* row = base + (int64_t)m * ldz;
        float p[CPT];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = t + 256 * i;
            p[i] = c < cols ? row[c] : 0.f;          // this thread's own stores of the first pass
            s += p[i] * dprob[i];
        }
        s = block_red<false>(s, red, slot);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = t + 256 * i;
            if (c < cols) row[c] = p[i] * (dprob[i] - s);
        }
    }
}

// Counter-based standard-normal draws (the explicit stand-in for tf.random_normal(seed=123) of the reparameterisation):
// element i depends only on (seed, offset + i) -- Box-Muller on two 24-bit uniforms of one splitmix64 hash.
__device__ __forceinline__ uint64_t mix64n(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void normal_noise_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    const uint64_t key = mix64n(seed ^ 0x6E6F726D616Cull);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t r = mix64n(key ^ (offset + (uint64_t)i));
        const float u1 = ((float)((r >> 40) & 0xFFFFFFu) + 1.0f) * (1.0f / 16777216.0f);       // (0, 1]
        const float u2 = (float)((r >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);                 // [0, 1)
        out[i] = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
}

// report[13] = mean_b extra_row, [14] = weight * that, [15] = report[0] + [14]; stats[b, 15] = extra_row[b]
__global__ __launch_bounds__(256) void extra_report_kernel(const float* __restrict__ extra_row, float* __restrict__ stats,
                                                           int B, float weight, float* __restrict__ report) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float v = extra_row[b];
        stats[(int64_t)b * VQA_STAT_COUNT + 15] = v;
        acc += v;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float mean = acc / (float)B;
        report[13] = mean;
        report[14] = weight * mean;
        report[15] = report[0] + weight * mean;
    }
}

}  // namespace

extern "C" int vqa_reparam_fwd(const float* mean, const float* log_sigma_sq, const float* noise, float* x, float* kl_row,
                               int B, int H, void* stream) {
    VQA_REQUIRE(mean && log_sigma_sq && noise && x && kl_row && B >= 0 && H > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mean, log_sigma_sq, noise, x, kl_row, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_reparam_bwd(const float* dx, const float* mean, const float* log_sigma_sq, const float* noise, float coef,
                               float* dmean, float* dlog_sigma_sq, int64_t n, void* stream) {
    VQA_REQUIRE(dx && mean && log_sigma_sq && noise && dmean && dlog_sigma_sq && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dx, mean, log_sigma_sq, noise, coef,
                       dmean, dlog_sigma_sq, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_outer_rows(const float* att, const float* dp, float* out, int B, int R, int H, void* stream) {
    VQA_REQUIRE(att && dp && out && B >= 0 && R > 0 && H > 0, VQA_ERR_ARG);
    VQA_REQUIRE((H & 3) != 0 || (vqa_aligned16(dp) && vqa_aligned16(out)), VQA_ERR_ALIGN);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(outer_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, att, dp, out, R, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_tile_mul_fwd(const float* pl, const float* ll, float* x, int B, int M, int H, void* stream) {
    VQA_REQUIRE(pl && ll && x && B >= 0 && M > 0 && H > 0, VQA_ERR_ARG);
    VQA_REQUIRE((H & 3) != 0 || (vqa_aligned16(pl) && vqa_aligned16(ll) && vqa_aligned16(x)), VQA_ERR_ALIGN);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(tile_mul_fwd_kernel, dim3((unsigned)((int64_t)B * M)), dim3(256), 0, (hipStream_t)stream, pl, ll, x, B, M, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_tile_mul_bwd(const float* dx, const float* pl, float* dll, int B, int M, int H, int accumulate,
                                void* stream) {
    VQA_REQUIRE(dx && pl && dll && B >= 0 && M > 0 && H > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(tile_mul_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dx, pl, dll, B, M, H, accumulate);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_marginal_entropy(float* tz, const float* train_mask, const float* exist_mask, float coef, float* marginal,
                                    float* ent_row, int B, int M, int cols, int ldz, int want_dz, void* stream) {
    VQA_REQUIRE(tz && train_mask && exist_mask && marginal && ent_row && B >= 0 && M > 0 && cols > 0 && ldz >= cols,
                VQA_ERR_ARG);
    VQA_REQUIRE(cols <= 16 * 256, VQA_ERR_UNSUPPORTED);
    if (B == 0) return VQA_OK;
    hipStream_t st = (hipStream_t)stream;
#define VQA_ME(cpt)                                                                                                      \
    hipLaunchKernelGGL(marginal_entropy_kernel<cpt>, dim3(B), dim3(256), 0, st, tz, train_mask, exist_mask, coef, marginal, \
                       ent_row, M, cols, ldz, want_dz)
    const int cpt = (cols + 255) / 256;
    if (cpt <= 1) VQA_ME(1); else if (cpt <= 2) VQA_ME(2); else if (cpt <= 4) VQA_ME(4); else if (cpt <= 8) VQA_ME(8);
    else if (cpt <= 12) VQA_ME(12); else VQA_ME(16);
#undef VQA_ME
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_normal_noise(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    VQA_REQUIRE(out && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(normal_noise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_extra_report(const float* extra_row, float* stats, int B, float weight, float* report, void* stream) {
    VQA_REQUIRE(extra_row && stats && report && B > 0, VQA_ERR_ARG);
    hipLaunchKernelGGL(extra_report_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, extra_row, stats, B, weight, report);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
