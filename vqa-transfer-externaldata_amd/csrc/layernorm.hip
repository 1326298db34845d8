// Group LayerNorm + ReLU (+ dropout) forward / backward for gfx950.
//
// Reference: modules.fc_layer(use_ln=True, activation_fn=relu)
// (vlmap/modules.py:647-650).  tf.contrib.layers.layer_norm normalises over ALL
// non-batch axes, so for v_linear_v ([B,36,1024]) one statistics group is a
// whole sample = 36 x 1024 values; for the 2-D layers a group
// is one row.  eps = 1e-12, biased variance, gamma/beta on the last axis.
//
// HBM-bound: one workgroup per group; a group (<= 147 KB) stays in L2 between the
// statistics passes and the normalise pass, so HBM sees one read + one write.
// Thread layout: (column unit cx, row lane ry); a column unit is 4 consecutive
// columns (float4) when N % 4 == 0.
#include "vqa_common.h"

namespace {

constexpr float LN_EPS = 1e-12f;

template <int U>
struct Vec;
template <>
struct Vec<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ __forceinline__ void load_mask(const uint8_t* p) {
        const uchar4 t = *reinterpret_cast<const uchar4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};
template <>
struct Vec<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
    __device__ __forceinline__ void store(float* p) const { p[0] = v[0]; }
    __device__ __forceinline__ void load_mask(const uint8_t* p) { v[0] = p[0]; }
};

// blockDim.x = CUt * RY ; cx = tid % CUt, ry = tid / CUt
template <int U>
__global__ void ln_relu_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const uint8_t* __restrict__ keepmask,
                                   float inv_keep, float* __restrict__ y, float* __restrict__ mean_out,
                                   float* __restrict__ rstd_out, int rows, int N, int CUt, int RY, int act) {
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int cx = threadIdx.x % CUt, ry = threadIdx.x / CUt;
    const int CU = N / U;
    const int64_t base = (int64_t)g * rows * N;
    const float invL = 1.f / ((float)rows * (float)N);

    float s = 0.f;
    for (int cu = cx; cu < CU; cu += CUt)
        for (int r = ry; r < rows; r += RY) {
            Vec<U> x; x.load(pre + base + (int64_t)r * N + cu * U);
#pragma unroll
            for (int j = 0; j < U; ++j) s += x.v[j];
        }
    const float mean = block_sum(s, red) * invL;
    float q = 0.f;
    for (int cu = cx; cu < CU; cu += CUt)
        for (int r = ry; r < rows; r += RY) {
            Vec<U> x; x.load(pre + base + (int64_t)r * N + cu * U);
#pragma unroll
            for (int j = 0; j < U; ++j) { const float d = x.v[j] - mean; q += d * d; }
        }
    const float var = block_sum(q, red) * invL;
    const float rstd = 1.f / sqrtf(var + LN_EPS);
    if (threadIdx.x == 0) { mean_out[g] = mean; rstd_out[g] = rstd; }

    for (int cu = cx; cu < CU; cu += CUt) {
        Vec<U> ga, be; ga.load(gamma + cu * U); be.load(beta + cu * U);
        for (int r = ry; r < rows; r += RY) {
            const int64_t off = base + (int64_t)r * N + cu * U;
            Vec<U> x, o; x.load(pre + off);
            Vec<U> km;
            if (keepmask != nullptr) km.load_mask(keepmask + off);
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const float ln = (x.v[j] - mean) * rstd * ga.v[j] + be.v[j];
                float v = act == 0 ? fmaxf(ln, 0.f) : tanhf(ln);
                if (keepmask != nullptr) v = v * km.v[j] * inv_keep;
                o.v[j] = v;
            }
            o.store(y + off);
        }
    }
}

// LDS column reduce over the RY row lanes: part[col] = sum_ry val.  buf: [RY][CUt*U]
template <int U>
__device__ __forceinline__ void col_reduce_store(float* buf, const float (&val)[U], float* dst_row, int cu, int CU,
                                                 int cx, int ry, int CUt, int RY) {
    if (RY == 1) {
        if (cu < CU && dst_row != nullptr) {
#pragma unroll
            for (int j = 0; j < U; ++j) dst_row[cu * U + j] = val[j];
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < U; ++j) buf[(ry * CUt + cx) * U + j] = val[j];
    __syncthreads();
    if (ry == 0 && cu < CU && dst_row != nullptr) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            float t = 0.f;
            for (int k = 0; k < RY; ++k) t += buf[(k * CUt + cx) * U + j];
            dst_row[cu * U + j] = t;
        }
    }
}

template <int U>
__global__ void ln_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                   const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ dpre,
                                   float* __restrict__ part_dgamma, float* __restrict__ part_dbeta,
                                   float* __restrict__ part_dbias, int rows, int N, int CUt, int RY, int act) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];  // [2][RY*CUt*U] when RY > 1
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int cx = threadIdx.x % CUt, ry = threadIdx.x / CUt;
    const int CU = N / U;
    const int iters = (CU + CUt - 1) / CUt;
    const int64_t base = (int64_t)g * rows * N;
    const float invL = 1.f / ((float)rows * (float)N);
    const float mean = mean_in[g], rstd = rstd_in[g];
    float* buf0 = dyn;
    float* buf1 = dyn + (size_t)RY * CUt * U;
    const bool want_params = (part_dgamma != nullptr);

    float s1 = 0.f, s2 = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int cu = cx + it * CUt;
        float cb[U], cg[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { cb[j] = 0.f; cg[j] = 0.f; }
        if (cu < CU) {
            Vec<U> ga, be; ga.load(gamma + cu * U); be.load(beta + cu * U);
            for (int r = ry; r < rows; r += RY) {
                const int64_t off = base + (int64_t)r * N + cu * U;
                Vec<U> x, d; x.load(pre + off); d.load(dy + off);
                Vec<U> km;
                if (keepmask != nullptr) km.load_mask(keepmask + off);
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const float xh = (x.v[j] - mean) * rstd;
                    const float ln = xh * ga.v[j] + be.v[j];
                    float gg = d.v[j];
                    if (keepmask != nullptr) gg = gg * km.v[j] * inv_keep;
                    float dln;
                    if (act == 0) dln = ln > 0.f ? gg : 0.f;
                    else { const float th = tanhf(ln); dln = gg * (1.f - th * th); }
                    const float dxh = dln * ga.v[j];
                    s1 += dxh;
                    s2 += dxh * xh;
                    cb[j] += dln;
                    cg[j] += dln * xh;
                }
            }
        }
        if (want_params) {
            col_reduce_store<U>(buf0, cb, part_dbeta + (int64_t)g * N, cu, CU, cx, ry, CUt, RY);
            col_reduce_store<U>(buf1, cg, part_dgamma + (int64_t)g * N, cu, CU, cx, ry, CUt, RY);
        }
    }
    const float m1 = block_sum(s1, red) * invL;
    const float m2 = block_sum(s2, red) * invL;

    for (int it = 0; it < iters; ++it) {
        const int cu = cx + it * CUt;
        float cbias[U];
#pragma unroll
        for (int j = 0; j < U; ++j) cbias[j] = 0.f;
        if (cu < CU) {
            Vec<U> ga, be; ga.load(gamma + cu * U); be.load(beta + cu * U);
            for (int r = ry; r < rows; r += RY) {
                const int64_t off = base + (int64_t)r * N + cu * U;
                Vec<U> x, d, o; x.load(pre + off); d.load(dy + off);
                Vec<U> km;
                if (keepmask != nullptr) km.load_mask(keepmask + off);
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const float xh = (x.v[j] - mean) * rstd;
                    const float ln = xh * ga.v[j] + be.v[j];
                    float gg = d.v[j];
                    if (keepmask != nullptr) gg = gg * km.v[j] * inv_keep;
                    float dln;
                    if (act == 0) dln = ln > 0.f ? gg : 0.f;
                    else { const float th = tanhf(ln); dln = gg * (1.f - th * th); }
                    const float dxh = dln * ga.v[j];
                    const float dp = rstd * (dxh - m1 - xh * m2);
                    o.v[j] = dp;
                    cbias[j] += dp;
                }
                o.store(dpre + off);
            }
        }
        if (part_dbias != nullptr)
            col_reduce_store<U>(buf0, cbias, part_dbias + (int64_t)g * N, cu, CU, cx, ry, CUt, RY);
    }
}

struct Shape { int U, CUt, RY, threads; };
Shape pick(const void* a, const void* b, const void* c, int rows, int N, bool mask) {
    Shape s;
    const bool vec = (N % 4 == 0) && vqa_aligned16(a) && vqa_aligned16(b) && (c == nullptr || vqa_aligned16(c));
    (void)mask;
    s.U = vec ? 4 : 1;
    const int CU = N / s.U;
    s.CUt = std::min(CU, 256);
    // round CUt up to a multiple of 64 so every wave is full (threads with cu >= CU idle)
    s.CUt = std::min(256, ((s.CUt + 63) / 64) * 64);
    s.RY = rows >= 4 ? 4 : 1;
    s.threads = s.CUt * s.RY;
    return s;
}

}  // namespace

extern "C" int vqa_ln_relu_fwd(const float* pre, const float* gamma, const float* beta, const uint8_t* keepmask,
                               float keep_prob, float* y, float* mean, float* rstd, int G, int rows, int N,
                               void* stream) {
    return vqa_ln_act_fwd(pre, gamma, beta, keepmask, keep_prob, y, mean, rstd, G, rows, N, 0, stream);
}

extern "C" int vqa_ln_act_fwd(const float* pre, const float* gamma, const float* beta, const uint8_t* keepmask,
                              float keep_prob, float* y, float* mean, float* rstd, int G, int rows, int N, int act,
                              void* stream) {
    VQA_REQUIRE(act == 0 || act == 1, VQA_ERR_ARG);
    VQA_REQUIRE(pre && gamma && beta && y && mean && rstd && G >= 0 && rows > 0 && N > 0, VQA_ERR_ARG);
    VQA_REQUIRE(keepmask == nullptr || keep_prob > 0.f, VQA_ERR_ARG);
    if (G == 0) return VQA_OK;
    Shape s = pick(pre, y, nullptr, rows, N, keepmask != nullptr);
    if (s.U == 4 && !(vqa_aligned16(gamma) && vqa_aligned16(beta) &&
                      (keepmask == nullptr || (reinterpret_cast<uintptr_t>(keepmask) & 3u) == 0)))
        s = Shape{1, std::min(256, ((std::min(N, 256) + 63) / 64) * 64), s.RY, 0};
    s.threads = s.CUt * s.RY;
    const float inv_keep = keepmask ? 1.f / keep_prob : 1.f;
    hipStream_t st = (hipStream_t)stream;
    if (s.U == 4)
        hipLaunchKernelGGL(ln_relu_fwd_kernel<4>, dim3(G), dim3(s.threads), 0, st, pre, gamma, beta, keepmask, inv_keep,
                           y, mean, rstd, rows, N, s.CUt, s.RY, act);
    else
        hipLaunchKernelGGL(ln_relu_fwd_kernel<1>, dim3(G), dim3(s.threads), 0, st, pre, gamma, beta, keepmask, inv_keep,
                           y, mean, rstd, rows, N, s.CUt, s.RY, act);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_ln_relu_bwd(const float* dy, const float* pre, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, const uint8_t* keepmask, float keep_prob,
                               float* dpre, float* part_dgamma, float* part_dbeta, float* part_dbias, int G, int rows,
                               int N, void* stream) {
    return vqa_ln_act_bwd(dy, pre, mean, rstd, gamma, beta, keepmask, keep_prob, dpre, part_dgamma, part_dbeta,
                          part_dbias, G, rows, N, 0, stream);
}

extern "C" int vqa_ln_act_bwd(const float* dy, const float* pre, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, const uint8_t* keepmask, float keep_prob,
                              float* dpre, float* part_dgamma, float* part_dbeta, float* part_dbias, int G, int rows,
                              int N, int act, void* stream) {
    VQA_REQUIRE(act == 0 || act == 1, VQA_ERR_ARG);
    VQA_REQUIRE(dy && pre && mean && rstd && gamma && beta && dpre && G >= 0 && rows > 0 && N > 0, VQA_ERR_ARG);
    VQA_REQUIRE((part_dgamma == nullptr) == (part_dbeta == nullptr), VQA_ERR_ARG);
    VQA_REQUIRE(keepmask == nullptr || keep_prob > 0.f, VQA_ERR_ARG);
    if (G == 0) return VQA_OK;
    Shape s = pick(pre, dy, dpre, rows, N, keepmask != nullptr);
    if (s.U == 4 && !(vqa_aligned16(gamma) && vqa_aligned16(beta) &&
                      (keepmask == nullptr || (reinterpret_cast<uintptr_t>(keepmask) & 3u) == 0) &&
                      (part_dgamma == nullptr || (vqa_aligned16(part_dgamma) && vqa_aligned16(part_dbeta))) &&
                      (part_dbias == nullptr || vqa_aligned16(part_dbias))))
        s = Shape{1, std::min(256, ((std::min(N, 256) + 63) / 64) * 64), s.RY, 0};
    s.threads = s.CUt * s.RY;
    const float inv_keep = keepmask ? 1.f / keep_prob : 1.f;
    const size_t dyn = s.RY > 1 ? (size_t)2 * s.RY * s.CUt * s.U * sizeof(float) : 0;
    hipStream_t st = (hipStream_t)stream;
    if (s.U == 4)
        hipLaunchKernelGGL(ln_relu_bwd_kernel<4>, dim3(G), dim3(s.threads), dyn, st, dy, pre, mean, rstd, gamma, beta,
                           keepmask, inv_keep, dpre, part_dgamma, part_dbeta, part_dbias, rows, N, s.CUt, s.RY, act);
    else
        hipLaunchKernelGGL(ln_relu_bwd_kernel<1>, dim3(G), dim3(s.threads), dyn, st, dy, pre, mean, rstd, gamma, beta,
                           keepmask, inv_keep, dpre, part_dgamma, part_dbeta, part_dbias, rows, N, s.CUt, s.RY, act);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
