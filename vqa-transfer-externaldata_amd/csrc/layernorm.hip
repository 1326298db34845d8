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

// Register-resident forms for wide groups with N == 1024 (v_linear_v: 36 x 1024 per sample): thread (cx, ry) owns the
// float4 column unit cx of rows ry, ry + 4, ... (exactly RPT of them: rows == 4 RPT) and fetches ALL of them up front -- one memory
// latency for the whole group instead of one per row and pass, and no second / third read of the group.  Same
// per-thread element order as the generic kernels.  (Native vector type + clamped, unconditional loads: see
// attention.hip on what hipcc does with guarded float4 loads.)
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int REG_CUT = 256, REG_RY = 4;

// RPT rows and CPT float4 column units (cx, cx + 256) per thread; EXACT: rows == REG_RY * RPT, no clamp and no guard
// (the 36-row kernel must not grow: 117 VGPRs in the backward); otherwise rows <= REG_RY * RPT, row indices are
// clamped for the (unconditional) loads and rows past the group count for nothing.  Per-thread element order as in
// the generic kernels: column units outer, rows inner.
template <int RPT, int CPT, bool MASK, bool EXACT>
__global__ __launch_bounds__(REG_CUT * REG_RY) void ln_fwd_reg_kernel(
    const float* __restrict__ pre, const float* __restrict__ gamma, const float* __restrict__ beta,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int rows) {
    constexpr int N4 = REG_CUT * CPT;
    __shared__ float red[16];
    const int g = blockIdx.x, cx = threadIdx.x % REG_CUT, ry = threadIdx.x / REG_CUT;
    const int64_t base4 = (int64_t)g * rows * N4;
    const f32x4v* p4 = reinterpret_cast<const f32x4v*>(pre) + base4;
    const unsigned* m4 = MASK ? reinterpret_cast<const unsigned*>(keepmask) + base4 : nullptr;
    const float invL = 1.f / ((float)rows * (float)(N4 * 4));
    f32x4v x[CPT][RPT];
    unsigned km[CPT][RPT];
    bool ok[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) ok[i] = EXACT || (ry + REG_RY * i < rows);
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = EXACT ? ry + REG_RY * i : min(ry + REG_RY * i, rows - 1);
            const int o = r * N4 + cx + REG_CUT * c;
            x[c][i] = p4[o];
            if (MASK) km[c][i] = m4[o];
        }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            if (ok[i]) { s += x[c][i].x; s += x[c][i].y; s += x[c][i].z; s += x[c][i].w; }
    const float mean = block_sum(s, red) * invL;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            if (!ok[i]) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = x[c][i][j] - mean; q += d * d; }
        }
    const float var = block_sum(q, red) * invL;
    const float rstd = 1.f / sqrtf(var + LN_EPS);
    if (threadIdx.x == 0) { mean_out[g] = mean; rstd_out[g] = rstd; }
    f32x4v* y4 = reinterpret_cast<f32x4v*>(y) + base4;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const f32x4v ga = reinterpret_cast<const f32x4v*>(gamma)[cx + REG_CUT * c];
        const f32x4v be = reinterpret_cast<const f32x4v*>(beta)[cx + REG_CUT * c];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            f32x4v o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ln = (x[c][i][j] - mean) * rstd * ga[j] + be[j];
                float v = fmaxf(ln, 0.f);                           // ReLU only (tanh layers use the generic kernel)
                if (MASK) v = v * (float)((km[c][i] >> (8 * j)) & 0xFFu) * inv_keep;
                o[j] = v;
            }
            if (ok[i]) y4[(ry + REG_RY * i) * N4 + cx + REG_CUT * c] = o;
        }
    }
}

template <int RPT, int CPT, bool MASK, bool EXACT>
__global__ __launch_bounds__(REG_CUT * REG_RY) void ln_bwd_reg_kernel(
    const float* __restrict__ dy, const float* __restrict__ pre, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ gamma, const float* __restrict__ beta,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ dpre, float* __restrict__ part_dgamma,
    float* __restrict__ part_dbeta, float* __restrict__ part_dbias, int rows) {
    constexpr int N4 = REG_CUT * CPT, N = N4 * 4;
    extern __shared__ __attribute__((aligned(16))) float dyn[];  // [2][RY*CUt*4]
    __shared__ float red[16];
    const int g = blockIdx.x, cx = threadIdx.x % REG_CUT, ry = threadIdx.x / REG_CUT;
    const int64_t base4 = (int64_t)g * rows * N4;
    const f32x4v* p4 = reinterpret_cast<const f32x4v*>(pre) + base4;
    const f32x4v* d4 = reinterpret_cast<const f32x4v*>(dy) + base4;
    const unsigned* m4 = MASK ? reinterpret_cast<const unsigned*>(keepmask) + base4 : nullptr;
    const float invL = 1.f / ((float)rows * (float)N);
    const float mean = mean_in[g], rstd = rstd_in[g];
    float* buf0 = dyn;
    float* buf1 = dyn + (size_t)REG_RY * REG_CUT * 4;
    f32x4v x[CPT][RPT], d[CPT][RPT];
    unsigned km[CPT][RPT];
    bool ok[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) ok[i] = EXACT || (ry + REG_RY * i < rows);
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = EXACT ? ry + REG_RY * i : min(ry + REG_RY * i, rows - 1);
            const int o = r * N4 + cx + REG_CUT * c;
            x[c][i] = p4[o];
            d[c][i] = d4[o];
            if (MASK) km[c][i] = m4[o];
        }
    float s1 = 0.f, s2 = 0.f;
    // pass 1 (registers only): d is overwritten by dxh = dln * gamma, x by xhat
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const f32x4v ga = reinterpret_cast<const f32x4v*>(gamma)[cx + REG_CUT * c];
        const f32x4v be = reinterpret_cast<const f32x4v*>(beta)[cx + REG_CUT * c];
        float cb[4] = {0.f, 0.f, 0.f, 0.f}, cg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (x[c][i][j] - mean) * rstd;
                const float ln = xh * ga[j] + be[j];
                float gg = ok[i] ? d[c][i][j] : 0.f;
                if (MASK) gg = gg * (float)((km[c][i] >> (8 * j)) & 0xFFu) * inv_keep;
                const float dln = ln > 0.f ? gg : 0.f;                    // ReLU only (tanh layers use the generic kernel)
                const float dxh = dln * ga[j];
                s1 += dxh; s2 += dxh * xh; cb[j] += dln; cg[j] += dln * xh;
                x[c][i][j] = xh;
                d[c][i][j] = dxh;
            }
        }
        if (part_dgamma != nullptr) {
            col_reduce_store<4>(buf0, cb, part_dbeta + (int64_t)g * N, cx + REG_CUT * c, N4, cx, ry, REG_CUT, REG_RY);
            col_reduce_store<4>(buf1, cg, part_dgamma + (int64_t)g * N, cx + REG_CUT * c, N4, cx, ry, REG_CUT, REG_RY);
        }
    }
    const float m1 = block_sum(s1, red) * invL;
    const float m2 = block_sum(s2, red) * invL;
    f32x4v* o4 = reinterpret_cast<f32x4v*>(dpre) + base4;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        float cbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            f32x4v o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dp = rstd * (d[c][i][j] - m1 - x[c][i][j] * m2);
                o[j] = dp;
                cbias[j] += ok[i] ? dp : 0.f;
            }
            if (ok[i]) o4[(ry + REG_RY * i) * N4 + cx + REG_CUT * c] = o;
        }
        if (part_dbias != nullptr)
            col_reduce_store<4>(buf0, cbias, part_dbias + (int64_t)g * N, cx + REG_CUT * c, N4, cx, ry, REG_CUT, REG_RY);
    }
}

int g_ln_reg = 1;   // A/B switch (vqa_ln_set_fast)

// register-resident kernels apply (16-byte path, ReLU): 1 = N 1024, rows == 36 (v_linear_v's block per sample);
// 2 = N 1024, rows <= 8; 3 = N 2048, rows <= 8 (the pre-training model's groups of 5 key boxes per image)
inline int reg_mode(int U, int rows, int N) {
    if (!g_ln_reg || U != 4) return 0;
    if (N == REG_CUT * 4 && rows == REG_RY * 9) return 1;
    if (N == REG_CUT * 4 && rows >= 2 && rows <= REG_RY * 2) return 2;
    if (N == REG_CUT * 8 && rows >= 2 && rows <= REG_RY * 2) return 3;
    return 0;
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

// ---------------------------------------------------------------------------------------------------------------
// pooled_linear_l and q_linear_l end in the same place: joint_fc reads their PRODUCT (vqa/model_vlmap_answer.py:163-177).
// One launch finishes both layers and forms the product -- LayerNorm + ReLU of two [B, N] pre-activations (one row per
// group) and z = y_a * y_b -- instead of two LayerNorm launches and an element-wise one; the backward twin takes dz
// (+ an optional extra gradient on y_b) back to both pre-activations.  One workgroup per sample, every value in registers.
template <int VPT>
__device__ __forceinline__ void pair_stats(const float4 (&a)[VPT], const float4 (&b)[VPT], int N, float* red, float& mean_a,
                                           float& rstd_a, float& mean_b, float& rstd_b, const bool (&in)[VPT]) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i)
        if (in[i]) { sa += (a[i].x + a[i].y) + (a[i].z + a[i].w); sb += (b[i].x + b[i].y) + (b[i].z + b[i].w); }
    const float invN = 1.f / (float)N;
    mean_a = block_sum(sa, red) * invN;
    mean_b = block_sum(sb, red) * invN;
    float qa = 0.f, qb = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i)
        if (in[i]) {
            const float a0 = a[i].x - mean_a, a1 = a[i].y - mean_a, a2 = a[i].z - mean_a, a3 = a[i].w - mean_a;
            const float b0 = b[i].x - mean_b, b1 = b[i].y - mean_b, b2 = b[i].z - mean_b, b3 = b[i].w - mean_b;
            qa += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            qb += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
        }
    rstd_a = 1.f / sqrtf(block_sum(qa, red) * invN + LN_EPS);
    rstd_b = 1.f / sqrtf(block_sum(qb, red) * invN + LN_EPS);
}

template <int VPT>
__global__ __launch_bounds__(256) void ln_pair_mul_fwd_kernel(const float* __restrict__ pre_a, const float* __restrict__ pre_b,
                                                              const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                                                              const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
                                                              float* __restrict__ y_a, float* __restrict__ y_b,
                                                              float* __restrict__ z, float* __restrict__ mean_a_out,
                                                              float* __restrict__ rstd_a_out, float* __restrict__ mean_b_out,
                                                              float* __restrict__ rstd_b_out, int N) {
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int64_t base = (int64_t)g * N;
    float4 a[VPT], b[VPT];
    bool in[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        in[i] = col < N;
        const int cc = in[i] ? col : 0;
        a[i] = *reinterpret_cast<const float4*>(pre_a + base + cc);
        b[i] = *reinterpret_cast<const float4*>(pre_b + base + cc);
    }
    float ma, ra, mb, rb;
    pair_stats<VPT>(a, b, N, red, ma, ra, mb, rb, in);
    if (threadIdx.x == 0) { mean_a_out[g] = ma; rstd_a_out[g] = ra; mean_b_out[g] = mb; rstd_b_out[g] = rb; }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (!in[i]) continue;
        const int col = (threadIdx.x + 256 * i) * 4;
        const float4 ga = *reinterpret_cast<const float4*>(gamma_a + col), ba = *reinterpret_cast<const float4*>(beta_a + col);
        const float4 gb = *reinterpret_cast<const float4*>(gamma_b + col), bb = *reinterpret_cast<const float4*>(beta_b + col);
        float4 ya, yb;
        ya.x = fmaxf((a[i].x - ma) * ra * ga.x + ba.x, 0.f); ya.y = fmaxf((a[i].y - ma) * ra * ga.y + ba.y, 0.f);
        ya.z = fmaxf((a[i].z - ma) * ra * ga.z + ba.z, 0.f); ya.w = fmaxf((a[i].w - ma) * ra * ga.w + ba.w, 0.f);
        yb.x = fmaxf((b[i].x - mb) * rb * gb.x + bb.x, 0.f); yb.y = fmaxf((b[i].y - mb) * rb * gb.y + bb.y, 0.f);
        yb.z = fmaxf((b[i].z - mb) * rb * gb.z + bb.z, 0.f); yb.w = fmaxf((b[i].w - mb) * rb * gb.w + bb.w, 0.f);
        *reinterpret_cast<float4*>(y_a + base + col) = ya;
        *reinterpret_cast<float4*>(y_b + base + col) = yb;
        *reinterpret_cast<float4*>(z + base + col) = make_float4(ya.x * yb.x, ya.y * yb.y, ya.z * yb.z, ya.w * yb.w);
    }
}

// one tensor's LayerNorm + ReLU backward on register-resident values: dy -> dpre (+ the per-sample partial rows)
template <int VPT>
__device__ __forceinline__ void pair_bwd_one(const float (&dy)[VPT][4], const float (&xh)[VPT][4], const float (&gam)[VPT][4],
                                             const bool (&pos)[VPT][4], const bool (&in)[VPT], float rstd, int N, float* red,
                                             float* __restrict__ dpre, float* __restrict__ pdg, float* __restrict__ pdb,
                                             float* __restrict__ pdbias, int64_t base) {
    float dxh[VPT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dln = (in[i] && pos[i][j]) ? dy[i][j] : 0.f;
            dxh[i][j] = dln * gam[i][j];
            s1 += dxh[i][j];
            s2 += dxh[i][j] * xh[i][j];
        }
    const float invN = 1.f / (float)N;
    const float m1 = block_sum(s1, red) * invN;
    const float m2 = block_sum(s2, red) * invN;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        if (!in[i]) continue;
        const int col = (threadIdx.x + 256 * i) * 4;
        float o[4], dl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = rstd * (dxh[i][j] - m1 - xh[i][j] * m2);
            dl[j] = pos[i][j] ? dy[i][j] : 0.f;
        }
        *reinterpret_cast<float4*>(dpre + base + col) = make_float4(o[0], o[1], o[2], o[3]);
        if (pdg != nullptr) {
            *reinterpret_cast<float4*>(pdg + base + col) = make_float4(dl[0] * xh[i][0], dl[1] * xh[i][1], dl[2] * xh[i][2], dl[3] * xh[i][3]);
            *reinterpret_cast<float4*>(pdb + base + col) = make_float4(dl[0], dl[1], dl[2], dl[3]);
        }
        if (pdbias != nullptr) *reinterpret_cast<float4*>(pdbias + base + col) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <int VPT>
__global__ __launch_bounds__(256) void ln_pair_mul_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ add_b, const float* __restrict__ pre_a, const float* __restrict__ pre_b,
    const float* __restrict__ mean_a, const float* __restrict__ rstd_a, const float* __restrict__ mean_b,
    const float* __restrict__ rstd_b, const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
    const float* __restrict__ gamma_b, const float* __restrict__ beta_b, float* __restrict__ dpre_a, float* __restrict__ dpre_b,
    float* __restrict__ pdg_a, float* __restrict__ pdb_a, float* __restrict__ pdbias_a, float* __restrict__ pdg_b,
    float* __restrict__ pdb_b, float* __restrict__ pdbias_b, int N) {
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int64_t base = (int64_t)g * N;
    const float ma = mean_a[g], ra = rstd_a[g], mb = mean_b[g], rb = rstd_b[g];
    float dya[VPT][4], dyb[VPT][4], xha[VPT][4], xhb[VPT][4], ga[VPT][4], gb[VPT][4];
    bool pa[VPT][4], pb[VPT][4], in[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        in[i] = col < N;
        const int cc = in[i] ? col : 0;
        const float4 d = *reinterpret_cast<const float4*>(dz + base + cc);
        const float4 a = *reinterpret_cast<const float4*>(pre_a + base + cc), b = *reinterpret_cast<const float4*>(pre_b + base + cc);
        const float4 gA = *reinterpret_cast<const float4*>(gamma_a + cc), bA = *reinterpret_cast<const float4*>(beta_a + cc);
        const float4 gB = *reinterpret_cast<const float4*>(gamma_b + cc), bB = *reinterpret_cast<const float4*>(beta_b + cc);
        float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add_b != nullptr) ad = *reinterpret_cast<const float4*>(add_b + base + cc);
        const float dv[4] = {d.x, d.y, d.z, d.w}, av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
        const float gav[4] = {gA.x, gA.y, gA.z, gA.w}, bav[4] = {bA.x, bA.y, bA.z, bA.w};
        const float gbv[4] = {gB.x, gB.y, gB.z, gB.w}, bbv[4] = {bB.x, bB.y, bB.z, bB.w}, adv[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xha[i][j] = (av[j] - ma) * ra;
            xhb[i][j] = (bv[j] - mb) * rb;
            const float lna = xha[i][j] * gav[j] + bav[j], lnb = xhb[i][j] * gbv[j] + bbv[j];
            pa[i][j] = lna > 0.f;
            pb[i][j] = lnb > 0.f;
            const float ya = fmaxf(lna, 0.f), yb = fmaxf(lnb, 0.f);
            dya[i][j] = dv[j] * yb;
            dyb[i][j] = dv[j] * ya + adv[j];
            ga[i][j] = gav[j];
            gb[i][j] = gbv[j];
        }
    }
    pair_bwd_one<VPT>(dya, xha, ga, pa, in, ra, N, red, dpre_a, pdg_a, pdb_a, pdbias_a, base);
    pair_bwd_one<VPT>(dyb, xhb, gb, pb, in, rb, N, red, dpre_b, pdg_b, pdb_b, pdbias_b, base);
}

}  // namespace

extern "C" int vqa_ln_pair_mul_supported(int N, const void* const* ptrs, int n_ptrs) {
    if (N <= 0 || N % 4 != 0 || N > 4096) return 0;
    for (int i = 0; i < n_ptrs; ++i)
        if (ptrs[i] != nullptr && !vqa_aligned16(ptrs[i])) return 0;
    return 1;
}

extern "C" int vqa_ln_pair_mul_fwd(const float* pre_a, const float* pre_b, const float* gamma_a, const float* beta_a,
                                   const float* gamma_b, const float* beta_b, float* y_a, float* y_b, float* z, float* mean_a,
                                   float* rstd_a, float* mean_b, float* rstd_b, int G, int N, void* stream) {
    VQA_REQUIRE(pre_a && pre_b && gamma_a && beta_a && gamma_b && beta_b && y_a && y_b && z && mean_a && rstd_a && mean_b && rstd_b &&
                    G >= 0,
                VQA_ERR_ARG);
    const void* ptrs[9] = {pre_a, pre_b, gamma_a, beta_a, gamma_b, beta_b, y_a, y_b, z};
    VQA_REQUIRE(vqa_ln_pair_mul_supported(N, ptrs, 9), VQA_ERR_ALIGN);
    if (G == 0) return VQA_OK;
    hipStream_t st = (hipStream_t)stream;
#define VQA_LNP_FWD(v)                                                                                                         \
    hipLaunchKernelGGL(ln_pair_mul_fwd_kernel<v>, dim3(G), dim3(256), 0, st, pre_a, pre_b, gamma_a, beta_a, gamma_b, beta_b, y_a, \
                       y_b, z, mean_a, rstd_a, mean_b, rstd_b, N)
    if (N <= 1024) VQA_LNP_FWD(1); else if (N <= 2048) VQA_LNP_FWD(2); else VQA_LNP_FWD(4);
#undef VQA_LNP_FWD
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_ln_pair_mul_bwd(const float* dz, const float* add_b, const float* pre_a, const float* pre_b, const float* mean_a,
                                   const float* rstd_a, const float* mean_b, const float* rstd_b, const float* gamma_a,
                                   const float* beta_a, const float* gamma_b, const float* beta_b, float* dpre_a, float* dpre_b,
                                   float* part_dgamma_a, float* part_dbeta_a, float* part_dbias_a, float* part_dgamma_b,
                                   float* part_dbeta_b, float* part_dbias_b, int G, int N, void* stream) {
    VQA_REQUIRE(dz && pre_a && pre_b && mean_a && rstd_a && mean_b && rstd_b && gamma_a && beta_a && gamma_b && beta_b && dpre_a &&
                    dpre_b && G >= 0,
                VQA_ERR_ARG);
    VQA_REQUIRE((part_dgamma_a == nullptr) == (part_dbeta_a == nullptr) && (part_dgamma_b == nullptr) == (part_dbeta_b == nullptr),
                VQA_ERR_ARG);
    const void* ptrs[18] = {dz, add_b, pre_a, pre_b, gamma_a, beta_a, gamma_b, beta_b, dpre_a, dpre_b, part_dgamma_a, part_dbeta_a,
                            part_dbias_a, part_dgamma_b, part_dbeta_b, part_dbias_b, nullptr, nullptr};
    VQA_REQUIRE(vqa_ln_pair_mul_supported(N, ptrs, 16), VQA_ERR_ALIGN);
    if (G == 0) return VQA_OK;
    hipStream_t st = (hipStream_t)stream;
#define VQA_LNP_BWD(v)                                                                                                          \
    hipLaunchKernelGGL(ln_pair_mul_bwd_kernel<v>, dim3(G), dim3(256), 0, st, dz, add_b, pre_a, pre_b, mean_a, rstd_a, mean_b, rstd_b, \
                       gamma_a, beta_a, gamma_b, beta_b, dpre_a, dpre_b, part_dgamma_a, part_dbeta_a, part_dbias_a, part_dgamma_b, \
                       part_dbeta_b, part_dbias_b, N)
    if (N <= 1024) VQA_LNP_BWD(1); else if (N <= 2048) VQA_LNP_BWD(2); else VQA_LNP_BWD(4);
#undef VQA_LNP_BWD
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_ln_set_fast(int on) {
    g_ln_reg = on ? 1 : 0;
    return VQA_OK;
}

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
    const int rm = act == 0 ? reg_mode(s.U, rows, N) : 0;
    if (rm != 0) {
#define VQA_LN_FWD_REG(rpt, cpt, exact)                                                                                    \
    do {                                                                                                                 \
        if (keepmask != nullptr)                                                                                         \
            hipLaunchKernelGGL((ln_fwd_reg_kernel<rpt, cpt, true, exact>), dim3(G), dim3(REG_CUT * REG_RY), 0, st, pre,    \
                               gamma, beta, keepmask, inv_keep, y, mean, rstd, rows);                                    \
        else                                                                                                             \
            hipLaunchKernelGGL((ln_fwd_reg_kernel<rpt, cpt, false, exact>), dim3(G), dim3(REG_CUT * REG_RY), 0, st, pre,   \
                               gamma, beta, keepmask, inv_keep, y, mean, rstd, rows);                                    \
    } while (0)
        if (rm == 1) VQA_LN_FWD_REG(9, 1, true); else if (rm == 2) VQA_LN_FWD_REG(2, 1, false); else VQA_LN_FWD_REG(2, 2, false);
#undef VQA_LN_FWD_REG
    } else if (s.U == 4)
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
    const int rm = act == 0 ? reg_mode(s.U, rows, N) : 0;
    if (rm != 0) {
        const size_t dyn_reg = (size_t)2 * REG_RY * REG_CUT * 4 * sizeof(float);
#define VQA_LN_BWD_REG(rpt, cpt, exact)                                                                                    \
    do {                                                                                                                 \
        if (keepmask != nullptr)                                                                                         \
            hipLaunchKernelGGL((ln_bwd_reg_kernel<rpt, cpt, true, exact>), dim3(G), dim3(REG_CUT * REG_RY), dyn_reg, st,   \
                               dy, pre, mean, rstd, gamma, beta, keepmask, inv_keep, dpre, part_dgamma, part_dbeta,      \
                               part_dbias, rows);                                                                        \
        else                                                                                                             \
            hipLaunchKernelGGL((ln_bwd_reg_kernel<rpt, cpt, false, exact>), dim3(G), dim3(REG_CUT * REG_RY), dyn_reg, st,  \
                               dy, pre, mean, rstd, gamma, beta, keepmask, inv_keep, dpre, part_dgamma, part_dbeta,      \
                               part_dbias, rows);                                                                        \
    } while (0)
        if (rm == 1) VQA_LN_BWD_REG(9, 1, true); else if (rm == 2) VQA_LN_BWD_REG(2, 1, false); else VQA_LN_BWD_REG(2, 2, false);
#undef VQA_LN_BWD_REG
    } else if (s.U == 4)
        hipLaunchKernelGGL(ln_relu_bwd_kernel<4>, dim3(G), dim3(s.threads), dyn, st, dy, pre, mean, rstd, gamma, beta,
                           keepmask, inv_keep, dpre, part_dgamma, part_dbeta, part_dbias, rows, N, s.CUt, s.RY, act);
    else
        hipLaunchKernelGGL(ln_relu_bwd_kernel<1>, dim3(G), dim3(s.threads), dyn, st, dy, pre, mean, rstd, gamma, beta,
                           keepmask, inv_keep, dpre, part_dgamma, part_dbeta, part_dbias, rows, N, s.CUt, s.RY, act);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
