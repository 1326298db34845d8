// Row kernels of the oldest registry model (vqa/model_vqa.py): GloVe_vocab lookup (constant rows + 3 trainable ones,
// vlmap/modules.py:451-467), the BasicLSTMCell recurrence under dynamic_rnn(sequence_length) (vlmap/modules.py:124-140 with
// the default cell_type 'LSTM'; gate order i, j, f, o and forget_bias 1.0 of tf.contrib.rnn.BasicLSTMCell), ReLU, and the
// broadcast scoring layer logit[b,a] = w . tanh(al[a] + pq[b]) + bias (vqa/model_vqa.py:232-257) without its [B,A,L]
// intermediate.  The matrix products around them are vqa_gemm_f32 calls in csrc/legacy_vqa.inc.  fp32, HBM- or
// VALU-bound; this model is a coverage item, not the benchmarked path.
#include <algorithm>

#include "vqa_common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return sigmoidf_stable(x); }

// x_tm[t, n, :] = (id < Vq - 3 ? fixed[id] : learn[id - (Vq - 3)]),  id = ids[n, t]  (batch-major ids, time-major rows)
__global__ __launch_bounds__(256) void embed2_fwd_kernel(const float* __restrict__ fixed, const float* __restrict__ learn,
                                                         const int32_t* __restrict__ ids, float* __restrict__ x, int N, int T,
                                                         int W, int Vq) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= N * T) return;
    const int t = tok / N, n = tok - t * N;
    int id = ids[n * T + t];
    id = min(max(id, 0), Vq - 1);
    const float* s = id < Vq - 3 ? fixed + (int64_t)id * W : learn + (int64_t)(id - (Vq - 3)) * W;
    float* d = x + (int64_t)tok * W;
    for (int i = threadIdx.x & 63; i < W; i += 64) d[i] = s[i];
}

// scatter-add of the rows that reach GloVe/learn (ids >= Vq - 3) + the sum of squares of those un-aggregated slices
__global__ __launch_bounds__(256) void embed2_bwd_kernel(const float* __restrict__ dx, const int32_t* __restrict__ ids,
                                                         float* __restrict__ dlearn, float* __restrict__ slice_sq, int N, int T,
                                                         int W, int Vq) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= N * T) return;
    const int t = tok / N, n = tok - t * N;
    const int id = min(max(ids[n * T + t], 0), Vq - 1);
    if (id < Vq - 3) return;
    const float* s = dx + (int64_t)tok * W;
    float* d = dlearn + (int64_t)(id - (Vq - 3)) * W;
    float sq = 0.f;
    for (int i = threadIdx.x & 63; i < W; i += 64) {
        const float g = s[i];
        atomicAdd(d + i, g);
        sq += g * g;
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) atomicAdd(slice_sq, sq);
}

// one step of the cell on pre-activations g [N, 4L] = [x, h] K + b (IN PLACE: left holding the activated gates
// i, j, f', o with f' = sigmoid(f + 1)); rows with t >= len carry (c, h) through
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float* __restrict__ g, const float* __restrict__ c_prev,
                                                            const float* __restrict__ h_prev, const int32_t* __restrict__ len,
                                                            int t, float* __restrict__ c_new, float* __restrict__ h_new, int N,
                                                            int L) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)N * L; e += (int64_t)gridDim.x * 256) {
        const int n = (int)(e / L), k = (int)(e - (int64_t)n * L);
        float* gr = g + (int64_t)n * 4 * L;
        const float i = sigm(gr[k]), j = tanhf(gr[L + k]), f = sigm(gr[2 * L + k] + 1.f), o = sigm(gr[3 * L + k]);
        gr[k] = i; gr[L + k] = j; gr[2 * L + k] = f; gr[3 * L + k] = o;
        const float cp = c_prev[e], hp = h_prev[e];
        const bool live = t < len[n];
        const float cn = cp * f + i * j;
        c_new[e] = live ? cn : cp;
        h_new[e] = live ? tanhf(cn) * o : hp;
    }
}

// backward of one step: dh / dc = gradients wrt (h, c) after step t; writes the pre-activation gradients dg [N, 4L],
// dc_prev, and dh_carry (= dh for finished rows, 0 for live ones: the part of dh_prev that does not go through the cell)
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc,
                                                            const float* __restrict__ g, const float* __restrict__ c_prev,
                                                            const float* __restrict__ c_new, const int32_t* __restrict__ len,
                                                            int t, float* __restrict__ dg, float* __restrict__ dc_prev,
                                                            float* __restrict__ dh_carry, int N, int L) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)N * L; e += (int64_t)gridDim.x * 256) {
        const int n = (int)(e / L), k = (int)(e - (int64_t)n * L);
        float* dgr = dg + (int64_t)n * 4 * L;
        if (t >= len[n]) {
            dgr[k] = 0.f; dgr[L + k] = 0.f; dgr[2 * L + k] = 0.f; dgr[3 * L + k] = 0.f;
            dc_prev[e] = dc[e];
            dh_carry[e] = dh[e];
            continue;
        }
        const float* gr = g + (int64_t)n * 4 * L;
        const float i = gr[k], j = gr[L + k], f = gr[2 * L + k], o = gr[3 * L + k];
        const float tc = tanhf(c_new[e]);
        const float dhv = dh[e];
        const float dct = dc[e] + dhv * o * (1.f - tc * tc);
        dgr[k] = dct * j * i * (1.f - i);
        dgr[L + k] = dct * i * (1.f - j * j);
        dgr[2 * L + k] = dct * c_prev[e] * f * (1.f - f);
        dgr[3 * L + k] = dhv * tc * o * (1.f - o);
        dc_prev[e] = dct * f;
        dh_carry[e] = 0.f;
    }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}

// z[b, a] = sum_k w[k] tanh(al[a, k] + pq[b, k]) + bias.  One workgroup per (b, chunk of answers): pq[b] and w in LDS,
// one wave per answer, lanes over k.
constexpr int SCORE_MAX_L = 1024;
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ al, const float* __restrict__ pq,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ z, int A, int L, int a_per_block) {
    __shared__ float s_pq[SCORE_MAX_L], s_w[SCORE_MAX_L];
    const int b = blockIdx.x, a0 = blockIdx.y * a_per_block, a1 = min(a0 + a_per_block, A);
    for (int k = threadIdx.x; k < L; k += 256) { s_pq[k] = pq[(int64_t)b * L + k]; s_w[k] = w[k]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float bc = bias[0];
    for (int a = a0 + wave; a < a1; a += 4) {
        const float* ar = al + (int64_t)a * L;
        float acc = 0.f;
        for (int k = lane; k < L; k += 64) acc += s_w[k] * tanhf(ar[k] + s_pq[k]);
        acc = wave_sum(acc);
        if (lane == 0) z[(int64_t)b * A + a] = acc + bc;
    }
}

// per question b: d_pq[b, k] = sum_a dz[b,a] w[k] (1 - t^2);  part_dw[b, k] = sum_a dz[b,a] t,  t = tanh(al[a,k] + pq[b,k]).
// Thread k-owner layout: thread handles columns k = threadIdx.x + 256 i, loops over all answers (al rows stream through L2).
__global__ __launch_bounds__(256) void score_bwd_q_kernel(const float* __restrict__ dz, const float* __restrict__ al,
                                                          const float* __restrict__ pq, const float* __restrict__ w,
                                                          float* __restrict__ d_pq, float* __restrict__ part_dw, int A, int L) {
    const int b = blockIdx.x;
    const float* dzb = dz + (int64_t)b * A;
    for (int k = threadIdx.x; k < L; k += 256) {
        const float p = pq[(int64_t)b * L + k], wk = w[k];
        float g = 0.f, dw = 0.f;
        for (int a = 0; a < A; ++a) {
            const float t = tanhf(al[(int64_t)a * L + k] + p);
            const float d = dzb[a];
            g += d * (1.f - t * t);
            dw += d * t;
        }
        d_pq[(int64_t)b * L + k] = g * wk;
        part_dw[(int64_t)b * L + k] = dw;
    }
}

// per answer a: d_al[a, k] = sum_b dz[b,a] w[k] (1 - t^2)
__global__ __launch_bounds__(256) void score_bwd_a_kernel(const float* __restrict__ dz, const float* __restrict__ al,
                                                          const float* __restrict__ pq, const float* __restrict__ w,
                                                          float* __restrict__ d_al, int B, int A, int L) {
    const int a = blockIdx.x;
    for (int k = threadIdx.x; k < L; k += 256) {
        const float av = al[(int64_t)a * L + k], wk = w[k];
        float g = 0.f;
        for (int b = 0; b < B; ++b) {
            const float t = tanhf(av + pq[(int64_t)b * L + k]);
            g += dz[(int64_t)b * A + a] * (1.f - t * t);
        }
        d_al[(int64_t)a * L + k] = g * wk;
    }
}

inline int grid_n(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

}  // namespace

extern "C" int vqa_embed2_fwd(const float* fixed, const float* learn, const int32_t* ids, float* x_tm, int N, int T, int W,
                              int Vq, void* stream) {
    VQA_REQUIRE(fixed && learn && ids && x_tm && N >= 0 && T >= 0 && W > 0 && Vq > 3, VQA_ERR_ARG);
    if (N * T == 0) return VQA_OK;
    hipLaunchKernelGGL(embed2_fwd_kernel, dim3((N * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, fixed, learn, ids, x_tm, N, T, W, Vq);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_embed2_bwd(const float* dx_tm, const int32_t* ids, float* dlearn, float* slice_sq, int N, int T, int W,
                              int Vq, void* stream) {
    VQA_REQUIRE(dx_tm && ids && dlearn && slice_sq && N >= 0 && T >= 0 && W > 0 && Vq > 3, VQA_ERR_ARG);
    if (N * T == 0) return VQA_OK;
    hipLaunchKernelGGL(embed2_bwd_kernel, dim3((N * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, dx_tm, ids, dlearn, slice_sq, N, T, W, Vq);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_lstm_step_fwd(float* gates, const float* c_prev, const float* h_prev, const int32_t* len, int t, float* c_new,
                                 float* h_new, int N, int L, void* stream) {
    VQA_REQUIRE(gates && c_prev && h_prev && len && c_new && h_new && N >= 0 && L > 0, VQA_ERR_ARG);
    if (N == 0) return VQA_OK;
    hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(grid_n((int64_t)N * L)), dim3(256), 0, (hipStream_t)stream, gates, c_prev, h_prev,
                       len, t, c_new, h_new, N, L);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_lstm_step_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* c_new,
                                 const int32_t* len, int t, float* dgates, float* dc_prev, float* dh_carry, int N, int L,
                                 void* stream) {
    VQA_REQUIRE(dh && dc && gates && c_prev && c_new && len && dgates && dc_prev && dh_carry && N >= 0 && L > 0, VQA_ERR_ARG);
    if (N == 0) return VQA_OK;
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(grid_n((int64_t)N * L)), dim3(256), 0, (hipStream_t)stream, dh, dc, gates, c_prev,
                       c_new, len, t, dgates, dc_prev, dh_carry, N, L);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    VQA_REQUIRE(x && y && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_n(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    VQA_REQUIRE(dy && y && dx && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_n(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_fill(float* x, int64_t n, float value, void* stream) {
    VQA_REQUIRE(x && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_n(n)), dim3(256), 0, (hipStream_t)stream, x, n, value);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_score_fwd(const float* al, const float* pq, const float* w, const float* bias, float* z, int B, int A, int L,
                             void* stream) {
    VQA_REQUIRE(al && pq && w && bias && z && B >= 0 && A > 0 && L > 0, VQA_ERR_ARG);
    VQA_REQUIRE(L <= SCORE_MAX_L, VQA_ERR_UNSUPPORTED);
    if (B == 0) return VQA_OK;
    const int per = 256, ny = (A + per - 1) / per;
    hipLaunchKernelGGL(score_fwd_kernel, dim3(B, ny), dim3(256), 0, (hipStream_t)stream, al, pq, w, bias, z, A, L, per);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_score_bwd(const float* dz, const float* al, const float* pq, const float* w, float* d_al, float* d_pq,
                             float* part_dw, int B, int A, int L, void* stream) {
    VQA_REQUIRE(dz && al && pq && w && d_al && d_pq && part_dw && B >= 0 && A > 0 && L > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(score_bwd_q_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dz, al, pq, w, d_pq, part_dw, A, L);
    VQA_CHECK_LAUNCH();
    hipLaunchKernelGGL(score_bwd_a_kernel, dim3(A), dim3(256), 0, (hipStream_t)stream, dz, al, pq, w, d_al, B, A, L);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
