// Row kernels of the bi-directional question encoder (modules.encode_L_bidirection, vlmap/modules.py:100-122, as used by
// vqa/model_vlmap_finetune.py / model_vlmap_only.py): tf.nn.bidirectional_dynamic_rnn runs the backward cell on
// reverse_sequence(inputs, len) and reverses its outputs back; outputs past a row's length are zero.  The recurrences
// themselves are the fused GRU-step kernels of gemm_f32.hip on time-major buffers; these kernels move between the
// reference's batch-major tensors and those.  All HBM-bound, fp32 / int32.
#include <algorithm>

#include "vqa_common.h"

namespace {

// q_rev[b, t] = q[b, len_b - 1 - t] for t < len_b, q[b, t] otherwise   (tf.reverse_sequence on the token ids)
__global__ __launch_bounds__(256) void reverse_tokens_kernel(const int32_t* __restrict__ q, const int32_t* __restrict__ len,
                                                             int32_t* __restrict__ q_rev, int B, int T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    const int n = min(max(len[b], 0), T);
    q_rev[i] = q[b * T + (t < n ? n - 1 - t : t)];
}

// q_map[b, t, :h]  = t < len ? hs_fw[t + 1, b, :]   : 0        (forward cell's output at token t)
// q_map[b, t, h:]  = t < len ? hs_bw[len - t, b, :] : 0        (backward cell's output after reversed step len - 1 - t)
// q_ft[b, :]       = [hs_fw[T, b, :], hs_bw[T, b, :]]          (final states: carried through past the length)
__global__ __launch_bounds__(256) void bi_outputs_fwd_kernel(const float* __restrict__ hs_fw, const float* __restrict__ hs_bw,
                                                             const int32_t* __restrict__ len, float* __restrict__ q_map,
                                                             float* __restrict__ q_ft, int B, int T, int h) {
    const int b = blockIdx.x;
    const int n = min(max(len[b], 0), T);
    const int64_t Bh = (int64_t)B * h;
    for (int i = threadIdx.x; i < (T + 1) * 2 * h; i += 256) {
        const int t = i / (2 * h), col = i - t * 2 * h;
        const bool fw = col < h;
        const int c = fw ? col : col - h;
        if (t == T) {
            q_ft[(int64_t)b * 2 * h + col] = (fw ? hs_fw : hs_bw)[(int64_t)T * Bh + (int64_t)b * h + c];
            continue;
        }
        float v = 0.f;
        if (t < n) v = fw ? hs_fw[(int64_t)(t + 1) * Bh + (int64_t)b * h + c] : hs_bw[(int64_t)(n - t) * Bh + (int64_t)b * h + c];
        q_map[((int64_t)b * T + t) * 2 * h + col] = v;
    }
}

// the transpose of the above: per-step output gradients of the two recurrences (time-major, zero past the length)
// and the gradients wrt their final states
__global__ __launch_bounds__(256) void bi_outputs_bwd_kernel(const float* __restrict__ d_map, const float* __restrict__ d_ft,
                                                             const int32_t* __restrict__ len, float* __restrict__ dout_fw,
                                                             float* __restrict__ dout_bw, float* __restrict__ dhT_fw,
                                                             float* __restrict__ dhT_bw, int B, int T, int h) {
    const int b = blockIdx.x;
    const int n = min(max(len[b], 0), T);
    const int64_t Bh = (int64_t)B * h;
    for (int i = threadIdx.x; i < (T + 1) * 2 * h; i += 256) {
        const int t = i / (2 * h), col = i - t * 2 * h;
        const bool fw = col < h;
        const int c = fw ? col : col - h;
        if (t == T) {
            (fw ? dhT_fw : dhT_bw)[(int64_t)b * h + c] = d_ft[(int64_t)b * 2 * h + col];
            continue;
        }
        // step index s of the recurrence whose output this is: fw s = t; bw s = len - 1 - t (t < len), else unused -> 0
        if (fw) dout_fw[(int64_t)t * Bh + (int64_t)b * h + c] = t < n ? d_map[((int64_t)b * T + t) * 2 * h + col] : 0.f;
        else dout_bw[(int64_t)t * Bh + (int64_t)b * h + c] = t < n ? d_map[((int64_t)b * T + (n - 1 - t)) * 2 * h + col] : 0.f;
    }
}

// dx[t, b, :] = dx_fw[t, b, :] + dx_bw[len - 1 - t, b, :] for t < len (the backward cell saw token t at reversed step
// len - 1 - t), dx_fw + dx_bw at t >= len (both zero there): the gradient wrt embedding_lookup's output, whose rows are
// the un-aggregated IndexedSlices of the embedding gradient
__global__ __launch_bounds__(256) void bi_dx_combine_kernel(const float* __restrict__ dx_fw, const float* __restrict__ dx_bw,
                                                            const int32_t* __restrict__ len, float* __restrict__ dx, int B,
                                                            int T, int W) {
    const int tok = blockIdx.x;                 // t * B + b
    const int t = tok / B, b = tok - t * B;
    const int n = min(max(len[b], 0), T);
    const int src = t < n ? n - 1 - t : t;
    const float* f = dx_fw + (int64_t)tok * W;
    const float* g = dx_bw + ((int64_t)src * B + b) * W;
    float* o = dx + (int64_t)tok * W;
    for (int i = threadIdx.x; i < W; i += 256) o[i] = f[i] + g[i];
}

}  // namespace

extern "C" int vqa_reverse_tokens(const int32_t* q, const int32_t* len, int32_t* q_rev, int B, int T, void* stream) {
    VQA_REQUIRE(q && len && q_rev && B >= 0 && T >= 0, VQA_ERR_ARG);
    if (B * T == 0) return VQA_OK;
    hipLaunchKernelGGL(reverse_tokens_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, len, q_rev, B, T);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_bi_outputs_fwd(const float* hs_fw, const float* hs_bw, const int32_t* len, float* q_map, float* q_ft,
                                  int B, int T, int h, void* stream) {
    VQA_REQUIRE(hs_fw && hs_bw && len && q_map && q_ft && B >= 0 && T >= 0 && h > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(bi_outputs_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, hs_fw, hs_bw, len, q_map, q_ft, B, T, h);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_bi_outputs_bwd(const float* d_map, const float* d_ft, const int32_t* len, float* dout_fw, float* dout_bw,
                                  float* dhT_fw, float* dhT_bw, int B, int T, int h, void* stream) {
    VQA_REQUIRE(d_map && d_ft && len && dout_fw && dout_bw && dhT_fw && dhT_bw && B >= 0 && T >= 0 && h > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(bi_outputs_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, d_map, d_ft, len, dout_fw, dout_bw,
                       dhT_fw, dhT_bw, B, T, h);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_bi_dx_combine(const float* dx_fw, const float* dx_bw, const int32_t* len, float* dx, int B, int T, int W,
                                 void* stream) {
    VQA_REQUIRE(dx_fw && dx_bw && len && dx && B >= 0 && T >= 0 && W > 0, VQA_ERR_ARG);
    if (B * T == 0) return VQA_OK;
    hipLaunchKernelGGL(bi_dx_combine_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, dx_fw, dx_bw, len, dx, B, T, W);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
