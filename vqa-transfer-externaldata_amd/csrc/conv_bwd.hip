// Backward of the extractor's convolution op (the only consumer in the reference is the legacy in-graph
// CNN fine-tune, vlmap/model_vlmap.py:675-690 --ft_enc_I): gradients of
//     y = [relu]( conv(x, w) * scale[co] + shift[co] + residual )          (vqa_conv2d_nhwc, folded inference BatchNorm)
// with respect to the input, the filter, the shift and the residual.  NHWC, f32, HWIO filters.
//
// Both big products run on the f32 MFMA GEMM of the forward path (vqa_gemm_f32):
//     dW [kh*kw*Ci, Co]  = cols^T   * dz        cols = im2col(x)  [B*Ho*Wo, kh*kw*Ci]
//     dcols              = dz       * W^T       dx = col2im(dcols)                       dz = dy * relu'(y) * scale
// over chunks of images sized by the caller's workspace; 1x1 / stride-1 convolutions skip im2col / col2im altogether
// (cols = x, dx = dcols).  col2im is written as a GATHER (every input pixel sums the filter taps that touched it), so
// it needs no atomics and is run-to-run deterministic.
#include <algorithm>

#include "vqa_common.h"

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int grid_for(int64_t items, int cap = 65536) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(items, 256), cap));
}

// dz = dy * (relu ? y > 0 : 1) * scale ; dres = dy * (relu ? y > 0 : 1)   (either output may be NULL)
__global__ __launch_bounds__(256) void conv_bwd_dz_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ scale, float* __restrict__ dz,
                                                          float* __restrict__ dres, int64_t total, int Co, int relu) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float g = dy[i];
        if (relu && !(y[i] > 0.f)) g = 0.f;
        if (dres != nullptr) dres[i] = g;
        if (dz != nullptr) dz[i] = scale != nullptr ? g * scale[i % Co] : g;
    }
}

// dshift[co] = sum over pixels of dy * relu'(y): one workgroup per 64 channels, rows strided over the threads
__global__ __launch_bounds__(256) void conv_bwd_dshift_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              float* __restrict__ dshift, int64_t M, int Co, int relu) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Co)
        for (int64_t m = rg; m < M; m += 4) {
            const float g = dy[m * Co + c];
            s += (relu && !(y[m * Co + c] > 0.f)) ? 0.f : g;
        }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < Co) dshift[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dx[b, iy, ix, ci] = sum over the taps (ky, kx) whose output pixel (oy, ox) = ((iy + pt - ky) / s, (ix + pl - kx) / s) exists
//                     of dcol[(b, oy, ox), (ky * kw + kx) * Ci + ci]
__global__ __launch_bounds__(256) void col2im_gather_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B,
                                                            int Hi, int Wi, int Ci, int kh, int kw, int stride, int pad_t,
                                                            int pad_l, int Ho, int Wo) {
    const int64_t total = (int64_t)B * Hi * Wi * Ci;
    const int K = kh * kw * Ci;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % Ci);
        int64_t t = i / Ci;
        const int ix = (int)(t % Wi); t /= Wi;
        const int iy = (int)(t % Hi);
        const int b = (int)(t / Hi);
        float s = 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const int ny = iy + pad_t - ky;
            if (ny < 0 || ny % stride != 0) continue;
            const int oy = ny / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int nx = ix + pad_l - kx;
                if (nx < 0 || nx % stride != 0) continue;
                const int ox = nx / stride;
                if (ox >= Wo) continue;
                s += dcol[(((int64_t)b * Ho + oy) * Wo + ox) * K + (ky * kw + kx) * Ci + ci];
            }
        }
        dx[i] = s;
    }
}

#define TRY(x)                           \
    do {                                 \
        int rc__ = (x);                  \
        if (rc__ != VQA_OK) return rc__; \
    } while (0)

}  // namespace

// floats of scratch vqa_conv2d_nhwc_bwd needs to process `chunk_images` images at a time (>= 1; B = everything at once)
extern "C" int64_t vqa_conv2d_bwd_workspace_floats(int B, int Ho, int Wo, int Ci, int kh, int kw, int Co, int chunk_images) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Ci <= 0 || kh <= 0 || kw <= 0 || Co <= 0 || chunk_images <= 0) return VQA_ERR_ARG;
    const int64_t K = (int64_t)kh * kw * Ci, Mc = (int64_t)std::min(chunk_images, B) * Ho * Wo, M = (int64_t)B * Ho * Wo;
    const int64_t g1 = vqa_gemm_workspace_floats(1, 0, (int)K, Co, (int)Mc, 0), g2 = vqa_gemm_workspace_floats(0, 1, (int)Mc, (int)K, Co, 0);
    return M * Co + 2 * Mc * K + std::max<int64_t>(std::max(g1, g2), 4) + 64;
}

extern "C" int vqa_conv2d_nhwc_bwd(const float* x, int B, int Hi, int Wi, int Ci, const float* w, int kh, int kw, int Co,
                                   int stride, int pad_t, int pad_l, int Ho, int Wo, const float* scale, const float* y,
                                   int relu, const float* dy, float* dx, float* dw, float* dshift, float* dresidual,
                                   float* workspace, int64_t workspace_floats, void* stream) {
    VQA_REQUIRE(x && w && dy && workspace && B > 0 && Hi > 0 && Wi > 0 && Ci > 0 && kh > 0 && kw > 0 && Co > 0 && stride > 0 &&
                    Ho > 0 && Wo > 0,
                VQA_ERR_ARG);
    VQA_REQUIRE(!relu || y != nullptr, VQA_ERR_ARG);
    VQA_REQUIRE(Ci % 4 == 0 && Co % 4 == 0, VQA_ERR_ALIGN);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t K = (int64_t)kh * kw * Ci, HW = (int64_t)Ho * Wo, M = (int64_t)B * HW;
    const bool pointwise = kh == 1 && kw == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == Hi && Wo == Wi;
    // the largest chunk of images the workspace holds
    int chunk = 0;
    for (int cnd = B; cnd >= 1; cnd = cnd > 1 ? cnd / 2 : 0)
        if (vqa_conv2d_bwd_workspace_floats(B, Ho, Wo, Ci, kh, kw, Co, cnd) <= workspace_floats) { chunk = cnd; break; }
    VQA_REQUIRE(chunk >= 1, VQA_ERR_WORKSPACE);
    float* dz = workspace;
    float* cols = dz + M * Co;
    float* dcols = cols + (int64_t)chunk * HW * K;
    float* gws = dcols + (int64_t)chunk * HW * K;
    const int64_t gws_floats = workspace + workspace_floats - gws;

    hipLaunchKernelGGL(conv_bwd_dz_kernel, dim3(grid_for(M * Co)), dim3(256), 0, st, dy, y, scale, dz, dresidual, M * Co, Co, relu);
    VQA_CHECK_LAUNCH();
    if (dshift != nullptr) {
        hipLaunchKernelGGL(conv_bwd_dshift_kernel, dim3((unsigned)cdiv(Co, 64)), dim3(256), 0, st, dy, y, dshift, M, Co, relu);
        VQA_CHECK_LAUNCH();
    }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        const int64_t Mc = (int64_t)nb * HW;
        const float* xc = x + (int64_t)b0 * Hi * Wi * Ci;
        const float* dzc = dz + (int64_t)b0 * HW * Co;
        const float* a = xc;                 // left operand of the weight gradient: the columns, or x itself
        if (!pointwise && dw != nullptr) {
            TRY(vqa_im2col_nhwc(xc, nb, Hi, Wi, Ci, kh, kw, stride, pad_t, pad_l, Ho, Wo, nullptr, cols, (int)K, stream));
            a = cols;
        }
        if (dw != nullptr)   // dW (+)= cols^T * dz  (later chunks accumulate onto the first)
            TRY(vqa_gemm_f32(1, 0, (int)K, Co, (int)Mc, a, (int)K, dzc, Co, dw, Co, nullptr, b0 > 0 ? dw : nullptr, Co, 0, gws,
                             gws_floats, stream));
        if (dx != nullptr) {
            float* dxc = dx + (int64_t)b0 * Hi * Wi * Ci;
            float* target = pointwise ? dxc : dcols;
            TRY(vqa_gemm_f32(0, 1, (int)Mc, (int)K, Co, dzc, Co, w, Co, target, (int)K, nullptr, nullptr, 0, 0, gws, gws_floats,
                             stream));
            if (!pointwise) {
                hipLaunchKernelGGL(col2im_gather_kernel, dim3(grid_for((int64_t)nb * Hi * Wi * Ci)), dim3(256), 0, st, dcols,
                                   dxc, nb, Hi, Wi, Ci, kh, kw, stride, pad_t, pad_l, Ho, Wo);
                VQA_CHECK_LAUNCH();
            }
        }
    }
    return VQA_OK;
}
