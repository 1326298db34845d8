// HBM-bound helpers of the region-feature extractor (NHWC, gfx950): conv1's im2col with
// the RGB mean subtraction fused, pool1, the bottleneck shortcut subsample and
// tf.image.crop_and_resize.  Reference: vlmap/modules.py:143-216 (slim resnet_v1 +
// crop_and_resize), TF semantics per SURVEY.md 5.2-10/11.
#include "vqa_common.h"

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int grid_for(int64_t items, int cap = 8192) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(items, 256), cap));
}

// col[m, (ky*kw + kx)*Ci + ci] = x[b, oy*s - pt + ky, ox*s - pl + kx, ci] - mean[ci]   (0 outside, 0 in the K pad)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B,
                                                     int Hi, int Wi, int Ci, int kh, int kw, int stride, int pad_t,
                                                     int pad_l, int Ho, int Wo, int Kpad, float m0, float m1,
                                                     float m2, int has_mean) {
    const int64_t total = (int64_t)B * Ho * Wo * Kpad;
    const int K = kh * kw * Ci;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % Kpad);
        const int64_t m = i / Kpad;
        float v = 0.f;
        if (k < K) {
            const int ci = k % Ci, tap = k / Ci;
            const int ky = tap / kw, kx = tap % kw;
            const int ox = (int)(m % Wo);
            const int64_t t = m / Wo;
            const int oy = (int)(t % Ho), b = (int)(t / Ho);
            const int iy = oy * stride - pad_t + ky, ix = ox * stride - pad_l + kx;
            if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
                v = x[(((int64_t)b * Hi + iy) * Wi + ix) * Ci + ci];
                if (has_mean) v -= (ci == 0 ? m0 : (ci == 1 ? m1 : m2));
            }
        }
        col[i] = v;
    }
}

// slim pool1: 3x3 / stride 2 / 'SAME' (extra padding element at the end; padding never wins)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                           int Hi, int Wi, int C, int Ho, int Wo, int pad_t,
                                                           int pad_l) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - pad_t + ky, ix = ox * 2 - pad_l + kx;
                if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
                    const float4 v = reinterpret_cast<const float4*>(x + (((int64_t)b * Hi + iy) * Wi + ix) * C)[c4];
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        reinterpret_cast<float4*>(y + (((int64_t)b * Ho + oy) * Wo + ox) * C)[c4] = m;
    }
}

// resnet_utils.subsample: y[b, oy, ox, :] = x[b, oy*f, ox*f, :]
__global__ __launch_bounds__(256) void subsample_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                        int Hi, int Wi, int C, int Ho, int Wo, int f) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        reinterpret_cast<float4*>(y)[i] =
            reinterpret_cast<const float4*>(x + (((int64_t)b * Hi + oy * f) * Wi + ox * f) * C)[c4];
    }
}

// tf.image.crop_and_resize, bilinear, extrapolation value 0.  One workgroup per (box, crop pixel)
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ fmap,
                                                          const float* __restrict__ boxes,
                                                          const int32_t* __restrict__ box_ind,
                                                          float* __restrict__ out, int H, int W, int C, int ch,
                                                          int cw) {
    const int n = blockIdx.x / (ch * cw);
    const int pix = blockIdx.x % (ch * cw);
    const int yy = pix / cw, xx = pix % cw;
    const float y1 = boxes[n * 4 + 0], x1 = boxes[n * 4 + 1], y2 = boxes[n * 4 + 2], x2 = boxes[n * 4 + 3];
    const int b = box_ind[n];
    const float hs = ch > 1 ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.f;
    const float ws = cw > 1 ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.f;
    const float in_y = ch > 1 ? y1 * (float)(H - 1) + (float)yy * hs : 0.5f * (y1 + y2) * (float)(H - 1);
    const float in_x = cw > 1 ? x1 * (float)(W - 1) + (float)xx * ws : 0.5f * (x1 + x2) * (float)(W - 1);
    float* o = out + ((int64_t)n * ch * cw + pix) * C;
    const bool inside = !(in_y < 0.f || in_y > (float)(H - 1) || in_x < 0.f || in_x > (float)(W - 1));
    if (!inside) {
        for (int c = threadIdx.x; c < C; c += 256) o[c] = 0.f;
        return;
    }
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
    const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
    const float ly = in_y - (float)top, lx = in_x - (float)left;
    const float* base = fmap + (int64_t)b * H * W * C;
    const float* tl = base + ((int64_t)top * W + left) * C;
    const float* tr = base + ((int64_t)top * W + right) * C;
    const float* bl = base + ((int64_t)bot * W + left) * C;
    const float* br = base + ((int64_t)bot * W + right) * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float t = tl[c] + (tr[c] - tl[c]) * lx;
        const float bt = bl[c] + (br[c] - bl[c]) * lx;
        o[c] = t + (bt - t) * ly;
    }
}

// y[b,h,w,0:3] = x[b,h,w,0:3] - mean, y[b,h,w,3] = 0: one 16-byte pixel per lane, so that a filter ROW of conv1
// (7 taps x 3 channels) becomes 8 x 4 = 32 contiguous floats and conv1 runs as an implicit GEMM (no im2col)
__global__ __launch_bounds__(256) void pad_c3c4_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix,
                                                       float m0, float m1, float m2) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const float* s = x + i * 3;
        reinterpret_cast<float4*>(y)[i] = make_float4(s[0] - m0, s[1] - m1, s[2] - m2, 0.f);
    }
}

}  // namespace

extern "C" int vqa_pad_c3c4_nhwc(const float* x, int B, int Hi, int Wi, const float* mean_host, float* y, void* stream) {
    VQA_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_aligned16(y), VQA_ERR_ALIGN);
    const int64_t npix = (int64_t)B * Hi * Wi;
    hipLaunchKernelGGL(pad_c3c4_kernel, dim3(grid_for(npix, 65536)), dim3(256), 0, (hipStream_t)stream, x, y, npix,
                       mean_host ? mean_host[0] : 0.f, mean_host ? mean_host[1] : 0.f, mean_host ? mean_host[2] : 0.f);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_im2col_nhwc(const float* x, int B, int Hi, int Wi, int Ci, int kh, int kw, int stride, int pad_t,
                               int pad_l, int Ho, int Wo, const float* mean_host, float* col, int Kpad, void* stream) {
    VQA_REQUIRE(x && col && B > 0 && Hi > 0 && Wi > 0 && Ci > 0 && kh > 0 && kw > 0 && stride > 0 && Ho > 0 && Wo > 0,
                VQA_ERR_ARG);
    VQA_REQUIRE(Kpad >= kh * kw * Ci && (mean_host == nullptr || Ci == 3), VQA_ERR_ARG);
    const int64_t total = (int64_t)B * Ho * Wo * Kpad;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, (hipStream_t)stream, x, col, B, Hi,
                       Wi, Ci, kh, kw, stride, pad_t, pad_l, Ho, Wo, Kpad, mean_host ? mean_host[0] : 0.f,
                       mean_host ? mean_host[1] : 0.f, mean_host ? mean_host[2] : 0.f, mean_host ? 1 : 0);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_maxpool3x3s2_same_nhwc(const float* x, int B, int Hi, int Wi, int C, float* y, void* stream) {
    VQA_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && C > 0, VQA_ERR_ARG);
    VQA_REQUIRE(C % 4 == 0 && vqa_aligned16(x) && vqa_aligned16(y), VQA_ERR_ALIGN);
    const int Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
    const int ph = std::max((Ho - 1) * 2 + 3 - Hi, 0), pw = std::max((Wo - 1) * 2 + 3 - Wi, 0);
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, (hipStream_t)stream, x, y, B,
                       Hi, Wi, C, Ho, Wo, ph / 2, pw / 2);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_subsample_nhwc(const float* x, int B, int Hi, int Wi, int C, int factor, float* y, void* stream) {
    VQA_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && C > 0 && factor > 0, VQA_ERR_ARG);
    VQA_REQUIRE(C % 4 == 0 && vqa_aligned16(x) && vqa_aligned16(y), VQA_ERR_ALIGN);
    const int Ho = (Hi - 1) / factor + 1, Wo = (Wi - 1) / factor + 1;
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(subsample_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, (hipStream_t)stream, x, y, B, Hi,
                       Wi, C, Ho, Wo, factor);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_crop_and_resize_nhwc(const float* fmap, int B, int H, int W, int C, const float* boxes,
                                        const int32_t* box_ind, int n_boxes, int crop_h, int crop_w, float* out,
                                        void* stream) {
    VQA_REQUIRE(fmap && boxes && box_ind && out && B > 0 && H > 0 && W > 0 && C > 0 && n_boxes >= 0 && crop_h > 0 &&
                    crop_w > 0,
                VQA_ERR_ARG);
    if (n_boxes == 0) return VQA_OK;
    hipLaunchKernelGGL(crop_resize_kernel, dim3(n_boxes * crop_h * crop_w), dim3(256), 0, (hipStream_t)stream, fmap,
                       boxes, box_ind, out, H, W, C, crop_h, crop_w);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
