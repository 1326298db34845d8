// Loss / metrics and the optimiser (row a12) for gfx950.
//
// Loss: tf.nn.sigmoid_cross_entropy_with_logits x train mask, argmax (first
// maximum on ties), one-hot VQA scores and the masked max-scores that feed the 13
// report scalars -- vqa/model_vlmap_answer.py:192-288 (model_standard.py:281-374).
// One workgroup per sample streams logits+targets once (HBM-bound, 24 KB/sample)
// and optionally emits dz = (sigmoid(z)-t)*mask/B for the backward pass in the
// same sweep.
//
// Optimiser: tf.contrib.layers.optimize_loss(Adam, clip_gradients=20.0)
// (vqa/trainer.py:106-114) = clip_by_global_norm then Adam, on flat buffers.
#include "vqa_common.h"

namespace {

struct ArgMax { float v; int i; };

__device__ __forceinline__ ArgMax argmax_combine(ArgMax a, ArgMax b) {
    // larger value wins; on ties the LOWER index (tf.argmax returns the first maximum)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// TWO = 1: the two-headed form of vqa/model_vlmap_answer_vqa_all2.py:226-244 -- z = fixed head, z2 = tuned head:
// loss = ce(z) * train_mask + ce(z2), report loss = ce(z) + ce(z2), pred = argmax(z * test_mask + z2 * train_mask),
// dz = d ce(z) * train_mask, dz2 = d ce(z2) (unmasked), zsum = z + z2 (output['logit'])
// TWO = 2: vqa/model_vlmap_answer_vqa_all.py:234-244 -- the tuned term sees the SUM and both terms are train-masked:
// loss = (ce(z) + ce(z + z2)) * train_mask, report loss the same unmasked, pred = argmax(z + z2),
// dz = (d ce(z) + d ce(z + z2)) * train_mask, dz2 = d ce(z + z2) * train_mask
template <int TWO>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ z, const float* __restrict__ z2,
                                                       const float* __restrict__ tgt,
                                                       const float* __restrict__ train_m,
                                                       const float* __restrict__ obj_m,
                                                       const float* __restrict__ attr_m,
                                                       const float* __restrict__ exist_m, int use_train_mask,
                                                       float inv_batch, float* __restrict__ stats,
                                                       int32_t* __restrict__ pred, float* __restrict__ dz,
                                                       float* __restrict__ dz2, float* __restrict__ zsum, int A) {
    __shared__ float redv[4];
    __shared__ int redi[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* zb = z + (int64_t)b * A;
    const float* tb = tgt + (int64_t)b * A;

    float l_train = 0.f, l_all = 0.f;
    // masked maxima start at the value of a fully masked row: max(tgt*0) = 0 only if A > 0
    float mx_exist = -INFINITY, mx_train_exist = -INFINITY, mx_tobj = -INFINITY, mx_tattr = -INFINITY,
          mx_test = -INFINITY, mx_test_exist = -INFINITY, mx_train = -INFINITY;
    ArgMax am{-INFINITY, 0x7fffffff};
    for (int a = threadIdx.x; a < A; a += 256) {
        float x = zb[a];
        const float t = tb[a];
        const float tr = train_m[a], te = 1.f - tr, ob = obj_m[a], at = attr_m[a], ex = exist_m[a];
        const float ell = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
        if (dz != nullptr) {
            float g = (sigmoidf_stable(x) - t) * inv_batch;
            if (use_train_mask || TWO) g *= tr;
            if (TWO != 2) dz[(int64_t)b * A + a] = g;
        }
        if (TWO == 2) {
            const float y = x + z2[(int64_t)b * A + a];
            const float ell2 = fmaxf(y, 0.f) - y * t + log1pf(expf(-fabsf(y)));
            l_all += ell + ell2;
            l_train += (ell + ell2) * tr;
            if (dz2 != nullptr) {
                const float g2 = (sigmoidf_stable(y) - t) * inv_batch * tr;
                dz2[(int64_t)b * A + a] = g2;
                dz[(int64_t)b * A + a] = (sigmoidf_stable(x) - t) * inv_batch * tr + g2;
            }
            if (zsum != nullptr) zsum[(int64_t)b * A + a] = y;
            x = y;                             // pred = argmax(logit + tuned_logit) (:244)
        } else if (TWO == 1) {
            const float y = z2[(int64_t)b * A + a];
            const float ell2 = fmaxf(y, 0.f) - y * t + log1pf(expf(-fabsf(y)));
            l_all += ell + ell2;
            l_train += ell * tr + ell2;
            if (dz2 != nullptr) dz2[(int64_t)b * A + a] = (sigmoidf_stable(y) - t) * inv_batch;
            if (zsum != nullptr) zsum[(int64_t)b * A + a] = x + y;
            x = x * te + y * tr;               // the logit the prediction is taken from (:243-244)
        } else {
            l_all += ell;
            l_train += ell * tr;
        }
        if (x > am.v) { am.v = x; am.i = a; }  // strided ascending a per thread: first max kept
        mx_exist = fmaxf(mx_exist, t * ex);
        mx_train_exist = fmaxf(mx_train_exist, t * ex * tr);
        mx_tobj = fmaxf(mx_tobj, t * te * ob);
        mx_tattr = fmaxf(mx_tattr, t * te * at);
        mx_test = fmaxf(mx_test, t * te);
        mx_test_exist = fmaxf(mx_test_exist, t * ex * te);
        mx_train = fmaxf(mx_train, t * tr);
    }
    // argmax: wave shuffle reduce then across the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax other{__shfl_xor(am.v, o, 64), __shfl_xor(am.i, o, 64)};
        am = argmax_combine(am, other);
    }
    // every block-wide reduction of the row behind ONE barrier: per-wave partials to LDS, then each
    // thread combines the four waves (fixed order, so the sums are deterministic)
    __shared__ float part[4][9];
    {
        const float v[9] = {wave_sum(l_all), wave_sum(l_train), wave_max(mx_exist), wave_max(mx_train_exist),
                            wave_max(mx_tobj), wave_max(mx_tattr), wave_max(mx_test), wave_max(mx_test_exist),
                            wave_max(mx_train)};
        if (lane == 0) {
            redv[wave] = am.v; redi[wave] = am.i;
#pragma unroll
            for (int k = 0; k < 9; ++k) part[wave][k] = v[k];
        }
    }
    __syncthreads();
    l_all = ((part[0][0] + part[1][0]) + part[2][0]) + part[3][0];
    l_train = ((part[0][1] + part[1][1]) + part[2][1]) + part[3][1];
    auto bmax = [&](int k) { return fmaxf(fmaxf(part[0][k], part[1][k]), fmaxf(part[2][k], part[3][k])); };
    mx_exist = bmax(2); mx_train_exist = bmax(3); mx_tobj = bmax(4); mx_tattr = bmax(5);
    mx_test = bmax(6); mx_test_exist = bmax(7); mx_train = bmax(8);
    ArgMax best{redv[0], redi[0]};
    for (int k = 1; k < 4; ++k) best = argmax_combine(best, ArgMax{redv[k], redi[k]});
    int p = best.i;
    if (p < 0 || p >= A) p = 0;  // all-NaN row: tf.argmax returns 0
    if (threadIdx.x == 0) {
        const float tp = tb[p];
        const float tr = train_m[p], te = 1.f - tr, ob = obj_m[p], at = attr_m[p], ex = exist_m[p];
        float* s = stats + (int64_t)b * VQA_STAT_COUNT;
        s[VQA_STAT_LOSS_TRAIN] = (use_train_mask || TWO) ? l_train : l_all;
        s[VQA_STAT_LOSS_REPORT] = l_all;
        s[VQA_STAT_ALL_SCORE] = tp;
        s[VQA_STAT_EXIST_SCORE] = tp * ex;
        s[VQA_STAT_TEST_SCORE] = tp * te;
        s[VQA_STAT_TEST_OBJ_SCORE] = tp * te * ob;
        s[VQA_STAT_TEST_ATTR_SCORE] = tp * te * at;
        s[VQA_STAT_TRAIN_EXIST_SCORE] = tp * ex * tr;
        s[VQA_STAT_MAX_EXIST] = mx_exist;
        s[VQA_STAT_MAX_TRAIN_EXIST] = mx_train_exist;
        s[VQA_STAT_TEST_OBJ_MAX] = mx_tobj;
        s[VQA_STAT_TEST_ATTR_MAX] = mx_tattr;
        s[VQA_STAT_TEST_MAX] = mx_test;
        s[VQA_STAT_TEST_MAX_EXIST] = mx_test_exist;
        s[VQA_STAT_MAX_TRAIN] = mx_train;
        s[15] = 0.f;
        pred[b] = p;
    }
}

// report[13]: means over the batch + guarded ratios  where(den == 0, den, num/den)
__global__ __launch_bounds__(1024) void report_reduce_kernel(const float* __restrict__ stats, int B,
                                                             float* __restrict__ report) {
    // one wave per statistic (16 waves >= VQA_STAT_COUNT): fixed summation order, no block-wide reductions
    static_assert(VQA_STAT_COUNT <= 16, "one wave per statistic");
    __shared__ float mean[VQA_STAT_COUNT];
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    if (k < VQA_STAT_COUNT) {
        float s = 0.f;
        for (int b = lane; b < B; b += 64) s += stats[(int64_t)b * VQA_STAT_COUNT + k];
        s = wave_sum(s);
        if (lane == 0) mean[k] = s / (float)B;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        auto ratio = [](float num, float den) { return den == 0.f ? den : num / den; };
        report[0] = mean[VQA_STAT_LOSS_TRAIN];
        report[1] = mean[VQA_STAT_LOSS_REPORT];
        report[2] = mean[VQA_STAT_ALL_SCORE];
        report[3] = mean[VQA_STAT_EXIST_SCORE];
        report[4] = mean[VQA_STAT_TEST_SCORE];
        report[5] = ratio(mean[VQA_STAT_TEST_SCORE], mean[VQA_STAT_TEST_MAX]);
        report[6] = ratio(mean[VQA_STAT_TEST_OBJ_SCORE], mean[VQA_STAT_TEST_OBJ_MAX]);
        report[7] = ratio(mean[VQA_STAT_TEST_ATTR_SCORE], mean[VQA_STAT_TEST_ATTR_MAX]);
        report[8] = ratio(mean[VQA_STAT_EXIST_SCORE], mean[VQA_STAT_MAX_EXIST]);
        report[9] = ratio(mean[VQA_STAT_TRAIN_EXIST_SCORE], mean[VQA_STAT_MAX_TRAIN_EXIST]);
        report[10] = mean[VQA_STAT_MAX_EXIST];
        report[11] = mean[VQA_STAT_TEST_MAX];
        report[12] = mean[VQA_STAT_TEST_MAX_EXIST];
    }
}

// n-way softmax cross-entropy with a validity mask, top-1 and top-k hit (cfg-5 pre-training model:
// n_way_classification_loss, vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:675-706).
// One workgroup per row.  stats[row] = {ce*valid, (argmax == label)*valid, (label in top-k)*valid, valid};
// dz[row,:] = (softmax - onehot) * valid * inv_valid_sum[0]   (inv_valid_sum = 1 / sum(valid), on device)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ z, const int32_t* __restrict__ label,
                                                         const float* __restrict__ valid, int topk,
                                                         const float* __restrict__ inv_valid_sum,
                                                         float* __restrict__ stats, float* __restrict__ dz, int A) {
    __shared__ float red[16];
    __shared__ float redv[4];
    __shared__ int redi[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* zb = z + (int64_t)b * A;
    const int lab = label[b];
    const float zl = zb[min(max(lab, 0), A - 1)];
    float mx = -INFINITY;
    ArgMax am{-INFINITY, 0x7fffffff};
    int rank = 0;   // entries that tf.nn.top_k orders before the label: larger, or equal with a lower index
    for (int a = threadIdx.x; a < A; a += 256) {
        const float x = zb[a];
        mx = fmaxf(mx, x);
        if (x > am.v) { am.v = x; am.i = a; }
        rank += (x > zl || (x == zl && a < lab)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax other{__shfl_xor(am.v, o, 64), __shfl_xor(am.i, o, 64)};
        am = argmax_combine(am, other);
    }
    mx = wave_max(mx);
    if (lane == 0) { redv[wave] = am.v; redi[wave] = am.i; red[8 + wave] = mx; }
    const float frank = block_sum((float)rank, red);      // (its barriers also publish redv/redi/red[8..])
    mx = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
    ArgMax best{redv[0], redi[0]};
    for (int k = 1; k < 4; ++k) best = argmax_combine(best, ArgMax{redv[k], redi[k]});
    float se = 0.f;
    for (int a = threadIdx.x; a < A; a += 256) se += expf(zb[a] - mx);
    se = block_sum(se, red);
    const float lse = mx + logf(se);
    const float vm = valid[b];
    if (dz != nullptr) {
        const float sc = vm * inv_valid_sum[0];
        for (int a = threadIdx.x; a < A; a += 256)
            dz[(int64_t)b * A + a] = (expf(zb[a] - lse) - (a == lab ? 1.f : 0.f)) * sc;
    }
    if (threadIdx.x == 0) {
        float* s = stats + (int64_t)b * 4;
        s[0] = (lse - zl) * vm;
        s[1] = (best.i == lab ? 1.f : 0.f) * vm;
        s[2] = (frank < (float)topk ? 1.f : 0.f) * vm;
        s[3] = vm;
    }
}

// The same with the row held in registers (A <= 4096, A % 4 == 0, 16-byte aligned rows): thread t owns the float4s
// t, t+256, ... (NV of them), so the logits are read ONCE with 16-byte loads, all of a thread's loads in flight
// together, instead of three dependent scalar passes over the row, and dz leaves as 16-byte stores.  Same arithmetic per
// element as softmax_ce_kernel (exp(z - lse) for dz); only the order of the sum of exponentials differs.
template <int NV>
__global__ __launch_bounds__(256) void softmax_ce_reg_kernel(const float* __restrict__ z, const int32_t* __restrict__ label,
                                                             const float* __restrict__ valid, int topk,
                                                             const float* __restrict__ inv_valid_sum,
                                                             float* __restrict__ stats, float* __restrict__ dz, int A) {
    __shared__ float red[16];
    __shared__ float redv[4];
    __shared__ int redi[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A4 = A >> 2;
    const float4* zb = reinterpret_cast<const float4*>(z + (int64_t)b * A);
    float x[NV][4];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x + 256 * k;
        const float4 v = i < A4 ? zb[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        x[k][0] = v.x; x[k][1] = v.y; x[k][2] = v.z; x[k][3] = v.w;
    }
    const int lab = label[b];
    const float zl = z[(int64_t)b * A + min(max(lab, 0), A - 1)];
    float mx = -INFINITY;
    ArgMax am{-INFINITY, 0x7fffffff};
    int rank = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = 4 * (threadIdx.x + 256 * k) + j;      // padding (a >= A) holds -inf: never a maximum, never ranked
            const float v = x[k][j];
            mx = fmaxf(mx, v);
            if (v > am.v) { am.v = v; am.i = a; }
            rank += (v > zl || (v == zl && a < lab)) ? 1 : 0;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax other{__shfl_xor(am.v, o, 64), __shfl_xor(am.i, o, 64)};
        am = argmax_combine(am, other);
    }
    mx = wave_max(mx);
    if (lane == 0) { redv[wave] = am.v; redi[wave] = am.i; red[8 + wave] = mx; }
    const float frank = block_sum((float)rank, red);
    mx = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
    ArgMax best{redv[0], redi[0]};
    for (int k = 1; k < 4; ++k) best = argmax_combine(best, ArgMax{redv[k], redi[k]});
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) se += expf(x[k][j] - mx);     // exp(-inf) = 0 for the padding
    se = block_sum(se, red);
    const float lse = mx + logf(se);
    const float vm = valid[b];
    if (dz != nullptr) {
        const float sc = vm * inv_valid_sum[0];
        float4* db = reinterpret_cast<float4*>(dz + (int64_t)b * A);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < A4) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (expf(x[k][j] - lse) - (4 * i + j == lab ? 1.f : 0.f)) * sc;
                db[i] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (threadIdx.x == 0) {
        float* s = stats + (int64_t)b * 4;
        s[0] = (lse - zl) * vm;
        s[1] = (best.i == lab ? 1.f : 0.f) * vm;
        s[2] = (frank < (float)topk ? 1.f : 0.f) * vm;
        s[3] = vm;
    }
}
int g_softmax_reg = 1;   // A/B switch (vqa_softmax_set_fast)

// ------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n,
                                                            float* __restrict__ partial) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = reinterpret_cast<const float4*>(g)[i];
        s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nparts,
                                                          const float* __restrict__ extra, float* __restrict__ out) {
    __shared__ double redd[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = redd[0] + redd[1] + redd[2] + redd[3];
        if (extra != nullptr) t += (double)extra[0];
        out[0] = (float)t;
    }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const float* __restrict__ norm_sq, float clip, float lr_host,
                                                        const float* __restrict__ lr_dev, float b1, float b2, float eps) {
    // lr_t from device memory when the launch is replayed from a captured graph (the bias-corrected rate changes every
    // step and a replay cannot change a by-value argument)
    const float lr_t = lr_dev != nullptr ? lr_dev[0] : lr_host;
    // clip_by_global_norm: g * clip / max(norm, clip)
    const float scale = (norm_sq != nullptr) ? clip / fmaxf(sqrtf(norm_sq[0]), clip) : 1.f;
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* Vv = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gs = G[j] * scale;
            M[j] = b1 * M[j] + (1.f - b1) * gs;
            Vv[j] = b2 * Vv[j] + (1.f - b2) * gs * gs;
            P[j] -= lr_t * M[j] / (sqrtf(Vv[j]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gs = g[i] * scale;
            m[i] = b1 * m[i] + (1.f - b1) * gs;
            v[i] = b2 * v[i] + (1.f - b2) * gs * gs;
            p[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
        }
}

// step <- step + 1; lr_t = lr * sqrt(1 - b2^step) / (1 - b1^step): the host arithmetic of an eager step, on the device so
// that a captured train-step graph carries its own step count
__global__ void adam_lr_step_kernel(int64_t* __restrict__ step, const double* __restrict__ lr, double b1, double b2,
                                    float* __restrict__ lr_t) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t t = step[0] + 1;
    step[0] = t;
    // the host's expression, term by term in double: lr * sqrt(1 - b2^t) / (1 - b1^t)
    lr_t[0] = (float)(lr[0] * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int sumsq_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n / 4, 256 * 4), 1024)); }

const char* const kReportKeys[VQA_REPORT_COUNT] = {
    "answer_train_loss", "answer_report_loss", "answer_acc", "exist_acc", "test_acc", "normal_test_acc",
    "normal_test_object_acc", "normal_test_attribute_acc", "normal_exist_acc", "normal_train_exist_acc",
    "max_exist_acc", "test_max_acc", "test_max_exist_acc"};

}  // namespace

extern "C" const char* vqa_report_key(int i) { return (i >= 0 && i < VQA_REPORT_COUNT) ? kReportKeys[i] : nullptr; }

extern "C" int vqa_loss_fwd(const float* z, const float* target, const float* train_mask, const float* obj_mask,
                            const float* attr_mask, const float* exist_mask, int use_train_mask_in_loss,
                            float inv_batch, float* stats, int32_t* pred, float* dz, int B, int A, void* stream) {
    VQA_REQUIRE(z && target && train_mask && obj_mask && attr_mask && exist_mask && stats && pred, VQA_ERR_ARG);
    VQA_REQUIRE(B >= 0 && A > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(loss_fwd_kernel<0>, dim3(B), dim3(256), 0, (hipStream_t)stream, z, (const float*)nullptr, target,
                       train_mask, obj_mask, attr_mask, exist_mask, use_train_mask_in_loss, inv_batch, stats, pred, dz,
                       (float*)nullptr, (float*)nullptr, A);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

// z1m[b, a] = exist[a] ? z[b, a] : min_a z[b, a] ; rowmin[b] = that minimum   (vqa/model_vlmap_answer_vqa_all.py:192-194)
__global__ __launch_bounds__(256) void rowmin_mask_fwd_kernel(const float* __restrict__ z, const float* __restrict__ exist,
                                                              float* __restrict__ zm, float* __restrict__ rowmin, int A) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* zb = z + (int64_t)b * A;
    float m = INFINITY;
    for (int a = threadIdx.x; a < A; a += 256) m = fminf(m, zb[a]);
    m = -wave_max(-m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    for (int a = threadIdx.x; a < A; a += 256) zm[(int64_t)b * A + a] = exist[a] != 0.f ? zb[a] : m;
    if (threadIdx.x == 0) rowmin[b] = m;
}

// back through it, in place: dz[b, a] = dz[b, a] * exist[a] + (z[b, a] == rowmin[b]) * (sum_a' dz[b, a'] * (1 - exist[a'])) / #ties
// (tf.reduce_min hands its gradient to the minimum, split evenly over ties)
__global__ __launch_bounds__(256) void rowmin_mask_bwd_kernel(float* __restrict__ dz, const float* __restrict__ z,
                                                              const float* __restrict__ rowmin, const float* __restrict__ exist,
                                                              int A) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    float* db = dz + (int64_t)b * A;
    const float* zb = z + (int64_t)b * A;
    const float m = rowmin[b];
    float s = 0.f, cnt = 0.f;
    for (int a = threadIdx.x; a < A; a += 256) {
        s += db[a] * (1.f - exist[a]);
        cnt += (zb[a] == m) ? 1.f : 0.f;
    }
    s = block_sum(s, red);
    cnt = block_sum(cnt, red);
    const float share = cnt > 0.f ? s / cnt : 0.f;
    for (int a = threadIdx.x; a < A; a += 256) db[a] = db[a] * exist[a] + ((zb[a] == m) ? share : 0.f);
}

extern "C" int vqa_rowmin_mask_fwd(const float* z, const float* exist_mask, float* z_masked, float* rowmin, int B, int A,
                                   void* stream) {
    VQA_REQUIRE(z && exist_mask && z_masked && rowmin && B >= 0 && A > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(rowmin_mask_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, z, exist_mask, z_masked, rowmin, A);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_rowmin_mask_bwd(float* dz, const float* z, const float* rowmin, const float* exist_mask, int B, int A,
                                   void* stream) {
    VQA_REQUIRE(dz && z && rowmin && exist_mask && B >= 0 && A > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(rowmin_mask_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dz, z, rowmin, exist_mask, A);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_loss2_fwd(const float* z_fixed, const float* z_tuned, const float* target, const float* train_mask,
                             const float* obj_mask, const float* attr_mask, const float* exist_mask, float inv_batch,
                             float* stats, int32_t* pred, float* dz_fixed, float* dz_tuned, float* z_sum, int sum_mode,
                             int B, int A, void* stream) {
    VQA_REQUIRE(z_fixed && z_tuned && target && train_mask && obj_mask && attr_mask && exist_mask && stats && pred,
                VQA_ERR_ARG);
    VQA_REQUIRE(B >= 0 && A > 0 && (dz_fixed == nullptr) == (dz_tuned == nullptr), VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    if (sum_mode)
        hipLaunchKernelGGL(loss_fwd_kernel<2>, dim3(B), dim3(256), 0, (hipStream_t)stream, z_fixed, z_tuned, target,
                           train_mask, obj_mask, attr_mask, exist_mask, 1, inv_batch, stats, pred, dz_fixed, dz_tuned, z_sum, A);
    else
        hipLaunchKernelGGL(loss_fwd_kernel<1>, dim3(B), dim3(256), 0, (hipStream_t)stream, z_fixed, z_tuned, target,
                           train_mask, obj_mask, attr_mask, exist_mask, 1, inv_batch, stats, pred, dz_fixed, dz_tuned, z_sum, A);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_softmax_ce_fwd(const float* z, const int32_t* label, const float* valid, int topk,
                                  const float* inv_valid_sum, float* stats, float* dz, int rows, int A, void* stream) {
    VQA_REQUIRE(z && label && valid && stats && rows >= 0 && A > 0 && topk > 0, VQA_ERR_ARG);
    VQA_REQUIRE(dz == nullptr || inv_valid_sum != nullptr, VQA_ERR_ARG);
    if (rows == 0) return VQA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (g_softmax_reg && A % 4 == 0 && A <= 4096 && vqa_aligned16(z) && (dz == nullptr || vqa_aligned16(dz))) {
        switch ((A / 4 + 255) / 256) {
        case 1: hipLaunchKernelGGL(softmax_ce_reg_kernel<1>, dim3(rows), dim3(256), 0, st, z, label, valid, topk, inv_valid_sum, stats, dz, A); break;
        case 2: hipLaunchKernelGGL(softmax_ce_reg_kernel<2>, dim3(rows), dim3(256), 0, st, z, label, valid, topk, inv_valid_sum, stats, dz, A); break;
        case 3: hipLaunchKernelGGL(softmax_ce_reg_kernel<3>, dim3(rows), dim3(256), 0, st, z, label, valid, topk, inv_valid_sum, stats, dz, A); break;
        default: hipLaunchKernelGGL(softmax_ce_reg_kernel<4>, dim3(rows), dim3(256), 0, st, z, label, valid, topk, inv_valid_sum, stats, dz, A); break;
        }
    } else {
        hipLaunchKernelGGL(softmax_ce_kernel, dim3(rows), dim3(256), 0, st, z, label, valid, topk, inv_valid_sum, stats, dz, A);
    }
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_softmax_set_fast(int on) {
    g_softmax_reg = on ? 1 : 0;
    return VQA_OK;
}

extern "C" int vqa_report_reduce(const float* stats, int B, float* report, void* stream) {
    VQA_REQUIRE(stats && report && B > 0, VQA_ERR_ARG);
    hipLaunchKernelGGL(report_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stats, B, report);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int64_t vqa_sumsq_workspace_floats(int64_t n) { return sumsq_blocks(n); }

extern "C" int vqa_sumsq(const float* g, int64_t n, const float* extra_sq, float* norm_sq_out, float* partial,
                         int64_t partial_floats, void* stream) {
    VQA_REQUIRE(g && norm_sq_out && partial && n >= 0, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_aligned16(g), VQA_ERR_ALIGN);
    const int nb = sumsq_blocks(n);
    VQA_REQUIRE(partial_floats >= nb, VQA_ERR_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial);
    VQA_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, extra_sq, norm_sq_out);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* norm_sq,
                             float clip, float lr_t, float beta1, float beta2, float eps, void* stream) {
    VQA_REQUIRE(p && g && m && v && n >= 0, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_aligned16(p) && vqa_aligned16(g) && vqa_aligned16(m) && vqa_aligned16(v), VQA_ERR_ALIGN);
    if (n == 0) return VQA_OK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n / 4 + 1, 256), 4096));
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, norm_sq, clip,
                       lr_t, (const float*)nullptr, beta1, beta2, eps);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_adam_lr_step(int64_t* step_dev, const double* lr_dev, double beta1, double beta2, float* lr_t_dev,
                                void* stream) {
    VQA_REQUIRE(step_dev && lr_dev && lr_t_dev, VQA_ERR_ARG);
    hipLaunchKernelGGL(adam_lr_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, lr_dev, beta1, beta2, lr_t_dev);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_clip_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* norm_sq,
                                 float clip, const float* lr_t_dev, float beta1, float beta2, float eps, void* stream) {
    VQA_REQUIRE(p && g && m && v && lr_t_dev && n >= 0, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_aligned16(p) && vqa_aligned16(g) && vqa_aligned16(m) && vqa_aligned16(v), VQA_ERR_ALIGN);
    if (n == 0) return VQA_OK;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n / 4 + 1, 256), 4096));
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, norm_sq, clip,
                       0.f, lr_t_dev, beta1, beta2, eps);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
