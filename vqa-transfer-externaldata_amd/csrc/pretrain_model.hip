// Host-side composition of the cfg-5 pre-training model: one call enqueues the whole forward
// (or backward) pass of vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py on the caller's stream -- the
// counterpart of vqa_fusion_forward / vqa_fusion_backward for the fusion model.
//
// Per category k in {object, attribute} (reference lines relative to that file):
//   build_*_V_ft      :323-364, :414-455   6-d box -> FC 1024 + LN + ReLU for the 36 regions (once per image) and the
//                                          n = 5 key boxes, Hadamard attention + pooling over the raw features with the
//                                          x5 tile of V_ft never materialised (attn_pool_*_rep)
//   build_*_blank_fill:505-609             L_GloVe embedding -> GRU over the caption with a blank (length-sorted live
//                                          prefix) -> fusion MLP -> classifier -> masked softmax-CE, top-1 / top-5
//   build_*_wordset   :366-412, :457-503   tanh(word-set embedding) -> FC + LN + tanh -> the same fusion MLP
//   n_way_classification_loss :675-706
// LayerNorm variables of an fc_layer scope entered by several call sites: with VQA_FLAG_SHARED_LN (what TF 1.x builds:
// leaving the string-named scope zeroes its sub-scope counts, so the un-scoped layers.layer_norm gets the un-suffixed
// name again and AUTO_REUSE shares it -- oracle/pretrain_oracle.py) every call site uses slot 0 and the
// call sites' d_gamma / d_beta are accumulated; without the flag each call site owns slot [ki] (V_ft / blank-fill /
// wordset_ft) or [2 t + ki] (the four heads of the shared fusion MLP) in TF graph build order.
//
// The workspace layout is a function of the dims alone, so the host views named intermediates without copies.
#include <string.h>

#include <set>
#include <string>
#include <vector>

#include "vqa_common.h"

namespace {

// ---------------------------------------------------------------- small kernels of this model
// key6 = (y1, x1, y2, x2, y2 - y1, x2 - x1)   (:330-333)
__global__ __launch_bounds__(256) void box6_kernel(const float* __restrict__ key, float* __restrict__ key6, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 b = reinterpret_cast<const float4*>(key)[i];
    float* o = key6 + (int64_t)i * 6;
    o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = b.z - b.x; o[5] = b.w - b.y;
}

// valid[b, j] = j < num[b] ; inv_valid = 1 / sum(valid)     (tf.sequence_mask + reduce_sum, :675-706); under data
// parallelism the sum runs over the GLOBAL batch and is handed in (global_valid > 0)
__global__ __launch_bounds__(256) void valid_kernel(const int32_t* __restrict__ num, float* __restrict__ valid,
                                                    float* __restrict__ inv_valid, int B, int n, float global_valid) {
    __shared__ float red[16];
    float tot = 0.f;
    for (int i = threadIdx.x; i < B * n; i += 256) {
        const float v = (i % n) < num[i / n] ? 1.f : 0.f;
        valid[i] = v;
        tot += v;
    }
    tot = block_sum(tot, red);
    if (threadIdx.x == 0) inv_valid[0] = 1.f / (global_valid > 0.f ? global_valid : tot);
}

// out[i, :] = in[index[i], :]   (rows of `cols` 4-byte words; index == NULL copies)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* __restrict__ in, const int32_t* __restrict__ index,
                                                          uint32_t* __restrict__ out, int rows, int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        out[i] = in[(int64_t)(index ? index[r] : r) * cols + c];
    }
}

// out[i, :] = (in0 ++ in1)[index[i], :]: rows 0..rows0-1 of the virtual concatenation live in in0, the rest in in1
// (the captions of the two categories arrive as two arrays and are encoded as ONE batch)
__global__ __launch_bounds__(256) void gather_rows2_kernel(const uint32_t* __restrict__ in0, const uint32_t* __restrict__ in1,
                                                           const int32_t* __restrict__ index, uint32_t* __restrict__ out,
                                                           int rows, int cols, int rows0) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        int s = index ? index[r] : r;
        s = min(max(s, 0), rows - 1);
        out[i] = s < rows0 ? in0[(int64_t)s * cols + c] : in1[(int64_t)(s - rows0) * cols + c];
    }
}

// report[3 h + j] = sum_rows stats_h[:, j] * inv_valid_h (j = loss, top-1, top-k), report[12] = sum of the 4 losses
struct ReportArgs { const float* stats[4]; const float* inv[4]; int rows; };
__global__ __launch_bounds__(256) void pretrain_report_kernel(ReportArgs a, float* __restrict__ report) {
    __shared__ float red[16];
    __shared__ float loss[4];
    for (int h = 0; h < 4; ++h) {
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int i = threadIdx.x; i < a.rows; i += 256) s += a.stats[h][(int64_t)i * 4 + j];
            s = block_sum(s, red);
            if (threadIdx.x == 0) {
                const float v = s * a.inv[h][0];
                report[3 * h + j] = v;
                if (j == 0) loss[h] = v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) report[12] = ((loss[0] + loss[1]) + loss[2]) + loss[3];
}

// row stride of the time-major GRU inputs: W word-vector columns, the constant 1 (its row of the x-part weight gradient
// is the bias gradient), zero padding to 16 bytes
inline int64_t x_stride(int64_t W) { return ((W + 1 + 3) / 4) * 4; }

// ---------------------------------------------------------------- workspace layout
struct Entry { std::string name; int64_t off, n; };
struct Layout {
    std::vector<Entry> e;
    int64_t total = 0;
    void add(const std::string& name, int64_t n) {
        e.push_back({name, total, n});
        total += ((n * 4 + 255) / 256) * 256;
    }
    // a named view of `cnt` floats inside an existing entry (the four heads' tensors are slices of stacked blocks)
    void alias(const std::string& name, const std::string& base, int64_t off_floats, int64_t cnt) {
        const Entry* b = find(base);
        e.push_back({name, b->off + off_floats * 4, cnt});
    }
    const Entry* find(const std::string& name) const {
        for (const auto& x : e)
            if (x.name == name) return &x;
        return nullptr;
    }
};

const char* const KIND[2] = {"obj", "attr"};
const char* const HEAD[2] = {"bf", "ws"};      // blank fill, word set

int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

Layout make_layout(const vqa_pretrain_dims_t& d) {
    Layout L;
    const int64_t B = d.B, n = d.n, R = d.R, D = d.D, H = d.H, W = d.W, A = d.A, T = d.L, Bn = B * n;
    // The four heads (ln = 2 t + k: obj/bf, attr/bf, obj/ws, attr/ws) share pooled_linear_l, q_linear_l, joint_fc and the
    // classifier, so their rows are stacked and every shared FC is ONE GEMM over 4 Bn (2 Bn for pooled_linear_l) rows:
    // M = 2560 leaves the last round of tiles half empty (1260 tiles of 128x64 on 512 workgroup slots = 2.46 rounds),
    // M = 10240 does not (9.84).  The per-head names below are views into the stacked blocks.
    L.add("S/pooled", 2 * Bn * D); L.add("S/vl_pre", 2 * Bn * H); L.add("S/lft", 4 * Bn * H);
    L.add("S/vl", 4 * Bn * H); L.add("S/ll_pre", 4 * Bn * H); L.add("S/ll", 4 * Bn * H); L.add("S/jin", 4 * Bn * H);
    L.add("S/j_pre", 4 * Bn * 2 * H); L.add("S/j", 4 * Bn * 2 * H);
    L.add("S/z", 4 * Bn * A); L.add("S/dz", 4 * Bn * A); L.add("S/stats", 4 * Bn * 4);
    for (int k = 0; k < 2; ++k) {
        const std::string p = std::string(KIND[k]) + "/";
        L.add(p + "key6", Bn * 6);
        L.add(p + "v_pre", B * R * H); L.add(p + "v", B * R * H); L.add(p + "v_mean", B); L.add(p + "v_rstd", B);
        L.add(p + "qv_pre", Bn * H); L.add(p + "qv", Bn * H); L.add(p + "qv_mean", B); L.add(p + "qv_rstd", B);
        L.add(p + "att", Bn * R);
        L.alias(p + "pooled", "S/pooled", k * Bn * D, Bn * D); L.alias(p + "vl_pre", "S/vl_pre", k * Bn * H, Bn * H);
        L.add(p + "valid", Bn); L.add(p + "inv_valid", 4);
        L.alias(p + "bf_state", "S/lft", k * Bn * H, Bn * H);
        L.add(p + "wse", Bn * W); L.add(p + "ws", Bn * W);
        L.add(p + "wf_pre", Bn * H); L.alias(p + "wf", "S/lft", (2 + k) * Bn * H, Bn * H);
        L.add(p + "wf_mean", B); L.add(p + "wf_rstd", B);
        for (int t = 0; t < 2; ++t) {
            const std::string q = p + HEAD[t] + "/";
            const int64_t ln = 2 * t + k;
            L.alias(q + "vl", "S/vl", ln * Bn * H, Bn * H); L.add(q + "vl_mean", B); L.add(q + "vl_rstd", B);
            L.alias(q + "ll_pre", "S/ll_pre", ln * Bn * H, Bn * H); L.alias(q + "ll", "S/ll", ln * Bn * H, Bn * H);
            L.add(q + "ll_mean", B); L.add(q + "ll_rstd", B);
            L.alias(q + "jin", "S/jin", ln * Bn * H, Bn * H);
            L.alias(q + "j_pre", "S/j_pre", ln * Bn * 2 * H, Bn * 2 * H); L.alias(q + "j", "S/j", ln * Bn * 2 * H, Bn * 2 * H);
            L.add(q + "j_mean", B); L.add(q + "j_rstd", B);
            L.alias(q + "z", "S/z", ln * Bn * A, Bn * A); L.alias(q + "dz", "S/dz", ln * Bn * A, Bn * A);
            L.alias(q + "stats", "S/stats", ln * Bn * 4, Bn * 4);
        }
    }
    // blank-fill captions of BOTH categories as one batch of 2 Bn rows (object rows first), time-major: the GRU weights are
    // shared, so one recurrence over 5120 rows replaces two over 2560 (half the step launches, tiles that fill the chip)
    L.add("J/blanks_s", 2 * Bn * T); L.add("J/lens_s", 2 * Bn);
    L.add("J/x_tm", T * 2 * Bn * x_stride(W)); L.add("J/xp", T * 2 * Bn * 3 * H); L.add("J/hs", (T + 1) * 2 * Bn * H);
    L.add("J/gru_r", T * 2 * Bn * H); L.add("J/gru_u", T * 2 * Bn * H); L.add("J/gru_c", T * 2 * Bn * H);
    L.add("J/gru_rh", T * 2 * Bn * H);
    L.add("wx_cat", W * 3 * H); L.add("bx_cat", 3 * H); L.add("dwx_cat", x_stride(W) * 3 * H);      // packed x rows of the GRU kernels
    L.add("report", 16);
    // backward scratch, shared by the two categories
    L.add("d_j", 4 * Bn * 2 * H); L.add("d_jpre", 4 * Bn * 2 * H); L.add("d_jin", 4 * Bn * H);
    L.add("d_vl", 4 * Bn * H); L.add("d_ll", 4 * Bn * H); L.add("d_vlpre", 4 * Bn * H); L.add("d_llpre", 4 * Bn * H);
    L.add("d_lft", 4 * Bn * H); L.add("d_pooled", 2 * Bn * D);
    L.add("d_state_s", 2 * Bn * H); L.add("d_hscratch", 2 * Bn * H);
    L.add("dxp", T * 2 * Bn * 3 * H); L.add("dx", T * 2 * Bn * W);
    L.add("d_wfpre", Bn * H); L.add("d_ws", Bn * W); L.add("d_wse", Bn * W);
    L.add("d_v", B * R * H); L.add("d_vpre", B * R * H); L.add("d_qv", Bn * H); L.add("d_qvpre", Bn * H);
    L.add("part_a", B * 2 * H); L.add("part_b", B * 2 * H); L.add("part_c", B * 2 * H);
    L.add("part_dw", Bn * H); L.add("part_db", Bn);
    L.add("vec_a", max64(max64(A, 3 * H), 16)); L.add("vec_b", max64(max64(A, 3 * H), 16));
    L.add("vec_c", max64(max64(A, 3 * H), 16));
    L.add("sq", 16);
    int64_t gw = 4;
    auto g = [&](int tA, int tB, int64_t M, int64_t N, int64_t K) {
        gw = max64(gw, vqa_gemm_workspace_floats(tA, tB, (int)M, (int)N, (int)K, 0));
    };
    g(0, 0, B * R, H, 6); g(0, 0, Bn, H, 6); g(0, 0, Bn, H, D); g(0, 0, Bn, H, H); g(0, 0, Bn, 2 * H, H);
    g(0, 0, Bn, A, 2 * H);
    g(0, 0, 2 * T * Bn, 3 * H, W); g(0, 1, 2 * T * Bn, W, 3 * H); g(1, 0, x_stride(W), 3 * H, 2 * T * Bn);      // packed x-projection
    g(1, 0, H, 2 * H, 2 * T * Bn); g(1, 0, H, H, 2 * T * Bn);
    g(0, 0, 2 * Bn, H, D); g(0, 0, 4 * Bn, H, H); g(0, 0, 4 * Bn, 2 * H, H); g(0, 0, 4 * Bn, A, 2 * H);      // stacked heads
    g(1, 0, 2 * H, A, 4 * Bn); g(0, 1, 4 * Bn, 2 * H, A); g(1, 0, H, 2 * H, 4 * Bn); g(0, 1, 4 * Bn, H, 2 * H);
    g(1, 0, D, H, 2 * Bn); g(0, 1, 2 * Bn, D, H); g(1, 0, H, H, 4 * Bn); g(0, 1, 4 * Bn, H, H); g(0, 0, T * Bn, 2 * H, W); g(0, 0, T * Bn, H, W); g(0, 0, Bn, H, W);
    g(1, 0, 2 * H, A, Bn); g(0, 1, Bn, 2 * H, A); g(1, 0, H, 2 * H, Bn); g(0, 1, Bn, H, 2 * H); g(1, 0, D, H, Bn);
    g(0, 1, Bn, D, H); g(1, 0, H, H, Bn); g(0, 1, Bn, H, H); g(1, 0, W, H, Bn); g(0, 1, Bn, W, H);
    g(1, 0, W, 2 * H, T * Bn); g(1, 0, H, 2 * H, T * Bn); g(1, 0, W, H, T * Bn); g(1, 0, H, H, T * Bn);
    g(0, 1, T * Bn, W, 2 * H); g(0, 1, T * Bn, W, H); g(1, 0, 6, H, B * R); g(1, 0, 6, H, Bn);
    L.add("gemm_ws", gw);
    int64_t cw = 4;
    cw = max64(cw, vqa_colsum_workspace_floats((int)B, (int)(2 * H)));
    cw = max64(cw, vqa_colsum_workspace_floats((int)Bn, (int)A));
    cw = max64(cw, vqa_colsum_workspace_floats((int)(4 * Bn), (int)A));
    cw = max64(cw, vqa_colsum_workspace_floats((int)(T * Bn), (int)(3 * H)));
    cw = max64(cw, vqa_colsum_workspace_floats((int)Bn, (int)H));
    L.add("colsum_ws", 3 * cw);
    L.add("sumsq_ws", max64(vqa_sumsq_workspace_floats(2 * T * Bn * W), 4));
    return L;
}

bool dims_ok(const vqa_pretrain_dims_t* d) {
    return d && d->B > 0 && d->n > 0 && d->n <= 8 && d->R > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->A > 0 &&
           d->Vq > 0 && d->n_ws > 0 && d->L > 0 && d->H % 4 == 0 && d->D % 4 == 0;
}

#define TRY(x)                           \
    do {                                 \
        int rc__ = (x);                  \
        if (rc__ != VQA_OK) return rc__; \
    } while (0)

struct Ctx {
    const vqa_pretrain_dims_t& d;
    const Layout& L;
    char* ws;
    hipStream_t st;
    float* f(const std::string& name) const { return reinterpret_cast<float*>(ws + L.find(name)->off); }
    int32_t* i32(const std::string& name) const { return reinterpret_cast<int32_t*>(ws + L.find(name)->off); }
    int64_t count(const std::string& name) const { return L.find(name)->n; }
    int gemm(int tA, int tB, int64_t M, int64_t N, int64_t K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias = nullptr, const float* D = nullptr, int ldd = 0) const {
        return vqa_gemm_f32(tA, tB, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, bias, D, ldd, 0, f("gemm_ws"),
                            count("gemm_ws"), st);
    }
};

// modules.fc_layer forward: FC on the last axis, layer_norm over groups of `rows` rows, activation (0 relu, 1 tanh)
int fc_ln_fwd(const Ctx& c, const float* x, int64_t M, int64_t K, int64_t N, const vqa_pt_fc_t& p, int ln, int rows,
              int act, const std::string& pre, const std::string& y, const std::string& mean, const std::string& rstd,
              const uint8_t* keep, float keep_prob) {
    TRY(c.gemm(0, 0, M, N, K, x, (int)K, p.w, (int)N, c.f(pre), (int)N, p.b));
    return vqa_ln_act_fwd(c.f(pre), p.gamma[ln], p.beta[ln], keep, keep_prob, c.f(y), c.f(mean), c.f(rstd),
                          (int)(M / rows), rows, (int)N, act, c.st);
}

// Gradient accumulation over the call sites that share a variable: the first contribution overwrites (gradient
// buffers are not cleared between steps), later ones add.
struct Acc {
    const Ctx& c;
    std::set<const float*> touched;
    int vec(float* grad, const float* value, int64_t n) {      // value already computed somewhere else
        if (touched.insert(grad).second)
            return hipMemcpyAsync(grad, value, (size_t)n * 4, hipMemcpyDeviceToDevice, c.st) == hipSuccess ? VQA_OK : VQA_ERR_LAUNCH;
        return vqa_add_inplace(grad, value, n, c.st);
    }
    // dW (+)= x^T * dpre
    int weight(float* gw, const float* x, int ldx, const float* dpre, int ldp, int64_t K, int64_t N, int64_t M) {
        const bool first = touched.insert(gw).second;
        return c.gemm(1, 0, K, N, M, x, ldx, dpre, ldp, gw, (int)N, nullptr, first ? nullptr : gw, (int)N);
    }
    // three column sums (d_gamma, d_beta, d_bias partials [G, N]) into their gradients: first touch overwrites, later
    // ones add inside the reduction's last pass (no temporaries, no add kernels)
    int colsum3(const float* p0, const float* p1, const float* p2, int64_t G, int64_t N, float* g0, float* g1, float* g2) {
        const int mask = (touched.count(g0) ? 1 : 0) | (touched.count(g1) ? 2 : 0) | (touched.count(g2) ? 4 : 0);
        TRY(vqa_colsum3_acc(p0, p1, p2, (int)G, (int)N, (int)N, g0, g1, g2, mask, c.f("colsum_ws"), c.count("colsum_ws"),
                            c.st));
        touched.insert(g0); touched.insert(g1); touched.insert(g2);
        return VQA_OK;
    }
    int colsum(const float* X, int64_t M, int64_t N, int ldx, float* grad) {
        const bool first = touched.insert(grad).second;
        return vqa_colsum_acc(X, (int)M, (int)N, ldx, grad, first ? 0 : 1, c.f("colsum_ws"), c.count("colsum_ws"), c.st);
    }
};

// backward of the LayerNorm / activation half of fc_ln_fwd: dy -> d_pre (named); gamma, beta and bias gradients accumulated
int ln_bwd(const Ctx& c, Acc& acc, const float* dy, int64_t M, int64_t N, const vqa_pt_fc_t& p, const vqa_pt_fc_t& g, int ln,
           int rows, int act, const std::string& pre, const std::string& mean, const std::string& rstd, const uint8_t* keep,
           float keep_prob, const std::string& d_pre) {
    const int64_t G = M / rows;
    TRY(vqa_ln_act_bwd(dy, c.f(pre), c.f(mean), c.f(rstd), p.gamma[ln], p.beta[ln], keep, keep_prob, c.f(d_pre),
                       c.f("part_a"), c.f("part_b"), c.f("part_c"), (int)G, rows, (int)N, act, c.st));
    return acc.colsum3(c.f("part_a"), c.f("part_b"), c.f("part_c"), G, N, g.gamma[ln], g.beta[ln], g.b);
}

// the same on pointers (slices of the stacked head blocks)
int ln_bwd_p(const Ctx& c, Acc& acc, const float* dy, int64_t M, int64_t N, const vqa_pt_fc_t& p, const vqa_pt_fc_t& g, int ln,
             int rows, int act, const float* pre, const float* mean, const float* rstd, const uint8_t* keep, float keep_prob,
             float* d_pre) {
    const int64_t G = M / rows;
    TRY(vqa_ln_act_bwd(dy, pre, mean, rstd, p.gamma[ln], p.beta[ln], keep, keep_prob, d_pre, c.f("part_a"), c.f("part_b"),
                       c.f("part_c"), (int)G, rows, (int)N, act, c.st));
    return acc.colsum3(c.f("part_a"), c.f("part_b"), c.f("part_c"), G, N, g.gamma[ln], g.beta[ln], g.b);
}

// backward of the FC half: dW (+)= x^T d_pre, optional dx = d_pre * W^T
int fc_bwd(const Ctx& c, Acc& acc, const std::string& d_pre, const float* x, int64_t M, int64_t K, int64_t N,
           const vqa_pt_fc_t& p, const vqa_pt_fc_t& g, float* dx) {
    TRY(acc.weight(g.w, x, (int)K, c.f(d_pre), (int)N, K, N, M));
    if (dx != nullptr) TRY(c.gemm(0, 1, M, K, N, c.f(d_pre), (int)N, p.w, (int)N, dx, (int)K));
    return VQA_OK;
}

// backward of fc_ln_fwd
int fc_ln_bwd(const Ctx& c, Acc& acc, const float* dy, const float* x, int64_t M, int64_t K, int64_t N, const vqa_pt_fc_t& p,
              const vqa_pt_fc_t& g, int ln, int rows, int act, const std::string& pre, const std::string& mean,
              const std::string& rstd, const uint8_t* keep, float keep_prob, const std::string& d_pre, float* dx) {
    TRY(ln_bwd(c, acc, dy, M, N, p, g, ln, rows, act, pre, mean, rstd, keep, keep_prob, d_pre));
    return fc_bwd(c, acc, d_pre, x, M, K, N, p, g, dx);
}

int gather_rows(const void* in, const int32_t* index, void* out, int64_t rows, int64_t cols, hipStream_t st) {
    if (rows * cols == 0) return VQA_OK;
    const int grid = (int)std::min<int64_t>((rows * cols + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, st, static_cast<const uint32_t*>(in), index,
                       static_cast<uint32_t*>(out), (int)rows, (int)cols);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

int gather_rows2(const void* in0, const void* in1, const int32_t* index, void* out, int64_t rows, int64_t cols,
                 int64_t rows0, hipStream_t st) {
    if (rows * cols == 0) return VQA_OK;
    const int grid = (int)std::min<int64_t>((rows * cols + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_rows2_kernel, dim3(grid), dim3(256), 0, st, static_cast<const uint32_t*>(in0),
                       static_cast<const uint32_t*>(in1), index, static_cast<uint32_t*>(out), (int)rows, (int)cols, (int)rows0);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

const char* const REPORT_KEYS[13] = {
    "obj_blank_fill_loss", "obj_blank_fill_acc", "obj_blank_fill_top_5_acc", "obj_wordset_loss", "obj_wordset_acc",
    "obj_wordset_top_5_acc", "attr_blank_fill_loss", "attr_blank_fill_acc", "attr_blank_fill_top_5_acc",
    "attr_wordset_loss", "attr_wordset_acc", "attr_wordset_top_5_acc", "total_loss"};

}  // namespace

extern "C" const char* vqa_pretrain_report_key(int i) { return (i >= 0 && i < 13) ? REPORT_KEYS[i] : nullptr; }

extern "C" int64_t vqa_pretrain_workspace_bytes(const vqa_pretrain_dims_t* dims) {
    if (!dims_ok(dims)) return VQA_ERR_ARG;
    return make_layout(*dims).total;
}

extern "C" int vqa_pretrain_tensor(const vqa_pretrain_dims_t* dims, const char* name, int64_t* offset_bytes,
                                   int64_t* n_elems) {
    if (!dims_ok(dims) || name == nullptr) return VQA_ERR_ARG;
    const Layout L = make_layout(*dims);
    const Entry* e = L.find(name);
    if (e == nullptr) return VQA_ERR_ARG;
    if (offset_bytes) *offset_bytes = e->off;
    if (n_elems) *n_elems = e->n;
    return VQA_OK;
}

extern "C" int vqa_pretrain_forward(const vqa_pretrain_dims_t* dims, const vqa_pretrain_params_t* P,
                                    const vqa_pretrain_batch_t* bt, void* workspace, int64_t workspace_bytes,
                                    int want_dz, void* stream) {
    VQA_REQUIRE(dims_ok(dims) && P && bt && workspace, VQA_ERR_ARG);
    const Layout L = make_layout(*dims);
    VQA_REQUIRE(workspace_bytes >= L.total, VQA_ERR_WORKSPACE);
    VQA_REQUIRE(vqa_aligned16(workspace), VQA_ERR_ALIGN);
    const Ctx c{*dims, L, static_cast<char*>(workspace), static_cast<hipStream_t>(stream)};
    const int64_t B = dims->B, n = dims->n, R = dims->R, D = dims->D, H = dims->H, W = dims->W, A = dims->A, T = dims->L;
    const int64_t Bn = B * n;
    const bool ln_shared = (dims->flags & VQA_FLAG_SHARED_LN) != 0;
    auto li = [ln_shared](int site) { return ln_shared ? 0 : site; };
    VQA_REQUIRE(bt->image_ft && bt->spatial_ft && bt->num_boxes, VQA_ERR_ARG);
    VQA_REQUIRE((bt->perm == nullptr) == (bt->inv == nullptr) && (bt->perm == nullptr) == (bt->live_rows == nullptr), VQA_ERR_ARG);
    ReportArgs ra{};
    ra.rows = (int)Bn;
    ProbeScope ps_all("pretrain.forward", c.st);
    // the x rows of the two GRU kernels side by side (vqa_gru_pack_wx): one projection GEMM per category
    TRY(vqa_gru_pack_wx(P->gru_wg, P->gru_wc, P->gru_bg, P->gru_bc, c.f("wx_cat"), c.f("bx_cat"), (int)W, (int)H, c.st));
    for (int k = 0; k < 2; ++k) {
        const vqa_pretrain_kind_t& kb = bt->kind[k];
        VQA_REQUIRE(kb.normal_boxes && kb.fills && kb.blanks && kb.blanks_len && kb.wordsets && kb.num, VQA_ERR_ARG);
        const std::string p = std::string(KIND[k]) + "/";
        // ---- build_*_V_ft: spatial attention over the regions
        ProbeScope ps_sp("pt.spatial_wordset.fwd", c.st);
        hipLaunchKernelGGL(box6_kernel, dim3((unsigned)((Bn + 255) / 256)), dim3(256), 0, c.st, kb.normal_boxes,
                           c.f(p + "key6"), (int)Bn);
        VQA_CHECK_LAUNCH();
        TRY(fc_ln_fwd(c, bt->spatial_ft, B * R, 6, H, P->spat_v_linear_v, li(k), (int)R, 0, p + "v_pre", p + "v", p + "v_mean",
                      p + "v_rstd", nullptr, 1.f));
        TRY(fc_ln_fwd(c, c.f(p + "key6"), Bn, 6, H, P->spat_q_linear_v, li(k), (int)n, 0, p + "qv_pre", p + "qv", p + "qv_mean",
                      p + "qv_rstd", nullptr, 1.f));
        TRY(vqa_attn_pool_fwd_rep(c.f(p + "v"), c.f(p + "qv"), bt->image_ft, bt->num_boxes, P->spat_att_score.w,
                                  P->spat_att_score.b, kb.keep_att, dims->keep_att, c.f(p + "att"), c.f(p + "pooled"),
                                  (int)B, (int)n, (int)R, (int)H, (int)D, c.st));
        hipLaunchKernelGGL(valid_kernel, dim3(1), dim3(256), 0, c.st, kb.num, c.f(p + "valid"), c.f(p + "inv_valid"),
                           (int)B, (int)n, dims->global_valid[k]);
        VQA_CHECK_LAUNCH();

        // ---- build_*_wordset
        TRY(vqa_embed_fwd(P->wordset_map, kb.wordsets, c.f(p + "wse"), (int)Bn, 1, (int)W, dims->n_ws, c.st));
        TRY(vqa_tanh_fwd(c.f(p + "wse"), c.f(p + "ws"), Bn * W, c.st));
        TRY(fc_ln_fwd(c, c.f(p + "ws"), Bn, W, H, P->wordset_ft, li(k), (int)n, 1, p + "wf_pre", p + "wf", p + "wf_mean",
                      p + "wf_rstd", nullptr, 1.f));
    }
    // ---- build_*_blank_fill of BOTH categories as one batch (shared L_GloVe / GRU weights): captions in length order
    // when the host sorted them (live prefix recurrence); the final states go back to caption order as rows
    // [0, Bn) = obj/bf_state and [Bn, 2 Bn) = attr/bf_state of the stacked "S/lft" block
    {
        const int64_t B2 = 2 * Bn;
        ProbeScope ps_g("pt.caption_gru.fwd", c.st);
        TRY(gather_rows2(bt->kind[0].blanks, bt->kind[1].blanks, bt->perm, c.i32("J/blanks_s"), B2, T, Bn, c.st));
        TRY(gather_rows2(bt->kind[0].blanks_len, bt->kind[1].blanks_len, bt->perm, c.i32("J/lens_s"), B2, 1, Bn, c.st));
        TRY(vqa_embed_fwd_ld(P->l_glove, c.i32("J/blanks_s"), c.f("J/x_tm"), (int)B2, (int)T, (int)W, dims->Vq,
                             (int)x_stride(W), c.st));
        float* xp = c.f("J/xp");
        TRY(c.gemm(0, 0, T * B2, 3 * H, W, c.f("J/x_tm"), (int)x_stride(W), c.f("wx_cat"), (int)(3 * H), xp, (int)(3 * H),
                   c.f("bx_cat")));
        float* hs = c.f("J/hs");
        if (hipMemsetAsync(hs, 0, (size_t)B2 * H * sizeof(float), c.st) != hipSuccess) return VQA_ERR_LAUNCH;
        const float* Wg_h = P->gru_wg + W * 2 * H;
        const float* Wc_h = P->gru_wc + W * H;
        if (bt->live_rows != nullptr)
            TRY(vqa_gru_seq_fwd_live(xp, Wg_h, Wc_h, c.i32("J/lens_s"), bt->live_rows, hs, c.f("J/gru_r"), c.f("J/gru_u"),
                                     c.f("J/gru_c"), c.f("J/gru_rh"), (int)T, (int)B2, (int)H, c.st));
        else
            TRY(vqa_gru_seq_fwd(xp, Wg_h, Wc_h, c.i32("J/lens_s"), hs, c.f("J/gru_r"), c.f("J/gru_u"), c.f("J/gru_c"),
                                c.f("J/gru_rh"), (int)T, (int)B2, (int)H, c.st));
        TRY(gather_rows(hs + T * B2 * H, bt->inv, c.f("S/lft"), B2, H, c.st));      // back to caption order
    }
    // ---- the four heads, stacked (ln = 2 t + k; see make_layout): every shared FC is one GEMM, LayerNorm (its own
    // gamma / beta per head) and the loss run per slice
    {
        const int64_t SH = Bn * H, SJ = Bn * 2 * H, SA = Bn * A;
        auto hname = [&](int ln) { return std::string(KIND[ln & 1]) + "/" + HEAD[ln >> 1] + "/"; };
        ProbeScope ps_h("pt.heads.fwd", c.st);
        // pooled_linear_l: both heads of a category apply the SAME FC to the same pooled features, so the product is
        // computed once per category (2 Bn rows)
        TRY(c.gemm(0, 0, 2 * Bn, H, D, c.f("S/pooled"), (int)D, P->pooled_linear_l.w, (int)H, c.f("S/vl_pre"), (int)H,
                   P->pooled_linear_l.b));
        TRY(c.gemm(0, 0, 4 * Bn, H, H, c.f("S/lft"), (int)H, P->q_linear_l.w, (int)H, c.f("S/ll_pre"), (int)H, P->q_linear_l.b));
        for (int ln = 0; ln < 4; ++ln) {
            const std::string q = hname(ln);
            TRY(vqa_ln_act_fwd(c.f("S/vl_pre") + (ln & 1) * SH, P->pooled_linear_l.gamma[li(ln)], P->pooled_linear_l.beta[li(ln)],
                               nullptr, 1.f, c.f("S/vl") + ln * SH, c.f(q + "vl_mean"), c.f(q + "vl_rstd"), (int)B, (int)n,
                               (int)H, 0, c.st));
            TRY(vqa_ln_act_fwd(c.f("S/ll_pre") + ln * SH, P->q_linear_l.gamma[li(ln)], P->q_linear_l.beta[li(ln)], nullptr, 1.f,
                               c.f("S/ll") + ln * SH, c.f(q + "ll_mean"), c.f(q + "ll_rstd"), (int)B, (int)n, (int)H, 0, c.st));
        }
        TRY(vqa_mul(c.f("S/vl"), c.f("S/ll"), c.f("S/jin"), 4 * SH, c.st));
        TRY(c.gemm(0, 0, 4 * Bn, 2 * H, H, c.f("S/jin"), (int)H, P->joint_fc.w, (int)(2 * H), c.f("S/j_pre"), (int)(2 * H),
                   P->joint_fc.b));
        for (int ln = 0; ln < 4; ++ln) {
            const std::string q = hname(ln);
            const vqa_pretrain_kind_t& kb = bt->kind[ln & 1];
            const uint8_t* jmask = (ln >> 1) == 0 ? kb.keep_bf_joint : kb.keep_ws_joint;
            TRY(vqa_ln_act_fwd(c.f("S/j_pre") + ln * SJ, P->joint_fc.gamma[li(ln)], P->joint_fc.beta[li(ln)], jmask, dims->keep_joint,
                               c.f("S/j") + ln * SJ, c.f(q + "j_mean"), c.f(q + "j_rstd"), (int)B, (int)n, (int)(2 * H), 0,
                               c.st));
        }
        TRY(c.gemm(0, 0, 4 * Bn, A, 2 * H, c.f("S/j"), (int)(2 * H), P->classifier.w, (int)A, c.f("S/z"), (int)A,
                   P->classifier.b));
        for (int ln = 0; ln < 4; ++ln) {
            const int k = ln & 1, t = ln >> 1;
            const std::string p = std::string(KIND[k]) + "/";
            TRY(vqa_softmax_ce_fwd(c.f("S/z") + ln * SA, bt->kind[k].fills, c.f(p + "valid"), 5, c.f(p + "inv_valid"),
                                   c.f("S/stats") + ln * Bn * 4, want_dz ? c.f("S/dz") + ln * SA : nullptr, (int)Bn, (int)A,
                                   c.st));
            ra.stats[2 * k + t] = c.f("S/stats") + ln * Bn * 4;
            ra.inv[2 * k + t] = c.f(p + "inv_valid");
        }
    }
    hipLaunchKernelGGL(pretrain_report_kernel, dim3(1), dim3(256), 0, c.st, ra, c.f("report"));
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

// Backward in dependency-ordered phases, so that a data-parallel caller can start reducing the gradients a phase
// completed while the next phase runs (the cfg-5 counterpart of vqa_fusion_backward_phases):
//   1  the four stacked heads: classifier, joint_fc, pooled_linear_l, q_linear_l (weights, biases, LayerNorms)
//   2  back-propagation through time of the joint caption batch + the GRU kernels' and biases' gradients
//   4  dx of the packed x-projection -> L_GloVe scatter-add (starts the running slice sum of squares)
//   8  per category: wordset_ft / wordset_map, spatial attention, spat_v_linear_v / spat_q_linear_v (finishes slice_sq)
// Each phase reads what the lower-numbered ones left in the workspace; a step runs them in this order.
extern "C" int vqa_pretrain_backward_phases(const vqa_pretrain_dims_t* dims, const vqa_pretrain_params_t* P,
                                            const vqa_pretrain_params_t* G, const vqa_pretrain_batch_t* bt,
                                            void* workspace, int64_t workspace_bytes, float* slice_sq, int phases,
                                            void* stream) {
    VQA_REQUIRE(dims_ok(dims) && P && G && bt && workspace, VQA_ERR_ARG);
    VQA_REQUIRE(phases > 0 && phases < 16, VQA_ERR_ARG);
    const Layout L = make_layout(*dims);
    VQA_REQUIRE(workspace_bytes >= L.total, VQA_ERR_WORKSPACE);
    const Ctx c{*dims, L, static_cast<char*>(workspace), static_cast<hipStream_t>(stream)};
    const int64_t B = dims->B, n = dims->n, R = dims->R, D = dims->D, H = dims->H, W = dims->W, A = dims->A, T = dims->L;
    const int64_t Bn = B * n;
    const bool ln_shared = (dims->flags & VQA_FLAG_SHARED_LN) != 0;
    auto li = [ln_shared](int site) { return ln_shared ? 0 : site; };
    ProbeScope ps_all("pretrain.backward", c.st);
    Acc acc{c, {}};      // no gradient is touched by two different phases, so the first-touch record may be per call
    // the two embedding tables are scatter-added: cleared in their phase; every other gradient is overwritten on first touch
    const float* sq_prev = nullptr;
    auto add_slice_sq = [&](const float* g, int64_t cnt) -> int {     // running sum of the un-aggregated slice norms
        TRY(vqa_sumsq(g, cnt, sq_prev, c.f("sq"), c.f("sumsq_ws"), c.count("sumsq_ws"), c.st));
        sq_prev = c.f("sq");
        return VQA_OK;
    };
    // ---- the four heads, stacked (ln = 2 t + k): every shared FC's dW and dx is one GEMM over the 4 Bn rows
    if (phases & 1) {
        ProbeScope ps_h("pt.heads.bwd", c.st);
        const int64_t SH = Bn * H, SJ = Bn * 2 * H;
        auto hname = [&](int ln) { return std::string(KIND[ln & 1]) + "/" + HEAD[ln >> 1] + "/"; };
        TRY(acc.weight(G->classifier.w, c.f("S/j"), (int)(2 * H), c.f("S/dz"), (int)A, 2 * H, A, 4 * Bn));
        TRY(acc.colsum(c.f("S/dz"), 4 * Bn, A, (int)A, G->classifier.b));
        TRY(c.gemm(0, 1, 4 * Bn, 2 * H, A, c.f("S/dz"), (int)A, P->classifier.w, (int)A, c.f("d_j"), (int)(2 * H)));
        for (int ln = 0; ln < 4; ++ln) {
            const std::string q = hname(ln);
            const vqa_pretrain_kind_t& kb = bt->kind[ln & 1];
            const uint8_t* jmask = (ln >> 1) == 0 ? kb.keep_bf_joint : kb.keep_ws_joint;
            TRY(ln_bwd_p(c, acc, c.f("d_j") + ln * SJ, Bn, 2 * H, P->joint_fc, G->joint_fc, li(ln), (int)n, 0,
                         c.f("S/j_pre") + ln * SJ, c.f(q + "j_mean"), c.f(q + "j_rstd"), jmask, dims->keep_joint,
                         c.f("d_jpre") + ln * SJ));
        }
        TRY(fc_bwd(c, acc, "d_jpre", c.f("S/jin"), 4 * Bn, H, 2 * H, P->joint_fc, G->joint_fc, c.f("d_jin")));
        TRY(vqa_mul_bwd(c.f("d_jin"), c.f("S/vl"), c.f("S/ll"), c.f("d_vl"), c.f("d_ll"), 4 * SH, c.st));
        for (int ln = 0; ln < 4; ++ln) {
            const std::string q = hname(ln);
            TRY(ln_bwd_p(c, acc, c.f("d_vl") + ln * SH, Bn, H, P->pooled_linear_l, G->pooled_linear_l, li(ln), (int)n, 0,
                         c.f("S/vl_pre") + (ln & 1) * SH, c.f(q + "vl_mean"), c.f(q + "vl_rstd"), nullptr, 1.f,
                         c.f("d_vlpre") + ln * SH));
            TRY(ln_bwd_p(c, acc, c.f("d_ll") + ln * SH, Bn, H, P->q_linear_l, G->q_linear_l, li(ln), (int)n, 0,
                         c.f("S/ll_pre") + ln * SH, c.f(q + "ll_mean"), c.f(q + "ll_rstd"), nullptr, 1.f,
                         c.f("d_llpre") + ln * SH));
        }
        // pooled_linear_l's product is shared by the two heads of a category: their d_pre (slices k and 2 + k) meet
        // before ONE dW and ONE dx GEMM over the 2 Bn pooled rows
        TRY(vqa_add_inplace(c.f("d_vlpre"), c.f("d_vlpre") + 2 * SH, 2 * SH, c.st));
        TRY(acc.weight(G->pooled_linear_l.w, c.f("S/pooled"), (int)D, c.f("d_vlpre"), (int)H, D, H, 2 * Bn));
        TRY(c.gemm(0, 1, 2 * Bn, D, H, c.f("d_vlpre"), (int)H, P->pooled_linear_l.w, (int)H, c.f("d_pooled"), (int)D));
        TRY(fc_bwd(c, acc, "d_llpre", c.f("S/lft"), 4 * Bn, H, H, P->q_linear_l, G->q_linear_l, c.f("d_lft")));
    }
    // ---- blank fill -> GRU -> L_GloVe, both categories as the one batch of the forward
    const int64_t B2 = 2 * Bn;
    float* dxp = c.f("dxp");
    const int ld3 = (int)(3 * H);
    if (phases & 2) {
        ProbeScope ps_g("pt.caption_gru.bwd", c.st);
        TRY(gather_rows(c.f("d_lft"), bt->perm, c.f("d_state_s"), B2, H, c.st));   // rows [0, 2 Bn) of d_lft, into the sorted order
        const float* Wg_h = P->gru_wg + W * 2 * H;
        const float* Wc_h = P->gru_wc + W * H;
        const float* hs = c.f("J/hs");
        if (bt->live_rows != nullptr)
            TRY(vqa_gru_seq_bwd_live(c.f("d_state_s"), Wg_h, Wc_h, c.i32("J/lens_s"), bt->live_rows, hs, c.f("J/gru_r"),
                                     c.f("J/gru_u"), c.f("J/gru_c"), dxp, c.f("d_hscratch"), (int)T, (int)B2, (int)H, c.st));
        else
            TRY(vqa_gru_seq_bwd(c.f("d_state_s"), Wg_h, Wc_h, c.i32("J/lens_s"), hs, c.f("J/gru_r"), c.f("J/gru_u"),
                                c.f("J/gru_c"), dxp, c.f("d_hscratch"), (int)T, (int)B2, (int)H, c.st));
        // x rows of both kernels' gradients as one GEMM into the packed [Wp, 3H] block; x_tm carries the constant 1 in
        // column W, so row W of the block is the two bias gradients and dxp is not read again for them.  The h rows
        // as before
        const int64_t Wp = x_stride(W);
        TRY(acc.weight(c.f("dwx_cat"), c.f("J/x_tm"), (int)Wp, dxp, ld3, Wp, 3 * H, T * B2));
        TRY(acc.weight(G->gru_wg + W * 2 * H, hs, (int)H, dxp, ld3, H, 2 * H, T * B2));
        TRY(acc.weight(G->gru_wc + W * H, c.f("J/gru_rh"), (int)H, dxp + 2 * H, ld3, H, H, T * B2));
        TRY(vqa_gru_unpack_dwx_bias(c.f("dwx_cat"), G->gru_wg, G->gru_wc, G->gru_bg, G->gru_bc, (int)W, (int)H, c.st));
    }
    if (phases & 4) {
        ProbeScope ps_e("pt.caption_embed.bwd", c.st);
        if (hipMemsetAsync(G->l_glove, 0, (size_t)dims->Vq * W * 4, c.st) != hipSuccess) return VQA_ERR_LAUNCH;
        float* dx = c.f("dx");
        // packed again here (one small kernel): no hidden dependence on the forward's copy of the weights
        TRY(vqa_gru_pack_wx(P->gru_wg, P->gru_wc, P->gru_bg, P->gru_bc, c.f("wx_cat"), c.f("bx_cat"), (int)W, (int)H, c.st));
        TRY(c.gemm(0, 1, T * B2, W, 3 * H, dxp, ld3, c.f("wx_cat"), (int)(3 * H), dx, (int)W));
        TRY(vqa_embed_bwd_len_det(dx, c.i32("J/blanks_s"), c.i32("J/lens_s"), G->l_glove, (int)B2, (int)T, (int)W,
                                  dims->Vq, (dims->flags & VQA_FLAG_DETERMINISTIC) ? 1 : 0, c.st));
        TRY(add_slice_sq(dx, T * B2 * W));
    }
    if (phases & 8) {
        if (hipMemsetAsync(G->wordset_map, 0, (size_t)dims->n_ws * W * 4, c.st) != hipSuccess) return VQA_ERR_LAUNCH;
        if (!(phases & 4)) sq_prev = c.f("sq");      // phase 4 of this step left the captions' slice sum of squares there
    }
    for (int k = 0; k < 2 && (phases & 8); ++k) {
        ProbeScope ps_sp("pt.spatial_wordset.bwd", c.st);
        const vqa_pretrain_kind_t& kb = bt->kind[k];
        const std::string p = std::string(KIND[k]) + "/";
        const float* d_pooled = c.f("d_pooled") + k * Bn * D;
        // ---- word set -> wordset_ft -> tanh -> wordset_map
        TRY(fc_ln_bwd(c, acc, c.f("d_lft") + (2 + k) * Bn * H, c.f(p + "ws"), Bn, W, H, P->wordset_ft, G->wordset_ft, li(k), (int)n, 1, p + "wf_pre",
                      p + "wf_mean", p + "wf_rstd", nullptr, 1.f, "d_wfpre", c.f("d_ws")));
        TRY(vqa_tanh_bwd(c.f("d_ws"), c.f(p + "ws"), c.f("d_wse"), Bn * W, c.st));
        TRY(vqa_embed_bwd_len_det(c.f("d_wse"), kb.wordsets, nullptr, G->wordset_map, (int)Bn, 1, (int)W, dims->n_ws,
                                  (dims->flags & VQA_FLAG_DETERMINISTIC) ? 1 : 0, c.st));
        TRY(add_slice_sq(c.f("d_wse"), Bn * W));
        // ---- spatial attention
        TRY(vqa_attn_pool_bwd_rep(d_pooled, c.f(p + "v"), c.f(p + "qv"), bt->image_ft, c.f(p + "att"),
                                  P->spat_att_score.w, kb.keep_att, dims->keep_att, c.f("d_v"), c.f("d_qv"),
                                  c.f("part_dw"), c.f("part_db"), (int)B, (int)n, (int)R, (int)H, (int)D, c.st));
        TRY(acc.colsum(c.f("part_dw"), Bn, H, (int)H, G->spat_att_score.w));
        TRY(acc.colsum(c.f("part_db"), Bn, 1, 1, G->spat_att_score.b));
        TRY(fc_ln_bwd(c, acc, c.f("d_v"), bt->spatial_ft, B * R, 6, H, P->spat_v_linear_v, G->spat_v_linear_v, li(k), (int)R, 0,
                      p + "v_pre", p + "v_mean", p + "v_rstd", nullptr, 1.f, "d_vpre", nullptr));
        TRY(fc_ln_bwd(c, acc, c.f("d_qv"), c.f(p + "key6"), Bn, 6, H, P->spat_q_linear_v, G->spat_q_linear_v, li(k), (int)n, 0,
                      p + "qv_pre", p + "qv_mean", p + "qv_rstd", nullptr, 1.f, "d_qvpre", nullptr));
    }
    if ((phases & 8) && slice_sq != nullptr && sq_prev != nullptr)
        if (hipMemcpyAsync(slice_sq, sq_prev, sizeof(float), hipMemcpyDeviceToDevice, c.st) != hipSuccess)
            return VQA_ERR_LAUNCH;
    return VQA_OK;
}

extern "C" int vqa_pretrain_backward(const vqa_pretrain_dims_t* dims, const vqa_pretrain_params_t* P,
                                     const vqa_pretrain_params_t* G, const vqa_pretrain_batch_t* bt, void* workspace,
                                     int64_t workspace_bytes, float* slice_sq, void* stream) {
    return vqa_pretrain_backward_phases(dims, P, G, bt, workspace, workspace_bytes, slice_sq, 15, stream);
}
