// Host-side composition of the fusion model's forward and backward passes:
// one call enqueues every kernel of a step on the caller's stream.
//
// Forward  = vqa/model_vlmap_answer.py:102-288 (model_standard.py:193-374):
//   feature gather -> v_linear_v -> embedding -> GRU -> q_linear_v -> Hadamard
//   attention + pooling -> pooled_linear_l / q_linear_l -> joint_fc (+dropout) ->
//   answer head -> sigmoid-CE loss, argmax, report.
// Backward = the autodiff that tf.contrib.layers.optimize_loss builds
//   (vqa/trainer.py:106-114), hand-derived; frozen variables of model_vlmap_answer
//   (filter_train_vars, model_vlmap_answer.py:81-89) still propagate dX.
//
// Workspace layout is computed from the dims alone, so the host can view named
// intermediates (mid_result / output of the reference Model) without copies.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "vqa_common.h"

namespace {

struct Entry { const char* name; int64_t off; int64_t n; int esz; };

struct Layout {
    std::vector<Entry> e;
    int64_t total = 0;
    int64_t add(const char* name, int64_t n, int esz = 4) {
        const int64_t off = total;
        e.push_back({name, off, n, esz});
        total += ((n * esz + 255) / 256) * 256;  // 256-B aligned carve
        return off;
    }
    const Entry* find(const char* name) const {
        for (const auto& x : e)
            if (strcmp(x.name, name) == 0) return &x;
        return nullptr;
    }
};

int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

void legacy_vqa_layout(Layout& L, const vqa_dims_t& d, int64_t& gw);      // csrc/legacy_vqa.inc (model_type 13)

Layout make_layout(const vqa_dims_t& d) {
    Layout L;
    const int64_t B = d.B, R = d.R, D = d.D, H = d.H, T = d.T, W = d.W, A = d.A;
    L.add("V_ft", B * R * D);
    L.add("num_V_ft", B);
    L.add("pre_v", B * R * H);
    L.add("v_linear_v", B * R * H);
    L.add("mean_v", B); L.add("rstd_v", B);
    const int64_t Wp = ((W + 1 + 3) / 4) * 4;      // row stride of x_tm: W inputs + the constant 1 (+ zero padding to 16 B)
    L.add("x_tm", T * B * Wp);
    L.add("xp", T * B * 3 * H);
    L.add("wx_cat", W * 3 * H); L.add("bx_cat", 3 * H); L.add("dwx_cat", Wp * 3 * H);
    L.add("hs", (T + 1) * B * H);
    L.add("gru_r", T * B * H); L.add("gru_u", T * B * H); L.add("gru_c", T * B * H); L.add("gru_rh", T * B * H);
    // hand-off buffers and flags of the weight-stationary recurrence (csrc/gru_ws.hip), where its shape limits hold
    if (H == 1024 && B <= 512) L.add("gru_ws", vqa_gru_ws_workspace_bytes((int)T) / 4);
    L.add("pre_qv", B * H); L.add("q_linear_v", B * H); L.add("mean_qv", B); L.add("rstd_qv", B);
    L.add("att_score", B * R);
    const int64_t Dp = d.model_type == VQA_MODEL_ADAPT ? H : D;      // width of what the attention pools (pooled_dim)
    L.add("pooled_V_ft", B * Dp);
    L.add("pre_pl", B * H); L.add("pooled_linear_l", B * H); L.add("mean_pl", B); L.add("rstd_pl", B);
    L.add("pre_ll", B * H); L.add("l_linear_l", B * H); L.add("mean_ll", B); L.add("rstd_ll", B);
    L.add("joint_in", B * H);
    L.add("pre_j", B * 2 * H); L.add("joint", B * 2 * H); L.add("mean_j", B); L.add("rstd_j", B);
    L.add("joint2", B * W);      // standard_word2vec: classifier output in the 300-d word space
    L.add("logit", B * A);
    if (d.model_type == 5) {     // vlmap_answer_noc: the l_joint branch ("joint" is v_joint)
        L.add("pre_jl", B * 2 * H); L.add("l_joint", B * 2 * H); L.add("mean_jl", B); L.add("rstd_jl", B);
        L.add("d_ljoint", B * 2 * H); L.add("d_pre_jl", B * 2 * H);
    }
    if (d.model_type == 4 || d.model_type == 6) {     // vlmap_answer_vqa_all2 / _vqa_all: the two heads' logits ("logit" = their sum), the tuned head's dz
        L.add("logit_fixed", B * A); L.add("logit_tuned", B * A); L.add("dlogit_tuned", B * A);
    }
    if (d.model_type == 6) { L.add("logit_raw", B * A); L.add("rowmin", B); }      // before the row-minimum substitution
    // the five older ablations (vqa_hot.h: VQA_MODEL_*)
    if (d.model_type == VQA_MODEL_ANSWER2) {
        L.add("pre_ft2", B * H); L.add("q_L_ft2", B * H); L.add("mean_ft2", B); L.add("rstd_ft2", B);
        L.add("d_ft2", B * H); L.add("d_pre_ft2", B * H);
    }
    if (d.model_type == VQA_MODEL_NO_NOISE || d.model_type == VQA_MODEL_FULL) { L.add("q_L_mean", B * H); L.add("d_qm", B * H); }
    if (d.model_type == VQA_MODEL_FULL) {
        L.add("q_L_log_sigma_sq", B * H); L.add("q_L_mean_noise", B * H); L.add("d_qs", B * H); L.add("d_lin", B * H);
    }
    if (d.model_type == VQA_MODEL_FULL || d.model_type == VQA_MODEL_ENT) L.add("extra_row", B);
    if (d.model_type == VQA_MODEL_ADAPT) {
        L.add("pre_va", B * R * H); L.add("v_adapt", B * R * H); L.add("mean_va", B); L.add("rstd_va", B);
        L.add("d_va", B * R * H); L.add("d_pre_va", B * R * H);
    }
    if (d.model_type == VQA_MODEL_BI) {
        // bi-directional question encoder + question self-attention (the forward cell reuses x_tm / xp / hs / gru_* / dxp /
        // wx_cat / bx_cat / dwx_cat above with h = H / 2 columns; the backward cell gets its own)
        const int64_t h = H / 2, Wq = ((W + 1 + 3) / 4) * 4;
        L.add("q_rev", B * T);
        L.add("x_tm_bw", T * B * Wq); L.add("xp_bw", T * B * 3 * h);
        L.add("wx_cat_bw", W * 3 * h); L.add("bx_cat_bw", 3 * h); L.add("dwx_cat_bw", Wq * 3 * h);
        L.add("hs_bw", (T + 1) * B * h);
        L.add("gru_r_bw", T * B * h); L.add("gru_u_bw", T * B * h); L.add("gru_c_bw", T * B * h); L.add("gru_rh_bw", T * B * h);
        L.add("q_L_map", B * T * H); L.add("q_L_ft", B * H);
        L.add("pre_key", B * T * H); L.add("q_att_key", B * T * H); L.add("mean_key", B); L.add("rstd_key", B);
        L.add("pre_query", B * H); L.add("q_att_query", B * H); L.add("mean_query", B); L.add("rstd_query", B);
        L.add("w_att_score", B * T);
        L.add("e2", B * T * W); L.add("pre_vw", B * T * H); L.add("q_v_ft", B * T * H); L.add("mean_vw", B); L.add("rstd_vw", B);
        L.add("pooled_q_v", B * H);
        L.add("d_pooled_qv", B * H); L.add("d_key", B * T * H); L.add("d_pre_key", B * T * H);
        L.add("d_query", B * H); L.add("d_pre_query", B * H);
        L.add("d_qvft", B * T * H); L.add("d_pre_vw", B * T * H); L.add("d_e2", B * T * W); L.add("sq_e2", 4);
        L.add("d_qmap", B * T * H);
        L.add("dout_fw", T * B * h); L.add("dout_bw", T * B * h);
        L.add("dhT_fw", B * h); L.add("dhT_bw", B * h); L.add("dhS_fw", B * h); L.add("dhS_bw", B * h);
        L.add("dxp_bw", T * B * 3 * h); L.add("dx_fw", T * B * W); L.add("dx_bw", T * B * W);
        L.add("part_wdw", B * H); L.add("part_wdb", B);
    }
    if (d.model_type == VQA_MODEL_ENT) {
        const int64_t M = d.num_marginal, C = d.ent_cols;
        L.add("tile_in", B * M * H); L.add("pre_tj", B * M * 2 * H); L.add("tile_joint", B * M * 2 * H);
        L.add("mean_tj", B); L.add("rstd_tj", B);
        L.add("tile_z", B * M * C); L.add("marginal_prob", B * C);
        L.add("d_tile_joint", B * M * 2 * H); L.add("d_pre_tj", B * M * 2 * H); L.add("d_tile_in", B * M * H);
    }
    L.add("stats", B * VQA_STAT_COUNT);
    L.add("pred", B);
    L.add("report", 16);
    L.add("dlogit", B * A);
    // backward scratch
    L.add("d_joint2", B * W);
    L.add("d_joint", B * 2 * H); L.add("d_pre_j", B * 2 * H);
    L.add("d_joint_in", B * H);
    L.add("d_pl", B * H); L.add("d_ll", B * H); L.add("d_pre_pl", B * H); L.add("d_pre_ll", B * H);
    L.add("d_pooled", B * Dp);
    L.add("d_h0", B * H); L.add("d_h1", B * H);
    L.add("d_v", B * R * H); L.add("d_pre_v", B * R * H);
    L.add("d_qv", B * H); L.add("d_pre_qv", B * H);
    L.add("part_a", B * 2 * H); L.add("part_b", B * 2 * H); L.add("part_c", B * 2 * H);
    L.add("part_dw", B * H); L.add("part_db", B);
    L.add("dxp", T * B * 3 * H);
    L.add("d_rh", B * H);
    L.add("dx_embed", T * B * W);
    // shared scratch: split-k slabs, colsum partials, sumsq partials
    int64_t gw = 0;
    auto g = [&](int tA, int tB, int64_t M, int64_t N, int64_t K) {
        gw = max64(gw, vqa_gemm_workspace_floats(tA, tB, (int)M, (int)N, (int)K, 0));
    };
    g(0, 0, B * R, H, D); g(0, 0, T * B, 2 * H, W); g(0, 0, T * B, H, W); g(0, 0, B, H, H); g(0, 0, B, H, D);
    g(0, 0, B, 2 * H, H); g(0, 0, B, A, 2 * H);                                   // forward
    g(0, 1, B, 2 * H, A); g(1, 0, 2 * H, A, B); g(0, 1, B, H, 2 * H); g(1, 0, H, 2 * H, B); g(0, 1, B, D, H);
    g(1, 0, D, H, B); g(1, 0, H, H, B); g(0, 1, B, H, H); g(1, 0, D, H, B * R); g(1, 0, W, 2 * H, T * B);
    g(1, 0, H, 2 * H, T * B); g(1, 0, W, H, T * B); g(1, 0, H, H, T * B); g(0, 1, T * B, W, 2 * H);
    g(0, 1, T * B, W, H);                                                         // backward
    g(0, 0, T * B, 3 * H, W); g(0, 1, T * B, W, 3 * H); g(1, 0, W, 3 * H, T * B);  // packed x-projection
    g(1, 0, Wp, 3 * H, T * B);
    g(0, 0, B, W, 2 * H); g(0, 0, B, A, W); g(0, 1, B, W, A); g(1, 0, 2 * H, W, B); g(0, 1, B, 2 * H, W);  // word2vec head
    if (d.model_type == VQA_MODEL_BI) {
        const int64_t h = H / 2;
        g(0, 0, T * B, 3 * h, W); g(0, 1, T * B, W, 3 * h); g(1, 0, Wp, 3 * h, T * B); g(1, 0, h, 2 * h, T * B); g(1, 0, h, h, T * B);
        g(0, 0, B * T, H, H); g(0, 1, B * T, H, H); g(1, 0, H, H, B * T);
        g(0, 0, B * T, H, W); g(0, 1, B * T, W, H); g(1, 0, W, H, B * T);
    }
    if (d.model_type == VQA_MODEL_ENT) {
        const int64_t M = d.num_marginal, C = d.ent_cols;
        g(0, 0, B * M, 2 * H, H); g(0, 0, B * M, C, 2 * H); g(0, 1, B * M, 2 * H, C); g(0, 1, B * M, H, 2 * H);
    }
    if (d.model_type == VQA_MODEL_LEGACY_VQA) legacy_vqa_layout(L, d, gw);
    L.add("gemm_ws", max64(gw, 4));
    L.add("gemm_ws1", max64(gw, 4));            // scratch of the side stream (v_linear_v branch)
    int64_t cw = 0;
    cw = max64(cw, vqa_colsum_workspace_floats((int)B, (int)(2 * H)));
    cw = max64(cw, vqa_colsum_workspace_floats((int)(T * B), (int)(3 * H)));
    cw = max64(cw, vqa_colsum_workspace_floats((int)B, (int)A));
    cw = max64(cw, vqa_colsum_workspace_floats((int)B, (int)W));
    cw = max64(cw, vqa_colsum_workspace_floats((int)(B * T), (int)H));
    if (d.model_type == VQA_MODEL_LEGACY_VQA)
        cw = max64(cw, max64(max64(vqa_colsum_workspace_floats((int)(T * B), (int)(4 * H)), vqa_colsum_workspace_floats((int)(d.La * A), (int)(4 * H))),
                             vqa_colsum_workspace_floats((int)A, 1)));
    L.add("colsum_ws", max64(3 * cw, 4));    // x3: vqa_colsum3 reduces three partial matrices per launch
    L.add("colsum_ws1", max64(3 * cw, 4));
    L.add("part_a1", B * H); L.add("part_b1", B * H); L.add("part_c1", B * H);
    L.add("sumsq_ws", max64(vqa_sumsq_workspace_floats(T * B * W), 4));
    L.add("sumsq_ws2", max64(vqa_sumsq_workspace_floats(T * B * W), 4));
    return L;
}

#define TRY(x)                      \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != VQA_OK) return rc__; \
    } while (0)

struct Ctx {
    const vqa_dims_t& d;
    const Layout& L;
    char* ws;
    hipStream_t st;
    int lane;   // 0 = caller's stream, 1 = side stream (own scratch so the two never share a buffer)
    float* f(const char* name) const { return reinterpret_cast<float*>(ws + L.find(name)->off); }
    int32_t* i32(const char* name) const { return reinterpret_cast<int32_t*>(ws + L.find(name)->off); }
    float* gemm_ws() const { return f(lane ? "gemm_ws1" : "gemm_ws"); }
    float* colsum_ws() const { return f(lane ? "colsum_ws1" : "colsum_ws"); }
    float* part(int i) const {
        static const char* const n0[3] = {"part_a", "part_b", "part_c"};
        static const char* const n1[3] = {"part_a1", "part_b1", "part_c1"};
        return f(lane ? n1[i] : n0[i]);
    }
    int64_t gemm_ws_floats() const { return L.find("gemm_ws")->n; }
    int64_t colsum_ws_floats() const { return L.find("colsum_ws")->n; }
};

// ---- side stream: the v_linear_v branch (one big GEMM) runs beside the latency-bound GRU
// recurrence so its workgroups fill the CUs the small per-step GEMMs leave idle.  Fork/join
// with events keeps everything ordered with respect to the caller's stream (capture-safe).
struct Side {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ok = false, tried = false;
};
Side& side_stream() {
    static Side sd[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    Side& x = sd[dev & 63];
    if (!x.tried) {
        x.tried = true;
        // Off by default: measured on MI355X the overlap is worth ~1-2 % of the step (both sides are
        // MFMA-bound) while it inflates each overlapped kernel's own duration; VQA_HOT_OVERLAP=1 enables it.
        const char* env = getenv("VQA_HOT_OVERLAP");
        if (env != nullptr && env[0] == '1') {
            x.ok = hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess;
        }
    }
    return x;
}
int side_max_blocks() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VQA_HOT_SIDE_BLOCKS");
        v = e ? atoi(e) : 0;
    }
    return v;
}
// The recurrence is latency-bound (M = batch rows only): every one of its 2 T dependent step kernels pays ~7 us that
// is not matrix work (drain, launch gap, kernel arguments, first tile, fused epilogue) against 7-14 us that is
// (profiles/r3_gru_loop_variants.txt).  The rows of a batch are independent sequences, so the batch is cut into CHAINS
// of rows that run on their own streams with small co-resident workgroups (32x32 tiles, 4 waves: several workgroups
// of different chains share a CU) and -- the point -- in ANTI-PHASE: chain i starts i * delay later, so one chain's
// boundary falls into the other chain's matrix work.  In phase (no delay) both chains compute at half rate and wait
// together, and nothing is gained (profiles/r3_gru_split2.txt: one chain 558 us, two chains in phase 511-540, two chains
// 8-16 us apart 491).  Inside the step the second chain starts behind a cross-stream event anyway, and a small explicit
// delay is enough (profiles/r3_gru_chains.txt: 0 / 3 us 3.592 ms per step, 6 / 10 us 3.602-3.604).
// VQA_HOT_GRU_CHAINS (default 2; 1 = one chain on the caller's stream), VQA_HOT_GRU_CHAIN_DELAY_US (default 3).
struct Chains {
    static constexpr int MAXC = 4;
    hipStream_t s[MAXC] = {};
    hipEvent_t fork = nullptr, join[MAXC] = {};
    int n = 1;
    float delay_us = 3.f;
    bool tried = false;
};
Chains& gru_chains() {
    static Chains ch[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    Chains& x = ch[dev & 63];
    if (!x.tried) {
        x.tried = true;
        const char* e = getenv("VQA_HOT_GRU_CHAINS");
        int want = e ? atoi(e) : 2;
        if (want > Chains::MAXC) want = Chains::MAXC;
        const char* d = getenv("VQA_HOT_GRU_CHAIN_DELAY_US");
        if (d != nullptr) x.delay_us = (float)atof(d);
        bool ok = want > 1 && hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 1; ok && i < want; ++i)
            ok = hipStreamCreateWithFlags(&x.s[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&x.join[i], hipEventDisableTiming) == hipSuccess;
        x.n = ok ? want : 1;
    }
    return x;
}
// rows [lo, hi) of chain i of n over B rows: boundaries on multiples of 32 (the step kernels' row tile)
inline void chain_rows(int64_t B, int n, int i, int64_t* lo, int64_t* hi) {
    const int64_t tiles = (B + 31) / 32;
    const int64_t q = tiles / n, r = tiles % n;
    const int64_t t0 = i * q + (i < r ? i : r), t1 = t0 + q + (i < r ? 1 : 0);
    *lo = t0 * 32 < B ? t0 * 32 : B;
    *hi = t1 * 32 < B ? t1 * 32 : B;
}
// runs fn(chain, row0, rows, stream) for every chain: chain 0 on the caller's stream, chain i > 0 on its own stream behind
// a fork event and a delay of i * delay_us; the caller's stream waits for all of them at the end
template <typename Fn>
int run_chains(const Ctx& c, int64_t B, Fn fn) {
    Chains& ch = gru_chains();
    int n = ch.n;
    while (n > 1 && B < 64 * (int64_t)n) --n;         // at least two row tiles per chain
    if (n <= 1) return fn(0, (int64_t)0, B, c.st);
    if (hipEventRecord(ch.fork, c.st) != hipSuccess) return VQA_ERR_LAUNCH;
    // Whatever fails after the fork, the caller's stream is still ordered behind every side stream that may have work
    // in flight before the error code goes back: the caller owns the workspace and may free or reuse it on return.
    int rc = VQA_OK, launched = 0;
    for (int i = 1; i < n && rc == VQA_OK; ++i) {
        int64_t lo, hi;
        chain_rows(B, n, i, &lo, &hi);
        if (hipStreamWaitEvent(ch.s[i], ch.fork, 0) != hipSuccess) { rc = VQA_ERR_LAUNCH; break; }
        launched = i;
        if (ch.delay_us > 0.f) rc = vqa_stream_delay_us(ch.delay_us * (float)i, ch.s[i]);
        if (rc == VQA_OK) rc = fn(i, lo, hi - lo, ch.s[i]);
    }
    if (rc == VQA_OK) {
        int64_t lo, hi;
        chain_rows(B, n, 0, &lo, &hi);
        rc = fn(0, lo, hi - lo, c.st);
    }
    for (int i = 1; i <= launched; ++i)
        if (hipEventRecord(ch.join[i], ch.s[i]) != hipSuccess || hipStreamWaitEvent(c.st, ch.join[i], 0) != hipSuccess)
            rc = rc == VQA_OK ? VQA_ERR_LAUNCH : rc;
    return rc;
}
// how V_ft = features[image_idx] is produced: 0 = a gather pass in front of v_linear_v's GEMM (default), 1 = fused
// into that GEMM's operand load (VQA_FLAG_FUSED_GATHER or VQA_HOT_GATHER=fused).  Measured at bs 512
// (profiles/r2_gather_mode_ab.txt, r2_gather_gemm_bench.txt): 3.73 ms per step either way -- the 46 us gather pass
// doubles as a prefetch of V_ft into the Infinity Cache, from where the GEMM streams its left operand 37 us faster than
// the fused form streams it from HBM.  A third form, the gather on a helper stream beside the question branch's
// projection GEMMs, lost 0.13 ms to the cross-stream fork / join.
int gather_mode(const vqa_dims_t* dims) {
    static int env = -2;
    if (env == -2) {
        const char* e = getenv("VQA_HOT_GATHER");
        env = e == nullptr ? -1 : (strcmp(e, "fused") == 0 ? 1 : 0);
    }
    if (env >= 0) return env;
    return (dims->flags & VQA_FLAG_FUSED_GATHER) ? 1 : 0;
}
bool fork_side(const Ctx& c, Side& sd) {
    if (!sd.ok) return false;
    return hipEventRecord(sd.fork, c.st) == hipSuccess && hipStreamWaitEvent(sd.s, sd.fork, 0) == hipSuccess;
}
bool join_side_record(Side& sd) { return hipEventRecord(sd.join, sd.s) == hipSuccess; }
bool join_side(const Ctx& c, Side& sd) {
    return join_side_record(sd) && hipStreamWaitEvent(c.st, sd.join, 0) == hipSuccess;
}


int gemm(const Ctx& c, int tA, int tB, int64_t M, int64_t N, int64_t K, const float* A, int lda, const float* B,
         int ldb, float* C, int ldc, const float* bias = nullptr, const float* D = nullptr, int ldd = 0) {
    // side-stream GEMMs: one workgroup per CU (persistent), so the recurrence on the caller's
    // stream keeps finding free LDS / wave slots on every CU
    return vqa_gemm_f32_ex(tA, tB, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, bias, D, ldd, 0, c.gemm_ws(),
                           c.gemm_ws_floats(), c.lane ? side_max_blocks() : 0, c.st);
}
int colsum(const Ctx& c, const float* X, int64_t M, int64_t N, int ldx, float* out) {
    return vqa_colsum(X, (int)M, (int)N, ldx, out, c.colsum_ws(), c.colsum_ws_floats(), c.st);
}

bool dims_ok(const vqa_dims_t* d) {
    if (!(d && d->B > 0 && d->R > 0 && d->D > 0 && d->H > 0 && d->T > 0 && d->W > 0 && d->A > 0 && d->Vq > 0 &&
          d->N_img > 0 && d->model_type >= 0 && d->model_type <= VQA_MODEL_LEGACY_VQA))
        return false;
    if (d->model_type == VQA_MODEL_LEGACY_VQA)      // 16-byte rows everywhere; the scoring kernel keeps one H-row in LDS
        return d->map_dim > 0 && d->La > 0 && d->H % 4 == 0 && d->D % 4 == 0 && d->map_dim % 4 == 0 && d->W % 4 == 0 && d->H <= 1024 && d->Vq > 3;
    if (d->model_type == VQA_MODEL_BI) return d->H % 8 == 0;      // two cells of H / 2 units, 16-byte rows each
    if (d->model_type == VQA_MODEL_ENT)       // the pairings' tensors are addressed with 32-bit element counts
        return d->num_marginal > 0 && d->ent_cols > 0 && d->ent_cols <= d->A && d->ent_cols <= 4096 &&
               (int64_t)d->B * d->num_marginal * (2 * (int64_t)d->H > d->ent_cols ? 2 * (int64_t)d->H : d->ent_cols) < (1ll << 30);
    return true;
}
// K of pooled_linear_l = width of what the attention pools: v_adapt [R,H] for vlmap_answer_adapt, V_ft [R,D] otherwise
inline int64_t pooled_dim(const vqa_dims_t& d) { return d.model_type == VQA_MODEL_ADAPT ? d.H : d.D; }

// FC + LN + ReLU forward (modules.fc_layer, vlmap/modules.py:630-650)
int fc_ln_relu_fwd(const Ctx& c, const float* x, int64_t M, int64_t K, int64_t N, const vqa_fc_t& p, int rows,
                   const char* pre, const char* y, const char* mean, const char* rstd, const uint8_t* keep,
                   float keep_prob) {
    {
        ProbeScope ps(rows > 1 ? "v_linear_v.fwd_gemm" : "fc.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, M, N, K, x, (int)K, p.w, (int)N, c.f(pre), (int)N, p.b));
    }
    ProbeScope ps(rows > 1 ? "v_linear_v.ln_fwd" : "fc.ln_fwd", c.st);
    return vqa_ln_relu_fwd(c.f(pre), p.gamma, p.beta, keep, keep_prob, c.f(y), c.f(mean), c.f(rstd), (int)(M / rows),
                           rows, (int)N, c.st);
}

// the FC half of the backward once d_pre is known: parameter gradients from the per-group partials (g.w == NULL => frozen
// layer) and the optional dx = d_pre * W^T (+ dx)
int fc_bwd_tail(const Ctx& c, const float* x, int64_t M, int64_t K, int64_t N, const vqa_fc_t& p, const vqa_fc_t* g, int rows,
                const float* d_pre, const float* pa, const float* pb, const float* pc, float* dx, bool dx_accumulate) {
    const bool train = g != nullptr && g->w != nullptr;
    const int64_t G = M / rows;
    if (train) {
        {
            ProbeScope ps(rows > 1 ? "v_linear_v.ln_bwd" : "fc.ln_bwd", c.st);
            TRY(vqa_colsum3(pa, pb, pc, (int)G, (int)N, (int)N, g->gamma, g->beta, g->b, c.colsum_ws(), c.colsum_ws_floats(), c.st));
        }
        ProbeScope ps(rows > 1 ? "v_linear_v.dw_gemm" : "fc.dw_gemm", c.st);
        TRY(gemm(c, 1, 0, K, N, M, x, (int)K, d_pre, (int)N, g->w, (int)N));  // dW = x^T * d_pre
    }
    if (dx != nullptr) {
        ProbeScope ps("fc.dx_gemm", c.st);
        TRY(gemm(c, 0, 1, M, K, N, d_pre, (int)N, p.w, (int)N, dx, (int)K, nullptr, dx_accumulate ? dx : nullptr, (int)K));
    }
    return VQA_OK;
}

// backward of the same block.  dy -> d_pre (named buffer); optional parameter
// grads (g.w == NULL => frozen layer); optional dx = d_pre * W^T (+ dx_add).
int fc_ln_relu_bwd(const Ctx& c, const float* dy, const float* x, int64_t M, int64_t K, int64_t N, const vqa_fc_t& p,
                   const vqa_fc_t* g, int rows, const char* pre, const char* mean, const char* rstd,
                   const uint8_t* keep, float keep_prob, const char* d_pre, float* dx, bool dx_accumulate) {
    const bool train = g != nullptr && g->w != nullptr;
    const int64_t G = M / rows;
    {
        ProbeScope ps(rows > 1 ? "v_linear_v.ln_bwd" : "fc.ln_bwd", c.st);
        TRY(vqa_ln_relu_bwd(dy, c.f(pre), c.f(mean), c.f(rstd), p.gamma, p.beta, keep, keep_prob, c.f(d_pre),
                            train ? c.part(0) : nullptr, train ? c.part(1) : nullptr,
                            train ? c.part(2) : nullptr, (int)G, rows, (int)N, c.st));
    }
    return fc_bwd_tail(c, x, M, K, N, p, g, rows, c.f(d_pre), c.part(0), c.part(1), c.part(2), dx, dx_accumulate);
}

// pooled_linear_l and q_linear_l finish in one launch that also forms their product (vqa_ln_pair_mul_*): two LayerNorm
// launches and the element-wise one less per direction.  VQA_HOT_LN_PAIR=0 restores the three-launch form (A/B).
inline bool ln_pair_enabled() {
    static const bool v = [] { const char* e = getenv("VQA_HOT_LN_PAIR"); return e == nullptr || atoi(e) != 0; }();
    return v;
}
bool ln_pair_ok(const vqa_dims_t& d, const vqa_params_t* P) {
    if (!ln_pair_enabled() || d.model_type == 5) return false;       // vlmap_answer_noc has no product
    const void* ptrs[4] = {P->pooled_linear_l.gamma, P->pooled_linear_l.beta, P->q_linear_l.gamma, P->q_linear_l.beta};
    return vqa_ln_pair_mul_supported(d.H, ptrs, 4) != 0;
}

#include "legacy_vqa.inc"

// ---- model_type 12: bi-directional question encoder + question self-attention (vqa/model_vlmap_finetune.py:119-150) ----
struct BiGru { const float *wg, *bg, *wc, *bc; };

int bi_question_fwd(const Ctx& c, const vqa_params_t* P, const vqa_batch_t* bt) {
    const vqa_dims_t& d = c.d;
    const int64_t B = d.B, H = d.H, T = d.T, W = d.W, h = H / 2;
    const int64_t Wp = ((W + 1 + 3) / 4) * 4;
    VQA_REQUIRE(P->embed2 && P->gru_bw_wg && P->gru_bw_bg && P->gru_bw_wc && P->gru_bw_bc && P->q_att_key.w &&
                    P->q_att_key.gamma && P->q_att_query.w && P->q_att_query.gamma && P->word_score.w && P->word_score.b &&
                    P->v_word_fc.w && P->v_word_fc.gamma,
                VQA_ERR_ARG);
    int32_t* q_rev = c.i32("q_rev");
    {
        ProbeScope ps("embed.fwd", c.st);
        TRY(vqa_reverse_tokens(bt->q_intseq, bt->q_intseq_len, q_rev, (int)B, (int)T, c.st));
        TRY(vqa_embed_fwd_ld(P->embed, bt->q_intseq, c.f("x_tm"), (int)B, (int)T, (int)W, d.Vq, (int)Wp, c.st));
        TRY(vqa_embed_fwd_ld(P->embed, q_rev, c.f("x_tm_bw"), (int)B, (int)T, (int)W, d.Vq, (int)Wp, c.st));
        // the second embedding, batch-major rows (b, t): one "sequence" per token
        TRY(vqa_embed_fwd_ld(P->embed2, bt->q_intseq, c.f("e2"), (int)(B * T), 1, (int)W, d.Vq, (int)W, c.st));
    }
    const BiGru cell[2] = {{P->gru_wg, P->gru_bg, P->gru_wc, P->gru_bc}, {P->gru_bw_wg, P->gru_bw_bg, P->gru_bw_wc, P->gru_bw_bc}};
    static const char* const nm[2][9] = {{"x_tm", "xp", "wx_cat", "bx_cat", "hs", "gru_r", "gru_u", "gru_c", "gru_rh"},
                                         {"x_tm_bw", "xp_bw", "wx_cat_bw", "bx_cat_bw", "hs_bw", "gru_r_bw", "gru_u_bw",
                                          "gru_c_bw", "gru_rh_bw"}};
    for (int k = 0; k < 2; ++k) {
        {
            ProbeScope ps("gru.xp_gemm", c.st);
            TRY(vqa_gru_pack_wx(cell[k].wg, cell[k].wc, cell[k].bg, cell[k].bc, c.f(nm[k][2]), c.f(nm[k][3]), (int)W, (int)h, c.st));
            TRY(gemm(c, 0, 0, T * B, 3 * h, W, c.f(nm[k][0]), (int)Wp, c.f(nm[k][2]), (int)(3 * h), c.f(nm[k][1]), (int)(3 * h),
                     c.f(nm[k][3])));
        }
        float* hs = c.f(nm[k][4]);
        if (hipMemsetAsync(hs, 0, (size_t)B * h * sizeof(float), c.st) != hipSuccess) return VQA_ERR_LAUNCH;
        ProbeScope ps("gru.fwd", c.st);
        TRY(vqa_gru_seq_fwd_rows(c.f(nm[k][1]), cell[k].wg + W * 2 * h, cell[k].wc + W * h, bt->q_intseq_len, hs, c.f(nm[k][5]),
                                 c.f(nm[k][6]), c.f(nm[k][7]), c.f(nm[k][8]), (int)T, (int)B, (int)h, 0, (int)B, c.st));
    }
    TRY(vqa_bi_outputs_fwd(c.f("hs"), c.f("hs_bw"), bt->q_intseq_len, c.f("q_L_map"), c.f("q_L_ft"), (int)B, (int)T, (int)h, c.st));
    // keys: fc_layer on [B,T,H] -- LayerNorm over the whole [T,H] block of a question, padded positions included
    TRY(fc_ln_relu_fwd(c, c.f("q_L_map"), B * T, H, H, P->q_att_key, (int)T, "pre_key", "q_att_key", "mean_key", "rstd_key",
                       nullptr, 1.f));
    TRY(fc_ln_relu_fwd(c, c.f("q_L_ft"), B, H, H, P->q_att_query, 1, "pre_query", "q_att_query", "mean_query", "rstd_query",
                       nullptr, 1.f));
    TRY(fc_ln_relu_fwd(c, c.f("e2"), B * T, W, H, P->v_word_fc, (int)T, "pre_vw", "q_v_ft", "mean_vw", "rstd_vw", nullptr, 1.f));
    ProbeScope ps("attn_pool.fwd", c.st);
    return vqa_attn_pool_fwd(c.f("q_att_key"), c.f("q_att_query"), c.f("q_v_ft"), bt->q_intseq_len, P->word_score.w,
                             P->word_score.b, bt->keep_word, d.keep_att, c.f("w_att_score"), c.f("pooled_q_v"), (int)B, (int)T,
                             (int)H, (int)H, c.st);
}

// everything between d(pooled_q_v) / d(q_L_ft) and the two recurrences' inputs: word attention, v_word_fc (+ the second
// embedding's slices), q_att_query, q_att_key.  dh = gradient wrt q_L_ft so far (q_linear_l's share), accumulated into.
int bi_question_bwd_head(const Ctx& c, const vqa_params_t* P, const vqa_params_t* G, const vqa_batch_t* bt, float* dh) {
    const vqa_dims_t& d = c.d;
    const int64_t B = d.B, H = d.H, T = d.T, W = d.W;
    {
        ProbeScope ps("attn_pool.bwd", c.st);
        TRY(vqa_attn_pool_bwd(c.f("d_pooled_qv"), c.f("q_att_key"), c.f("q_att_query"), c.f("q_v_ft"), c.f("w_att_score"),
                              P->word_score.w, bt->keep_word, d.keep_att, c.f("d_key"), c.f("d_query"), c.f("part_wdw"),
                              c.f("part_wdb"), (int)B, (int)T, (int)H, (int)H, c.st));
        if (G->word_score.w != nullptr) {
            TRY(colsum(c, c.f("part_wdw"), B, H, (int)H, G->word_score.w));
            TRY(colsum(c, c.f("part_wdb"), B, 1, 1, G->word_score.b));
        }
        // the pooled memory is a trainable layer's output here: d q_v_ft = w_att (x) d pooled_q_v
        TRY(vqa_outer_rows(c.f("w_att_score"), c.f("d_pooled_qv"), c.f("d_qvft"), (int)B, (int)T, (int)H, c.st));
    }
    const bool e2_train = G->embed2 != nullptr;
    if (G->v_word_fc.w != nullptr || e2_train)
        TRY(fc_ln_relu_bwd(c, c.f("d_qvft"), c.f("e2"), B * T, W, H, P->v_word_fc, &G->v_word_fc, (int)T, "pre_vw", "mean_vw",
                           "rstd_vw", nullptr, 1.f, "d_pre_vw", e2_train ? c.f("d_e2") : nullptr, false));
    if (e2_train) {      // V_WordMap: scatter-add of the batch-major slices (every position, padding included) + their norm
        ProbeScope ps("embed.bwd", c.st);
        TRY(vqa_embed_bwd_len_det(c.f("d_e2"), bt->q_intseq, nullptr, G->embed2, (int)(B * T), 1, (int)W, d.Vq,
                                  (d.flags & VQA_FLAG_DETERMINISTIC) ? 1 : 0, c.st));
        TRY(vqa_sumsq(c.f("d_e2"), B * T * W, nullptr, c.f("sq_e2"), c.f("sumsq_ws2"), c.L.find("sumsq_ws2")->n, c.st));
    }
    TRY(fc_ln_relu_bwd(c, c.f("d_query"), c.f("q_L_ft"), B, H, H, P->q_att_query, &G->q_att_query, 1, "pre_query", "mean_query",
                       "rstd_query", nullptr, 1.f, "d_pre_query", dh, true));
    return fc_ln_relu_bwd(c, c.f("d_key"), c.f("q_L_map"), B * T, H, H, P->q_att_key, &G->q_att_key, (int)T, "pre_key", "mean_key",
                          "rstd_key", nullptr, 1.f, "d_pre_key", c.f("d_qmap"), false);
}

// phase 2: both BPTTs, the gradient wrt the looked-up embeddings, LearnGloVe's scatter-add and the slice norm
int bi_question_bwd_bptt(const Ctx& c, const vqa_params_t* P, const vqa_params_t* G, const vqa_batch_t* bt, const float* dh,
                         float* embed_slice_sq) {
    const vqa_dims_t& d = c.d;
    const int64_t B = d.B, H = d.H, T = d.T, W = d.W, h = H / 2;
    TRY(vqa_bi_outputs_bwd(c.f("d_qmap"), dh, bt->q_intseq_len, c.f("dout_fw"), c.f("dout_bw"), c.f("dhT_fw"), c.f("dhT_bw"),
                           (int)B, (int)T, (int)h, c.st));
    {
        ProbeScope ps("gru.bwd", c.st);
        TRY(vqa_gru_seq_bwd_outs(c.f("dhT_fw"), P->gru_wg + W * 2 * h, P->gru_wc + W * h, bt->q_intseq_len, c.f("hs"), c.f("gru_r"),
                                 c.f("gru_u"), c.f("gru_c"), c.f("dout_fw"), c.f("dxp"), c.f("dhS_fw"), (int)T, (int)B, (int)h, c.st));
        TRY(vqa_gru_seq_bwd_outs(c.f("dhT_bw"), P->gru_bw_wg + W * 2 * h, P->gru_bw_wc + W * h, bt->q_intseq_len, c.f("hs_bw"),
                                 c.f("gru_r_bw"), c.f("gru_u_bw"), c.f("gru_c_bw"), c.f("dout_bw"), c.f("dxp_bw"), c.f("dhS_bw"),
                                 (int)T, (int)B, (int)h, c.st));
    }
    {
        ProbeScope ps("gru.dx_gemm", c.st);
        TRY(vqa_gru_pack_wx(P->gru_wg, P->gru_wc, P->gru_bg, P->gru_bc, c.f("wx_cat"), c.f("bx_cat"), (int)W, (int)h, c.st));
        TRY(gemm(c, 0, 1, T * B, W, 3 * h, c.f("dxp"), (int)(3 * h), c.f("wx_cat"), (int)(3 * h), c.f("dx_fw"), (int)W));
        TRY(vqa_gru_pack_wx(P->gru_bw_wg, P->gru_bw_wc, P->gru_bw_bg, P->gru_bw_bc, c.f("wx_cat_bw"), c.f("bx_cat_bw"), (int)W,
                            (int)h, c.st));
        TRY(gemm(c, 0, 1, T * B, W, 3 * h, c.f("dxp_bw"), (int)(3 * h), c.f("wx_cat_bw"), (int)(3 * h), c.f("dx_bw"), (int)W));
        TRY(vqa_bi_dx_combine(c.f("dx_fw"), c.f("dx_bw"), bt->q_intseq_len, c.f("dx_embed"), (int)B, (int)T, (int)W, c.st));
    }
    ProbeScope ps("embed.bwd", c.st);
    if (G->embed != nullptr)
        TRY(vqa_embed_bwd_len_det(c.f("dx_embed"), bt->q_intseq, bt->q_intseq_len, G->embed, (int)B, (int)T, (int)W, d.Vq,
                                  (d.flags & VQA_FLAG_DETERMINISTIC) ? 1 : 0, c.st));
    if (embed_slice_sq != nullptr)
        TRY(vqa_sumsq(c.f("dx_embed"), T * B * W, G->embed2 != nullptr ? c.f("sq_e2") : nullptr, embed_slice_sq, c.f("sumsq_ws"),
                      c.L.find("sumsq_ws")->n, c.st));
    return VQA_OK;
}

// phases 4 / 8: the two cells' gate (incl. every x row and bias) / candidate recurrent weight gradients
int bi_question_bwd_weights(const Ctx& c, const vqa_params_t* G, int phases) {
    const vqa_dims_t& d = c.d;
    const int64_t B = d.B, H = d.H, T = d.T, W = d.W, h = H / 2;
    const int64_t Wp = ((W + 1 + 3) / 4) * 4;
    float* const gw[2][4] = {{G->gru_wg, G->gru_wc, G->gru_bg, G->gru_bc}, {G->gru_bw_wg, G->gru_bw_wc, G->gru_bw_bg, G->gru_bw_bc}};
    static const char* const nm[2][5] = {{"x_tm", "dxp", "dwx_cat", "hs", "gru_rh"}, {"x_tm_bw", "dxp_bw", "dwx_cat_bw", "hs_bw", "gru_rh_bw"}};
    for (int k = 0; k < 2; ++k) {
        if (gw[k][0] == nullptr) continue;
        float* dxp = c.f(nm[k][1]);
        if (phases & 4) {
            {
                ProbeScope ps("gru.dwx_gemm", c.st);
                TRY(gemm(c, 1, 0, Wp, 3 * h, T * B, c.f(nm[k][0]), (int)Wp, dxp, (int)(3 * h), c.f(nm[k][2]), (int)(3 * h)));
                TRY(vqa_gru_unpack_dwx_bias(c.f(nm[k][2]), gw[k][0], gw[k][1], gw[k][2], gw[k][3], (int)W, (int)h, c.st));
            }
            ProbeScope ps("gru.dwh_gemm", c.st);
            TRY(gemm(c, 1, 0, h, 2 * h, T * B, c.f(nm[k][3]), (int)h, dxp, (int)(3 * h), gw[k][0] + W * 2 * h, (int)(2 * h)));
        }
        if (phases & 8) {
            ProbeScope ps("gru.dwh_gemm", c.st);
            TRY(gemm(c, 1, 0, h, h, T * B, c.f(nm[k][4]), (int)h, dxp + 2 * h, (int)(3 * h), gw[k][1] + W * h, (int)h));
        }
    }
    return VQA_OK;
}

}  // namespace

extern "C" int64_t vqa_fusion_workspace_bytes(const vqa_dims_t* dims) {
    if (!dims_ok(dims)) return VQA_ERR_ARG;
    return make_layout(*dims).total;
}

extern "C" int vqa_fusion_tensor(const vqa_dims_t* dims, const char* name, int64_t* offset_bytes, int64_t* n_elems) {
    if (!dims_ok(dims) || name == nullptr) return VQA_ERR_ARG;
    const Layout L = make_layout(*dims);
    const char* key = name;
    if (strcmp(name, "condition") == 0 && dims->model_type == VQA_MODEL_ANSWER2) key = "q_L_ft2";   // model_vlmap_answer2.py:131
    else if (strcmp(name, "condition") == 0 && dims->model_type == VQA_MODEL_BI) key = "q_L_ft";  // concat of the two final states
    else if (dims->model_type == VQA_MODEL_LEGACY_VQA && (strcmp(name, "q_L_ft") == 0 || strcmp(name, "answer_ft") == 0)) {
        // final LSTM states: the last [N,H] block of the time-major state tapes
        const bool q = name[0] == 'q';
        const Entry* hsn = L.find(q ? "lv_hsq" : "lv_hsa");
        const int64_t N = q ? dims->B : dims->A, Tn = q ? dims->T : dims->La;
        if (offset_bytes) *offset_bytes = hsn->off + Tn * N * dims->H * 4;
        if (n_elems) *n_elems = N * dims->H;
        return VQA_OK;
    }
    else if (strcmp(name, "condition") == 0) {  // heavy_output['condition'] = final GRU state = hs[T]
        const Entry* h = L.find("hs");
        if (offset_bytes) *offset_bytes = h->off + (int64_t)dims->T * dims->B * dims->H * 4;
        if (n_elems) *n_elems = (int64_t)dims->B * dims->H;
        return VQA_OK;
    }
    const Entry* e = L.find(key);
    if (e == nullptr) return VQA_ERR_ARG;
    if (offset_bytes) *offset_bytes = e->off;
    if (n_elems) *n_elems = e->n;
    return VQA_OK;
}

namespace {
inline bool visual_late_enabled() {       // VQA_HOT_VISUAL_LATE=0: the round-1 order (A/B)
    static const bool v = [] { const char* e = getenv("VQA_HOT_VISUAL_LATE"); return e == nullptr || atoi(e) != 0; }();
    return v;
}
}  // namespace

namespace {
inline bool gru_ws_on() {          // VQA_HOT_GRU_WS=0: the per-step recurrence kernels (A/B)
    static const bool v = [] { const char* e = getenv("VQA_HOT_GRU_WS"); return e == nullptr || atoi(e) != 0; }();
    return v;
}
inline bool xcat_enabled() {       // VQA_HOT_XCAT=0: the two-GEMM form (A/B)
    static const bool v = [] { const char* e = getenv("VQA_HOT_XCAT"); return e == nullptr || atoi(e) != 0; }();
    return v;
}
}  // namespace

extern "C" int vqa_fusion_forward(const vqa_dims_t* dims, const vqa_params_t* P, const vqa_batch_t* bt,
                                  void* workspace, int64_t workspace_bytes, int want_dz, void* stream) {
    VQA_REQUIRE(dims_ok(dims) && P && bt && workspace, VQA_ERR_ARG);
    const Layout L = make_layout(*dims);
    Ctx c{*dims, L, static_cast<char*>(workspace), static_cast<hipStream_t>(stream), 0};
    VQA_REQUIRE(workspace_bytes >= c.L.total, VQA_ERR_WORKSPACE);
    VQA_REQUIRE(vqa_aligned16(workspace), VQA_ERR_ALIGN);
    ProbeScope ps_all("forward", c.st);
    if (dims->model_type == VQA_MODEL_LEGACY_VQA) return legacy_vqa_forward(c, P, bt, want_dz);
    const int64_t B = dims->B, R = dims->R, D = dims->D, H = dims->H, T = dims->T, W = dims->W, A = dims->A;

    // visual branch (a1 + a2) on the side stream (VQA_HOT_OVERLAP=1 only), question branch (a3-a5) on the caller's
    Side& sd = side_stream();
    const bool forked = fork_side(c, sd);
    Ctx cv{*dims, L, c.ws, forked ? sd.s : c.st, forked ? 1 : 0};
    const bool fuse_gather = gather_mode(dims) == 1 && (D % 32 == 0) && (H % 4 == 0) && vqa_aligned16(bt->table);
    const int mt = dims->model_type;
    // The visual branch (a1 + a2).  With the side stream it is launched first and overlaps the recurrence; on one
    // stream it runs AFTER the question branch, right before the attention that consumes it: V_ft (151 MB) and
    // v_linear_v (75 MB) are then still in the 256 MB Infinity Cache when the attention kernel reads them, instead
    // of having been pushed out by the recurrence's traffic.
    auto visual_branch = [&]() -> int {
        // a1: V_ft = features[image_idx] (a pass of its own, or inside the GEMM below), num_V_ft = num_boxes[image_idx]
        {
            ProbeScope ps("gather", cv.st);
            TRY(vqa_gather_features(bt->table, bt->nbox_table, bt->image_idx, fuse_gather ? nullptr : cv.f("V_ft"),
                                    cv.i32("num_V_ft"), (int)B, (int)R, (int)D, dims->N_img, cv.st));
        }
        // a2: v_linear_v, LN statistics over the whole [R,H] block of a sample
        if (fuse_gather) {
            {
                ProbeScope ps("v_linear_v.fwd_gemm", cv.st);
                TRY(vqa_gemm_f32_gather((int)(B * R), (int)H, (int)D, bt->table, (int)D, bt->image_idx, (int)R, dims->N_img,
                                        P->v_linear_v.w, (int)H, cv.f("pre_v"), (int)H, P->v_linear_v.b, cv.f("V_ft"), (int)D,
                                        cv.st));
            }
            ProbeScope ps("v_linear_v.ln_fwd", cv.st);
            TRY(vqa_ln_relu_fwd(cv.f("pre_v"), P->v_linear_v.gamma, P->v_linear_v.beta, nullptr, 1.f, cv.f("v_linear_v"),
                                cv.f("mean_v"), cv.f("rstd_v"), (int)B, (int)R, (int)H, cv.st));
        } else {
            TRY(fc_ln_relu_fwd(cv, cv.f("V_ft"), B * R, D, H, P->v_linear_v, (int)R, "pre_v", "v_linear_v", "mean_v", "rstd_v",
                               nullptr, 1.f));
        }
        if (mt == VQA_MODEL_ADAPT) {     // v_adapt: a second FC + LN[R,H] + ReLU on the same V_ft (:132-135)
            VQA_REQUIRE(P->v_adapt.w != nullptr && P->v_adapt.gamma != nullptr, VQA_ERR_ARG);
            TRY(fc_ln_relu_fwd(cv, cv.f("V_ft"), B * R, D, H, P->v_adapt, (int)R, "pre_va", "v_adapt", "mean_va", "rstd_va",
                               nullptr, 1.f));
        }
        return VQA_OK;
    };
    const bool visual_late = !forked && visual_late_enabled();
    if (!visual_late) TRY(visual_branch());
    if (forked && !join_side_record(sd)) return VQA_ERR_LAUNCH;
    const float* h = nullptr;            // the question code q_L_ft [B,H]
    const float* qv_in = nullptr;        // what q_linear_v reads
    if (mt == VQA_MODEL_BI) {
        TRY(bi_question_fwd(c, P, bt));
        h = c.f("q_L_ft");
        qv_in = c.f("pooled_q_v");
    } else {
    // a3: embedding lookup, time-major
    // x_tm rows carry the constant 1 after the W inputs (vqa_embed_fwd_ld): the x-part weight-gradient GEMM then also
    // delivers the bias gradients, and the two passes over dxp that summed its columns are gone
    const int64_t Wp = ((W + 1 + 3) / 4) * 4;
    {
        ProbeScope ps("embed.fwd", c.st);
        TRY(vqa_embed_fwd_ld(P->embed, bt->q_intseq, c.f("x_tm"), (int)B, (int)T, (int)W, dims->Vq, (int)Wp, c.st));
    }
    // a4: GRU.  Input projections of all steps as two big GEMMs ...
    float* xp = c.f("xp");
    {
    ProbeScope ps("gru.xp_gemm", c.st);
    if (xcat_enabled()) {
        TRY(vqa_gru_pack_wx(P->gru_wg, P->gru_wc, P->gru_bg, P->gru_bc, c.f("wx_cat"), c.f("bx_cat"), (int)W, (int)H, c.st));
        TRY(gemm(c, 0, 0, T * B, 3 * H, W, c.f("x_tm"), (int)Wp, c.f("wx_cat"), (int)(3 * H), xp, (int)(3 * H), c.f("bx_cat")));
    } else {
        TRY(gemm(c, 0, 0, T * B, 2 * H, W, c.f("x_tm"), (int)Wp, P->gru_wg, (int)(2 * H), xp, (int)(3 * H), P->gru_bg));
        TRY(gemm(c, 0, 0, T * B, H, W, c.f("x_tm"), (int)Wp, P->gru_wc, (int)H, xp + 2 * H, (int)(3 * H), P->gru_bc));
    }
    }
    float* hs = c.f("hs");
    if (hipMemsetAsync(hs, 0, (size_t)B * H * sizeof(float), c.st) != hipSuccess) return VQA_ERR_LAUNCH;
    const float* Wg_h = P->gru_wg + W * 2 * H;
    const float* Wc_h = P->gru_wc + W * H;
    {
        ProbeScope ps("gru.fwd", c.st);
        // (a length-sorted batch whose longest row runs to T: the weight-stationary launch computes every row of every step
        // under its length mask -- same results -- and is faster than the shrinking per-step kernels at these sizes)
        const bool ws_ok = gru_ws_on() && (bt->live_rows == nullptr || (T > 0 && bt->live_rows[T - 1] > 0));
        if (ws_ok && vqa_gru_ws_supported((int)T, (int)B, (int)H) == 1) {
            // one launch, recurrent weights resident in registers and LDS, eight XCD-local chains (csrc/gru_ws.hip)
            TRY(vqa_gru_seq_fwd_ws(xp, Wg_h, Wc_h, bt->q_intseq_len, hs, c.f("gru_r"), c.f("gru_u"), c.f("gru_c"), c.f("gru_rh"),
                                   (int)T, (int)B, (int)H, c.f("gru_ws"), c.st));
        } else if (bt->live_rows != nullptr) {          // rows sorted by length: skip finished sequences
            TRY(vqa_gru_seq_fwd_live(xp, Wg_h, Wc_h, bt->q_intseq_len, bt->live_rows, hs, c.f("gru_r"), c.f("gru_u"),
                                     c.f("gru_c"), c.f("gru_rh"), (int)T, (int)B, (int)H, c.st));
        } else {                                  // independent row chains in anti-phase (gru_chains)
            float* gr = c.f("gru_r"); float* gu = c.f("gru_u"); float* gc = c.f("gru_c"); float* grh = c.f("gru_rh");
            TRY(run_chains(c, B, [&](int, int64_t row0, int64_t rows, hipStream_t st) {
                return vqa_gru_seq_fwd_rows(xp, Wg_h, Wc_h, bt->q_intseq_len, hs, gr, gu, gc, grh, (int)T, (int)B, (int)H,
                                            (int)row0, (int)rows, st);
            }));
        }
    }
    h = hs + T * B * H;
    qv_in = h;
    }
    // what q_linear_l reads: the GRU state, or one of the ablations' layers on top of it
    const float* lin_in = h;
    if (mt == VQA_MODEL_ANSWER2) {           // q_L_ft2 = tanh(LN(fc(q_L_ft)))   (vqa/model_vlmap_answer2.py:127-130)
        VQA_REQUIRE(P->q_L_ft2.w != nullptr && P->q_L_ft2.gamma != nullptr, VQA_ERR_ARG);
        ProbeScope ps("fc.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, B, H, H, h, (int)H, P->q_L_ft2.w, (int)H, c.f("pre_ft2"), (int)H, P->q_L_ft2.b));
        TRY(vqa_ln_act_fwd(c.f("pre_ft2"), P->q_L_ft2.gamma, P->q_L_ft2.beta, nullptr, 1.f, c.f("q_L_ft2"), c.f("mean_ft2"),
                           c.f("rstd_ft2"), (int)B, 1, (int)H, 1, c.st));
        lin_in = c.f("q_L_ft2");
    } else if (mt == VQA_MODEL_NO_NOISE || mt == VQA_MODEL_FULL) {     // q_L_mean: a plain linear layer (:122-125)
        VQA_REQUIRE(P->q_L_mean.w != nullptr, VQA_ERR_ARG);
        ProbeScope ps("fc.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, B, H, H, h, (int)H, P->q_L_mean.w, (int)H, c.f("q_L_mean"), (int)H, P->q_L_mean.b));
        lin_in = c.f("q_L_mean");
        if (mt == VQA_MODEL_FULL) {          // reparameterisation (vqa/model_vlmap_answer_full.py:128-134)
            VQA_REQUIRE(P->q_L_log_sigma_sq.w != nullptr && bt->noise != nullptr, VQA_ERR_ARG);
            TRY(gemm(c, 0, 0, B, H, H, h, (int)H, P->q_L_log_sigma_sq.w, (int)H, c.f("q_L_log_sigma_sq"), (int)H,
                     P->q_L_log_sigma_sq.b));
            TRY(vqa_reparam_fwd(c.f("q_L_mean"), c.f("q_L_log_sigma_sq"), bt->noise, c.f("q_L_mean_noise"), c.f("extra_row"),
                                (int)B, (int)H, c.st));
            lin_in = c.f("q_L_mean_noise");
        }
    }
    // a5
    TRY(fc_ln_relu_fwd(c, qv_in, B, H, H, P->q_linear_v, 1, "pre_qv", "q_linear_v", "mean_qv", "rstd_qv", nullptr, 1.f));
    if (forked && hipStreamWaitEvent(c.st, sd.join, 0) != hipSuccess) return VQA_ERR_LAUNCH;
    if (visual_late) TRY(visual_branch());
    // a6 + a7 (vlmap_answer_adapt pools v_adapt [R,H] instead of V_ft [R,D])
    const int64_t Dp = pooled_dim(*dims);
    {
    ProbeScope ps("attn_pool.fwd", c.st);
    TRY(vqa_attn_pool_fwd(c.f("v_linear_v"), c.f("q_linear_v"), c.f(mt == VQA_MODEL_ADAPT ? "v_adapt" : "V_ft"),
                          c.i32("num_V_ft"), P->score.w, P->score.b, bt->keep_att, dims->keep_att, c.f("att_score"),
                          c.f("pooled_V_ft"), (int)B, (int)R, (int)H, (int)Dp, c.st));
    }
    // a8
    const bool pair = ln_pair_ok(*dims, P);
    if (pair) {
        {
            ProbeScope ps("fc.fwd_gemm", c.st);
            TRY(gemm(c, 0, 0, B, H, Dp, c.f("pooled_V_ft"), (int)Dp, P->pooled_linear_l.w, (int)H, c.f("pre_pl"), (int)H,
                     P->pooled_linear_l.b));
            TRY(gemm(c, 0, 0, B, H, H, lin_in, (int)H, P->q_linear_l.w, (int)H, c.f("pre_ll"), (int)H, P->q_linear_l.b));
        }
        ProbeScope ps("fc.ln_fwd", c.st);
        TRY(vqa_ln_pair_mul_fwd(c.f("pre_pl"), c.f("pre_ll"), P->pooled_linear_l.gamma, P->pooled_linear_l.beta, P->q_linear_l.gamma,
                                P->q_linear_l.beta, c.f("pooled_linear_l"), c.f("l_linear_l"), c.f("joint_in"), c.f("mean_pl"),
                                c.f("rstd_pl"), c.f("mean_ll"), c.f("rstd_ll"), (int)B, (int)H, c.st));
    } else {
    TRY(fc_ln_relu_fwd(c, c.f("pooled_V_ft"), B, Dp, H, P->pooled_linear_l, 1, "pre_pl", "pooled_linear_l", "mean_pl",
                       "rstd_pl", nullptr, 1.f));
    TRY(fc_ln_relu_fwd(c, lin_in, B, H, H, P->q_linear_l, 1, "pre_ll", "l_linear_l", "mean_ll", "rstd_ll", nullptr, 1.f));
    }
    // a9
    if (dims->model_type == 5) {
        // vlmap_answer_noc (vqa/model_vlmap_answer_noc.py:177-188): no composition -- joint_v on pooled_linear_l and joint_l
        // on l_linear_l, each FC + LN + ReLU + dropout .5 ("joint" holds v_joint)
        VQA_REQUIRE(P->joint2.w != nullptr && P->head2.w != nullptr, VQA_ERR_ARG);
        TRY(fc_ln_relu_fwd(c, c.f("pooled_linear_l"), B, H, 2 * H, P->joint_fc, 1, "pre_j", "joint", "mean_j", "rstd_j",
                           bt->keep_joint, dims->keep_joint));
        TRY(fc_ln_relu_fwd(c, c.f("l_linear_l"), B, H, 2 * H, P->joint2, 1, "pre_jl", "l_joint", "mean_jl", "rstd_jl",
                           bt->keep_joint2, dims->keep_joint));
    } else {
    if (!pair) {
        ProbeScope ps("eltwise", c.st);
        TRY(vqa_mul(c.f("pooled_linear_l"), c.f("l_linear_l"), c.f("joint_in"), B * H, c.st));
    }
    TRY(fc_ln_relu_fwd(c, c.f("joint_in"), B, H, 2 * H, P->joint_fc, 1, "pre_j", "joint", "mean_j", "rstd_j",
                       bt->keep_joint, dims->keep_joint));
    }
    // a10
    if (dims->model_type == 2) {
        // standard_word2vec (vqa/model_standard_word2vec.py:180-188): classifier FC into the 300-d word space, then
        // logits = joint2 x the constant [W, A] GloVe matrix of the answers
        VQA_REQUIRE(P->answer_glove != nullptr, VQA_ERR_ARG);
        ProbeScope ps("head.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, B, W, 2 * H, c.f("joint"), (int)(2 * H), P->head.w, (int)W, c.f("joint2"), (int)W, P->head.b));
        TRY(gemm(c, 0, 0, B, A, W, c.f("joint2"), (int)W, P->answer_glove, (int)A, c.f("logit"), (int)A));
    } else if (dims->model_type == 5) {
        // logit = WordWeightAnswerV(v_joint) + WordWeightAnswerL(l_joint)   (:190-204): the second GEMM adds onto the first
        ProbeScope ps("head.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, B, A, 2 * H, c.f("joint"), (int)(2 * H), P->head.w, (int)A, c.f("logit"), (int)A, P->head.b));
        TRY(gemm(c, 0, 0, B, A, 2 * H, c.f("l_joint"), (int)(2 * H), P->head2.w, (int)A, c.f("logit"), (int)A, P->head2.b,
                 c.f("logit"), (int)A));
    } else if (dims->model_type == 4 || dims->model_type == 6) {
        // vlmap_answer_vqa_all2 (vqa/model_vlmap_answer_vqa_all2.py:196-227): the fixed WordWeightAnswer head and the
        // trainable TunedWordWeightAnswer head, BOTH on `joint` (the reference's tuned head reads `joint`, :216-217);
        // _vqa_all: the same with the fixed logits of unknown answers moved to the row minimum (:192-194)
        VQA_REQUIRE(P->head2.w != nullptr && P->head2.b != nullptr, VQA_ERR_ARG);
        ProbeScope ps("head.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, B, A, 2 * H, c.f("joint"), (int)(2 * H), P->head.w, (int)A,
                 c.f(dims->model_type == 6 ? "logit_raw" : "logit_fixed"), (int)A, P->head.b));
        if (dims->model_type == 6)
            TRY(vqa_rowmin_mask_fwd(c.f("logit_raw"), bt->exist_mask, c.f("logit_fixed"), c.f("rowmin"), (int)B, (int)A, c.st));
        TRY(gemm(c, 0, 0, B, A, 2 * H, c.f("joint"), (int)(2 * H), P->head2.w, (int)A, c.f("logit_tuned"), (int)A, P->head2.b));
    } else {
        ProbeScope ps("head.fwd_gemm", c.st);
        TRY(gemm(c, 0, 0, B, A, 2 * H, c.f("joint"), (int)(2 * H), P->head.w, (int)A, c.f("logit"), (int)A, P->head.b));
    }
    // a11 (the train loss is masked by the train-answer mask in vlmap_answer and standard_word2vec, not in standard)
    ProbeScope ps("loss.fwd", c.st);
    if (dims->model_type == 4 || dims->model_type == 6)
        TRY(vqa_loss2_fwd(c.f("logit_fixed"), c.f("logit_tuned"), bt->answer_target, bt->train_mask, bt->obj_mask,
                          bt->attr_mask, bt->exist_mask, dims->inv_global_batch, c.f("stats"), c.i32("pred"),
                          want_dz ? c.f("dlogit") : nullptr, want_dz ? c.f("dlogit_tuned") : nullptr, c.f("logit"),
                          dims->model_type == 6 ? 1 : 0, (int)B, (int)A, c.st));
    else
        TRY(vqa_loss_fwd(c.f("logit"), bt->answer_target, bt->train_mask, bt->obj_mask, bt->attr_mask, bt->exist_mask,
                         dims->model_type != 1 ? 1 : 0, dims->inv_global_batch, c.f("stats"), c.i32("pred"),
                         want_dz ? c.f("dlogit") : nullptr, (int)B, (int)A, c.st));
    TRY(vqa_report_reduce(c.f("stats"), (int)B, c.f("report"), c.st));
    if (mt == VQA_MODEL_ENT) {
        // Maximum entropy regularisation (vqa/model_vlmap_answer_ent.py:191-211, 281-292): joint_fc (+ its own dropout)
        // and the head's first ent_cols columns on num_marginal pairings of every question; tile_z ends up holding
        // d loss / d logit of the pairings (want_dz) or their probabilities
        ProbeScope ps("ent.fwd", c.st);
        const int64_t M = dims->num_marginal, C = dims->ent_cols;
        TRY(vqa_tile_mul_fwd(c.f("pooled_linear_l"), c.f("l_linear_l"), c.f("tile_in"), (int)B, (int)M, (int)H, c.st));
        TRY(gemm(c, 0, 0, B * M, 2 * H, H, c.f("tile_in"), (int)H, P->joint_fc.w, (int)(2 * H), c.f("pre_tj"), (int)(2 * H),
                 P->joint_fc.b));
        TRY(vqa_ln_relu_fwd(c.f("pre_tj"), P->joint_fc.gamma, P->joint_fc.beta, bt->keep_tile, dims->keep_joint,
                            c.f("tile_joint"), c.f("mean_tj"), c.f("rstd_tj"), (int)B, (int)M, (int)(2 * H), c.st));
        TRY(gemm(c, 0, 0, B * M, C, 2 * H, c.f("tile_joint"), (int)(2 * H), P->head.w, (int)A, c.f("tile_z"), (int)C, P->head.b));
        TRY(vqa_marginal_entropy(c.f("tile_z"), bt->train_mask, bt->exist_mask, dims->extra_weight * dims->inv_global_batch,
                                 c.f("marginal_prob"), c.f("extra_row"), (int)B, (int)M, (int)C, (int)C, want_dz, c.st));
    }
    if (mt == VQA_MODEL_FULL || mt == VQA_MODEL_ENT)      // latent_loss | entropy, its weighted form and the total loss
        TRY(vqa_extra_report(c.f("extra_row"), c.f("stats"), (int)B, dims->extra_weight, c.f("report"), c.st));
    return VQA_OK;
}

extern "C" int vqa_fusion_backward(const vqa_dims_t* dims, const vqa_params_t* P, const vqa_params_t* G,
                                   const vqa_batch_t* bt, void* workspace, int64_t workspace_bytes,
                                   float* embed_slice_sq, void* stream) {
    return vqa_fusion_backward_phases(dims, P, G, bt, workspace, workspace_bytes, embed_slice_sq, 15, stream);
}

// phases (bit mask), in dependency order:
//   1  head .. attention .. v_linear_v / q_linear_v / score gradients      (complete after this phase)
//   2  GRU back-propagation through time, dx, embedding scatter-add, slice sum of squares
//   4  GRU gate weight / bias gradients (the larger half, 10.8 MB at H 1024)
//   8  GRU candidate weight / bias gradients (5.4 MB) -- the only all-reduce nothing is left to overlap with
// so a data-parallel caller can start all-reducing each gradient bucket while the next phase runs.
extern "C" int vqa_fusion_backward_phases(const vqa_dims_t* dims, const vqa_params_t* P, const vqa_params_t* G,
                                          const vqa_batch_t* bt, void* workspace, int64_t workspace_bytes,
                                          float* embed_slice_sq, int phases, void* stream) {
    VQA_REQUIRE(dims_ok(dims) && P && G && bt && workspace, VQA_ERR_ARG);
    const Layout L = make_layout(*dims);
    Ctx c{*dims, L, static_cast<char*>(workspace), static_cast<hipStream_t>(stream), 0};
    VQA_REQUIRE(workspace_bytes >= c.L.total, VQA_ERR_WORKSPACE);
    ProbeScope ps_all("backward", c.st);
    if (dims->model_type == VQA_MODEL_LEGACY_VQA) return (phases & 1) ? legacy_vqa_backward(c, P, G, bt, embed_slice_sq) : VQA_OK;
    const int64_t B = dims->B, R = dims->R, D = dims->D, H = dims->H, T = dims->T, W = dims->W, A = dims->A;
    const int64_t Wp = ((W + 1 + 3) / 4) * 4;      // row stride of x_tm (make_layout)
    const float* hs = c.f("hs");
    const float* h = hs + T * B * H;
    float* dh = c.f("d_h0");
    float* dxp = c.f("dxp");
    const int mt = dims->model_type;
    const int64_t Dp = pooled_dim(*dims);
    const bool pair = ln_pair_ok(*dims, P);

    if (phases & 1) {
    {
    ProbeScope ps_head("head.bwd_gemm", c.st);
    if (dims->model_type == 2) {
        // word2vec head: d_joint2 = dlogit * G^T (the GloVe matrix is a constant), then the classifier FC
        VQA_REQUIRE(P->answer_glove != nullptr, VQA_ERR_ARG);
        TRY(gemm(c, 0, 1, B, W, A, c.f("dlogit"), (int)A, P->answer_glove, (int)A, c.f("d_joint2"), (int)W));
        if (G->head.w != nullptr) {
            TRY(gemm(c, 1, 0, 2 * H, W, B, c.f("joint"), (int)(2 * H), c.f("d_joint2"), (int)W, G->head.w, (int)W));
            TRY(colsum(c, c.f("d_joint2"), B, W, (int)W, G->head.b));
        }
        TRY(gemm(c, 0, 1, B, 2 * H, W, c.f("d_joint2"), (int)W, P->head.w, (int)W, c.f("d_joint"), (int)(2 * H)));
    } else {
    if (dims->model_type == 6)       // back through the row-minimum substitution, in place (dlogit was taken wrt the masked logits)
        TRY(vqa_rowmin_mask_bwd(c.f("dlogit"), c.f("logit_raw"), c.f("rowmin"), bt->exist_mask, (int)B, (int)A, c.st));
    // head: logit = joint*W + b
    if (G->head.w != nullptr) {
        TRY(gemm(c, 1, 0, 2 * H, A, B, c.f("joint"), (int)(2 * H), c.f("dlogit"), (int)A, G->head.w, (int)A));
        TRY(colsum(c, c.f("dlogit"), B, A, (int)A, G->head.b));
    }
    TRY(gemm(c, 0, 1, B, 2 * H, A, c.f("dlogit"), (int)A, P->head.w, (int)A, c.f("d_joint"), (int)(2 * H)));
    if (dims->model_type == 4 || dims->model_type == 6) {     // the tuned head: its own weights train, and its dz joins d_joint (unmasked term of the loss)
        VQA_REQUIRE(P->head2.w != nullptr, VQA_ERR_ARG);
        if (G->head2.w != nullptr) {
            TRY(gemm(c, 1, 0, 2 * H, A, B, c.f("joint"), (int)(2 * H), c.f("dlogit_tuned"), (int)A, G->head2.w, (int)A));
            TRY(colsum(c, c.f("dlogit_tuned"), B, A, (int)A, G->head2.b));
        }
        TRY(gemm(c, 0, 1, B, 2 * H, A, c.f("dlogit_tuned"), (int)A, P->head2.w, (int)A, c.f("d_joint"), (int)(2 * H), nullptr,
                 c.f("d_joint"), (int)(2 * H)));
    }
    }
    }
    if (dims->model_type == 5) {
        // vlmap_answer_noc: the two branches separately, straight into d_pl / d_ll (no product to differentiate)
        if (G->head2.w != nullptr) {
            TRY(gemm(c, 1, 0, 2 * H, A, B, c.f("l_joint"), (int)(2 * H), c.f("dlogit"), (int)A, G->head2.w, (int)A));
            TRY(colsum(c, c.f("dlogit"), B, A, (int)A, G->head2.b));
        }
        TRY(gemm(c, 0, 1, B, 2 * H, A, c.f("dlogit"), (int)A, P->head2.w, (int)A, c.f("d_ljoint"), (int)(2 * H)));
        TRY(fc_ln_relu_bwd(c, c.f("d_joint"), c.f("pooled_linear_l"), B, H, 2 * H, P->joint_fc, &G->joint_fc, 1, "pre_j",
                           "mean_j", "rstd_j", bt->keep_joint, dims->keep_joint, "d_pre_j", c.f("d_pl"), false));
        TRY(fc_ln_relu_bwd(c, c.f("d_ljoint"), c.f("l_linear_l"), B, H, 2 * H, P->joint2, &G->joint2, 1, "pre_jl", "mean_jl",
                           "rstd_jl", bt->keep_joint2, dims->keep_joint, "d_pre_jl", c.f("d_ll"), false));
    } else {
    // joint_fc (dropout mask folded into the LN/ReLU backward)
    TRY(fc_ln_relu_bwd(c, c.f("d_joint"), c.f("joint_in"), B, H, 2 * H, P->joint_fc, &G->joint_fc, 1, "pre_j", "mean_j",
                       "rstd_j", bt->keep_joint, dims->keep_joint, "d_pre_j", c.f("d_joint_in"), false));
    if (!pair) {
        ProbeScope ps("eltwise", c.st);
        TRY(vqa_mul_bwd(c.f("d_joint_in"), c.f("pooled_linear_l"), c.f("l_linear_l"), c.f("d_pl"), c.f("d_ll"), B * H,
                        c.st));
    }
    }
    if (mt == VQA_MODEL_ENT) {
        // the regulariser's path back to l_linear_l (pooled_linear_l is behind tf.stop_gradient, :197): tile_z holds
        // d loss / d logit; head and joint_fc are frozen (filter_train_vars :86-94), so only dX products run
        ProbeScope ps("ent.bwd", c.st);
        const int64_t M = dims->num_marginal, C = dims->ent_cols;
        TRY(gemm(c, 0, 1, B * M, 2 * H, C, c.f("tile_z"), (int)C, P->head.w, (int)A, c.f("d_tile_joint"), (int)(2 * H)));
        TRY(vqa_ln_relu_bwd(c.f("d_tile_joint"), c.f("pre_tj"), c.f("mean_tj"), c.f("rstd_tj"), P->joint_fc.gamma,
                            P->joint_fc.beta, bt->keep_tile, dims->keep_joint, c.f("d_pre_tj"), nullptr, nullptr, nullptr,
                            (int)B, (int)M, (int)(2 * H), c.st));
        TRY(gemm(c, 0, 1, B * M, H, 2 * H, c.f("d_pre_tj"), (int)(2 * H), P->joint_fc.w, (int)(2 * H), c.f("d_tile_in"), (int)H));
        // (paired LayerNorm backward: d_ll holds only this extra gradient and is added inside that kernel)
        TRY(vqa_tile_mul_bwd(c.f("d_tile_in"), c.f("pooled_linear_l"), c.f("d_ll"), (int)B, (int)M, (int)H, pair ? 0 : 1, c.st));
    }
    // pooled_linear_l and q_linear_l: LayerNorm backward (paired with the product's, or each on its own), then the FC halves
    const bool tr_pl = G->pooled_linear_l.w != nullptr, tr_ll = G->q_linear_l.w != nullptr;
    if (pair) {
        ProbeScope ps("fc.ln_bwd", c.st);
        TRY(vqa_ln_pair_mul_bwd(c.f("d_joint_in"), mt == VQA_MODEL_ENT ? c.f("d_ll") : nullptr, c.f("pre_pl"), c.f("pre_ll"),
                                c.f("mean_pl"), c.f("rstd_pl"), c.f("mean_ll"), c.f("rstd_ll"), P->pooled_linear_l.gamma,
                                P->pooled_linear_l.beta, P->q_linear_l.gamma, P->q_linear_l.beta, c.f("d_pre_pl"), c.f("d_pre_ll"),
                                tr_pl ? c.part(0) : nullptr, tr_pl ? c.part(1) : nullptr, tr_pl ? c.part(2) : nullptr,
                                tr_ll ? c.f("part_a1") : nullptr, tr_ll ? c.f("part_b1") : nullptr, tr_ll ? c.f("part_c1") : nullptr,
                                (int)B, (int)H, c.st));
        TRY(fc_bwd_tail(c, c.f("pooled_V_ft"), B, Dp, H, P->pooled_linear_l, &G->pooled_linear_l, 1, c.f("d_pre_pl"), c.part(0),
                        c.part(1), c.part(2), c.f("d_pooled"), false));
    } else {
    TRY(fc_ln_relu_bwd(c, c.f("d_pl"), c.f("pooled_V_ft"), B, Dp, H, P->pooled_linear_l, &G->pooled_linear_l, 1, "pre_pl",
                       "mean_pl", "rstd_pl", nullptr, 1.f, "d_pre_pl", c.f("d_pooled"), false));
    }
    // q_linear_l's FC half (after its LayerNorm backward, unless the paired kernel did it): x = what the layer read
    auto ll_bwd = [&](const float* x, float* dx) -> int {
        if (pair)
            return fc_bwd_tail(c, x, B, H, H, P->q_linear_l, &G->q_linear_l, 1, c.f("d_pre_ll"), c.f("part_a1"), c.f("part_b1"),
                               c.f("part_c1"), dx, false);
        return fc_ln_relu_bwd(c, c.f("d_ll"), x, B, H, H, P->q_linear_l, &G->q_linear_l, 1, "pre_ll", "mean_ll", "rstd_ll", nullptr,
                              1.f, "d_pre_ll", dx, false);
    };
    if (mt == VQA_MODEL_ANSWER2) {
        // q_linear_l read q_L_ft2: back through it, then through tanh + LN + FC (trainable) into dh
        TRY(ll_bwd(c.f("q_L_ft2"), c.f("d_ft2")));
        const bool train = G->q_L_ft2.w != nullptr;
        TRY(vqa_ln_act_bwd(c.f("d_ft2"), c.f("pre_ft2"), c.f("mean_ft2"), c.f("rstd_ft2"), P->q_L_ft2.gamma, P->q_L_ft2.beta,
                           nullptr, 1.f, c.f("d_pre_ft2"), train ? c.part(0) : nullptr, train ? c.part(1) : nullptr,
                           train ? c.part(2) : nullptr, (int)B, 1, (int)H, 1, c.st));
        if (train) {
            TRY(vqa_colsum3(c.part(0), c.part(1), c.part(2), (int)B, (int)H, (int)H, G->q_L_ft2.gamma, G->q_L_ft2.beta,
                            G->q_L_ft2.b, c.colsum_ws(), c.colsum_ws_floats(), c.st));
            TRY(gemm(c, 1, 0, H, H, B, h, (int)H, c.f("d_pre_ft2"), (int)H, G->q_L_ft2.w, (int)H));
        }
        TRY(gemm(c, 0, 1, B, H, H, c.f("d_pre_ft2"), (int)H, P->q_L_ft2.w, (int)H, dh, (int)H));
    } else if (mt == VQA_MODEL_NO_NOISE || mt == VQA_MODEL_FULL) {
        // q_linear_l read q_L_mean (+ noise * sigma): linear layers on the GRU state
        const bool full = mt == VQA_MODEL_FULL;
        float* d_in = c.f(full ? "d_lin" : "d_qm");
        TRY(ll_bwd(c.f(full ? "q_L_mean_noise" : "q_L_mean"), d_in));
        if (full)      // through x = mean + noise * sigma, plus the KL term's own gradient (weight / global batch)
            TRY(vqa_reparam_bwd(d_in, c.f("q_L_mean"), c.f("q_L_log_sigma_sq"), bt->noise,
                                dims->extra_weight * dims->inv_global_batch, c.f("d_qm"), c.f("d_qs"), B * H, c.st));
        if (G->q_L_mean.w != nullptr) {
            TRY(gemm(c, 1, 0, H, H, B, h, (int)H, c.f("d_qm"), (int)H, G->q_L_mean.w, (int)H));
            TRY(colsum(c, c.f("d_qm"), B, H, (int)H, G->q_L_mean.b));
        }
        TRY(gemm(c, 0, 1, B, H, H, c.f("d_qm"), (int)H, P->q_L_mean.w, (int)H, dh, (int)H));
        if (full) {
            if (G->q_L_log_sigma_sq.w != nullptr) {
                TRY(gemm(c, 1, 0, H, H, B, h, (int)H, c.f("d_qs"), (int)H, G->q_L_log_sigma_sq.w, (int)H));
                TRY(colsum(c, c.f("d_qs"), B, H, (int)H, G->q_L_log_sigma_sq.b));
            }
            TRY(gemm(c, 0, 1, B, H, H, c.f("d_qs"), (int)H, P->q_L_log_sigma_sq.w, (int)H, dh, (int)H, nullptr, dh, (int)H));
        }
    } else {
    TRY(ll_bwd(mt == VQA_MODEL_BI ? c.f("q_L_ft") : h, dh));
    }
    // attention + pooling
    {
    ProbeScope ps("attn_pool.bwd", c.st);
    TRY(vqa_attn_pool_bwd(c.f("d_pooled"), c.f("v_linear_v"), c.f("q_linear_v"), c.f(mt == VQA_MODEL_ADAPT ? "v_adapt" : "V_ft"),
                          c.f("att_score"), P->score.w, bt->keep_att, dims->keep_att, c.f("d_v"), c.f("d_qv"), c.f("part_dw"),
                          c.f("part_db"), (int)B, (int)R, (int)H, (int)Dp, c.st));
    }
    if (mt == VQA_MODEL_ADAPT) {
        // the pooled memory is trainable here: d v_adapt = att (x) d pooled, then LN[R,H] + ReLU + FC backward (V_ft is an
        // input: parameters only), like v_linear_v's
        TRY(vqa_outer_rows(c.f("att_score"), c.f("d_pooled"), c.f("d_va"), (int)B, (int)R, (int)H, c.st));
        TRY(fc_ln_relu_bwd(c, c.f("d_va"), c.f("V_ft"), B * R, D, H, P->v_adapt, &G->v_adapt, (int)R, "pre_va", "mean_va",
                           "rstd_va", nullptr, 1.f, "d_pre_va", nullptr, false));
    }
    if (G->score.w != nullptr) {
        ProbeScope ps("attn_pool.bwd", c.st);
        TRY(colsum(c, c.f("part_dw"), B, H, (int)H, G->score.w));
        TRY(colsum(c, c.f("part_db"), B, 1, 1, G->score.b));
    }
    // v_linear_v: parameters only (V_ft is an input).  77 GFLOP of dW on the side stream, beside
    // the latency-bound back-propagation through time below.
    Side& sd = side_stream();
    const bool forked = fork_side(c, sd);
    {
        Ctx cv{*dims, L, c.ws, forked ? sd.s : c.st, forked ? 1 : 0};
        TRY(fc_ln_relu_bwd(cv, cv.f("d_v"), cv.f("V_ft"), B * R, D, H, P->v_linear_v, &G->v_linear_v, (int)R, "pre_v",
                           "mean_v", "rstd_v", nullptr, 1.f, "d_pre_v", nullptr, false));
        if (forked && !join_side_record(sd)) return VQA_ERR_LAUNCH;
    }
    if (mt == VQA_MODEL_BI) {
        // q_linear_v read pooled_q_v; from there back through the question self-attention into d q_L_ft (dh) / d q_L_map
        TRY(fc_ln_relu_bwd(c, c.f("d_qv"), c.f("pooled_q_v"), B, H, H, P->q_linear_v, &G->q_linear_v, 1, "pre_qv", "mean_qv",
                           "rstd_qv", nullptr, 1.f, "d_pre_qv", c.f("d_pooled_qv"), false));
        TRY(bi_question_bwd_head(c, P, G, bt, dh));
    } else {
    // q_linear_v: dh += ...
    TRY(fc_ln_relu_bwd(c, c.f("d_qv"), h, B, H, H, P->q_linear_v, &G->q_linear_v, 1, "pre_qv", "mean_qv", "rstd_qv",
                       nullptr, 1.f, "d_pre_qv", dh, true));
    }
    if (forked && hipStreamWaitEvent(c.st, sd.join, 0) != hipSuccess) return VQA_ERR_LAUNCH;
    }   // phase 1

    if (mt == VQA_MODEL_BI) {
        if (phases & 2) TRY(bi_question_bwd_bptt(c, P, G, bt, dh, embed_slice_sq));
        if (phases & 12) TRY(bi_question_bwd_weights(c, G, phases));
        return VQA_OK;
    }
    if (phases & 2) {
    // GRU back-propagation through time (gate math fused into the GEMM epilogues)
    const float* Wg_h = P->gru_wg + W * 2 * H;
    const float* Wc_h = P->gru_wc + W * H;
    {
        ProbeScope ps("gru.bwd", c.st);
        const bool ws_ok = gru_ws_on() && (bt->live_rows == nullptr || (T > 0 && bt->live_rows[T - 1] > 0));
        if (ws_ok && vqa_gru_ws_bwd_supported((int)T, (int)B, (int)H) == 1) {
            TRY(vqa_gru_seq_bwd_ws(dh, nullptr, Wg_h, Wc_h, bt->q_intseq_len, hs, c.f("gru_r"), c.f("gru_u"), c.f("gru_c"), dxp,
                                   (int)T, (int)B, (int)H, c.f("gru_ws"), c.st));
        } else if (bt->live_rows != nullptr) {
            TRY(vqa_gru_seq_bwd_live(dh, Wg_h, Wc_h, bt->q_intseq_len, bt->live_rows, hs, c.f("gru_r"), c.f("gru_u"),
                                     c.f("gru_c"), dxp, c.f("d_h1"), (int)T, (int)B, (int)H, c.st));
        } else {
            const float* gr = c.f("gru_r"); const float* gu = c.f("gru_u"); const float* gc = c.f("gru_c");
            float* dh1 = c.f("d_h1");
            TRY(run_chains(c, B, [&](int, int64_t row0, int64_t rows, hipStream_t st) {
                return vqa_gru_seq_bwd_rows(dh, Wg_h, Wc_h, bt->q_intseq_len, hs, gr, gu, gc, dxp, dh1, (int)T, (int)B, (int)H,
                                            (int)row0, (int)rows, st);
            }));
        }
    }
    // embedding: un-aggregated slices dx [T,B,W], then scatter-add
    float* dx = c.f("dx_embed");
    {
    ProbeScope ps("gru.dx_gemm", c.st);
    if (xcat_enabled()) {
        // wx_cat: the forward's packed copy of the x rows still sits in the workspace (backward follows the forward of the
        // same step on the same weights: every gradient here is meaningless otherwise); VQA_HOT_REPACK=1 packs it again
        static const bool repack = [] { const char* e = getenv("VQA_HOT_REPACK"); return e != nullptr && atoi(e) != 0; }();
        if (repack)
            TRY(vqa_gru_pack_wx(P->gru_wg, P->gru_wc, P->gru_bg, P->gru_bc, c.f("wx_cat"), c.f("bx_cat"), (int)W, (int)H, c.st));
        TRY(gemm(c, 0, 1, T * B, W, 3 * H, dxp, (int)(3 * H), c.f("wx_cat"), (int)(3 * H), dx, (int)W));
    } else {
        TRY(gemm(c, 0, 1, T * B, W, 2 * H, dxp, (int)(3 * H), P->gru_wg, (int)(2 * H), dx, (int)W));
        TRY(gemm(c, 0, 1, T * B, W, H, dxp + 2 * H, (int)(3 * H), P->gru_wc, (int)H, dx, (int)W, nullptr, dx, (int)W));
    }
    }
    ProbeScope ps_embed("embed.bwd", c.st);
    if (G->embed != nullptr)
        TRY(vqa_embed_bwd_len_det(dx, bt->q_intseq, bt->q_intseq_len, G->embed, (int)B, (int)T, (int)W, dims->Vq,
                                  (dims->flags & VQA_FLAG_DETERMINISTIC) ? 1 : 0, c.st));
    if (embed_slice_sq != nullptr)
        TRY(vqa_sumsq(dx, T * B * W, nullptr, embed_slice_sq, c.f("sumsq_ws"), c.L.find("sumsq_ws")->n, c.st));
    }   // phase 2

    if ((phases & 4) && G->gru_wg != nullptr) {
        {
        ProbeScope ps("gru.dwx_gemm", c.st);
        if (xcat_enabled()) {   // x rows of BOTH kernels' gradients (the candidate's bucket is reduced after phase 4)
            // rows 0..W-1: x rows of both kernels' gradients; row W (the constant input): both bias gradients
            TRY(gemm(c, 1, 0, Wp, 3 * H, T * B, c.f("x_tm"), (int)Wp, dxp, (int)(3 * H), c.f("dwx_cat"), (int)(3 * H)));
            TRY(vqa_gru_unpack_dwx_bias(c.f("dwx_cat"), G->gru_wg, G->gru_wc, G->gru_bg, G->gru_bc, (int)W, (int)H, c.st));
        } else {
            TRY(gemm(c, 1, 0, W, 2 * H, T * B, c.f("x_tm"), (int)Wp, dxp, (int)(3 * H), G->gru_wg, (int)(2 * H)));
            TRY(colsum(c, dxp, T * B, 2 * H, (int)(3 * H), G->gru_bg));
        }
        }
        ProbeScope ps("gru.dwh_gemm", c.st);
        TRY(gemm(c, 1, 0, H, 2 * H, T * B, hs, (int)H, dxp, (int)(3 * H), G->gru_wg + W * 2 * H, (int)(2 * H)));
    }   // phase 3: gates
    if ((phases & 8) && G->gru_wg != nullptr) {
        if (!xcat_enabled()) {
            ProbeScope ps("gru.dwx_gemm", c.st);
            TRY(gemm(c, 1, 0, W, H, T * B, c.f("x_tm"), (int)Wp, dxp + 2 * H, (int)(3 * H), G->gru_wc, (int)H));
            TRY(colsum(c, dxp + 2 * H, T * B, H, (int)(3 * H), G->gru_bc));
        }
        ProbeScope ps("gru.dwh_gemm", c.st);
        TRY(gemm(c, 1, 0, H, H, T * B, c.f("gru_rh"), (int)H, dxp + 2 * H, (int)(3 * H), G->gru_wc + W * H, (int)H));
    }   // phase 4: candidate
    return VQA_OK;
}

extern "C" int vqa_hot_version(void) { return VQA_HOT_ABI_VERSION; }

extern "C" const char* vqa_hot_error_string(int code) {
    switch (code) {
        case VQA_OK: return "ok";
        case VQA_ERR_ARG: return "bad argument (size or null pointer)";
        case VQA_ERR_ALIGN: return "pointer / leading dimension not aligned";
        case VQA_ERR_LAUNCH: return "kernel launch failed";
        case VQA_ERR_UNSUPPORTED: return "unsupported configuration";
        case VQA_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
