/* libvqahot.so -- C ABI of the MI355X-native VQA hot path.
 *
 * The reference (HyeonwooNoh/VQA-Transfer-ExternalData) has NO FFI/plugin
 * boundary: its hot path is a Python class contract inside one TF-1.6 graph
 * (SURVEY.md 8b).  Each entry point below therefore cites the reference graph
 * code whose stock TF ops it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer
 *     unless the name ends in _host.  No torch types.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued and the call
 *     returns immediately.  No allocation, no synchronisation inside.
 *   - Return 0 on success, a negative VQA_ERR_* otherwise; never throws.
 *   - Row-major tensors exactly as in the reference (NHWC for images).
 *   - fp32 everywhere the reference is fp32; matrix products run on the exact
 *     f32-input MFMA (v_mfma_f32_32x32x2_f32).
 */
#ifndef VQA_HOT_H
#define VQA_HOT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQA_HOT_ABI_VERSION 5

enum {
    VQA_OK = 0,
    VQA_ERR_ARG = -1,     /* bad size / null pointer */
    VQA_ERR_ALIGN = -2,   /* pointer or leading dimension not 16-byte aligned where required */
    VQA_ERR_LAUNCH = -3,  /* hipGetLastError() after a launch */
    VQA_ERR_UNSUPPORTED = -4,
    VQA_ERR_WORKSPACE = -5 /* workspace too small */
};

int vqa_hot_version(void);
const char* vqa_hot_error_string(int code);

/* ---------------------------------------------------------------- a1 / K1
 * V_ft = np.take(features, image_idx); num_V_ft = gather(num_boxes, image_idx)
 * vqa/model_vlmap_answer.py:110-123 (tf.py_func on /cpu:0 + H2D copy).
 * table [N,R,D] f32 resident in HBM, idx i64[B] -> V [B,R,D], nb i32[B].  V == NULL gathers num_boxes only
 * (the whole-model forward fuses the feature rows into v_linear_v's GEMM, vqa_gemm_f32_gather). */
int vqa_gather_features(const float* table, const int32_t* nbox_table, const int64_t* idx,
                        float* V, int32_t* nb, int B, int R, int D, int64_t N, void* stream);

/* ---------------------------------------------------------------- a3 / K3
 * tf.nn.embedding_lookup(glove_map, q_intseq)  vqa/model_vlmap_answer.py:134.
 * q i32[B,T] (batch-major, zero padded) -> x [T,B,W] TIME-major (the GRU
 * consumes one contiguous [B,W] slab per step). */
int vqa_embed_fwd(const float* E, const int32_t* q, float* x_tm, int B, int T, int W, int Vq, void* stream);
/* The x rows (first W) of the two GRU kernels -- gates [W+H, 2H] and candidate [W+H, H] of tf.contrib.rnn.GRUCell
 * (vlmap/modules.py:124-140) -- side by side as one [W, 3H] matrix, the two biases as one [3H] vector: the input
 * projection of all time steps, its gradient and the x-part weight gradient are then one GEMM each.
 * vqa_gru_unpack_dwx writes the gradient of the packed matrix into the x rows of the two kernels' gradients. */
int vqa_gru_pack_wx(const float* wg, const float* wc, const float* bg, const float* bc, float* wx, float* bx, int W, int H,
                    void* stream);
int vqa_gru_unpack_dwx(const float* dwx, float* gwg, float* gwc, int W, int H, void* stream);
/* vqa_embed_fwd with a row stride ldx >= W: columns W .. ldx-1 of every row are 1, 0, 0, ... -- the constant input whose
 * weight row is the bias.  The x-part weight-gradient GEMM over ldx rows (x_tm^T [ldx, T*B] x dxp [T*B, 3H]) then has
 * the bias gradients (the column sums of dxp) in its row W: vqa_gru_unpack_dwx_bias writes that row into the gradients
 * of the two biases, the rows above it into the x rows of the two kernels' gradients. */
int vqa_embed_fwd_ld(const float* E, const int32_t* q, float* x_tm, int B, int T, int W, int Vq, int ldx, void* stream);
int vqa_gru_unpack_dwx_bias(const float* dwx, float* gwg, float* gwc, float* gbg, float* gbc, int W, int H, void* stream);
/* backward: dE[q[b,t],:] += dx[t,b,:] (dE must be zeroed by the caller);
 * the IndexedSlices gradient of the gather.  Float atomics by default (order of the adds, hence the last bit,
 * varies run to run -- as in the reference); after vqa_set_deterministic(1) an atomic-free, run-to-run bitwise
 * reproducible form is used for W <= 512 (about 30 us slower at bs 512). */
int vqa_set_deterministic(int on);   /* process-wide; every other kernel is deterministic already */
int vqa_embed_bwd(const float* dx_tm, const int32_t* q, float* dE, int B, int T, int W, int Vq, void* stream);
/* Same, skipping the zero-padded positions t >= len[b]: dynamic_rnn(sequence_length) makes their dx exactly
 * zero (vlmap/modules.py:124-140), so the result is identical and the padding id is not a hot row. */
int vqa_embed_bwd_len(const float* dx_tm, const int32_t* q, const int32_t* len, float* dE, int B, int T, int W,
                      int Vq, void* stream);
/* Same with the summation form chosen per call instead of by the process-wide default: deterministic = 1 atomic-free
 * and run-to-run bitwise reproducible, 0 float atomics, -1 whatever vqa_set_deterministic() last selected.  The
 * whole-model entry points use this with (vqa_dims_t.flags & VQA_FLAG_DETERMINISTIC), so two engines in one process
 * never share the setting. */
int vqa_embed_bwd_len_det(const float* dx_tm, const int32_t* q, const int32_t* len, float* dE, int B, int T, int W,
                          int Vq, int deterministic, void* stream);

/* ------------------------------------------------- GEMM (layers.fully_connected)
 * C[M,N] = op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ D[M,N]),  f32 MFMA.
 *   transA = 0: A is [M,K] row-major (lda >= K);  1: A is [K,M] row-major (lda >= M)
 *   transB = 0: B is [K,N] row-major (ldb >= N);  1: B is [N,K] row-major (ldb >= K)
 * Supported: (0,0) forward FC, (0,1) dX = dY*W^T, (1,0) dW = X^T*dY.
 * bias, D may be NULL; D may alias C.  split_k > 1 needs `workspace` of
 * split_k*M*N floats (deterministic slab reduce); split_k = 0 lets the library
 * choose.  Replaces tf.contrib.layers.fully_connected / its autodiff
 * (vlmap/modules.py:635-641) and the GRUCell matmuls (vlmap/modules.py:129-135). */
int vqa_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                 int ldb, float* C, int ldc, const float* bias, const float* D, int ldd, int split_k,
                 float* workspace, int64_t workspace_floats, void* stream);
/* Same, launched with at most max_blocks workgroups that walk the tiles persistently
 * (0 = one workgroup per tile).  Used for the big GEMMs on the side stream: one
 * workgroup per CU leaves LDS and wave slots for the latency-bound GRU recurrence. */
int vqa_gemm_f32_ex(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                    int ldb, float* C, int ldc, const float* bias, const float* D, int ldd, int split_k,
                    float* workspace, int64_t workspace_floats, int max_blocks, void* stream);
/* The same product with the feature gather fused into the left operand's load ("or fused into K2
 * A-operand load"): row m of the left operand is row idx[m / R] * R + m % R of `table` [n_samples_in_table * R, K]
 * (ld = lda), i.e. C = features[image_idx].reshape(B*R, K) * B + bias without a gather pass
 * (vqa/model_vlmap_answer.py:110-117 + 126-129).  gathered_out (may be NULL; ld = ldg) receives the gathered rows
 * [M, K] as a by-product for the consumers that re-read V_ft (attention pooling, its backward, dW of v_linear_v).
 * Out-of-range indices are clamped like vqa_gather_features.  Needs K % 32 == 0, N % 4 == 0 and 16-byte aligned
 * operands; the table itself may be larger than 4 GiB. */
int vqa_gemm_f32_gather(int M, int N, int K, const float* table, int lda, const int64_t* idx, int R,
                        int64_t n_samples_in_table, const float* B, int ldb, float* C, int ldc, const float* bias,
                        float* gathered_out, int ldg, void* stream);
int vqa_gemm_set_max_blocks(int n);
int vqa_gemm_set_order(int order);   /* tuning: 0 n-fastest, 1 m-fastest, -1 automatic */
int64_t vqa_gemm_workspace_floats(int transA, int transB, int M, int N, int K, int split_k);
/* tuning hooks: force tile configuration `cfg` (0..23) for every later GEMM, -1 = automatic;
 * tile configuration (4, 7..11, 13, 16..18) of the fused GRU-step GEMMs, -1 = defaults */
int vqa_gemm_set_config(int cfg);
int vqa_gemm_set_gru_config(int cfg);
/* tile shape of the tall-activation GEMMs (M >= 2048, N >= 512, K >= 2048): 20 = 128x64 (default), 21 = 64x128 */
int vqa_gemm_set_tall_config(int cfg);

/* ------------------------------------------- a2,a5,a8,a9 : LN + ReLU (+dropout)
 * y = relu(layer_norm(pre)) [* keepmask / keep]  with statistics over groups of
 * `rows` consecutive rows (rows = R for v_linear_v: LN over all 36x1024 values
 * of a sample; rows = 1 otherwise), gamma/beta on the last axis, eps 1e-12.
 * vlmap/modules.py:647-650 (layers.layer_norm + relu), vqa/model_vlmap_answer.py:180
 * (tf.nn.dropout 0.5).  pre,y [G*rows,N]; mean,rstd [G]; keepmask u8 [G*rows,N] or NULL. */
/* tuning / A-B switch: 1 (default) = register-resident kernels for groups of 5..36 rows x 1024 columns (v_linear_v's
 * 36 x 1024 block per sample), 0 = generic kernels */
int vqa_ln_set_fast(int on);
int vqa_ln_relu_fwd(const float* pre, const float* gamma, const float* beta, const uint8_t* keepmask,
                    float keep_prob, float* y, float* mean, float* rstd, int G, int rows, int N, void* stream);
/* backward.  dy [G*rows,N] -> dpre; per-group partial sums of d(gamma), d(beta),
 * d(bias) are written to part_* [G,N] when non-NULL (reduce with vqa_colsum). */
int vqa_ln_relu_bwd(const float* dy, const float* pre, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const uint8_t* keepmask, float keep_prob,
                    float* dpre, float* part_dgamma, float* part_dbeta, float* part_dbias, int G, int rows,
                    int N, void* stream);
/* The same with an activation selector: act 0 = ReLU, 1 = tanh (fc_layer(activation_fn=tf.tanh) of the
 * pre-training model's 'wordset_ft', vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:376-378). */
int vqa_ln_act_fwd(const float* pre, const float* gamma, const float* beta, const uint8_t* keepmask, float keep_prob,
                   float* y, float* mean, float* rstd, int G, int rows, int N, int act, void* stream);
int vqa_ln_act_bwd(const float* dy, const float* pre, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, const uint8_t* keepmask, float keep_prob, float* dpre, float* part_dgamma,
                   float* part_dbeta, float* part_dbias, int G, int rows, int N, int act, void* stream);
/* pooled_linear_l and q_linear_l meet in a product (joint_fc's input, vqa/model_vlmap_answer.py:163-177): ONE launch does
 * LayerNorm + ReLU of both [G,N] pre-activations (one row per group) and z = y_a * y_b; the backward takes dz (+ add_b, an
 * optional extra gradient wrt y_b, or NULL) to both pre-activations and, when asked, the per-row partials of
 * d(gamma), d(beta), d(bias) [G,N] (both or neither of dgamma / dbeta per tensor).  N % 4 == 0, N <= 4096, 16-byte
 * aligned pointers (vqa_ln_pair_mul_supported). */
int vqa_ln_pair_mul_supported(int N, const void* const* ptrs, int n_ptrs);
int vqa_ln_pair_mul_fwd(const float* pre_a, const float* pre_b, const float* gamma_a, const float* beta_a, const float* gamma_b,
                        const float* beta_b, float* y_a, float* y_b, float* z, float* mean_a, float* rstd_a, float* mean_b,
                        float* rstd_b, int G, int N, void* stream);
int vqa_ln_pair_mul_bwd(const float* dz, const float* add_b, const float* pre_a, const float* pre_b, const float* mean_a,
                        const float* rstd_a, const float* mean_b, const float* rstd_b, const float* gamma_a, const float* beta_a,
                        const float* gamma_b, const float* beta_b, float* dpre_a, float* dpre_b, float* part_dgamma_a,
                        float* part_dbeta_a, float* part_dbias_a, float* part_dgamma_b, float* part_dbeta_b, float* part_dbias_b,
                        int G, int N, void* stream);
/* y = tanh(x) ; dx = dy * (1 - y^2)   (tanh of the word-set embedding, same file :373-375) */
int vqa_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int vqa_tanh_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* out[N] = sum_m X[m,:]  (deterministic two-stage; workspace >= vqa_colsum_workspace_floats). */
int vqa_colsum(const float* X, int M, int N, int ldx, float* out, float* workspace, int64_t workspace_floats,
               void* stream);
int64_t vqa_colsum_workspace_floats(int M, int N);
/* Three equally shaped reductions in one pair of launches (the d_gamma / d_beta / d_bias partials of one
 * fc_layer, vlmap/modules.py:630-650); workspace >= 3 * vqa_colsum_workspace_floats(M, N). */
int vqa_colsum3(const float* X0, const float* X1, const float* X2, int M, int N, int ldx, float* out0, float* out1,
                float* out2, float* workspace, int64_t workspace_floats, void* stream);
/* the same with accumulation: out (+)= column sums.  vqa_colsum_acc: accumulate != 0 adds; vqa_colsum3_acc: bit i of
 * acc_mask adds into out_i.  (Gradients of variables shared by several call sites: the cfg-5 model's heads.) */
int vqa_colsum_acc(const float* X, int M, int N, int ldx, float* out, int accumulate, float* workspace,
                   int64_t workspace_floats, void* stream);
int vqa_colsum3_acc(const float* X0, const float* X1, const float* X2, int M, int N, int ldx, float* out0, float* out1,
                    float* out2, int acc_mask, float* workspace, int64_t workspace_floats, void* stream);
/* z = a * b elementwise (pooled_linear_l * l_linear_l, vqa/model_vlmap_answer.py:177) */
int vqa_mul(const float* a, const float* b, float* z, int64_t n, void* stream);
/* da = dz*b ; db = dz*a */
int vqa_mul_bwd(const float* dz, const float* a, const float* b, float* da, float* db, int64_t n, void* stream);
int vqa_add_inplace(float* acc, const float* x, int64_t n, void* stream);

/* ---------------------------------------------------------------- a4 / K4
 * tf.contrib.rnn.GRUCell under tf.nn.dynamic_rnn(sequence_length)
 * (vlmap/modules.py:124-140).  The matmuls go through vqa_gemm_f32; these are
 * the fused gate kernels of one time step.
 *  gates:  r,u = sigmoid(gpre[:, :H]), sigmoid(gpre[:, H:]);  rh = r*h_prev
 *  cand :  c = tanh(cpre); h_new = (t < len) ? u*h_prev + (1-u)*c : h_prev   */
int vqa_gru_gates_fwd(const float* gpre, int ldg, const float* h_prev, float* r, float* u, float* rh, int B,
                      int H, void* stream);
int vqa_gru_cand_fwd(const float* cpre, int ldc, const float* u, const float* h_prev, const int32_t* len, int t,
                     float* c, float* h_new, int B, int H, void* stream);
/* backward of one step, part 1: from dh (grad wrt h_t):
 *  dc_pre = live ? dh*(1-u)*(1-c^2) : 0 ; du_pre = live ? dh*(h_prev-c)*u*(1-u) : 0
 *  dh_acc = live ? dh*u : dh                                                      */
int vqa_gru_bwd_a(const float* dh, const float* h_prev, const float* u, const float* c, const int32_t* len,
                  int t, float* dc_pre, int ld_dc, float* du_pre, int ld_du, float* dh_acc, int B, int H,
                  void* stream);
/* part 2: from drh = dc_pre * Wc_h^T:  dr_pre = drh*h_prev*r*(1-r); dh_acc += drh*r */
int vqa_gru_bwd_b(const float* drh, const float* h_prev, const float* r, float* dr_pre, int ld_dr,
                  float* dh_acc, int B, int H, void* stream);

/* Whole recurrence with the gate math fused into the GEMM epilogues (2 launches per
 * step).  xp [T,B,3H] holds x_t*W_x + b for (r|u|c) and is read only; hs [T+1,B,H]
 * with hs[0] = initial state (zeros); tape r,u,c,rh [T,B,H]. */
int vqa_gru_seq_fwd(float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs, float* r,
                    float* u, float* c, float* rh, int T, int B, int H, void* stream);
/* BPTT.  dh_T [B,H] = gradient wrt hs[T] (used as scratch afterwards); dxp [T,B,3H]
 * receives (dr_pre | du_pre | dc_pre); dh_scratch [B,H]. */
int vqa_gru_seq_bwd(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len, const float* hs,
                    const float* r, const float* u, const float* c, float* dxp, float* dh_scratch, int T, int B,
                    int H, void* stream);

/* The same over the live prefix only: rows sorted by length (longest first), live_rows = HOST int[T] with
 * live_rows[t] = #rows with len > t.  Step t runs on rows [0, live_rows[t]); finished rows are filled in
 * afterwards (state carried, r*h and pre-activation gradients zero) exactly as the masked recurrence leaves them
 * (tf.nn.dynamic_rnn(sequence_length=...), vlmap/modules.py:124-140).  Results are identical; the work shrinks
 * with the sequences still running. */
int vqa_gru_seq_fwd_live(float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, const int32_t* live_rows,
                         float* hs, float* r, float* u, float* c, float* rh, int T, int B, int H, void* stream);
int vqa_gru_seq_bwd_live(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len,
                         const int32_t* live_rows, const float* hs, const float* r, const float* u, const float* c,
                         float* dxp, float* dh_scratch, int T, int B, int H, void* stream);
/* helpers of the above: hs[t+1,b] = hs[len[b],b], rh[t,b] = 0 and dxp[t,b] = 0 for every t >= len[b] */
int vqa_gru_fill_finished(float* hs, float* rh, const int32_t* len, int T, int B, int H, void* stream);
int vqa_gru_zero_finished(float* dxp, const int32_t* len, int T, int B, int H, void* stream);

/* The whole forward recurrence in ONE persistent launch (csrc/gru_persistent.hip): the two halves of the batch run as
 * two independent chains on two co-resident workgroups per CU, a per-chain grid barrier replaces each kernel boundary
 * and hides behind the other chain's matrix work.  Same tensors as vqa_gru_seq_fwd plus `sync`, a device scratch of
 * vqa_gru_persistent_sync_bytes() that the call zeroes itself; after the stream has run, a non-zero 32-bit word at
 * byte offset 128 of `sync` reports a barrier time-out (results invalid).  VQA_ERR_UNSUPPORTED when the shape or the
 * device does not qualify (vqa_gru_fwd_persistent_supported: H % 512 == 0, B >= 64, and -- asked per device, for the
 * variant and LDS size that would be launched -- every workgroup of the grid co-resident).  An opt-in experiment
 * (DESIGN section 4): no model path uses it; a caller must check the error word after synchronising. */
int vqa_gru_seq_fwd_persistent(const float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs,
                               float* r, float* u, float* c, float* rh, int T, int B, int H, unsigned* sync,
                               void* stream);
int vqa_gru_fwd_persistent_supported(int T, int B, int H);
int64_t vqa_gru_persistent_sync_bytes(void);
int vqa_gru_set_persistent(int mode);   /* -1 automatic, 0 never, 1 whenever supported */
int vqa_gru_persistent_set_census(unsigned* dev_words);   /* placement study (tools/gru_tune.py); NULL = off */

/* Weight-stationary persistent recurrence (csrc/gru_ws.hip): the same tape as vqa_gru_seq_fwd in ONE launch in which
 * the recurrent weights are loaded once -- the gate slabs into the registers of the CU that owns 32 state columns, the
 * candidate slab into its LDS -- and only the state moves per step (eight XCD-local chains of 64 rows, each two 32-row
 * half-chains in anti-phase on the same waves).  Applies to H = 1024, B <= 512 on a device of 8 x 32 CUs
 * (vqa_gru_ws_supported).  `ws`: vqa_gru_ws_workspace_bytes(T) bytes of 16-byte-aligned device memory, contents
 * irrelevant (fragment-order hand-off buffers and counters; the call zeroes the counters).  After the stream has run
 * a non-zero 32-bit word at byte offset 2048 of `ws` reports a barrier time-out of this launch (results invalid);
 * the word at byte offset 4092 is set with it and cleared by no launch: zero it once, look whenever convenient.
 * Replaces the loop of vlmap/modules.py:124-140 (dynamic_rnn over GRUCell). */
int vqa_gru_seq_fwd_ws(const float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs, float* r,
                       float* u, float* c, float* rh, int T, int B, int H, void* ws, void* stream);
int vqa_gru_ws_supported(int T, int B, int H);
/* The back-propagation through time in the same frame (one launch, W_g^T / W_c^T slabs resident): dxp [T,B,3H] =
 * (dr_pre | du_pre | dc_pre) from dh_T [B,H] -- read only here, unlike vqa_gru_seq_bwd -- and the forward tape; d_outs
 * [T,B,H] or NULL as in vqa_gru_seq_bwd_outs.  256 < B <= 512, H = 1024 (vqa_gru_ws_bwd_supported); `ws` as above (one
 * buffer may serve both directions). */
int vqa_gru_seq_bwd_ws(const float* dh_T, const float* d_outs, const float* Wg_h, const float* Wc_h, const int32_t* len,
                       const float* hs, const float* r, const float* u, const float* c, float* dxp, int T, int B, int H,
                       void* ws, void* stream);
int vqa_gru_ws_bwd_supported(int T, int B, int H);
int64_t vqa_gru_ws_workspace_bytes(int T);
int vqa_gru_ws_set_mode(int mode);      /* bit 0: forward, bit 1: back-propagation, wherever they apply; -1 = 3 (default) */
int vqa_gru_ws_set_form(int form);      /* tuning: 0 = sub-phase tails inside the next matrix stream (default), 1 = plain order */
int vqa_gru_ws_set_stamps(unsigned long long* dev_words);   /* timing study (tools/gru_tune.py); NULL = off */

/* The same restricted to batch rows [row0, row0+rows): samples are independent, so disjoint
 * row windows may run concurrently on different streams. */
int vqa_gru_seq_fwd_rows(float* xp, const float* Wg_h, const float* Wc_h, const int32_t* len, float* hs, float* r,
                         float* u, float* c, float* rh, int T, int B, int H, int row0, int rows, void* stream);
int vqa_gru_seq_bwd_rows(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len, const float* hs,
                         const float* r, const float* u, const float* c, float* dxp, float* dh_scratch, int T, int B,
                         int H, int row0, int rows, void* stream);

/* ------------------------------------------------------------ a6+a7 / K6+K7
 * hadamard_attention + attention_pooling fused (vlmap/modules.py:67-97, 23-39):
 *  s[b,r] = sum_h v[b,r,h]*qv[b,h]*keep[b,r,h]/keep_prob*w[h] + bias
 *  s[r >= nb[b]] = -inf ; att = softmax_R(s) ; pooled[b,:] = sum_r att[b,r]*V[b,r,:]
 * keepmask u8 [B,R,H] (the explicit tf.nn.dropout(.,0.8) mask) or NULL. */
int vqa_attn_pool_fwd(const float* v, const float* qv, const float* V, const int32_t* nb, const float* w,
                      const float* bias, const uint8_t* keepmask, float keep_prob, float* att, float* pooled,
                      int B, int R, int H, int D, void* stream);
/* backward: dpooled [B,D] -> dv [B,R,H], dqv [B,H], per-sample partials
 * part_dw [B,H] and part_db [B] (reduce with vqa_colsum). */
int vqa_attn_pool_bwd(const float* dpooled, const float* v, const float* qv, const float* V, const float* att,
                      const float* w, const uint8_t* keepmask, float keep_prob, float* dv, float* dqv,
                      float* part_dw, float* part_db, int B, int R, int H, int D, void* stream);

/* `rep` queries per memory (the cfg-5 pre-training model attends n = 5 key boxes per image over the
 * same regions, vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:323-364, without materialising the
 * x5 tile): v [B,R,H], V [B,R,D], nb [B] per memory; qv [B*rep,H], keepmask [B*rep,R,H], att
 * [B*rep,R], pooled [B*rep,D] per query; dv [B,R,H] is summed over the queries of a memory. */
/* tuning / A-B switch: 0 = generic kernel; 1 (default) = the loads-in-flight forward kernel for the models' shapes when
 * rep == 1 and the one-workgroup-per-memory kernel when rep == 5; 2 = also the per-query fast kernel for other reps;
 * 3 = the per-query fast kernel for every rep */
int vqa_attn_set_fast(int on);
int vqa_attn_pool_fwd_rep(const float* v, const float* qv, const float* V, const int32_t* nb, const float* w,
                          const float* bias, const uint8_t* keepmask, float keep_prob, float* att, float* pooled,
                          int B, int rep, int R, int H, int D, void* stream);
int vqa_attn_pool_bwd_rep(const float* dpooled, const float* v, const float* qv, const float* V, const float* att,
                          const float* w, const uint8_t* keepmask, float keep_prob, float* dv, float* dqv,
                          float* part_dw, float* part_db, int B, int rep, int R, int H, int D, void* stream);

/* --------------------------------------------------------------- a11 / K11
 * sigmoid-CE loss, argmax, VQA scores (vqa/model_vlmap_answer.py:192-288).
 * Per sample stats[b, VQA_STAT_*]; dz = (sigmoid(z)-t)*(loss_mask?)/B_norm when
 * dz != NULL.  masks are float [A]; loss_mask = train mask for vlmap_answer,
 * NULL for model_standard (vqa/model_standard.py:285). */
enum {
    VQA_STAT_LOSS_TRAIN = 0, VQA_STAT_LOSS_REPORT, VQA_STAT_ALL_SCORE, VQA_STAT_EXIST_SCORE,
    VQA_STAT_TEST_SCORE, VQA_STAT_TEST_OBJ_SCORE, VQA_STAT_TEST_ATTR_SCORE, VQA_STAT_TRAIN_EXIST_SCORE,
    VQA_STAT_MAX_EXIST, VQA_STAT_MAX_TRAIN_EXIST, VQA_STAT_TEST_OBJ_MAX, VQA_STAT_TEST_ATTR_MAX,
    VQA_STAT_TEST_MAX, VQA_STAT_TEST_MAX_EXIST, VQA_STAT_MAX_TRAIN, VQA_STAT_COUNT = 16
};
int vqa_loss_fwd(const float* z, const float* target, const float* train_mask, const float* obj_mask,
                 const float* attr_mask, const float* exist_mask, int use_train_mask_in_loss, float inv_batch,
                 float* stats, int32_t* pred, float* dz, int B, int A, void* stream);
/* The two-headed loss of vqa/model_vlmap_answer_vqa_all2.py:226-339 (z_fixed = WordWeightAnswer logits, z_tuned =
 * TunedWordWeightAnswer logits): stats[.,LOSS_TRAIN] = sum_a ce(z_fixed)*train_mask + ce(z_tuned), [.,LOSS_REPORT] =
 * sum_a ce(z_fixed) + ce(z_tuned); pred = first argmax of z_fixed*(1-train_mask) + z_tuned*train_mask; the other
 * statistics as in vqa_loss_fwd on that prediction; dz_fixed = (sigmoid(z_fixed)-t)*train_mask*inv_batch, dz_tuned =
 * (sigmoid(z_tuned)-t)*inv_batch (both or neither NULL); z_sum (may be NULL) = z_fixed + z_tuned = output['logit'].
 * sum_mode != 0: the form of vqa/model_vlmap_answer_vqa_all.py:234-244 -- the tuned term is ce(z_fixed + z_tuned) and both
 * terms are train-masked in the training loss; pred = argmax(z_fixed + z_tuned); dz_fixed = d loss / d z_fixed (both terms),
 * dz_tuned = the tuned term's. */
int vqa_loss2_fwd(const float* z_fixed, const float* z_tuned, const float* target, const float* train_mask,
                  const float* obj_mask, const float* attr_mask, const float* exist_mask, float inv_batch, float* stats,
                  int32_t* pred, float* dz_fixed, float* dz_tuned, float* z_sum, int sum_mode, int B, int A, void* stream);
/* vqa/model_vlmap_answer_vqa_all.py:192-194: z_masked = z * exist + rowmin(z) * (1 - exist) (answers the word-weight
 * directory does not know sit at the row minimum), rowmin [B]; and its backward in place on dz (tf.reduce_min's
 * gradient goes to the minimum, split evenly over ties). */
int vqa_rowmin_mask_fwd(const float* z, const float* exist_mask, float* z_masked, float* rowmin, int B, int A, void* stream);
int vqa_rowmin_mask_bwd(float* dz, const float* z, const float* rowmin, const float* exist_mask, int B, int A, void* stream);
/* n-way softmax cross-entropy with a validity mask + top-1 / top-k hits (n_way_classification_loss,
 * vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:675-706).  z [rows,A], label i32[rows], valid f32[rows];
 * stats [rows,4] = {ce*valid, top1*valid, topk*valid, valid}; dz = (softmax-onehot)*valid*inv_valid_sum[0]
 * (inv_valid_sum is a DEVICE scalar = 1/sum(valid); dz may be NULL). */
int vqa_softmax_ce_fwd(const float* z, const int32_t* label, const float* valid, int topk,
                       const float* inv_valid_sum, float* stats, float* dz, int rows, int A, void* stream);
/* tuning / A-B switch: 1 (default) = rows of A <= 4096 (A % 4 == 0, 16-byte aligned) are held in registers and read
 * once, 0 = the three-pass kernel for every A */
int vqa_softmax_set_fast(int on);
/* report[13] in the order of vqa_report_key(i): means over B + guarded ratios. */
int vqa_report_reduce(const float* stats, int B, float* report, void* stream);
#define VQA_REPORT_COUNT 13
const char* vqa_report_key(int i);

/* --------------------------------------------------------------- a12 / K12
 * tf.contrib.layers.optimize_loss(Adam, clip_gradients=20.0) (vqa/trainer.py:106-114)
 * on FLAT parameter / gradient buffers.  norm_sq_out[0] = sum(g^2) over
 * g[0..n) + extra_sq[0] (extra = un-aggregated embedding-slice sumsq, may be
 * NULL).  partial needs >= vqa_sumsq_workspace_floats(n) floats. */
int vqa_sumsq(const float* g, int64_t n, const float* extra_sq, float* norm_sq_out, float* partial,
              int64_t partial_floats, void* stream);
int64_t vqa_sumsq_workspace_floats(int64_t n);
/* p,m,v updated in place: g' = g*clip/max(sqrt(norm_sq),clip); Adam(b1,b2,eps)
 * with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host and passed in. */
int vqa_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* norm_sq, float clip,
                  float lr_t, float beta1, float beta2, float eps, void* stream);

/* the same with lr_t read from DEVICE memory at run time: the form a captured train-step graph replays (vqa_graph_*) */
/* step_dev[0] += 1; lr_t_dev[0] = (float)(lr_dev[0] * sqrt(1 - beta2^step) / (1 - beta1^step)) (TF1 Adam's bias-corrected
 * rate, in double like the host's arithmetic of an eager step), computed on the device so that a replayed graph advances
 * its own step count */
int vqa_adam_lr_step(int64_t* step_dev, const double* lr_dev, double beta1, double beta2, float* lr_t_dev, void* stream);
int vqa_clip_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* norm_sq, float clip,
                      const float* lr_t_dev, float beta1, float beta2, float eps, void* stream);

/* explicit dropout keep-mask (counter-based, reproducible): out[i] = u(seed,i) < keep_prob */
int vqa_dropout_mask(uint8_t* out, int64_t n, uint64_t seed, uint64_t offset, float keep_prob, void* stream);

/* ------------------------------------------------------------------------
 * Whole fusion model (vqa/model_vlmap_answer.py:102-288 / vqa/model_standard.py:193-374)
 * and its backward, as one host call each.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t B, R, D, H, T, W, A, Vq;
    int64_t N_img;
    int32_t model_type;      /* 7..11: see VQA_MODEL_* below.  0 = vlmap_answer, 1 = standard, 2 = standard_word2vec, 3 = standard_testmask (= 1 with
                              * the training loss masked by the train-answer mask, vqa/model_standard_testmask.py:266-268),
                              * 4 = vlmap_answer_vqa_all2 (= 0 + the trainable TunedWordWeightAnswer head, summed logits,
                              * two-term loss, mixed-mask argmax: vqa/model_vlmap_answer_vqa_all2.py:196-244),
                              * 6 = vlmap_answer_vqa_all (= 4 with unknown answers' fixed logits at the row minimum, the
                              * tuned loss on the summed logits, both terms train-masked, argmax of the sum),
                              * 5 = vlmap_answer_noc / _nocarch (two un-composed branches joint_v(pooled_linear_l) and
                              * joint_l(l_linear_l) with their own heads, logits summed: vqa/model_vlmap_answer_noc.py:177-204) */
    float keep_att;          /* 0.8  vlmap/modules.py:82 */
    float keep_joint;        /* 0.5  vqa/model_vlmap_answer.py:180 */
    float inv_global_batch;  /* 1/B for one GPU, 1/(sum of shard sizes) under data parallel */
    int32_t flags;           /* VQA_FLAG_* bit mask, per call (no process-wide state) */
    /* ABI 5: the five older ablations of model_vlmap_answer (model_type 7..11, see VQA_MODEL_* below) */
    int32_t num_marginal;    /* 11 only: pairings per question, NUM_MARGINAL = 200 (vqa/model_vlmap_answer_ent.py:16) */
    int32_t ent_cols;        /* 11 only: number of leading head columns the regulariser needs = 1 + the last answer index
                              * with train_mask * exist_mask > 0.5 (the train mask is a prefix, :44-46), rounded up by the
                              * caller as it likes (<= A); columns the masks exclude are ignored */
    float extra_weight;      /* 10: latent_loss_weight 0.1 (vqa/model_vlmap_answer_full.py:33); 11: W_ENTROPY 0.1 (:14) */
    int32_t map_dim;         /* 13 only: MAP_DIM, the hidden width of L2V / V2L (vqa/model_vqa.py:12) */
    int32_t La;              /* 13 only: padded token length of the candidate answers (batch.answer_intseq [A, La]) */
} vqa_dims_t;
/* model_type 7..11 = model_vlmap_answer (0) with ONE change each:
 *   7  vlmap_answer2          q_L_ft2 = fc_layer(q_L_ft, LN, tanh) feeds q_linear_l and is heavy_output['condition']
 *                             (vqa/model_vlmap_answer2.py:127-131,164)
 *   8  vlmap_answer_no_noise  q_L_mean = linear fc_layer(q_L_ft) feeds q_linear_l (vqa/model_vlmap_answer_no_noise.py:122-125,157)
 *   9  vlmap_answer_adapt     v_adapt = fc_layer(V_ft, LN over [R,H], ReLU) is what the attention pools: pooled_V_ft is
 *                             [B,H], pooled_linear_l maps H -> H (vqa/model_vlmap_answer_adapt.py:132-142)
 *  10  vlmap_answer_full      q_linear_l reads q_L_mean + noise * sqrt(exp(q_L_log_sigma_sq)) with `noise` an explicit
 *                             input (tf.random_normal(seed=123) in the reference); loss += extra_weight * KL
 *                             (vqa/model_vlmap_answer_full.py:124-134,166,217-223,272-276)
 *  11  vlmap_answer_ent       marginal-entropy regulariser: joint_fc (+ own dropout) + head on num_marginal pairings of every
 *                             question with stop-gradient pooled_linear_l rows of the batch, softmax over the known training
 *                             answers, mean over the pairings, loss += extra_weight * mean_b sum_a p log(p + 1e-8)
 *                             (vqa/model_vlmap_answer_ent.py:191-211, 281-292)
 * Types 10 and 11 report three more scalars behind the 13 of vqa_report_key: report[13] = latent_loss | entropy,
 * report[14] = extra_weight * report[13], report[15] = the model's total loss (report[0] + report[14]); stats[b,15]
 * carries the per-sample term for data-parallel reporting. */
#define VQA_MODEL_ANSWER2 7
#define VQA_MODEL_NO_NOISE 8
#define VQA_MODEL_ADAPT 9
#define VQA_MODEL_FULL 10
#define VQA_MODEL_ENT 11
/* 12 = vlmap_finetune / vlmap_only (vqa/model_vlmap_finetune.py:89-211; model_vlmap_only.py differs in its train set
 * only, i.e. in which members of `grads` are NULL): the question is encoded by a bi-directional GRU of H/2 + H/2 units
 * (encode_L_bidirection; gru_* = forward cell, gru_bw_* = backward cell), q_L_ft = concat(final states) feeds
 * q_linear_l; the image attention's query is q_linear_v(pooled_q_v), pooled_q_v = a question SELF-attention
 * (word_attention: keys q_att_key(bi-GRU outputs) [B,T,H] with LayerNorm over [T,H], query q_att_query(q_L_ft), its own
 * dropout mask keep_word) pooling v_word_fc(V_WordMap[q]) [B,T,H].  report[0..2] = answer_train_loss,
 * answer_report_loss, answer_accuracy (:207-211; = entries 0, 1, 2 of vqa_report_key).  The gradient of V_WordMap is
 * scatter-added like the first table's (grads.embed2 pre-zeroed); embed_slice_sq receives the sum of squares of BOTH
 * tables' un-aggregated slices (the second one only when grads.embed2 != NULL). */
#define VQA_MODEL_BI 12
/* 13 = vqa (vqa/model_vqa.py:185-277, the registry's oldest model and vqa/trainer.py's default): a BasicLSTMCell
 * (H = L_DIM units; one set of weights) encodes the questions and every candidate answer's token sequence
 * (batch.answer_intseq [A, La]); L2V maps the question code to the D-wide features of the retired model_vfeat pipeline
 * (D == vfeat_dim, 512), a DOT-PRODUCT attention pools V_ft, V2L maps the pooled feature back, and
 * logit[b,a] = w . tanh(A1 answer_ft[a] + P1 pooled_map_L[b] + Q1 q_L_ft[b] + bq) + bc; loss = mean_B sum_A sigmoid-CE
 * (no train-answer mask); report[0] = answer_loss, report[2] = answer_accuracy.  Embedding = GloVe_vocab: `glove_fixed`
 * constant rows + the 3 trainable rows `glove_learn` (scatter-added: grads.glove_learn pre-zeroed; embed_slice_sq =
 * sum of squares of the un-aggregated slices that reach it).  All of backward runs in phase 1. */
#define VQA_MODEL_LEGACY_VQA 13
#define VQA_FLAG_DETERMINISTIC 1   /* embedding-gradient scatter-add without atomics: bitwise reproducible steps */
#define VQA_FLAG_FUSED_GATHER 2     /* no gather pass: v_linear_v's GEMM reads the table rows through image_idx
                                     * (vqa_gemm_f32_gather) and leaves V_ft behind as a by-product; default: a gather
                                     * pass in front of the GEMM (same step time, see csrc/fusion_model.hip) */
#define VQA_FLAG_SHARED_LN 4        /* cfg-5 model only: one LayerNorm per shared fc_layer scope (see vqa_pt_fc_t) */

/* One FC(+LN) layer: weights [in,out], biases [out], LayerNorm beta/gamma [out] (NULL if no LN). */
typedef struct { float *w, *b, *beta, *gamma; } vqa_fc_t;

typedef struct {
    float* embed;                       /* LearnGloVe/embed_map [Vq,W] */
    vqa_fc_t v_linear_v;                /* [D,H] */
    float *gru_wg, *gru_bg;             /* encode_L/rnn/gru_cell/gates/{kernel,bias} [W+H,2H],[2H] */
    float *gru_wc, *gru_bc;             /* .../candidate/{kernel,bias} [W+H,H],[H] */
    vqa_fc_t q_linear_v;                /* [H,H] */
    vqa_fc_t score;                     /* hadamard_attention/compute/score [H,1],[1] */
    vqa_fc_t pooled_linear_l;           /* [D,H] */
    vqa_fc_t q_linear_l;                /* [H,H] */
    vqa_fc_t joint_fc;                  /* [H,2H] */
    vqa_fc_t head;                      /* WordWeightAnswer | reasoning/classifier [2H,A]  ([2H,W] for standard_word2vec) */
    float* answer_glove;                /* standard_word2vec only: constant [W,A] GloVe matrix of the answers
                                         * (vqa/model_standard_word2vec.py:185-188); NULL otherwise */
    vqa_fc_t head2;                     /* vlmap_answer_vqa_all2: TunedWordWeightAnswer [2H,A], the trainable second head on
                                         * `joint` (vqa/model_vlmap_answer_vqa_all2.py:216-220); vlmap_answer_noc:
                                         * WordWeightAnswerL [2H,A] (`head` is then WordWeightAnswerV); NULL otherwise */
    vqa_fc_t joint2;                    /* vlmap_answer_noc only: joint_l [H,2H] on l_linear_l (`joint_fc` is then joint_v on
                                         * pooled_linear_l; vqa/model_vlmap_answer_noc.py:177-188); NULL otherwise */
    /* ABI 5 (model_type 7..11; NULL otherwise) */
    vqa_fc_t q_L_ft2;                   /* 7: [H,H] + LayerNorm */
    vqa_fc_t q_L_mean;                  /* 8, 10: [H,H], no LayerNorm */
    vqa_fc_t q_L_log_sigma_sq;          /* 10: [H,H], no LayerNorm */
    vqa_fc_t v_adapt;                   /* 9: [D,H] + LayerNorm over the [R,H] block (pooled_linear_l is then [H,H]) */
    /* model_type 12 (NULL otherwise); gru_wg .. gru_bc are then the FORWARD cell [W+H/2, H], [H], [W+H/2, H/2], [H/2] */
    float* embed2;                      /* V_WordMap/embed_map [Vq,W] */
    float *gru_bw_wg, *gru_bw_bg;       /* encode_L_bi/bidirectional_rnn/bw/gru_cell/gates/{kernel,bias} */
    float *gru_bw_wc, *gru_bw_bc;       /* .../bw/gru_cell/candidate/{kernel,bias} */
    vqa_fc_t q_att_key, q_att_query;    /* [H,H] + LayerNorm (over [T,H] / over [H]) */
    vqa_fc_t word_score;                /* word_attention/compute/score [H,1] */
    vqa_fc_t v_word_fc;                 /* [W,H] + LayerNorm over [T,H] */
    /* model_type 13 (NULL otherwise) */
    float* glove_fixed;                 /* constant GloVe rows [Vq-3, W] (a tf.constant: never a gradient) */
    float* glove_learn;                 /* GloVe/learn [3, W]: the last three vocabulary entries */
    float *lstm_k, *lstm_b;             /* encode_L/rnn/basic_lstm_cell/{kernel [W+H, 4H], bias [4H]} */
    vqa_fc_t l2v[3];                    /* L2V/{fc_1 [H,M], fc_2 [M,M], Linear [M,D]} (ReLU, ReLU, none; no LayerNorm) */
    vqa_fc_t v2l[3];                    /* V2L/{fc_1 [D,M], fc_2 [M,M], Linear [M,H]} (tanh, tanh, none) */
    vqa_fc_t answer_layer1, pooled_layer1, q_layer1;   /* reasoning/... [H,H]; only q_layer1 has a bias */
    vqa_fc_t classifier;                /* reasoning/classifier [H,1] + bias */
} vqa_params_t;

typedef struct {
    const float* table;                 /* [N_img,R,D] */
    const int32_t* nbox_table;          /* [N_img] */
    const int64_t* image_idx;           /* [B] */
    const int32_t* q_intseq;            /* [B,T] zero padded */
    const int32_t* q_intseq_len;        /* [B] */
    const float* answer_target;         /* [B,A] */
    const float *train_mask, *obj_mask, *attr_mask, *exist_mask; /* [A] */
    const uint8_t* keep_att;            /* [B,R,H] 0/1 or NULL (no dropout) */
    const uint8_t* keep_joint;          /* [B,2H] 0/1 or NULL */
    const uint8_t* keep_joint2;         /* vlmap_answer_noc only: keep-mask of l_joint [B,2H] (keep_joint is v_joint's) or NULL */
    const int32_t* live_rows;           /* HOST int[T] or NULL.  Non-NULL promises that the batch rows are sorted by
                                         * q_intseq_len, longest first, and live_rows[t] = #rows with len > t; the
                                         * recurrence then skips finished sequences (vqa_gru_seq_*_live). */
    /* ABI 5 */
    const float* noise;                 /* model_type 10: [B,H] standard-normal draws of the reparameterisation */
    const uint8_t* keep_tile;           /* model_type 11: keep-mask of tf.nn.dropout(tile_joint, 0.5) [B,num_marginal,2H] or NULL */
    const uint8_t* keep_word;           /* model_type 12: keep-mask of the word attention's dropout [B,T,H] or NULL */
    const int32_t* answer_intseq;       /* model_type 13: token ids of the candidate answers [A, La], zero padded */
    const int32_t* answer_intseq_len;   /* model_type 13: [A] */
} vqa_batch_t;

int64_t vqa_fusion_workspace_bytes(const vqa_dims_t* dims);
/* Byte offset / element count of a named intermediate inside the workspace
 * (names follow the reference's mid_result/output keys: V_ft, num_V_ft,
 * v_linear_v, condition, q_linear_v, att_score, pooled_V_ft, pooled_linear_l,
 * l_linear_l, joint, logit, pred, stats, report, dx_embed, ...). */
int vqa_fusion_tensor(const vqa_dims_t* dims, const char* name, int64_t* offset_bytes, int64_t* n_elems);
int vqa_fusion_forward(const vqa_dims_t* dims, const vqa_params_t* params, const vqa_batch_t* batch,
                       void* workspace, int64_t workspace_bytes, int want_dz, void* stream);
/* grads: same layout as params; a NULL member skips that gradient (frozen
 * variable).  Gradient buffers are OVERWRITTEN (embed must be pre-zeroed: it is
 * scatter-added).  embed_slice_sq receives sum(dx^2) of the un-aggregated
 * embedding IndexedSlices (for clip_by_global_norm). */
int vqa_fusion_backward(const vqa_dims_t* dims, const vqa_params_t* params, const vqa_params_t* grads,
                        const vqa_batch_t* batch, void* workspace, int64_t workspace_bytes,
                        float* embed_slice_sq, void* stream);

/* ------------------------------------------------------------------------
 * Whole cfg-5 pre-training model (vlmap_memft/model_vlmap_bf_or_wordset_withatt_sp.py:54-94,
 * 323-609, 675-706) and its backward, one host call each -- the same shape as vqa_fusion_forward / _backward.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t B, n, R, D, H, W, A, Vq, n_ws;  /* images, entries per category (5), regions, feature / hidden / word dims,
                                             * answers, caption vocabulary, word-set vocabulary */
    int32_t L;                               /* padded caption length of THIS batch (blanks [B,n,L]) */
    int32_t flags;                           /* VQA_FLAG_* */
    float keep_att, keep_joint;              /* 0.8 / 0.5 */
    float global_valid[2];                   /* data parallel: number of valid entries of the GLOBAL batch per category
                                              * (object, attribute) = the denominator of the masked mean losses
                                              * (:675-706); 0 = this batch's own count (one process) */
} vqa_pretrain_dims_t;

/* fc_layer scope entered by several call sites (vlmap/modules.py:630-650): one weight / bias and
 *   - with VQA_FLAG_SHARED_LN in vqa_pretrain_dims_t.flags: ONE LayerNorm (slot 0) used by every call site, its gradient
 *     the sum over the call sites.  This is what TF 1.x builds: leaving the string-named fc_layer scope zeroes the
 *     sub-scope counts (close_variable_subscopes), so the un-scoped layers.layer_norm is named `LayerNorm` again at the
 *     next call site and AUTO_REUSE shares it;
 *   - without the flag: one LayerNorm PER CALL SITE in TF graph order (`LayerNorm`, `LayerNorm_1`, ...), the other
 *     reading of the same code, kept selectable so that the variable names of a checkpoint decide.
 * Unused slots NULL. */
typedef struct { float *w, *b, *beta[4], *gamma[4]; } vqa_pt_fc_t;

typedef struct {
    float* wordset_map;                     /* wordset_map/learn [n_ws,W] */
    float* l_glove;                         /* L_GloVe/embed_map [Vq,W] */
    vqa_pt_fc_t spat_v_linear_v;            /* [6,H]   LN x2 (object, attribute) */
    vqa_pt_fc_t spat_q_linear_v;            /* [6,H]   LN x2 */
    vqa_pt_fc_t spat_att_score;             /* spat_att/compute/score [H,1] */
    float *gru_wg, *gru_bg, *gru_wc, *gru_bc; /* encode_L_blank/rnn/gru_cell/{gates,candidate}/{kernel,bias} */
    vqa_pt_fc_t pooled_linear_l;            /* [D,H]   LN x4 (obj bf, attr bf, obj ws, attr ws) */
    vqa_pt_fc_t q_linear_l;                 /* [H,H]   LN x4 */
    vqa_pt_fc_t joint_fc;                   /* [H,2H]  LN x4 */
    vqa_pt_fc_t wordset_ft;                 /* [W,H]   LN x2 */
    vqa_pt_fc_t classifier;                 /* [2H,A] */
} vqa_pretrain_params_t;

typedef struct {                            /* one blank-fill category (vlmap_memft/datasets/dataset_vlmap.py:128-236) */
    const float* normal_boxes;              /* [B,n,4] */
    const int32_t *fills, *blanks, *blanks_len, *wordsets;   /* [B,n], [B,n,L] zero padded, [B,n], [B,n] */
    const int32_t* num;                     /* [B] valid entries per image */
    const uint8_t *keep_att, *keep_bf_joint, *keep_ws_joint;  /* [B*n,R,H], [B*n,2H], [B*n,2H] 0/1 or NULL */
} vqa_pretrain_kind_t;

typedef struct {
    const float* image_ft;                  /* [B,R,D] */
    const float* spatial_ft;                /* [B,R,6] */
    const int32_t* num_boxes;               /* [B] */
    vqa_pretrain_kind_t kind[2];            /* 0 = object, 1 = attribute */
    /* The blank-fill captions of both categories are encoded as ONE batch of 2*B*n rows (L_GloVe and the GRU are shared):
     * row v < B*n is object caption v, row v >= B*n attribute caption v - B*n.  Optionally ordered by length, longest
     * first (all three NULL = as given): perm / inv device int32 [2*B*n] (sorted position -> row, row -> sorted
     * position), live_rows HOST int[L] = #captions longer than t -- the recurrence then runs on the live prefix only */
    const int32_t *perm, *inv, *live_rows;
} vqa_pretrain_batch_t;

int64_t vqa_pretrain_workspace_bytes(const vqa_pretrain_dims_t* dims);
/* Named intermediates inside the workspace: "<obj|attr>/{att,pooled,valid,bf_state,...}",
 * "<obj|attr>/<bf|ws>/{z,dz,stats,j,...}", "report" (13 floats in the order of vqa_pretrain_report_key). */
int vqa_pretrain_tensor(const vqa_pretrain_dims_t* dims, const char* name, int64_t* offset_bytes, int64_t* n_elems);
const char* vqa_pretrain_report_key(int i);
int vqa_pretrain_forward(const vqa_pretrain_dims_t* dims, const vqa_pretrain_params_t* params,
                         const vqa_pretrain_batch_t* batch, void* workspace, int64_t workspace_bytes, int want_dz,
                         void* stream);
/* grads: same layout as params, every member non-NULL.  Gradient buffers are OVERWRITTEN (the two embedding tables
 * are cleared and scatter-added).  slice_sq receives the sum of squares of the un-aggregated embedding slices. */
int vqa_pretrain_backward(const vqa_pretrain_dims_t* dims, const vqa_pretrain_params_t* params,
                          const vqa_pretrain_params_t* grads, const vqa_pretrain_batch_t* batch, void* workspace,
                          int64_t workspace_bytes, float* slice_sq, void* stream);
/* The same backward in dependency-ordered phases (bit mask; a step runs them in ascending order, each reads what the
 * lower ones left in the workspace) so that a data-parallel caller starts reducing a finished bucket while the next
 * phase runs -- the wrap of vlmap_memft/trainer.py:129-137, 202-263 (optimize_loss over ALL variables):
 *   1  stacked heads: classifier, joint_fc, pooled_linear_l, q_linear_l (weights, biases, LayerNorms)
 *   2  BPTT of the joint caption batch + GRU kernel / bias gradients
 *   4  dx of the packed x-projection -> L_GloVe scatter-add (starts the slice sum of squares)
 *   8  per category: wordset_ft, wordset_map, spat_att, spat_v_linear_v, spat_q_linear_v (writes slice_sq)
 * vqa_pretrain_backward == phases 15. */
int vqa_pretrain_backward_phases(const vqa_pretrain_dims_t* dims, const vqa_pretrain_params_t* params,
                                 const vqa_pretrain_params_t* grads, const vqa_pretrain_batch_t* batch, void* workspace,
                                 int64_t workspace_bytes, float* slice_sq, int phases, void* stream);

/* ------------------------------------------------------------------------
 * Region-feature extractor, NHWC fp32.
 * ------------------------------------------------------------------------ */
/* conv + folded inference BatchNorm (+ residual) (+ ReLU):
 *   y = [relu]( conv(x, w) * scale[co] + shift[co] + residual )
 * x [B,Hi,Wi,Ci], w HWIO [kh,kw,Ci,Co] (TF layout), explicit top/left zero padding, output
 * [B,Ho,Wo,Co]; scale/shift/residual may be NULL.  1x1/stride-1 -> plain MFMA GEMM, otherwise an
 * implicit GEMM (needs Ci % 32 == 0, or Ci == 4 with kh*kw*4 a multiple of 32).  Replaces slim conv2d/conv2d_same + batch_norm(is_training=
 * False) + relu of resnet_v1 (vlmap/modules.py:143-191) and modules.conv2d (:552-572). */
int vqa_conv2d_nhwc(const float* x, int B, int Hi, int Wi, int Ci, const float* w, int kh, int kw, int Co,
                    int stride, int pad_t, int pad_l, int Ho, int Wo, const float* scale, const float* shift,
                    const float* residual, int relu, float* y, void* stream);
/* Backward of vqa_conv2d_nhwc (the reference's only consumer is the legacy CNN fine-tune,
 * vlmap/model_vlmap.py:675-690 --ft_enc_I): with dy = d loss / d y and g = dy * (relu ? y > 0 : 1),
 *   dx [B,Hi,Wi,Ci] = conv_transpose(g * scale, w), dw HWIO [kh,kw,Ci,Co] = im2col(x)^T (g * scale), dshift [Co] = sum g,
 *   dresidual [B,Ho,Wo,Co] = g        (every output may be NULL; y is needed only when relu != 0; scale may be NULL).
 * Both products run on vqa_gemm_f32 over chunks of images that fit `workspace` (vqa_conv2d_bwd_workspace_floats(...,
 * chunk_images): pass B for one chunk, 1 for the minimum); the col2im step is a gather (no atomics: deterministic).
 * Needs Ci % 4 == 0 and Co % 4 == 0 (conv1 of the extractor runs on 4-channel padded pixels, vqa_pad_c3c4_nhwc). */
int64_t vqa_conv2d_bwd_workspace_floats(int B, int Ho, int Wo, int Ci, int kh, int kw, int Co, int chunk_images);
int vqa_conv2d_nhwc_bwd(const float* x, int B, int Hi, int Wi, int Ci, const float* w, int kh, int kw, int Co, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, const float* scale, const float* y, int relu, const float* dy,
                        float* dx, float* dw, float* dshift, float* dresidual, float* workspace, int64_t workspace_floats,
                        void* stream);
/* tuning: tile configuration of the implicit-GEMM path (-1 = chosen by shape, the default; 0 = 64x64 / 4 waves,
 * 1 = 128x64, 2 = 64x128, 3 = 128x128, all 8 waves; >= 0 also pins the 1x1 path to 64x64 unless vqa_gemm_set_config
 * forces another); process-wide, like vqa_gemm_set_config. */
int vqa_conv_set_config(int cfg);
/* y [B,Hi,Wi,4] = (x [B,Hi,Wi,3] - mean_host, 0): 16-byte pixels, so that conv1 (7x7/2 on RGB, vlmap/modules.py:170-190)
 * runs through vqa_conv2d_nhwc as an implicit GEMM with Ci = 4 and an 8-wide zero-padded filter row (K = 7*8*4). */
int vqa_pad_c3c4_nhwc(const float* x, int B, int Hi, int Wi, const float* mean_host, float* y, void* stream);
/* explicit im2col for the 3-channel conv1 (7x7/2): col [B*Ho*Wo, Kpad], zero in the K padding;
 * mean_host (3 floats on the HOST, may be NULL) is subtracted from in-bounds pixels only
 * (the RGB mean subtraction of vlmap/modules.py:170-174 precedes the zero padding). */
int vqa_im2col_nhwc(const float* x, int B, int Hi, int Wi, int Ci, int kh, int kw, int stride, int pad_t, int pad_l,
                    int Ho, int Wo, const float* mean_host, float* col, int Kpad, void* stream);
/* slim pool1: max_pool2d(3x3, stride 2, 'SAME'); y [B,ceil(Hi/2),ceil(Wi/2),C] */
int vqa_maxpool3x3s2_same_nhwc(const float* x, int B, int Hi, int Wi, int C, float* y, void* stream);
/* resnet_utils.subsample (1x1 max-pool with stride = strided slicing) */
int vqa_subsample_nhwc(const float* x, int B, int Hi, int Wi, int C, int factor, float* y, void* stream);
/* tf.image.crop_and_resize (bilinear, extrapolation 0): boxes [n,4] = normalised [y1,x1,y2,x2],
 * box_ind i32[n] -> out [n,crop_h,crop_w,C]   (modules.roi_pool, vlmap/modules.py:204-216) */
int vqa_crop_and_resize_nhwc(const float* fmap, int B, int H, int W, int C, const float* boxes,
                             const int32_t* box_ind, int n_boxes, int crop_h, int crop_w, float* out, void* stream);

/* ------------------------------------------------------------------------
 * Measurement probe (bench.py) and trace ranges (csrc/probe.hip).  Every launch group of the whole-model entry points
 * sits in a labelled scope on the group's own stream:
 *   forward   "forward" = { "embed.fwd", "gru.xp_gemm", "gru.fwd", "fc.fwd_gemm", "fc.ln_fwd", "gather",
 *             "v_linear_v.fwd_gemm", "v_linear_v.ln_fwd", "attn_pool.fwd", "eltwise", "head.fwd_gemm", "loss.fwd" }
 *   backward  "backward" = { "head.bwd_gemm", "fc.ln_bwd", "fc.dw_gemm", "fc.dx_gemm", "eltwise", "attn_pool.bwd",
 *             "v_linear_v.ln_bwd", "v_linear_v.dw_gemm", "gru.bwd", "gru.dx_gemm", "embed.bwd", "gru.dwx_gemm",
 *             "gru.dwh_gemm" }
 *   cfg-5     "pretrain.forward", "pretrain.backward" and "pt.*" groups (csrc/pretrain_model.hip)
 * vqa_probe_enable(labels, max_samples): `labels` is a comma-separated list ("*" = every scope met); HIP events are
 * recorded around every enabled group, at most max_samples per label.  vqa_probe_read_label synchronises the events of
 * one label and returns its per-sample durations in milliseconds (vqa_probe_read: the first label of the list).
 * vqa_probe_labels lists the labels ('\n'-separated; returns the buffer size needed).
 * vqa_roctx_enable(1) (or VQA_HOT_ROCTX=1 in the environment) additionally opens a roctx range per scope, so that
 * `rocprofv3 --kernel-trace --marker-trace` groups kernels by phase; VQA_ERR_UNSUPPORTED when no roctx library loads. */
int vqa_probe_enable(const char* labels, int max_samples);
int vqa_probe_read(float* ms_out, int capacity, int* n_out);
int vqa_probe_read_label(const char* label, float* ms_out, int capacity, int* n_out);
int vqa_probe_labels(char* buf, int capacity);
int vqa_probe_disable(void);
int vqa_roctx_enable(int on);
/* EXPERIMENT (not on the default path): C[M,N] = A[M,K] * B[K,N] (+ bias) with every f32 operand split into three bf16
 * pieces and six v_mfma_f32_32x32x16_bf16 products per a*b accumulated in f32 (f32-equivalent products at 6/16 of the f32
 * MFMA's matrix time; csrc/gemm_bf16x3.hip).  Whole 128 x 128 x 32 tiles only (vqa_gemm_bf16x3_supported);
 * VQA_ERR_UNSUPPORTED otherwise.  The replacement of layers.fully_connected stays vqa_gemm_f32 (exact f32 MFMA). */
int vqa_gemm_bf16x3_supported(int M, int N, int K);
int vqa_gemm_bf16x3_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                       const float* bias, void* stream);
/* The same with a transposed left operand (transA = 1: A stored [K,M] -- dW = X^T dY) and split k (split_k ranges whose
 * partial products meet in `workspace`, vqa_gemm_bf16x3_workspace_floats, and are summed in range order: deterministic).
 * vqa_gemm_bf16x3_set_mode(1) (or VQA_HOT_BF16X3=1) makes vqa_gemm_f32 route its big whole-tile NN / TN products here
 * (>= 2^32 multiply-adds, no addend): an opt-in numerics mode -- f32-equivalent products, not bit-identical sums -- that
 * bench.py reports as a SEPARATE leg; 0 = off (default), -1 = back to the environment. */
int64_t vqa_gemm_bf16x3_workspace_floats(int M, int N, int K, int split_k);
int vqa_gemm_bf16x3(int transA, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, int split_k, float* workspace, int64_t workspace_floats, void* stream);
int vqa_gemm_bf16x3_set_mode(int on);
/* Short-K GEMM with the left operand stationary in registers (csrc/gemm_shortk.hip):
 *   C[M,N] = [relu]( (A[M,K] * B[K,N]) * scale[n] + bias[n] + D[M,N] )     row-major, bias / scale / D optional (NULL)
 * for K <= 512 -- the packed x-projection of the GRU (K = 300; replaces the x half of GRUCell's two matmuls,
 * vlmap/modules.py:124-140) and the extractor's 1x1 expansion convolutions (K = Ci = 64 / 128 / 256) with their folded
 * BatchNorm, residual and ReLU (K = Ci = 128 / 256 / 512).  Exact f32 MFMA (v_mfma_f32_32x32x2_f32), k summed in ascending order per output.
 * vqa_gemm_shortk_supported: 1 when the shape qualifies (K <= 512, K % 4 == 0, N % 32 == 0, lda / ldb % 4 == 0, operands
 * below 4 GB); vqa_gemm_f32 (K <= 304, M >= 1024) and vqa_conv2d_nhwc (Ci 128..256, Co >= 2 Ci) route their calls here themselves (VQA_HOT_SHORTK=0 turns that off).
 * vqa_gemm_shortk_set_grid: tuning override of the workgroup count (0 = two per CU).
 * vqa_gemm_shortk_set_mode: which callers route to it -- bit 0 vqa_gemm_f32, bit 1 the 1x1 convolutions of
 * vqa_conv2d_nhwc (Ci 128..256); -1 = back to VQA_HOT_SHORTK / the default 3. */
int vqa_gemm_shortk_supported(int M, int N, int K, int lda, int ldb, int ldc);
int vqa_gemm_shortk_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                       const float* bias, const float* scale, const float* D, int ldd, int relu, void* stream);
int vqa_gemm_shortk_set_grid(int n);
int vqa_gemm_shortk_set_waves(int n);   /* tuning: 4 (two workgroups per CU) or 8 (one) waves per workgroup; 0 = default */
int vqa_gemm_shortk_set_mode(int mode);
/* Shader-clock sampler: enqueues, on `stream`, n_workgroups single-wave workgroups that for n_samples periods of
 * us_per_sample microseconds (100 MHz real-time counter) note the shader cycles that went by; ghz_out (device memory,
 * [n_workgroups][n_samples]) receives cycles / time in GHz.  Launched on a side stream beside the work to be measured
 * (bench.py: the clock the timed steps ran at; the published MFMA peak assumes 2.4 GHz).  At most 2 s in total. */
int vqa_clock_sample(float us_per_sample, int n_samples, int n_workgroups, float* ghz_out, void* stream);
/* Enqueues a delay of `us` microseconds ON `stream` (one wave polling the 100 MHz real-time counter; 0 <= us <= 1e5).
 * The whole-model entry points use it to start the recurrence's independent row chains in anti-phase
 * (VQA_HOT_GRU_CHAINS / VQA_HOT_GRU_CHAIN_DELAY_US, csrc/fusion_model.hip). */
int vqa_stream_delay_us(float us, void* stream);

/* ------------------------------------------------------------------------
 * Bi-directional question encoder (modules.encode_L_bidirection, vlmap/modules.py:100-122) of
 * vqa/model_vlmap_finetune.py / model_vlmap_only.py: helpers around the fused GRU recurrence (csrc/bi_ops.hip).
 * tf.nn.bidirectional_dynamic_rnn runs the backward cell on reverse_sequence(inputs, len) and reverses its outputs
 * back; outputs past a row's length are zero, final states are carried through.
 * ------------------------------------------------------------------------ */
/* q_rev[b,t] = q[b, len_b-1-t] for t < len_b, q[b,t] otherwise (tf.reverse_sequence on token ids [B,T]) */
int vqa_reverse_tokens(const int32_t* q, const int32_t* len, int32_t* q_rev, int B, int T, void* stream);
/* hs_fw / hs_bw [T+1,B,h] (time-major states of the two recurrences, the backward one over reversed tokens) ->
 * q_map [B,T,2h] = concat(outputs) in ORIGINAL token order, zero past the length; q_ft [B,2h] = concat(final states) */
int vqa_bi_outputs_fwd(const float* hs_fw, const float* hs_bw, const int32_t* len, float* q_map, float* q_ft, int B, int T,
                       int h, void* stream);
/* its transpose: d_map [B,T,2h], d_ft [B,2h] -> per-step output gradients dout_fw / dout_bw [T,B,h] in each recurrence's
 * own step order (zero past the length) and the final-state gradients dhT_fw / dhT_bw [B,h] */
int vqa_bi_outputs_bwd(const float* d_map, const float* d_ft, const int32_t* len, float* dout_fw, float* dout_bw,
                       float* dhT_fw, float* dhT_bw, int B, int T, int h, void* stream);
/* dx[t,b,:] = dx_fw[t,b,:] + dx_bw[len_b-1-t, b,:] (t < len_b; plain sum otherwise): gradient wrt the looked-up
 * embeddings [T,B,W] time-major = the un-aggregated IndexedSlices of the embedding gradient */
int vqa_bi_dx_combine(const float* dx_fw, const float* dx_bw, const int32_t* len, float* dx, int B, int T, int W,
                      void* stream);
/* vqa_gru_seq_bwd with a gradient on every step's output: d_outs [T,B,H], zero where t >= len */
int vqa_gru_seq_bwd_outs(float* dh_T, const float* Wg_h, const float* Wc_h, const int32_t* len, const float* hs,
                         const float* r, const float* u, const float* c, const float* d_outs, float* dxp,
                         float* dh_scratch, int T, int B, int H, void* stream);

/* ------------------------------------------------------------------------
 * Kernels of the oldest registry model, vqa/model_vqa.py (csrc/lstm_ops.hip); composed for model_type 13.
 * ------------------------------------------------------------------------ */
/* modules.GloVe_vocab lookup (vlmap/modules.py:451-467): x_tm[t,n,:] = id < Vq-3 ? fixed[id] : learn[id-(Vq-3)] for
 * id = ids[n,t] (ids batch-major [N,T], rows time-major [T,N,W]); backward: scatter-add of the rows with id >= Vq-3 into
 * dlearn [3,W] (float atomics) and slice_sq[0] += sum of squares of those rows. */
int vqa_embed2_fwd(const float* fixed, const float* learn, const int32_t* ids, float* x_tm, int N, int T, int W, int Vq,
                   void* stream);
int vqa_embed2_bwd(const float* dx_tm, const int32_t* ids, float* dlearn, float* slice_sq, int N, int T, int W, int Vq,
                   void* stream);
/* One step of tf.contrib.rnn.BasicLSTMCell under dynamic_rnn(sequence_length): gates [N,4L] holds [x,h] K + b on entry
 * (order i, j, f, o) and the activated gates (f with forget_bias 1.0 folded in) on return; c_new = c f + i j, h_new =
 * tanh(c_new) o for rows with t < len, (c, h) carried through otherwise. */
int vqa_lstm_step_fwd(float* gates, const float* c_prev, const float* h_prev, const int32_t* len, int t, float* c_new,
                      float* h_new, int N, int L, void* stream);
/* its backward: dh, dc wrt the step's (h, c) -> dgates [N,4L] wrt the pre-activations, dc_prev, and dh_carry (dh of the
 * finished rows; the caller adds dgates K_h^T) */
int vqa_lstm_step_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* c_new,
                      const int32_t* len, int t, float* dgates, float* dc_prev, float* dh_carry, int N, int L, void* stream);
int vqa_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int vqa_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
int vqa_fill(float* x, int64_t n, float value, void* stream);
/* vqa/model_vqa.py:232-257 without the [B,A,L] intermediate: z[b,a] = sum_k w[k] tanh(al[a,k] + pq[b,k]) + bias[0]
 * (al = answer_layer1 [A,L], pq = pooled_layer1 + q_layer1 [B,L]); backward: d_al [A,L], d_pq [B,L] and the per-question
 * partials part_dw [B,L] of the classifier weight gradient (reduce with vqa_colsum); L <= 1024. */
int vqa_score_fwd(const float* al, const float* pq, const float* w, const float* bias, float* z, int B, int A, int L,
                  void* stream);
int vqa_score_bwd(const float* dz, const float* al, const float* pq, const float* w, float* d_al, float* d_pq,
                  float* part_dw, int B, int A, int L, void* stream);

/* ------------------------------------------------------------------------
 * Whole-step hipGraph capture / replay (csrc/graph.hip).  Every whole-model entry point, vqa_sumsq and
 * vqa_clip_adam_dev only enqueue work on the caller's stream (side streams are forked / joined with events), so:
 *     vqa_graph_capture_begin(stream); <one whole train step on `stream`>; vqa_graph_capture_end(stream, &exec, &n);
 *     every later step: refresh the inputs IN PLACE (same device addresses), then vqa_graph_launch(exec, stream).
 * Kernel arguments are frozen at capture: buffers must stay where they are, per-step scalars live in device memory.
 * `stream` must not be the NULL stream.  n_nodes_out (may be NULL) receives the node count of the captured graph.
 * ------------------------------------------------------------------------ */
int vqa_graph_capture_begin(void* stream);
int vqa_graph_capture_end(void* stream, void** exec_out, int* n_nodes_out);
int vqa_graph_capture_abort(void* stream);
int vqa_graph_launch(void* exec, void* stream);
int vqa_graph_destroy(void* exec);
int vqa_stream_is_capturing(void* stream);      /* 1 while `stream` is in capture */

/* ------------------------------------------------------------------------
 * Kernels of the five older model_vlmap_answer ablations (csrc/ablation_ops.hip); the whole-model entry points
 * compose them for model_type 7..11.
 * ------------------------------------------------------------------------ */
/* vqa/model_vlmap_answer_full.py:128-134, 272-276: x = mean + noise * sqrt(exp(log_sigma_sq)) [B,H];
 * kl_row[b] = -0.5 * sum_h (1 + log_sigma_sq - mean^2 - exp(log_sigma_sq)) (latent_loss = mean_b kl_row). */
int vqa_reparam_fwd(const float* mean, const float* log_sigma_sq, const float* noise, float* x, float* kl_row, int B, int H,
                    void* stream);
/* its backward plus the KL term's own gradient, coef = latent weight / global batch:
 * dmean = dx + coef * mean; dlog_sigma_sq = dx * noise * sqrt(exp(ls)) / 2 + coef * (exp(ls) - 1) / 2. */
int vqa_reparam_bwd(const float* dx, const float* mean, const float* log_sigma_sq, const float* noise, float coef,
                    float* dmean, float* dlog_sigma_sq, int64_t n, void* stream);
/* out[b,r,:] = att[b,r] * dp[b,:]: gradient of attention_pooling (vlmap/modules.py:23-39) wrt the pooled memory, needed
 * when the memory is the trainable v_adapt (vqa/model_vlmap_answer_adapt.py:142) instead of the input V_ft. */
int vqa_outer_rows(const float* att, const float* dp, float* out, int B, int R, int H, void* stream);
/* vqa/model_vlmap_answer_ent.py:196-199: x[(b,m),:] = pl[(b*M+m) % B,:] * ll[b,:] -- the tf.tile([M,1]) + reshape pairing
 * times the broadcast l_linear_l, without materialising the tile; and the gradient wrt ll (pl is behind
 * tf.stop_gradient): dll[b,:] (+)= sum_m dx[(b,m),:] * pl[(b*M+m) % B,:]. */
int vqa_tile_mul_fwd(const float* pl, const float* ll, float* x, int B, int M, int H, void* stream);
int vqa_tile_mul_bwd(const float* dx, const float* pl, float* dll, int B, int M, int H, int accumulate, void* stream);
/* :205-211, 281-284 on the pairings' logits tz [B*M, ldz], IN PLACE: softmax over the answers a < cols with
 * train_mask[a] * exist_mask[a] > 0.5, marginal[b,a] = mean over the M pairings [B,cols] (0 for excluded answers),
 * ent_row[b] = sum_a marginal * log(marginal + 1e-8).  want_dz != 0: tz <- d loss / d logit with
 * coef = W_ENTROPY / global batch; otherwise tz <- the pairings' probabilities.  cols <= 4096. */
int vqa_marginal_entropy(float* tz, const float* train_mask, const float* exist_mask, float coef, float* marginal,
                         float* ent_row, int B, int M, int cols, int ldz, int want_dz, void* stream);
/* explicit, reproducible stand-in for tf.random_normal(seed=123) (vqa/model_vlmap_answer_full.py:133): out[i] is a
 * standard-normal draw that depends only on (seed, offset + i) */
int vqa_normal_noise(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* report[13] = mean_b extra_row[b], report[14] = weight * report[13], report[15] = report[0] + report[14] (run it after
 * vqa_report_reduce); stats[b,15] = extra_row[b]. */
int vqa_extra_report(const float* extra_row, float* stats, int B, float weight, float* report, void* stream);

/* The same in dependency-ordered phases (bit mask): 1 = head..attention..v_linear_v / q_linear_v /
 * score gradients, 2 = GRU BPTT + embedding gradient + slice sum of squares, 4 = GRU gate weight / bias
 * gradients, 8 = GRU candidate weight / bias gradients (15 = everything).  A data-parallel caller launches one phase, starts the all-reduce of the bucket that
 * phase completed, and launches the next phase meanwhile. */
int vqa_fusion_backward_phases(const vqa_dims_t* dims, const vqa_params_t* params, const vqa_params_t* grads,
                               const vqa_batch_t* batch, void* workspace, int64_t workspace_bytes,
                               float* embed_slice_sq, int phases, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQA_HOT_H */
