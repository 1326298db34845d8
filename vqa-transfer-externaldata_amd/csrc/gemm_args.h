// Argument blocks shared by the GEMM kernels of libvqahot.so (gemm_f32.hip and the register-streamed GRU step kernels of
// gru_stream.hip).  Internal: nothing here is part of the C ABI.
#pragma once
#include <stdint.h>

struct GemmArgs {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* D; int ldd;
    int k_per_split;      // multiple of 64 (>= every BK)
    int64_t slab_stride;  // floats between split-k slabs of C (0 when split_k == 1)
    int vecA, vecB;       // 16-byte global loads allowed for A / B
    int tiles_m, tiles_n, nsplit;   // 1-D grid: workgroups walk the (split, m, n) tiles
    int m_fastest;                  // tile order inside a split: m fastest (1) or n fastest (0)
    unsigned a_bytes, b_bytes;      // operand extents for the buffer descriptors (0 => use the generic loader)
    const float* scale;             // per-column scale (folded BatchNorm), applied to the accumulator
    int relu;                       // ReLU after scale / bias / addend
    // implicit-GEMM convolution (A = NHWC activations, row m = output pixel, k = (ky, kx, ci))
    int Hi, Wi, Ci, Ho, Wo, cstride, pad_t, pad_l, kw;
    int conv_taps;                  // kh * kw
    int vec_epi;                    // plain epilogue may use 16-byte accesses
    // row-gathered A (GATHER kernels): logical row m of the left operand is row g_idx[m / g_R] * g_R + m % g_R of the
    // table at A (the feature gather V_ft = features[image_idx] fused into v_linear_v's operand load); g_out, when
    // set, receives the gathered rows densely [M, K] as a by-product (written by the workgroups of column panel 0)
    const int64_t* g_idx; int g_R; int64_t g_N; float* g_out; int g_ldo;
};

// Fused epilogues of the GRU recurrence (tf.contrib.rnn.GRUCell, vlmap/modules.py:129-135):
//  EPI_GATES  g = acc + D ; s = sigmoid(g) ; col <  H: r = s, rh = s*h_prev ; col >= H: u = s
//  EPI_CAND   c = tanh(acc + D) ; h_new = (t < len) ? u*h_prev + (1-u)*c : h_prev
//  EPI_BWD_RH drh = acc ; dr_pre = drh*h_prev*r*(1-r) -> o0[row*ldo + col] ; dh_acc += drh*r
//  EPI_BWD_DH dh = acc + dh_acc (the gradient wrt h of step t-1) and, fused, the first half of
//             step t-1's backward: dc_pre, du_pre, dh_acc' (see vqa_gru_bwd_a)
enum { EPI_PLAIN = 0, EPI_GATES = 1, EPI_CAND = 2, EPI_BWD_RH = 3, EPI_BWD_DH = 4 };
struct EpiArgs {
    int H, t, ldo;
    const int32_t* len;
    const float* h_prev;   // [B,H]
    const float* i0;       // GATES: -      CAND: u        BWD_RH: r        BWD_DH: u (step t-1)
    const float* i1;       //                                               BWD_DH: c (step t-1)
    float* o0;             // GATES: r      CAND: c        BWD_RH: dr_pre   BWD_DH: dc_pre (ld ldo)
    float* o1;             // GATES: u      CAND: h_new    BWD_RH: dh_acc   BWD_DH: du_pre (ld ldo)
    float* o2;             // GATES: rh                                      BWD_DH: dh_acc
};

