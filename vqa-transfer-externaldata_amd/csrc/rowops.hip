// HBM-bound row / elementwise kernels of the fusion model (gfx950):
// feature-table gather (a1), embedding gather + scatter-add (a3), Hadamard
// products, GRU gate math (a4), column sums for bias / LN-parameter gradients,
// and the reproducible dropout keep-mask.
// All loads/stores are 16 B per lane where the row length allows it.
#include "vqa_common.h"

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int grid_for(int64_t work_items, int cap = 4096) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(work_items, 256), cap));
}

// ------------------------------------------------------------------ a1
// One workgroup per (sample, region-chunk): rows of D floats are copied with
// float4 loads; the table row index is wave-uniform.
__global__ __launch_bounds__(256) void gather_features_kernel(const float* __restrict__ table,
                                                              const int32_t* __restrict__ nbox_table,
                                                              const int64_t* __restrict__ idx, float* __restrict__ V,
                                                              int32_t* __restrict__ nb, int B, int64_t row_floats,
                                                              int64_t N) {
    const int b = blockIdx.x;
    int64_t src = idx[b];
    if (src < 0) src = 0;
    if (src >= N) src = N - 1;  // np.take would raise; clamp instead of faulting
    if (blockIdx.y == 0 && threadIdx.x == 0 && nb != nullptr) nb[b] = nbox_table[src];
    const float* s = table + src * row_floats;
    float* d = V + (int64_t)b * row_floats;
    const int64_t n4 = row_floats / 4;
    for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.y * 256)
        reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
    if (blockIdx.y == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < row_floats; i += 256) d[i] = s[i];
}

// ------------------------------------------------------------------ a3
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ E, const int32_t* __restrict__ q,
                                                        float* __restrict__ x, int B, int T, int W, int Vq, int ldx) {
    // one wave per token; x is time-major [T,B,ldx]; columns W .. ldx-1 (if any) are 1, 0, 0, ...: the constant input
    // whose "weight" row is the bias, so the x-part weight-gradient GEMM also delivers the bias gradient
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= B * T) return;
    const int t = tok / B, b = tok % B;
    int id = q[b * T + t];
    id = min(max(id, 0), Vq - 1);
    const float* s = E + (int64_t)id * W;
    float* d = x + (int64_t)tok * ldx;
    for (int i = threadIdx.x & 63; i < ldx; i += 64) d[i] = i < W ? s[i] : (i == W ? 1.f : 0.f);
}

// Deterministic mode (vqa_set_deterministic): scatter-add of the embedding gradient WITHOUT atomics -- one wave
// owns up to EMB_ROWS vocabulary rows, scans the B*T token ids once (a few tens of KB, cache resident) and adds the matching dx rows
// into its LDS accumulators in ascending token order -- deterministic, and a frequent word costs no
// atomic contention (its rows are fetched eight at a time: independent loads, fixed addition order).
// Positions past a sequence's length (zero padding; their dx is exactly zero) are skipped when `len`
// is given, so the padding id is not a hot row.
constexpr int EMB_ROWS = 16;
template <int NK>   // NK * 64 >= W
__global__ __launch_bounds__(256) void embed_bwd_owner_kernel(const float* __restrict__ dx, const int32_t* __restrict__ q,
                                                              const int32_t* __restrict__ len, float* __restrict__ dE,
                                                              int B, int T, int W, int Vq, int nw_log2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [4 waves][EMB_ROWS][NK * 64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave w of nw_log2-many owns rows {w, w + NW, w + 2 NW, ...}: frequent words (small ids in a
    // frequency-sorted vocabulary) land on different waves
    const int w = blockIdx.x * 4 + wave, NW = 1 << nw_log2;
    if (w >= NW) return;                                          // no workgroup barrier below
    float* acc = lds + (size_t)wave * EMB_ROWS * NK * 64;
#pragma unroll
    for (int i = 0; i < EMB_ROWS * NK; ++i) acc[i * 64 + lane] = 0.f;
    unsigned touched = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const int lim = (b < B) ? (len != nullptr ? min(len[b], T) : T) : 0;
        for (int t = 0; t < T; ++t) {
            int rid = -1;
            if (t < lim) {
                const int id = min(max(q[b * T + t], 0), Vq - 1);
                if ((id & (NW - 1)) == w) rid = id >> nw_log2;
            }
            unsigned long long m = __ballot(rid >= 0);
            while (m) {
                int j[8], rr[8];
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    j[u] = 0; rr[u] = 0;
                    if (m) {
                        j[u] = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        rr[u] = __shfl(rid, j[u], 64);
                        ++cnt;
                    }
                }
                float r[8][NK];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* s = dx + ((int64_t)t * B + b0 + j[u]) * W;
#pragma unroll
                    for (int k = 0; k < NK; ++k) r[u][k] = (u < cnt && lane + 64 * k < W) ? s[lane + 64 * k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < cnt) {
                        float* a = acc + rr[u] * NK * 64 + lane;
#pragma unroll
                        for (int k = 0; k < NK; ++k) a[64 * k] += r[u][k];
                        touched |= 1u << rr[u];
                    }
            }
        }
    }
    for (int rl = 0; rl < EMB_ROWS; ++rl) {
        if (!((touched >> rl) & 1u)) continue;
        float* d = dE + (int64_t)(w + (rl << nw_log2)) * W;
        const float* a = acc + rl * NK * 64 + lane;
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (lane + 64 * k < W) d[lane + 64 * k] += a[64 * k];
    }
}

// default: float atomics, one wave per token
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, const int32_t* __restrict__ q,
                                                        const int32_t* __restrict__ len, float* __restrict__ dE, int B,
                                                        int T, int W, int Vq) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= B * T) return;
    const int t = tok / B, b = tok % B;
    if (len != nullptr && t >= len[b]) return;
    int id = q[b * T + t];
    id = min(max(id, 0), Vq - 1);
    const float* s = dx + (int64_t)tok * W;
    float* d = dE + (int64_t)id * W;
    for (int i = threadIdx.x & 63; i < W; i += 64) atomicAdd(d + i, s[i]);
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ z, int64_t n) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) z[i] = a[i] * b[i];
}
__global__ __launch_bounds__(256) void mul_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                                      const float* __restrict__ b, float* __restrict__ da,
                                                      float* __restrict__ db, int64_t n) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = dz[i], av = a[i], bv = b[i];
        da[i] = g * bv;
        db[i] = g * av;
    }
}
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ acc, const float* __restrict__ x,
                                                          int64_t n) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc[i] += x[i];
}

__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = tanhf(x[i]);
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, int64_t n) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float t = y[i];
        dx[i] = dy[i] * (1.f - t * t);
    }
}

// ------------------------------------------------------------------ a4 (GRU gate math)
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float* __restrict__ gpre, int ldg,
                                                            const float* __restrict__ h_prev, float* __restrict__ r,
                                                            float* __restrict__ u, float* __restrict__ rh, int B,
                                                            int H) {
    const int64_t n = (int64_t)B * H;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / H), k = (int)(i % H);
        const float rv = sigmoidf_stable(gpre[(int64_t)b * ldg + k]);
        const float uv = sigmoidf_stable(gpre[(int64_t)b * ldg + H + k]);
        r[i] = rv;
        u[i] = uv;
        rh[i] = rv * h_prev[i];
    }
}
__global__ __launch_bounds__(256) void gru_cand_fwd_kernel(const float* __restrict__ cpre, int ldc,
                                                           const float* __restrict__ u,
                                                           const float* __restrict__ h_prev,
                                                           const int32_t* __restrict__ len, int t,
                                                           float* __restrict__ c, float* __restrict__ h_new, int B,
                                                           int H) {
    const int64_t n = (int64_t)B * H;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / H), k = (int)(i % H);
        const float cv = tanhf(cpre[(int64_t)b * ldc + k]);
        const float hp = h_prev[i], uv = u[i];
        c[i] = cv;
        h_new[i] = (t < len[b]) ? (uv * hp + (1.f - uv) * cv) : hp;
    }
}
__global__ __launch_bounds__(256) void gru_bwd_a_kernel(const float* __restrict__ dh, const float* __restrict__ h_prev,
                                                        const float* __restrict__ u, const float* __restrict__ c,
                                                        const int32_t* __restrict__ len, int t,
                                                        float* __restrict__ dc_pre, int ld_dc,
                                                        float* __restrict__ du_pre, int ld_du,
                                                        float* __restrict__ dh_acc, int B, int H) {
    const int64_t n = (int64_t)B * H;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / H), k = (int)(i % H);
        const bool live = t < len[b];
        const float g = dh[i], uv = u[i], cv = c[i], hp = h_prev[i];
        dc_pre[(int64_t)b * ld_dc + k] = live ? g * (1.f - uv) * (1.f - cv * cv) : 0.f;
        du_pre[(int64_t)b * ld_du + k] = live ? g * (hp - cv) * uv * (1.f - uv) : 0.f;
        dh_acc[i] = live ? g * uv : g;
    }
}
__global__ __launch_bounds__(256) void gru_bwd_b_kernel(const float* __restrict__ drh, const float* __restrict__ h_prev,
                                                        const float* __restrict__ r, float* __restrict__ dr_pre,
                                                        int ld_dr, float* __restrict__ dh_acc, int B, int H) {
    const int64_t n = (int64_t)B * H;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / H), k = (int)(i % H);
        const float g = drh[i], rv = r[i];
        dr_pre[(int64_t)b * ld_dr + k] = g * h_prev[i] * rv * (1.f - rv);
        dh_acc[i] += g * rv;
    }
}

// Finished sequences (t >= len[b]) when the recurrence skipped them (vqa_gru_seq_*_live): forward, their state
// is carried (hs[t+1] = hs[len]) and r*h zeroed so the weight-gradient GEMMs over all T*B rows see defined values;
// backward, their pre-activation gradients are zero.
__global__ __launch_bounds__(256) void gru_fill_finished_kernel(float* __restrict__ hs, float* __restrict__ rh,
                                                                const int32_t* __restrict__ len, int T, int B, int H) {
    const int b = blockIdx.x, t = blockIdx.y;
    const int L = min(max(len[b], 0), T);
    if (t < L) return;
    const float4* src = reinterpret_cast<const float4*>(hs + ((int64_t)L * B + b) * H);
    float4* dst = reinterpret_cast<float4*>(hs + ((int64_t)(t + 1) * B + b) * H);
    float4* z = reinterpret_cast<float4*>(rh + ((int64_t)t * B + b) * H);
    for (int i = threadIdx.x; i < H / 4; i += 256) {
        dst[i] = src[i];
        z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__global__ __launch_bounds__(256) void gru_zero_finished_kernel(float* __restrict__ dxp, const int32_t* __restrict__ len,
                                                                int T, int B, int H3) {
    const int b = blockIdx.x, t = blockIdx.y;
    if (t < len[b]) return;
    float4* z = reinterpret_cast<float4*>(dxp + ((int64_t)t * B + b) * H3);
    for (int i = threadIdx.x; i < H3 / 4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ column sums
// partial[y][n] = sum over the rows of chunk y.  Block = 64 columns x 4 row lanes.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                     float* __restrict__ out, int rows_per_chunk, int accumulate = 0) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    float s = 0.f;
    if (col < N)
        for (int m = m0 + ry; m < m1; m += 4) s += X[(int64_t)m * ldx + col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N) {
        const float v = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
        float* o = out + (int64_t)blockIdx.y * N + col;
        *o = accumulate ? *o + v : v;           // accumulate: single-chunk / final pass only (one writer per column)
    }
}

// 16-byte form of the two kernels (N, ldx multiples of 4, 16-byte aligned): a thread owns FOUR columns and keeps four
// rows in flight; every column is still summed over its thread's rows in the same order (ry, ry + 4, ...), so the
// result is bit for bit that of the scalar kernels.  The scalar form issued one 4-byte load per thread and iteration
// and ran at 2.2-3.6 TB/s on the [25600, 3H] gate gradients.
typedef float cs4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void colsum4_body(const float* __restrict__ X, float* __restrict__ out, int M, int N, int ldx,
                                             int64_t out_row, int rows_per_chunk, int accumulate, cs4 (*red)[64]) {
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cx) * 4;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    cs4 s = (cs4)(0.f);
    if (col < N) {
        const cs4* __restrict__ p = reinterpret_cast<const cs4*>(X + col);
        const int64_t ld4 = ldx / 4;
        int m = m0 + ry;
        for (; m + 12 < m1; m += 16) {
            const cs4 a = p[(int64_t)m * ld4], b = p[(int64_t)(m + 4) * ld4], c = p[(int64_t)(m + 8) * ld4],
                      d = p[(int64_t)(m + 12) * ld4];
            s += a; s += b; s += c; s += d;
        }
        for (; m < m1; m += 4) s += p[(int64_t)m * ld4];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N) {
        const cs4 v = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
        cs4* o = reinterpret_cast<cs4*>(out + out_row + col);
        *o = accumulate ? *o + v : v;
    }
}
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                      float* __restrict__ out, int rows_per_chunk, int accumulate) {
    __shared__ cs4 red[4][64];
    colsum4_body(X, out, M, N, ldx, (int64_t)blockIdx.y * N, rows_per_chunk, accumulate, red);
}

// three equally shaped matrices per launch (blockIdx.z): the (d_gamma, d_beta, d_bias) partials of one
// LayerNorm block -- these reductions are launch-latency bound, so batching them is the whole gain
struct Colsum3 { const float* X[3]; float* out[3]; int acc[3]; };
__global__ __launch_bounds__(256) void colsum3_kernel(Colsum3 a, int M, int N, int ldx, int64_t out_stride_rows,
                                                      int rows_per_chunk) {
    __shared__ float red[4][64];
    const float* __restrict__ X = a.X[blockIdx.z];
    float* __restrict__ out = a.out[blockIdx.z];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    float s = 0.f;
    if (col < N)
        for (int m = m0 + ry; m < m1; m += 4) s += X[(int64_t)m * ldx + col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N) {
        const float v = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
        float* o = out + (int64_t)blockIdx.y * out_stride_rows + col;
        *o = a.acc[blockIdx.z] ? *o + v : v;
    }
}

__global__ __launch_bounds__(256) void colsum3x4_kernel(Colsum3 a, int M, int N, int ldx, int64_t out_stride_rows,
                                                        int rows_per_chunk) {
    __shared__ cs4 red[4][64];
    colsum4_body(a.X[blockIdx.z], a.out[blockIdx.z], M, N, ldx, (int64_t)blockIdx.y * out_stride_rows, rows_per_chunk,
                 a.acc[blockIdx.z], red);
}

// ------------------------------------------------------------------ dropout keep-mask
// Counter-based generator (splitmix64 finaliser on seed ^ counter): the mask of
// element i depends only on (seed, offset + i), so forward, backward and the test
// harness regenerate identical masks.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// one 64-bit hash serves the four positions of an aligned group (16-bit uniforms: keep probabilities resolve to
// 1.5e-5): the generator, not the 1 byte per element it writes, bounded the kernel at 1.3 TB/s with a hash per position
__device__ __forceinline__ unsigned keep_thr(float keep) {
    const float t = keep * 65536.0f;
    return t <= 0.f ? 0u : (t >= 65536.0f ? 65536u : (unsigned)t);
}
__device__ __forceinline__ unsigned keep_bit(uint64_t key, uint64_t pos, unsigned thr) {
    const uint64_t r = mix64(key ^ (pos >> 2));
    return ((unsigned)(r >> (16 * (pos & 3))) & 0xFFFFu) < thr ? 1u : 0u;
}
// 16 mask bytes per lane and store (one 16-byte store instead of sixteen 1-byte ones); `out` 16-byte aligned
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n, uint64_t seed,
                                                           uint64_t offset, float keep) {
    const uint64_t key = mix64(seed);
    const unsigned thr = keep_thr(keep);
    const int64_t n16 = n / 16;
    if ((offset & 3) == 0) {          // position groups line up with the 4-byte words: one hash per word
        for (int64_t g = blockIdx.x * 256 + threadIdx.x; g < n16; g += (int64_t)gridDim.x * 256) {
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t r = mix64(key ^ ((offset + (uint64_t)(g * 16 + q * 4)) >> 2));
                unsigned v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) v |= ((((unsigned)(r >> (16 * b))) & 0xFFFFu) < thr ? 1u : 0u) << (8 * b);
                w[q] = v;
            }
            reinterpret_cast<uint4*>(out)[g] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int64_t g = blockIdx.x * 256 + threadIdx.x; g < n16; g += (int64_t)gridDim.x * 256) {
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) v |= keep_bit(key, offset + (uint64_t)(g * 16 + q * 4 + b), thr) << (8 * b);
                w[q] = v;
            }
            reinterpret_cast<uint4*>(out)[g] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    for (int64_t i = n16 * 16 + blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (uint8_t)keep_bit(key, offset + (uint64_t)i, thr);
}
__global__ __launch_bounds__(256) void dropout_mask_bytes_kernel(uint8_t* __restrict__ out, int64_t n, uint64_t seed,
                                                                 uint64_t offset, float keep) {
    const uint64_t key = mix64(seed);
    const unsigned thr = keep_thr(keep);
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (uint8_t)keep_bit(key, offset + (uint64_t)i, thr);
}

// The x rows (first W) of the two GRU kernels -- gates [W+H, 2H], candidate [W+H, H] -- side by side as one [W, 3H]
// matrix (+ the two biases as one [3H] vector), so that the input projection of all steps, its gradient dx and the
// x-part of the weight gradient are ONE GEMM each against the [T*B, 3H] projection block instead of two with
// K (or N) = 300: three launches and an accumulate pass less per step.  Repacked every step (the weights move).
__global__ __launch_bounds__(256) void pack_wx_kernel(const float* __restrict__ wg, const float* __restrict__ wc,
                                                      const float* __restrict__ bg, const float* __restrict__ bc,
                                                      float* __restrict__ wx, float* __restrict__ bx, int W, int H) {
    const int H3 = 3 * H;
    const int64_t n = (int64_t)W * H3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n + H3; i += (int64_t)gridDim.x * 256) {
        if (i < n) {
            const int r = (int)(i / H3), col = (int)(i - (int64_t)r * H3);
            wx[i] = col < 2 * H ? wg[(int64_t)r * 2 * H + col] : wc[(int64_t)r * H + (col - 2 * H)];
        } else {
            const int col = (int)(i - n);
            bx[col] = col < 2 * H ? bg[col] : bc[col - 2 * H];
        }
    }
}
// gradient of the packed matrix back into the x rows of the two kernels' gradients
__global__ __launch_bounds__(256) void unpack_dwx_kernel(const float* __restrict__ dwx, float* __restrict__ gwg,
                                                         float* __restrict__ gwc, float* __restrict__ gbg,
                                                         float* __restrict__ gbc, int W, int H) {
    const int H3 = 3 * H;
    const int64_t n = (int64_t)(W + (gbg != nullptr ? 1 : 0)) * H3;      // row W (if asked for): the bias gradients
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / H3), col = (int)(i - (int64_t)r * H3);
        if (r < W) {
            if (col < 2 * H) gwg[(int64_t)r * 2 * H + col] = dwx[i];
            else gwc[(int64_t)r * H + (col - 2 * H)] = dwx[i];
        } else {
            if (col < 2 * H) gbg[col] = dwx[i];
            else gbc[col - 2 * H] = dwx[i];
        }
    }
}
}  // namespace

extern "C" int vqa_gru_pack_wx(const float* wg, const float* wc, const float* bg, const float* bc, float* wx, float* bx, int W,
                               int H, void* stream) {
    VQA_REQUIRE(wg && wc && bg && bc && wx && bx && W > 0 && H > 0, VQA_ERR_ARG);
    const int64_t n = (int64_t)W * 3 * H + 3 * H;
    hipLaunchKernelGGL(pack_wx_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, wg, wc, bg, bc, wx, bx, W, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_unpack_dwx(const float* dwx, float* gwg, float* gwc, int W, int H, void* stream) {
    return vqa_gru_unpack_dwx_bias(dwx, gwg, gwc, nullptr, nullptr, W, H, stream);
}

extern "C" int vqa_gru_unpack_dwx_bias(const float* dwx, float* gwg, float* gwc, float* gbg, float* gbc, int W, int H,
                                       void* stream) {
    VQA_REQUIRE(dwx && gwg && gwc && W > 0 && H > 0 && (gbg == nullptr) == (gbc == nullptr), VQA_ERR_ARG);
    const int64_t n = (int64_t)(W + 1) * 3 * H;
    hipLaunchKernelGGL(unpack_dwx_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, dwx, gwg, gwc, gbg, gbc, W, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gather_features(const float* table, const int32_t* nbox_table, const int64_t* idx, float* V,
                                   int32_t* nb, int B, int R, int D, int64_t N, void* stream) {
    VQA_REQUIRE(table && idx && (V || nb) && B >= 0 && R > 0 && D > 0 && N > 0, VQA_ERR_ARG);
    VQA_REQUIRE(nb == nullptr || nbox_table != nullptr, VQA_ERR_ARG);
    VQA_REQUIRE(vqa_aligned16(table) && vqa_aligned16(V) && ((int64_t)R * D) % 4 == 0, VQA_ERR_ALIGN);
    if (B == 0) return VQA_OK;
    const int64_t row = V != nullptr ? (int64_t)R * D : 0;      // V == NULL: num_boxes only
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(row / 4, 256 * 4), 16));
    hipLaunchKernelGGL(gather_features_kernel, dim3(B, chunks), dim3(256), 0, (hipStream_t)stream, table, nbox_table,
                       idx, V, nb, B, row, N);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_embed_fwd(const float* E, const int32_t* q, float* x_tm, int B, int T, int W, int Vq,
                             void* stream) {
    VQA_REQUIRE(E && q && x_tm && B >= 0 && T >= 0 && W > 0 && Vq > 0, VQA_ERR_ARG);
    if (B * T == 0) return VQA_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, E, q, x_tm, B, T,
                       W, Vq, W);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_embed_fwd_ld(const float* E, const int32_t* q, float* x_tm, int B, int T, int W, int Vq, int ldx,
                                void* stream) {
    VQA_REQUIRE(E && q && x_tm && B >= 0 && T >= 0 && W > 0 && Vq > 0 && ldx >= W, VQA_ERR_ARG);
    if (B * T == 0) return VQA_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, E, q, x_tm, B, T,
                       W, Vq, ldx);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
namespace {
int g_deterministic = 0;
}
extern "C" int vqa_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return VQA_OK;
}
extern "C" int vqa_embed_bwd_len(const float* dx_tm, const int32_t* q, const int32_t* len, float* dE, int B, int T,
                                 int W, int Vq, void* stream) {
    return vqa_embed_bwd_len_det(dx_tm, q, len, dE, B, T, W, Vq, -1, stream);
}
extern "C" int vqa_embed_bwd_len_det(const float* dx_tm, const int32_t* q, const int32_t* len, float* dE, int B, int T,
                                     int W, int Vq, int deterministic, void* stream) {
    VQA_REQUIRE(dx_tm && q && dE && B >= 0 && T >= 0 && W > 0 && Vq > 0, VQA_ERR_ARG);
    if (B * T == 0) return VQA_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool det = deterministic < 0 ? g_deterministic != 0 : deterministic != 0;
    if (!det || W > 512) {
        // float atomics: the L2 serialises a frequent word's adds per address at ~10 ns each, so skew costs
        // little; the summation order (hence the last bit) varies from run to run, as it does in the reference
        hipLaunchKernelGGL(embed_bwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, st, dx_tm, q, len, dE, B, T, W, Vq);
        VQA_CHECK_LAUNCH();
        return VQA_OK;
    }
    int nw_log2 = 0;
    while ((EMB_ROWS << nw_log2) < Vq) ++nw_log2;
    const int grid = ((1 << nw_log2) + 3) / 4;
    if (W <= 320) {
        auto kern = embed_bwd_owner_kernel<5>;
        constexpr int lds = 4 * EMB_ROWS * 5 * 64 * (int)sizeof(float);     // 80 KiB > the 64 KiB default limit
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess)
                return VQA_ERR_LAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dx_tm, q, len, dE, B, T, W, Vq, nw_log2);
    } else {
        auto kern = embed_bwd_owner_kernel<8>;
        constexpr int lds = 4 * EMB_ROWS * 8 * 64 * (int)sizeof(float);     // 128 KiB
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess)
                return VQA_ERR_LAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dx_tm, q, len, dE, B, T, W, Vq, nw_log2);
    }
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_embed_bwd(const float* dx_tm, const int32_t* q, float* dE, int B, int T, int W, int Vq,
                             void* stream) {
    return vqa_embed_bwd_len(dx_tm, q, nullptr, dE, B, T, W, Vq, stream);
}

extern "C" int vqa_mul(const float* a, const float* b, float* z, int64_t n, void* stream) {
    VQA_REQUIRE(a && b && z && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, z, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_mul_bwd(const float* dz, const float* a, const float* b, float* da, float* db, int64_t n,
                           void* stream) {
    VQA_REQUIRE(dz && a && b && da && db && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(mul_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dz, a, b, da, db, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
    VQA_REQUIRE(x && y && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_tanh_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    VQA_REQUIRE(dy && y && dx && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_add_inplace(float* acc, const float* x, int64_t n, void* stream) {
    VQA_REQUIRE(acc && x && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, acc, x, n);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_gates_fwd(const float* gpre, int ldg, const float* h_prev, float* r, float* u, float* rh,
                                 int B, int H, void* stream) {
    VQA_REQUIRE(gpre && h_prev && r && u && rh && ldg >= 2 * H && B >= 0 && H > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(grid_for((int64_t)B * H)), dim3(256), 0, (hipStream_t)stream, gpre,
                       ldg, h_prev, r, u, rh, B, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_gru_cand_fwd(const float* cpre, int ldc, const float* u, const float* h_prev, const int32_t* len,
                                int t, float* c, float* h_new, int B, int H, void* stream) {
    VQA_REQUIRE(cpre && u && h_prev && len && c && h_new && ldc >= H && B >= 0 && H > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(gru_cand_fwd_kernel, dim3(grid_for((int64_t)B * H)), dim3(256), 0, (hipStream_t)stream, cpre,
                       ldc, u, h_prev, len, t, c, h_new, B, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_gru_bwd_a(const float* dh, const float* h_prev, const float* u, const float* c,
                             const int32_t* len, int t, float* dc_pre, int ld_dc, float* du_pre, int ld_du,
                             float* dh_acc, int B, int H, void* stream) {
    VQA_REQUIRE(dh && h_prev && u && c && len && dc_pre && du_pre && dh_acc && ld_dc >= H && ld_du >= H && B >= 0 &&
                    H > 0,
                VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(gru_bwd_a_kernel, dim3(grid_for((int64_t)B * H)), dim3(256), 0, (hipStream_t)stream, dh, h_prev,
                       u, c, len, t, dc_pre, ld_dc, du_pre, ld_du, dh_acc, B, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_gru_bwd_b(const float* drh, const float* h_prev, const float* r, float* dr_pre, int ld_dr,
                             float* dh_acc, int B, int H, void* stream) {
    VQA_REQUIRE(drh && h_prev && r && dr_pre && dh_acc && ld_dr >= H && B >= 0 && H > 0, VQA_ERR_ARG);
    if (B == 0) return VQA_OK;
    hipLaunchKernelGGL(gru_bwd_b_kernel, dim3(grid_for((int64_t)B * H)), dim3(256), 0, (hipStream_t)stream, drh,
                       h_prev, r, dr_pre, ld_dr, dh_acc, B, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_fill_finished(float* hs, float* rh, const int32_t* len, int T, int B, int H, void* stream) {
    VQA_REQUIRE(hs && rh && len && T >= 0 && B >= 0 && H > 0, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0 && vqa_aligned16(hs) && vqa_aligned16(rh), VQA_ERR_ALIGN);
    if (T == 0 || B == 0) return VQA_OK;
    hipLaunchKernelGGL(gru_fill_finished_kernel, dim3(B, T), dim3(256), 0, (hipStream_t)stream, hs, rh, len, T, B, H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_gru_zero_finished(float* dxp, const int32_t* len, int T, int B, int H, void* stream) {
    VQA_REQUIRE(dxp && len && T >= 0 && B >= 0 && H > 0, VQA_ERR_ARG);
    VQA_REQUIRE(H % 4 == 0 && vqa_aligned16(dxp), VQA_ERR_ALIGN);
    if (T == 0 || B == 0) return VQA_OK;
    hipLaunchKernelGGL(gru_zero_finished_kernel, dim3(B, T), dim3(256), 0, (hipStream_t)stream, dxp, len, T, B, 3 * H);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}

namespace {
int colsum_chunks(int M) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(M, 64), 64)); }
}
extern "C" int64_t vqa_colsum_workspace_floats(int M, int N) {
    const int ch = colsum_chunks(M);
    return ch > 1 ? (int64_t)ch * N : 0;
}
extern "C" int vqa_colsum_acc(const float* X, int M, int N, int ldx, float* out, int accumulate, float* workspace,
                              int64_t workspace_floats, void* stream) {
    VQA_REQUIRE(X && out && M >= 0 && N > 0 && ldx >= N, VQA_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    const int ch = colsum_chunks(std::max(M, 1));
    const int acc = accumulate ? 1 : 0;
    const bool v4 = (N % 4 == 0) && (ldx % 4 == 0) && vqa_aligned16(X) && vqa_aligned16(out) &&
                    (workspace == nullptr || vqa_aligned16(workspace));
    const int gx = v4 ? (N / 4 + 63) / 64 : (N + 63) / 64;
#define VQA_COLSUM_LAUNCH(gy, x, m, ld, o, rpc_, ac)                                                                    \
    do {                                                                                                                \
        if (v4) hipLaunchKernelGGL(colsum4_kernel, dim3(gx, gy), dim3(256), 0, st, x, m, N, ld, o, rpc_, ac);           \
        else hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, st, x, m, N, ld, o, rpc_, ac);               \
        VQA_CHECK_LAUNCH();                                                                                             \
    } while (0)
    if (ch == 1 || workspace == nullptr) {
        VQA_COLSUM_LAUNCH(1, X, M, ldx, out, std::max(M, 1), acc);
        return VQA_OK;
    }
    VQA_REQUIRE(workspace_floats >= (int64_t)ch * N, VQA_ERR_WORKSPACE);
    const int rpc = (int)cdiv(M, ch);
    const int ch2 = (int)cdiv(M, rpc);
    VQA_COLSUM_LAUNCH(ch2, X, M, ldx, workspace, rpc, 0);
    VQA_COLSUM_LAUNCH(1, workspace, ch2, N, out, ch2, acc);
#undef VQA_COLSUM_LAUNCH
    return VQA_OK;
}

extern "C" int vqa_colsum(const float* X, int M, int N, int ldx, float* out, float* workspace,
                          int64_t workspace_floats, void* stream) {
    return vqa_colsum_acc(X, M, N, ldx, out, 0, workspace, workspace_floats, stream);
}

extern "C" int vqa_colsum3(const float* X0, const float* X1, const float* X2, int M, int N, int ldx, float* out0,
                           float* out1, float* out2, float* workspace, int64_t workspace_floats, void* stream) {
    return vqa_colsum3_acc(X0, X1, X2, M, N, ldx, out0, out1, out2, 0, workspace, workspace_floats, stream);
}

extern "C" int vqa_colsum3_acc(const float* X0, const float* X1, const float* X2, int M, int N, int ldx, float* out0,
                               float* out1, float* out2, int acc_mask, float* workspace, int64_t workspace_floats,
                               void* stream) {
    VQA_REQUIRE(X0 && X1 && X2 && out0 && out1 && out2 && M >= 0 && N > 0 && ldx >= N, VQA_ERR_ARG);
    hipStream_t st = (hipStream_t)stream;
    const int ch = colsum_chunks(std::max(M, 1));
    const bool v4 = (N % 4 == 0) && (ldx % 4 == 0) && vqa_aligned16(X0) && vqa_aligned16(X1) && vqa_aligned16(X2) &&
                    vqa_aligned16(out0) && vqa_aligned16(out1) && vqa_aligned16(out2) &&
                    (workspace == nullptr || vqa_aligned16(workspace));
    const int gx = v4 ? (N / 4 + 63) / 64 : (N + 63) / 64;
    const int a0 = acc_mask & 1, a1 = (acc_mask >> 1) & 1, a2 = (acc_mask >> 2) & 1;
    Colsum3 a{{X0, X1, X2}, {out0, out1, out2}, {a0, a1, a2}};
#define VQA_COLSUM3_LAUNCH(gy, args, m, ld, rpc_)                                                                        \
    do {                                                                                                                \
        if (v4) hipLaunchKernelGGL(colsum3x4_kernel, dim3(gx, gy, 3), dim3(256), 0, st, args, m, N, ld, (int64_t)N, rpc_); \
        else hipLaunchKernelGGL(colsum3_kernel, dim3(gx, gy, 3), dim3(256), 0, st, args, m, N, ld, (int64_t)N, rpc_);   \
        VQA_CHECK_LAUNCH();                                                                                             \
    } while (0)
    if (ch == 1) {
        VQA_COLSUM3_LAUNCH(1, a, M, ldx, std::max(M, 1));
        return VQA_OK;
    }
    VQA_REQUIRE(workspace && workspace_floats >= 3 * (int64_t)ch * N, VQA_ERR_WORKSPACE);
    const int rpc = (int)cdiv(M, ch);
    const int ch2 = (int)cdiv(M, rpc);
    const int64_t zs = (int64_t)ch * N;
    Colsum3 p1{{X0, X1, X2}, {workspace, workspace + zs, workspace + 2 * zs}, {0, 0, 0}};
    VQA_COLSUM3_LAUNCH(ch2, p1, M, ldx, rpc);
    Colsum3 p2{{workspace, workspace + zs, workspace + 2 * zs}, {out0, out1, out2}, {a0, a1, a2}};
    VQA_COLSUM3_LAUNCH(1, p2, ch2, N, ch2);
#undef VQA_COLSUM3_LAUNCH
    return VQA_OK;
}

extern "C" int vqa_dropout_mask(uint8_t* out, int64_t n, uint64_t seed, uint64_t offset, float keep_prob,
                                void* stream) {
    VQA_REQUIRE(out && n >= 0, VQA_ERR_ARG);
    if (n == 0) return VQA_OK;
    if (vqa_aligned16(out))
        hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n / 16 + 1)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                           offset, keep_prob);
    else
        hipLaunchKernelGGL(dropout_mask_bytes_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                           offset, keep_prob);
    VQA_CHECK_LAUNCH();
    return VQA_OK;
}
