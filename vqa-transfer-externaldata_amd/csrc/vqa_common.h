// Shared helpers for the gfx950 kernels of libvqahot.so.  Device code is written
// for CDNA4 only: wave = 64 lanes, 256-thread workgroups unless stated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vqa_hot.h"

#define VQA_WAVE 64

#define VQA_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return VQA_ERR_LAUNCH;        \
    } while (0)

#define VQA_REQUIRE(cond, code) \
    do {                        \
        if (!(cond)) return (code); \
    } while (0)

// Measurement / trace scope around a launch group (csrc/probe.hip): HIP events on `st` when the label is probed, a roctx
// range when ranges are on.  Costs one branch when neither is.
int vqa_probe_begin(const char* label, hipStream_t st, void** handle);
void vqa_probe_end(int flags, hipStream_t st, void* handle);
struct ProbeScope {
    hipStream_t st;
    void* handle = nullptr;
    int flags;
    ProbeScope(const char* label, hipStream_t s) : st(s) { flags = vqa_probe_begin(label, s, &handle); }
    ~ProbeScope() {
        if (flags) vqa_probe_end(flags, st, handle);
    }
    ProbeScope(const ProbeScope&) = delete;
    ProbeScope& operator=(const ProbeScope&) = delete;
};

static inline bool vqa_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for 256..1024-thread blocks; red must hold >= 16 floats.  All
// threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float sigmoidf_stable(float x) {
    // same value as the oracle's piecewise logistic to rounding
    if (x >= 0.f) return 1.f / (1.f + expf(-x));
    const float e = expf(x);
    return e / (1.f + e);
}
