// Microbenchmark: how fast does one wave issue v_mfma_f32_32x32x2_f32 when consecutive instructions DEPEND on each other
// through the accumulator (1 accumulator), against 2 / 4 independent accumulators, at 1 / 2 / 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain ; run: ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(1024) void chain(float* out, int iters, float a0, float b0) {
    f32x16 acc[ACC];
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16 / ACC; ++j)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(int waves_per_simd, float* out) {
    const int threads = 256 * waves_per_simd, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<ACC>, dim3(256), dim3(threads), 0, 0, out, 100, 1.f, 0.f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain<ACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 0.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    const double cyc = best * 1e-3 * 2.4e9 / mfma_per_simd;
    const double tf = 256.0 * 4 * mfma_per_simd * 32 * 32 * 2 * 2 / (best * 1e-3) / 1e12;
    printf("accumulators %d, waves/SIMD %d: %.3f ms, %.1f cycles per MFMA at 2.4 GHz, %.1f TFLOP/s\n", ACC, waves_per_simd, best, cyc, tf);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int w : {1, 2, 4}) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
