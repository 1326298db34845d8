// Microbenchmark: what does one wave per SIMD pay for NON-matrix instructions placed between its
// v_mfma_f32_32x32x2_f32 (64 cycles of matrix core each)?  Per pair of MFMAs on two accumulators, N filler
// instructions of one kind: independent v_fma_f32, a dependent v_fma chain, v_exp_f32 (quarter rate), ds_write_b32,
// v_accvgpr_read, s_nop.  Cycles per MFMA from s_memtime (shader clock), so DVFS does not matter.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_gap.hip -o mfma_gap ; run: ./mfma_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void gap(float* out, long long* cyc, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    f32x16 accA, accB;
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    float a = a0 + threadIdx.x, b = b0;
    float f0 = a, f1 = b, f2 = a + 1.f, f3 = b + 2.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            MFMA(accA, a, b);
            MFMA(accB, a, b);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 0) {            // independent VALU (4 chains)
                    if ((n & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(b));
                    if ((n & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f1) : "v"(b));
                    if ((n & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f2) : "v"(b));
                    if ((n & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f3) : "v"(b));
                } else if (KIND == 1) {     // one dependent VALU chain
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(b));
                } else if (KIND == 2) {     // transcendental
                    asm volatile("v_exp_f32 %0, %0" : "+v"(f0));
                } else if (KIND == 3) {     // LDS write
                    asm volatile("ds_write_b32 %0, %1" ::"v"((int)(threadIdx.x * 4)), "v"(f0) : "memory");
                } else if (KIND == 4) {     // s_nop 0
                    asm volatile("s_nop 0");
                } else if (KIND == 5) {     // VALU that reads the accumulator being produced two MFMAs back? no: a plain mov
                    asm volatile("v_mov_b32 %0, %1" : "=v"(f1) : "v"(f0));
                }
            }
        }
    }
    const long long t1 = clock64();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = f0 + f1 + f2 + f3 + lds[threadIdx.x];
    for (int r = 0; r < 16; ++r) s += accA[r] + accB[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int N>
void run(const char* what, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((gap<KIND, N>), dim3(256), dim3(256), 0, 0, out, cyc, 50, 1.f, 0.f);
    hipLaunchKernelGGL((gap<KIND, N>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.f, 0.f);
    hipDeviceSynchronize();
    long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-28s N = %2d per MFMA pair: %6.1f shader cycles per MFMA\n", what, N, (double)c / (iters * 16.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&cyc, 64);
    run<0, 0>("no filler", out, cyc);
    run<0, 4>("independent v_fma", out, cyc);  run<0, 8>("independent v_fma", out, cyc);
    run<0, 16>("independent v_fma", out, cyc); run<0, 24>("independent v_fma", out, cyc); run<0, 32>("independent v_fma", out, cyc);
    run<1, 8>("dependent v_fma chain", out, cyc); run<1, 16>("dependent v_fma chain", out, cyc); run<1, 24>("dependent v_fma chain", out, cyc);
    run<2, 2>("v_exp_f32", out, cyc); run<2, 4>("v_exp_f32", out, cyc); run<2, 8>("v_exp_f32", out, cyc);
    run<3, 2>("ds_write_b32", out, cyc); run<3, 4>("ds_write_b32", out, cyc); run<3, 8>("ds_write_b32", out, cyc);
    run<4, 8>("s_nop 0", out, cyc); run<4, 16>("s_nop 0", out, cyc);
    run<5, 8>("v_mov_b32", out, cyc); run<5, 16>("v_mov_b32", out, cyc);
    return 0;
}
