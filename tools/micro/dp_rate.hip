// Micro-benchmark: issue cost of the double-precision instructions the equaliser forward's closing stage uses
// (v_cvt_f64_f32, v_add_f64, v_fma_f64) against v_fma_f32, alone and beside a chain of bf16 matrix instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf8 __attribute__((ext_vector_type(8)));

template <int KIND, int NM>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc;
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    float v[16]; double d[16];
    for (int q = 0; q < 16; ++q) { v[q] = threadIdx.x * 0.001f + q; d[q] = v[q]; }
    bf8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (short)(0x3f80 + threadIdx.x + q); b[q] = (short)(0x3f00 + q); }
    const double c1 = 1.0000001; const float f1 = 1.0001f, f2 = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s < NM) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = (4 * s + q) % 16;
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(f1), "v"(f2));
                if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[r]) : "v"(c1));
                if (KIND == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[r]) : "v"(v[r]));
                if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[r]) : "v"(c1));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < 16; ++g) s += acc[g] + v[g] + (float)d[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NM>
void run(const char* name, int waves, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND, NM>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
        if (rep == 2) (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
    }
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d waves/SIMD  %2d bf16 MFMA + 64 x %-14s: %8.1f cycles/iter  %8.1f ns/iter\n", waves / 4, NM, name, (double)c / iters, ms * 1e6 / iters);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<1, 0>("v_fma_f32", waves, out, cyc);
        run<2, 0>("v_add_f64", waves, out, cyc);
        run<3, 0>("v_cvt_f64_f32", waves, out, cyc);
        run<4, 0>("v_fma_f64", waves, out, cyc);
        run<1, 16>("v_fma_f32", waves, out, cyc);
        run<2, 16>("v_add_f64", waves, out, cyc);
        run<3, 16>("v_cvt_f64_f32", waves, out, cyc);
        run<4, 16>("v_fma_f64", waves, out, cyc);
    }
    return 0;
}
