// Micro-benchmark: do f32 MFMA chains and VALU work overlap on one SIMD (a) inside one wave when interleaved in program
// order, (b) across two waves that each run [MFMA chain][VALU block] phases?  Prints cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int MODE>  // 0 mfma only, 1 valu only, 2 serial phases, 3 interleaved
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc;
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = threadIdx.x * 0.001f + q;
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = MFMA32(a, b, acc);
        }
        if (MODE == 1 || MODE == 2) {
            PIN();
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = fmaf(v[q], 1.0001f, 0.5f);
            PIN();
        }
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                PIN();
                acc = MFMA32(a, b, acc);
                PIN();
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = fmaf(v[q], 1.0001f, 0.5f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < 16; ++g) s += acc[g] + v[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 512 * 4 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char* names[4] = {"mfma chain only (16 x 32x32x2 f32)", "valu only (256 fma)", "serial phases", "interleaved in program order"};
    // full-chip run of the MFMA chain: does the shader clock hold under load?  (ticks are shader cycles)
    for (int grid = 1; grid <= 1024; grid *= 4) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float* big; (void)hipMalloc(&big, (size_t)grid * 512 * 4);
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 2) (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, big, cyc, 10 * iters);
            if (rep == 2) (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
        }
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("grid %4d x 4 waves, mfma chain only: %8.1f ticks/iter, %7.1f ns/iter -> %.2f ticks/ns\n", grid, (double)c / (10 * iters),
               ms * 1e6 / (10 * iters), (double)c / (ms * 1e6));
        (void)hipFree(big);
    }
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (rep == 1) (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
            }
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            // s_memtime counts at 100 MHz: convert with the 2.4 GHz shader clock
            printf("%d waves/SIMD  %-40s %8.1f ticks/iter, %7.1f ns/iter (wall, incl. launch)\n", waves / 4, names[mode],
                   (double)c / iters, ms * 1e6 / iters);
        }
    }
    return 0;
}
