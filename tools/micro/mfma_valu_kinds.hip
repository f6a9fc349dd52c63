// Micro-benchmark: which vector instructions hide behind an f32 MFMA chain of the SAME wave?  Per loop iteration a chain of 16
// dependent v_mfma_f32_32x32x2_f32 with NF filler instructions after each MFMA (independent of the chain), fillers being
// (a) v_fma_f32, (b) v_pk_fma_f32, (c) v_add_f32 ... Whole chip (256 workgroups x 4 or 8 waves), cycles and wall time per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int KIND, int NF>   // KIND 0: none, 1: v_fma_f32, 2: v_pk_fma_f32, 3: v_pk_mul_f32
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc;
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    float v[16];
    f2 w[16];
    for (int q = 0; q < 16; ++q) { v[q] = threadIdx.x * 0.001f + q; w[q] = f2{v[q], v[q] + 1.f}; }
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    const f2 c1 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc = MFMA32(a, b, acc);
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q % 16]) : "v"(b), "v"(a));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[q % 16]) : "v"(c1), "v"(c2));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[q % 16]) : "v"(c1));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < 16; ++g) s += acc[g] + v[g] + w[g].x + w[g].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NF>
void run(const char* name, int waves, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
        if (rep == 2) (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
    }
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d waves/SIMD  %-14s x%2d per MFMA: %8.1f cycles/iter  %8.1f ns/iter  (%.2f GHz)\n", waves / 4, name, NF, (double)c / iters,
           ms * 1e6 / iters, (double)c / (ms * 1e6) );
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0, 0>("none", waves, out, cyc);
        run<1, 4>("v_fma_f32", waves, out, cyc);
        run<1, 8>("v_fma_f32", waves, out, cyc);
        run<1, 12>("v_fma_f32", waves, out, cyc);
        run<1, 16>("v_fma_f32", waves, out, cyc);
        run<2, 2>("v_pk_fma_f32", waves, out, cyc);
        run<2, 4>("v_pk_fma_f32", waves, out, cyc);
        run<2, 8>("v_pk_fma_f32", waves, out, cyc);
        run<3, 4>("v_pk_mul_f32", waves, out, cyc);
        run<3, 8>("v_pk_mul_f32", waves, out, cyc);
    }
    return 0;
}
