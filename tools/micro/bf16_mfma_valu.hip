// Micro-benchmark: do vector instructions run beside a chain of bf16 matrix instructions of the SAME wave (and of the SIMD's other
// wave)?  Per loop iteration 16 dependent v_mfma_f32_32x32x16_bf16 with NF independent filler instructions after each one.
// Fillers: (1) v_fma_f32, (2) v_pk_fma_f32, (3) v_cvt_pk_bf16_f32, (4) v_and_b32, (5) v_mfma_f32_32x32x2_f32 (the f32 matrix
// instruction, NF of them per 16 bf16 ones).  Whole chip (256 workgroups x 4 or 8 waves), cycles of one wave and wall time per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short bf8 __attribute__((ext_vector_type(8)));

template <int KIND, int NF, int NM>   // NM: bf16 matrix instructions per iteration (0 = fillers alone, 16 slots)
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc, acc2;
    for (int g = 0; g < 16; ++g) { acc[g] = 0.f; acc2[g] = 0.f; }
    float v[16];
    f2 w[16];
    unsigned u[16];
    for (int q = 0; q < 16; ++q) { v[q] = threadIdx.x * 0.001f + q; w[q] = f2{v[q], v[q] + 1.f}; u[q] = threadIdx.x * 77u + q; }
    bf8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (short)(0x3f80 + threadIdx.x + q); b[q] = (short)(0x3f00 + q); }
    float fa = threadIdx.x * 0.5f, fb = 1.0001f;
    const f2 c1 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s < NM) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            if (KIND == 5) {
                if (s < NF) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc2, 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q % 16]) : "v"(fb), "v"(fa));
                    if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[q % 16]) : "v"(c1), "v"(c2));
                    if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[q % 16]) : "v"(v[q % 16]), "v"(v[(q + 1) % 16]));
                    if (KIND == 4) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[q % 16]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < 16; ++g) s += acc[g] + acc2[g] + v[g] + w[g].x + w[g].y + (float)u[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NF, int NM>
void run(const char* name, int waves, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND, NF, NM>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
        if (rep == 2) (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
    }
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d waves/SIMD  %2d bf16 MFMA + %-18s x%2d each: %8.1f cycles/iter  %8.1f ns/iter\n", waves / 4, NM, name, NF, (double)c / iters,
           ms * 1e6 / iters);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0, 0, 16>("none", waves, out, cyc);
        run<1, 4, 0>("v_fma_f32 alone", waves, out, cyc);
        run<1, 4, 16>("v_fma_f32", waves, out, cyc);
        run<1, 6, 16>("v_fma_f32", waves, out, cyc);
        run<1, 8, 16>("v_fma_f32", waves, out, cyc);
        run<1, 12, 16>("v_fma_f32", waves, out, cyc);
        run<2, 4, 0>("v_pk_fma_f32 alone", waves, out, cyc);
        run<2, 2, 16>("v_pk_fma_f32", waves, out, cyc);
        run<2, 4, 16>("v_pk_fma_f32", waves, out, cyc);
        run<2, 6, 16>("v_pk_fma_f32", waves, out, cyc);
        run<2, 8, 16>("v_pk_fma_f32", waves, out, cyc);
        run<2, 12, 16>("v_pk_fma_f32", waves, out, cyc);
        run<3, 4, 0>("v_cvt_pk_bf16 alone", waves, out, cyc);
        run<3, 4, 16>("v_cvt_pk_bf16_f32", waves, out, cyc);
        run<3, 8, 16>("v_cvt_pk_bf16_f32", waves, out, cyc);
        run<4, 4, 16>("v_and_b32", waves, out, cyc);
        run<4, 8, 16>("v_and_b32", waves, out, cyc);
        run<5, 4, 0>("f32 MFMA alone", waves, out, cyc);
        run<5, 4, 16>("f32 MFMA 32x32x2", waves, out, cyc);
        run<5, 8, 16>("f32 MFMA 32x32x2", waves, out, cyc);
        run<5, 16, 16>("f32 MFMA 32x32x2", waves, out, cyc);
    }
    return 0;
}
