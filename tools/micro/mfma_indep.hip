// Micro-benchmark: issue rate of INDEPENDENT v_mfma_f32_32x32x2_f32 (NACC accumulators round-robin), operands in registers
// or fetched from LDS one step ahead, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NACC, int LDS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float tab[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 256) tab[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int g = 0; g < 16; ++g) acc[q][g] = 0.f;
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    const float* tl = tab + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float an[NACC];
    for (int q = 0; q < NACC; ++q) an[q] = a;
    if (LDS >= 2) {
        // 2: LDS reads issued next to the MFMAs but NOT used as operands; 3: used as operands, two register sets, no copies;
        // 4: like 3 with one ds_read_b128 per four operands
        float s0[NACC], s1[NACC];
        for (int q = 0; q < NACC; ++q) s0[q] = s1[q] = a;
        for (int it = 0; it < iters; it += 2) {
            const float* p0 = tl + (it & 31) * 128;
#pragma unroll
            for (int q = 0; q < NACC; ++q) s1[q] = p0[q * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = MFMA32(LDS == 2 ? a : s0[q], b, acc[q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NACC; ++q) s0[q] = p0[q * 64 + 32 * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = MFMA32(LDS == 2 ? a : s1[q], b, acc[q]);
            __builtin_amdgcn_sched_barrier(0);
            if (LDS == 2) {
#pragma unroll
                for (int q = 0; q < NACC; ++q) asm volatile("" ::"v"(s0[q]), "v"(s1[q]));
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int q = 0; q < NACC; ++q)
            for (int g = 0; g < 16; ++g) s += acc[q][g];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        float ac[NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q) ac[q] = an[q];
        if (LDS) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) an[q] = tl[((it + q) & 63) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = MFMA32(ac[q], b, acc[q]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q)
        for (int g = 0; g < 16; ++g) s += acc[q][g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
#define RUN(NACC, LDS)                                                                                       \
    do {                                                                                                     \
        for (int rep = 0; rep < 2; ++rep) {                                                                  \
            hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(256), 0, 0, out, cyc, iters);                 \
            (void)hipDeviceSynchronize();                                                                    \
        }                                                                                                    \
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                            \
        printf("%d independent accumulators, operands %s: %6.1f cycles per MFMA\n", NACC,                    \
               LDS == 0 ? "in registers" : LDS == 1 ? "from LDS (one step ahead, copies)" : LDS == 2 ? "in registers, LDS reads alongside" : "from LDS, two register sets", (double)c / iters / NACC);                \
    } while (0)
    RUN(1, 0); RUN(6, 0); RUN(6, 1); RUN(6, 2); RUN(6, 3);
    return 0;
}
