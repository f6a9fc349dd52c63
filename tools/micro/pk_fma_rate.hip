// Micro-benchmark: issue rate of v_pk_fma_f32 streams on one SIMD as a function of (a) the number of independent accumulator
// chains, (b) a scalar-register source, (c) op_sel / neg modifiers, at 1, 2 and 4 waves per SIMD.  Whole chip (256 x 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NCH, int KIND>   // KIND 0: vector sources; 1: scalar src0; 2: scalar src0 + op_sel/neg modifiers
__global__ __launch_bounds__(1024) void k(float* out, const unsigned long long* tab, int iters) {
    f2 acc[NCH];
    for (int q = 0; q < NCH; ++q) acc[q] = f2{threadIdx.x * 0.001f + q, 1.f};
    const f2 y = {1.0001f, 0.9999f}, c = {1e-9f, 1e-9f};
    const unsigned long long ks = tab[blockIdx.x & 1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            f2& a = acc[s % NCH];
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(c), "v"(y));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "s"(ks), "v"(y));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a) : "s"(ks), "v"(y));
        }
    }
    float s = 0.f;
    for (int q = 0; q < NCH; ++q) s += acc[q].x + acc[q].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH, int KIND>
void run(const char* name, int waves, float* out, unsigned long long* tab) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 2) (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NCH, KIND>), dim3(256), dim3(64 * waves), 0, 0, out, tab, iters);
        if (rep == 2) (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
    }
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_simd = (double)iters * 64 * (waves / 4);
    printf("%d waves/SIMD  %-22s chains %2d: %6.2f ns per pk op per SIMD\n", waves / 4, name, NCH, ms * 1e6 / ops_per_simd);
}

int main() {
    float* out; unsigned long long* tab;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&tab, 16); (void)hipMemset(tab, 0, 16);
    for (int waves = 4; waves <= 16; waves *= 2) {
        run<1, 0>("vector sources", waves, out, tab);
        run<2, 0>("vector sources", waves, out, tab);
        run<4, 0>("vector sources", waves, out, tab);
        run<16, 0>("vector sources", waves, out, tab);
        run<1, 1>("scalar src0", waves, out, tab);
        run<2, 1>("scalar src0", waves, out, tab);
        run<4, 1>("scalar src0", waves, out, tab);
        run<4, 2>("scalar src0 + op_sel", waves, out, tab);
        run<16, 2>("scalar src0 + op_sel", waves, out, tab);
    }
    return 0;
}
