// Micro-test: how exact is one v_mfma_f32_32x32x16_bf16?  D = C + sum over 16 of a_k b_k with bf16 inputs (products exact in
// float) against the same sum in double rounded once: error in ulps of the result, its mean (bias) and rms, for C = 0, C of
// the size of the sum, and C 256 times larger (the case of adding a small-piece product to a full accumulator).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf8 __attribute__((ext_vector_type(8)));

__global__ void k(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
    // A [32 rows][16 k], B [16 k][32 cols], C/D [32][32]; lane l: row/col l & 31, k-half l >> 5 (k = 8 * half + j)
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[r * 16 + 8 * h + j]; b[j] = (short)B[(8 * h + j) * 32 + r]; }
    f32x16 c;
    for (int g = 0; g < 16; ++g) c[g] = C[(8 * (g / 4) + 4 * h + (g & 3)) * 32 + r];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int g = 0; g < 16; ++g) D[(8 * (g / 4) + 4 * h + (g & 3)) * 32 + r] = c[g];
}

static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static double rnd() { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; }

int main() {
    unsigned short *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, 32 * 16 * 2); (void)hipMalloc(&dB, 16 * 32 * 2); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
    for (int mode = 0; mode < 4; ++mode) {
        double sum = 0, sq = 0, mx = 0; long n = 0;
        double sumf = 0, sqf = 0;
        for (int rep = 0; rep < 200; ++rep) {
            std::vector<unsigned short> A(512), B(512); std::vector<float> C(1024), D(1024);
            for (auto& x : A) x = f2bf((float)rnd());
            for (auto& x : B) x = f2bf((float)rnd());
            const double cs = mode == 0 ? 0.0 : mode == 1 ? 2.0 : mode == 2 ? 512.0 : 2.0;
            for (auto& x : C) x = (float)(rnd() * cs);
            if (mode == 3) for (auto& x : A) x = f2bf(bf2f(x) / 256.f);   // small products into a full accumulator
            (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
            (void)hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double ex = C[i * 32 + j];
                float chain = C[i * 32 + j];
                for (int kk = 0; kk < 16; ++kk) {
                    ex += (double)bf2f(A[i * 16 + kk]) * bf2f(B[kk * 32 + j]);
                    chain = fmaf(bf2f(A[i * 16 + kk]), bf2f(B[kk * 32 + j]), chain);
                }
                const double ulp = ldexp(1.0, ilogb(fabs(ex) > 1e-30 ? fabs(ex) : 1e-30) - 23);
                const double e = (D[i * 32 + j] - ex) / ulp, ef = (chain - ex) / ulp;
                const double sg = ex >= 0 ? 1.0 : -1.0;
                sum += e * sg; sq += e * e; if (fabs(e) > mx) mx = fabs(e); ++n;
                sumf += ef * sg; sqf += ef * ef;
            }
        }
        printf("mode %d: mfma error in ulps of the result: bias (toward larger magnitude) %+.3f rms %.3f max %.2f | float fma chain: bias %+.3f rms %.3f\n",
               mode, sum / n, sqrt(sq / n), mx, sumf / n, sqrt(sqf / n));
    }
    return 0;
}
