// Checks cx_antidiag (lane-shift anti-diagonal sums) against a host loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../volterra_vae_amd/csrc/v2vae_cx_common.h"
using namespace v2vae;
__global__ void k(const float* in, float* out) {
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = in[q * 64 + threadIdx.x];
    out[threadIdx.x] = cx_antidiag(v);
}
int main() {
    float h_in[8 * 64], h_out[64], ref[64] = {0};
    for (int q = 0; q < 8; ++q)
        for (int lane = 0; lane < 64; ++lane) {
            const float val = (float)((q * 131 + lane * 17) % 23) - 11.f;
            h_in[q * 64 + lane] = val;
            const int i = lane & 31, h = lane >> 5, x = (q & 3) + 8 * (q >> 2) + 4 * h;
            ref[i + x] += val;
        }
    float *d_in, *d_out;
    (void)hipMalloc(&d_in, sizeof(h_in)); (void)hipMalloc(&d_out, sizeof(h_out));
    (void)hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
    (void)hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 64; ++m) {
        if (std::fabs(h_out[m] - ref[m]) > 1e-4f) { if (bad < 10) printf("m=%d got %g want %g\n", m, h_out[m], ref[m]); ++bad; }
    }
    printf("%s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
