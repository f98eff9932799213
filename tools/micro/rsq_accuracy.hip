// Largest relative error of the hardware seed v_rsq_f64 and of one second-order (Newton) and one third-order correction
// of it, over 2^26 arguments spread over [2^-40, 2^40] (every binade, uniform mantissas), against 1 / sqrt(x) in double.
// Build: hipcc --offload-arch=gfx950 -O2 -o rsq_accuracy rsq_accuracy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k_err(double *out3, int per) {
    double m0 = 0, m2 = 0, m3 = 0;
    unsigned long long s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int i = 0; i < per; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double mant = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0);
        const double x = ldexp(mant, (int)((s >> 3) % 81) - 40);
        const double ref = 1.0 / sqrt(x); // (two correctly rounded operations: within 1 ulp)
        const double y0 = __builtin_amdgcn_rsq(x);
        const double e = fma(-x * y0, y0, 1.0);
        const double y2 = fma(y0 * e, 0.5, y0);
        const double y3 = fma(y0 * e, fma(e, 0.375, 0.5), y0);
        m0 = fmax(m0, fabs(y0 - ref) / ref);
        m2 = fmax(m2, fabs(y2 - ref) / ref);
        m3 = fmax(m3, fabs(y3 - ref) / ref);
    }
    for (int off = 32; off > 0; off >>= 1) {
        m0 = fmax(m0, __shfl_xor(m0, off)); m2 = fmax(m2, __shfl_xor(m2, off)); m3 = fmax(m3, __shfl_xor(m3, off));
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out3[3 * w] = m0; out3[3 * w + 1] = m2; out3[3 * w + 2] = m3;
    }
}

int main() {
    const int blocks = 1024, threads = 256, per = 256, waves = blocks * threads / 64;
    double *d;
    hipMalloc(&d, sizeof(double) * 3 * waves);
    hipLaunchKernelGGL(k_err, dim3(blocks), dim3(threads), 0, 0, d, per);
    double *h = new double[3 * waves];
    hipMemcpy(h, d, sizeof(double) * 3 * waves, hipMemcpyDeviceToHost);
    double m0 = 0, m2 = 0, m3 = 0;
    for (int w = 0; w < waves; w++) { m0 = fmax(m0, h[3 * w]); m2 = fmax(m2, h[3 * w + 1]); m3 = fmax(m3, h[3 * w + 2]); }
    printf("v_rsq_f64 seed: max relative error %.3e = 2^%.1f; second-order correction %.3e; third-order %.3e (2^-53 = %.3e)\n", m0, log2(m0), m2,
           m3, ldexp(1.0, -53));
    return 0;
}
