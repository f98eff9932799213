// Operand layout of v_mfma_f64_16x16x4f64 on gfx950, found by experiment: D = A (16 x 4) * B (4 x 16).
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out) {
    const int l = threadIdx.x;
    // hypothesis: lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]
    const int i = l % 16, kk = l / 16;
    const double a = 1.0 + i + 100.0 * kk;        // A[i][k] = 1 + i + 100 k
    const double b = (kk == 1) ? (l % 16 == 3 ? 1.0 : 0.0) : 0.0; // B[k][j] = delta(k, 1) delta(j, 3)
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main() {
    double *d; hipMalloc(&d, 64 * 4 * sizeof(double));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = A[i][1] * delta(j, 3) = (101 + i) at j = 3
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) if (h[l * 4 + r] != 0.0) printf("lane %d reg %d = %g\n", l, r, h[l * 4 + r]);
    return 0;
}
