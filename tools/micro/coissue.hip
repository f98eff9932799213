// Do FP64 matrix instructions of one wave overlap with FP64 vector instructions of other waves on the same SIMD?
// One workgroup of 256 * (1 + nv) threads per CU: the first four waves (one per SIMD) issue v_mfma_f64_16x16x4_f64
// back to back on four independent accumulators, the other 4 * nv waves (nv per SIMD) issue v_fma_f64 back to back on
// eight independent registers.  Each role is also run alone (the other role's waves leave at once).  If the two pipes
// overlap, the mixed run lasts as long as the longer of the two alone; if they share issue slots, as long as their sum.
// Build: hipcc --offload-arch=gfx950 -O2 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define ITER 400
#define MFMA_PER_ITER 8
#define FMA_PER_ITER 64

// mode bit 0: the matrix waves work, bit 1: the vector waves work
__global__ void k_coissue(double *out, long long *cyc, int mode, int mfma_iters, int fma_iters) {
    const int wave = threadIdx.x >> 6;
    const bool matrix_wave = wave < 4;
    double acc = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (matrix_wave) {
        if (mode & 1) {
            v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
            for (int it = 0; it < mfma_iters; it++) {
#pragma unroll
                for (int r = 0; r < MFMA_PER_ITER / 4; r++) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
                }
            }
            acc = c0.x + c1.y + c2.z + c3.w;
        }
    } else if (mode & 2) {
        double a0 = 1.5 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        for (int it = 0; it < fma_iters; it++) {
#pragma unroll
            for (int r = 0; r < FMA_PER_ITER / 8; r++)
                asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                             "v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

static float timed(int nv, int mode, int mi, int fi, double *out, long long *cyc, double *ticks_m, double *ticks_v) {
    const int threads = 256 * (1 + nv), blocks = 256;
    hipLaunchKernelGGL(k_coissue, dim3(blocks), dim3(threads), 0, 0, out, cyc, mode, mi, fi);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_coissue, dim3(blocks), dim3(threads), 0, 0, out, cyc, mode, mi, fi);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int wpb = threads / 64;
    std::vector<long long> h((size_t)blocks * wpb);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < wpb; w++) (w < 4 ? m : v) += (double)h[(size_t)b * wpb + w];
    *ticks_m = m / (blocks * 4.0);
    *ticks_v = nv ? v / (blocks * 4.0 * nv) : 0.0;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * 256 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 256 * 16);
    // alone, a matrix instruction holds its pipe for 16 passes of 4 cycles and a vector FMA for one: the vector waves get
    // 16 / nv times the instruction count so that both roles alone last about the same
    for (int nv : {1, 2, 3}) {
        const int mi = ITER, fi = ITER * MFMA_PER_ITER * 16 / FMA_PER_ITER / nv;
        double tm, tv, d0, d1;
        const float ms_m = timed(nv, 1, mi, fi, out, cyc, &tm, &d0);
        const float ms_v = timed(nv, 2, mi, fi, out, cyc, &d1, &tv);
        double bm, bv;
        const float ms_b = timed(nv, 3, mi, fi, out, cyc, &bm, &bv);
        const double n_m = (double)mi * MFMA_PER_ITER, n_v = (double)fi * FMA_PER_ITER * nv;
        printf("vector waves per SIMD %d: matrix alone %.3f ms (%.1f ns per v_mfma_f64_16x16x4 per SIMD), vector alone %.3f ms (%.2f ns per v_fma_f64 per SIMD), "
               "both %.3f ms = %.2f x the longer, %.2f x the sum; wave clocks alone -> together: matrix %.0f -> %.0f, vector %.0f -> %.0f\n",
               nv, ms_m, ms_m * 1e6 / n_m, ms_v, ms_v * 1e6 / n_v, ms_b, ms_b / (ms_m > ms_v ? ms_m : ms_v), ms_b / (ms_m + ms_v), tm, bm, tv, bv);
    }
    return 0;
}
