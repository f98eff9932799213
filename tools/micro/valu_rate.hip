// Issue cost of the vector instructions the pair kernel is made of, in shader cycles per wave-instruction per SIMD,
// at 1..8 waves per SIMD (one workgroup of 256 * w threads per CU).  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 200

template <int OP>
__global__ void k_rate(double *out, long long *cyc, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (OP == 0) { // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                             "v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) { // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            } else if (OP == 2) { // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                             "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
            } else if (OP == 3) { // v_rsq_f64
                asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                             "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 4) { // v_cvt_f32_f64
                asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            } else if (OP == 5) { // v_add_f64
                asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n"
                             "v_add_f64 %4, %4, %4\n v_add_f64 %5, %5, %5\n v_add_f64 %6, %6, %6\n v_add_f64 %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 6) { // v_rsq_f32
                asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                             "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            } else if (OP == 7) { // v_cndmask_b32 / integer
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                             "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
            } else if (OP == 8) { // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3\n"
                             "v_mul_f64 %4, %4, %4\n v_mul_f64 %5, %5, %5\n v_mul_f64 %6, %6, %6\n v_mul_f64 %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 9) { // v_cmp_lt_f64 + v_cndmask (hi word select)
                asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %4, 0, %4, vcc\n v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %5, 0, %5, vcc\n"
                             "v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %6, 0, %6, vcc\n v_cmp_lt_f64 vcc, %3, %0\n v_cndmask_b32 %7, 0, %7, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");
            } else if (OP == 10) { // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n"
                             "v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + i0 + i1 + i2 + i3;
    if ((threadIdx.x & 63) == 0) cyc[gid >> 6] = t1 - t0;
}

template <int OP>
static void run(const char *name, double *out, long long *cyc) {
    for (int w : {1, 2, 4}) { // waves per SIMD
        const int threads = 256 * w, blocks = 256;
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.5);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.5);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= h.size();
        // s_memtime ticks at 100 MHz here? report both: ticks per instr per wave and wall-time based ns per instr per SIMD
        const double ninstr = (double)REP * ITER;
        printf("%-16s waves/SIMD %d: memtime ticks/instr/wave %.3f  -> per SIMD %.3f ; wall ns per instr per SIMD %.3f (x2.4 = %.2f cyc)\n", name, w,
               mean / ninstr, mean / ninstr / w, ms * 1e6 / (ninstr * w), ms * 1e6 / (ninstr * w) * 2.4);
    }
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * 256 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 256 * 16);
    run<0>("v_fma_f64", out, cyc);
    run<8>("v_mul_f64", out, cyc);
    run<5>("v_add_f64", out, cyc);
    run<3>("v_rsq_f64", out, cyc);
    run<4>("v_cvt_f32_f64", out, cyc);
    run<9>("cmp_f64+cndmask", out, cyc);
    run<1>("v_fma_f32", out, cyc);
    run<2>("v_pk_fma_f32", out, cyc);
    run<10>("v_pk_add_f32", out, cyc);
    run<6>("v_rsq_f32", out, cyc);
    run<7>("v_add_u32", out, cyc);
    return 0;
}
