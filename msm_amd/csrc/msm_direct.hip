// Open-boundary direct sum on the GPU: u = 1/2 sum_{i != j} q_i q_j / r_ij and the gradient of the potential at
// every particle.  This is the computation of the reference's non-periodic driver (msm/nonperiodic.c:12-22, energy
// only, charges forced to 1 there) behind the entry point the reference declares but never defines
// (`direct`, msm/msmLibrary.h:212).
//
// Classic all-pairs tiling: a workgroup keeps 2 x 256 particles i in registers and streams all j through LDS
// (256 particles per tile, every lane reads the same j: broadcast reads).  No cutoff, so every lane of every
// instruction does useful work: 19 FP64 VALU instructions per pair.  Particle i's sum is accumulated in j order
// by one lane, so the result is bitwise reproducible.
#include "msm_kernels.hpp"
#include "msm_device.hpp"
#include "msm_host.hpp"

#include "msm_amd.h"

#include <cstring>
#include <mutex>

namespace msm {

#define DIR_THREADS 256
#define DIR_IPT 2 // particles i per thread

__device__ __forceinline__ double rsqrt_refined(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// acc_i = sum_j q_j grad_i (1/r_ij) = -sum_j q_j d / r^3 (scaled by the convention), upart[block] = 1/2 sum q_i phi_i
__global__ __launch_bounds__(DIR_THREADS) void k_direct(const double *__restrict__ r, const double *__restrict__ q, int N, int i0,
                                                        int i1, int convention, double *__restrict__ acc,
                                                        double *__restrict__ upart) {
    __shared__ double4 tile[DIR_THREADS];
    __shared__ double red[DIR_THREADS];
    const int tid = threadIdx.x;
    double xi[DIR_IPT], yi[DIR_IPT], zi[DIR_IPT], qi[DIR_IPT], u[DIR_IPT], fx[DIR_IPT], fy[DIR_IPT], fz[DIR_IPT];
    int idx[DIR_IPT];
#pragma unroll
    for (int k = 0; k < DIR_IPT; k++) {
        idx[k] = i0 + (blockIdx.x * DIR_IPT + k) * DIR_THREADS + tid;
        const bool ok = idx[k] < i1;
        const int s = ok ? idx[k] : i0;
        xi[k] = r[3 * (size_t)s]; yi[k] = r[3 * (size_t)s + 1]; zi[k] = r[3 * (size_t)s + 2];
        qi[k] = ok ? q[s] : 0.0;
        u[k] = fx[k] = fy[k] = fz[k] = 0.0;
    }
    for (int jb = 0; jb < N; jb += DIR_THREADS) {
        __syncthreads();
        {
            const int j = jb + tid;
            tile[tid] = j < N ? make_double4(r[3 * (size_t)j], r[3 * (size_t)j + 1], r[3 * (size_t)j + 2], q[j])
                              : make_double4(0.0, 0.0, 0.0, 0.0);
        }
        __syncthreads();
        const int cnt = min(DIR_THREADS, N - jb);
#pragma unroll 4
        for (int t = 0; t < cnt; t++) {
            const double4 pj = tile[t];
#pragma unroll
            for (int k = 0; k < DIR_IPT; k++) {
                const double dx = xi[k] - pj.x, dy = yi[k] - pj.y, dz = zi[k] - pj.z;
                const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                // the particle itself (and any exact duplicate) contributes nothing: charge with the high word cleared
                const double qj = __hiloint2double(r2 > 0.0 ? __double2hiint(pj.w) : 0, __double2loint(pj.w));
                const double ir = rsqrt_refined(r2 > 0.0 ? r2 : 1.0);
                const double qir = qj * ir;
                u[k] += qir;
                const double s = qir * (ir * ir);
                fx[k] = fma(s, dx, fx[k]);
                fy[k] = fma(s, dy, fy[k]);
                fz[k] = fma(s, dz, fz[k]);
            }
        }
    }
    double e = 0.0;
#pragma unroll
    for (int k = 0; k < DIR_IPT; k++) {
        if (idx[k] < i1) {
            // gradient of the potential: -sum q_j d / r^3
            const double w = convention == MSM_AMD_ACC_FORCE ? qi[k] : -1.0;
            const size_t o = 3 * (size_t)(idx[k] - i0);
            acc[o] = w * fx[k]; acc[o + 1] = w * fy[k]; acc[o + 2] = w * fz[k];
            e += 0.5 * qi[k] * u[k];
        }
    }
    red[tid] = e;
    __syncthreads();
#pragma unroll
    for (int off = DIR_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) upart[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_direct_sum(const double *__restrict__ upart, int n, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += upart[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

} // namespace msm

using namespace msm;

// Device-resident form: accelerations of particles [i0, i1) against all N (a rank of a multi-GPU run takes a
// slice of i; the energies of the slices add up).  d_acc has 3 (i1 - i0) entries; d_work needs
// msm_amd_direct_work_doubles(N) doubles; the slice's energy is left in d_work[0] and, if u != NULL, copied back
// (synchronises the stream).
extern "C" int msm_amd_direct_work_doubles(int N) { return (N + DIR_THREADS * DIR_IPT - 1) / (DIR_THREADS * DIR_IPT) + 8; }

extern "C" int msm_amd_direct_device(int N, const double *d_r, const double *d_q, int i0, int i1, int acc_convention,
                                     double *d_acc, double *d_work, void *stream, double *u) {
    if (N < 1 || !d_r || !d_q || !d_acc || !d_work || i0 < 0 || i1 > N || i0 >= i1) {
        set_error("msm_amd_direct_device: bad argument");
        return MSM_AMD_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (i1 - i0 + DIR_THREADS * DIR_IPT - 1) / (DIR_THREADS * DIR_IPT);
    hipLaunchKernelGGL(k_direct, dim3(nblk), dim3(DIR_THREADS), 0, st, d_r, d_q, N, i0, i1, acc_convention, d_acc, d_work + 8);
    hipLaunchKernelGGL(k_direct_sum, dim3(1), dim3(256), 0, st, d_work + 8, nblk, d_work);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("direct sum launch failed: %s", hipGetErrorString(e)); return MSM_AMD_EHIP; }
    if (u) {
        if (hipMemcpyAsync(u, d_work, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            set_error("direct sum: copy of the energy failed");
            return MSM_AMD_EHIP;
        }
    }
    return MSM_AMD_OK;
}

// ---------------------------------------------------------------------------
// The same sum for callers that cannot allocate (HIP graph capture, serving loops): the per-workgroup energy
// partials that msm_amd_direct_device keeps in its d_work argument live in a file-scope device array instead.
// Every GPU holds its own copy of that array at its own address until the process ends; the address is looked up
// once per device and only that pointer - never the array's name, which on the host is a stand-in - is handed to
// kernels.  Nothing in the array has to start at zero: k_direct overwrites the partials of every workgroup it
// launches and k_direct_sum reads exactly those.  Two calls that overlap on one GPU share the array and race.
// ---------------------------------------------------------------------------
#define DIR_SCRATCH_BLOCKS 8192 // 64 KiB of partials: slices of up to 8192 * 512 = 4 194 304 particles i
#define DIR_MAX_DEVICES 64
static __device__ __align__(256) double scratch[DIR_SCRATCH_BLOCKS];
static double *g_scratch_ptr[DIR_MAX_DEVICES]; // device address of `scratch` on each GPU, 0 until looked up
static std::mutex g_scratch_mu;

static int direct_scratch(double **out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DIR_MAX_DEVICES) {
        set_error("msm_amd_direct_device_noalloc: no current HIP device");
        return MSM_AMD_ENODEV;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("msm_amd_direct_device_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
            return MSM_AMD_EHIP;
        }
        g_scratch_ptr[dev] = (double *)p;
    }
    *out = g_scratch_ptr[dev];
    return MSM_AMD_OK;
}

extern "C" int msm_amd_direct_noalloc_max_slice(void) { return DIR_SCRATCH_BLOCKS * DIR_THREADS * DIR_IPT; }

extern "C" int msm_amd_direct_device_noalloc(int N, const double *d_r, const double *d_q, int i0, int i1, int acc_convention,
                                             double *d_acc, double *d_u, void *stream) {
    if (N < 1 || !d_r || !d_q || !d_acc || !d_u || i0 < 0 || i1 > N || i0 >= i1) {
        set_error("msm_amd_direct_device_noalloc: bad argument");
        return MSM_AMD_EINVAL;
    }
    const long nblk = ((long)i1 - i0 + DIR_THREADS * DIR_IPT - 1) / (DIR_THREADS * DIR_IPT);
    if (nblk > DIR_SCRATCH_BLOCKS) {
        set_error("msm_amd_direct_device_noalloc: a slice of %d particles exceeds the built-in scratch (at most %d); "
                  "split the slice or use msm_amd_direct_device with a work buffer", i1 - i0,
                  DIR_SCRATCH_BLOCKS * DIR_THREADS * DIR_IPT);
        return MSM_AMD_ERANGE;
    }
    double *part = nullptr;
    const int rc = direct_scratch(&part);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_direct, dim3((unsigned)nblk), dim3(DIR_THREADS), 0, st, d_r, d_q, N, i0, i1, acc_convention, d_acc, part);
    hipLaunchKernelGGL(k_direct_sum, dim3(1), dim3(256), 0, st, part, (int)nblk, d_u);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("direct sum launch failed: %s", hipGetErrorString(e)); return MSM_AMD_EHIP; }
    return MSM_AMD_OK;
}

namespace {
struct DirectStaging {
    int device = -1;
    size_t n = 0, nw = 0;
    double *r = nullptr, *q = nullptr, *acc = nullptr, *work = nullptr;
} g_stage;
std::mutex g_stage_mu;
void direct_release_staging_locked() {
    if (g_stage.device >= 0) {
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        if (have && cur != g_stage.device) (void)hipSetDevice(g_stage.device);
        (void)hipFree(g_stage.r); (void)hipFree(g_stage.q); (void)hipFree(g_stage.acc); (void)hipFree(g_stage.work);
        if (have && cur != g_stage.device) (void)hipSetDevice(cur);
    } else {
        (void)hipFree(g_stage.r); (void)hipFree(g_stage.q); (void)hipFree(g_stage.acc); (void)hipFree(g_stage.work);
    }
    g_stage = DirectStaging();
}
} // namespace
// (called by msm_amd_release_cached_plan)
namespace msm {
void direct_release_staging() {
    std::lock_guard<std::mutex> lock(g_stage_mu);
    direct_release_staging_locked();
}
} // namespace msm

// replaces direct(), declared at msm/msmLibrary.h:212 and defined nowhere in the reference; computation of
// msm/nonperiodic.c:12-22 plus the gradient.  Host pointers in and out; failures print to stderr (the reference's
// convention, no return code) and leave *ureal = 0.
extern "C" void direct(int N, double *r, double *q, double *ureal, double *freal) {
    if (ureal) *ureal = 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device is visible; msm_amd has no CPU path");
        fprintf(stderr, "direct: %s\n", msm_amd_last_error());
        return;
    }
    if (N < 1 || !r || !q) return;
    // device copies of the caller's arrays: kept from call to call (a driver calls direct() in a loop), grown when a
    // larger system arrives, released by msm_amd_release_cached_plan()
    std::lock_guard<std::mutex> lock(g_stage_mu);
    int dev = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess;
    const size_t nw = (size_t)msm_amd_direct_work_doubles(N);
    if (ok && (g_stage.device != dev || g_stage.n < (size_t)N || g_stage.nw < nw)) {
        direct_release_staging_locked();
        ok = hipMalloc((void **)&g_stage.r, sizeof(double) * 3 * (size_t)N) == hipSuccess &&
             hipMalloc((void **)&g_stage.q, sizeof(double) * (size_t)N) == hipSuccess &&
             hipMalloc((void **)&g_stage.acc, sizeof(double) * 3 * (size_t)N) == hipSuccess &&
             hipMalloc((void **)&g_stage.work, sizeof(double) * nw) == hipSuccess;
        if (ok) { g_stage.device = dev; g_stage.n = (size_t)N; g_stage.nw = nw; }
        else direct_release_staging_locked();
    }
    double *d_r = g_stage.r, *d_q = g_stage.q, *d_acc = g_stage.acc, *d_work = g_stage.work;
    ok = ok && hipMemcpy(d_r, r, sizeof(double) * 3 * (size_t)N, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_q, q, sizeof(double) * (size_t)N, hipMemcpyHostToDevice) == hipSuccess;
    double u = 0.0;
    if (ok) ok = msm_amd_direct_device(N, d_r, d_q, 0, N, MSM_AMD_ACC_GRADIENT, d_acc, d_work, nullptr, &u) == MSM_AMD_OK;
    if (ok && freal) ok = hipMemcpy(freal, d_acc, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
        if (!*msm_amd_last_error()) set_error("direct: a HIP call failed");
        fprintf(stderr, "direct: %s\n", msm_amd_last_error());
    } else if (ureal) {
        *ureal = u;
    }
}
