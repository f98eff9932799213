// Ewald sum on the GPU: the accuracy reference of the reference project (msm/ewaldClassical.c:5-109, energy) with
// the potential gradient added (the quantity msm/ewaldHernquist.c:113-172 tabulates for the gravitational sign).
//
//   phi_i = sum_{j, images} q_j erfc(alpha r)/r                                   (real space, inside rcut)
//         + sum_{k != 0} exp(-pi^2 k^2/alpha^2)/(pi k^2 V) Re[ e^{-2 pi i k.r_i} S(k) ]   (S = structure factor)
//         - q_i 2 alpha/sqrt(pi) - (sum q) pi/(alpha^2 V)
//
// Two modes.  FULL: the converged sum used as the denominator of accuracy figures: minimum image in real space
// (rcut <= half the shortest cell edge), every |k_d| <= ceil(kcut).  LITERAL: what ewaldClassical() computes: the
// real sum over the particles as given, without images or minimum image (ewaldClassical.c:36-48), and the
// reciprocal sum over kx^2+ky^2+kz^2 < cutoffFourier^2 (ewaldClassical.c:66-69).
//
// Structure factors: one thread per wave vector of the half space (S(-k) is the conjugate), all particles
// streamed through LDS in a few chunks whose partial sums are added in a fixed order.  Real space: one thread per
// particle i, all j streamed through LDS, again in chunks.  Reciprocal part of phi_i: one thread per i, the wave
// vectors in a fixed order (their coefficients are wave-uniform loads).  Nothing is accumulated with atomics, so
// the result is the same on every run.
#include "msm_kernels.hpp"
#include "msm_device.hpp"
#include "msm_host.hpp"

#include "msm_amd.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace msm {

#define EW_T 256
static constexpr double EW_PI = 3.14159265358979323846;

struct EwaldGeom {
    double A[3], invA[3];
    double alpha, rc2, kc2, vol;
    int kmax, cn, nkh; // wave vectors (kx, ky, kz) in [-kmax, kmax]^2 x [0, kmax], index kx + kmax + cn (ky + kmax + cn kz)
    int literal;
    int N;
};

__device__ __forceinline__ bool ew_decode(const EwaldGeom &g, int k, int &kx, int &ky, int &kz) {
    kx = k % g.cn - g.kmax;
    ky = (k / g.cn) % g.cn - g.kmax;
    kz = k / (g.cn * g.cn);
    // the half space kz > 0, or kz = 0 and ky > 0, or kz = ky = 0 and kx > 0
    if (kz == 0 && (ky < 0 || (ky == 0 && kx <= 0))) return false;
    if (g.literal && (double)(kx * kx + ky * ky + kz * kz) >= g.kc2) return false;
    return true;
}

// partial structure factors of particle chunk blockIdx.y: Cp/Sp[chunk][k] = sum_j q_j cos/sin(2 pi k.r_j)
__global__ __launch_bounds__(EW_T) void k_ewald_sk(EwaldGeom g, const double *__restrict__ r, const double *__restrict__ q,
                                                   int per_chunk, double *__restrict__ Cp, double *__restrict__ Sp) {
    __shared__ double4 tile[EW_T];
    const int k = blockIdx.x * EW_T + threadIdx.x;
    int kx = 0, ky = 0, kz = 0;
    const bool live = k < g.nkh && ew_decode(g, k, kx, ky, kz);
    const int j0 = blockIdx.y * per_chunk, j1 = min(g.N, j0 + per_chunk);
    double c = 0.0, s = 0.0;
    for (int jb = j0; jb < j1; jb += EW_T) {
        __syncthreads();
        const int j = jb + threadIdx.x;
        tile[threadIdx.x] = j < j1 ? make_double4(r[3 * (size_t)j] * g.invA[0], r[3 * (size_t)j + 1] * g.invA[1],
                                                  r[3 * (size_t)j + 2] * g.invA[2], q[j])
                                   : make_double4(0.0, 0.0, 0.0, 0.0);
        __syncthreads();
        if (live) {
            const int cnt = min(EW_T, j1 - jb);
            for (int t = 0; t < cnt; t++) {
                const double4 p = tile[t];
                double sn, cs;
                sincospi(2.0 * fma((double)kx, p.x, fma((double)ky, p.y, (double)kz * p.z)), &sn, &cs);
                c = fma(p.w, cs, c);
                s = fma(p.w, sn, s);
            }
        }
    }
    if (k < g.nkh) {
        Cp[(size_t)blockIdx.y * g.nkh + k] = c;
        Sp[(size_t)blockIdx.y * g.nkh + k] = s;
    }
}

// chunks added in order; F(k) = 2 exp(-pi^2 k^2/alpha^2)/(pi k^2 V) for the wave vectors that take part, else 0
__global__ void k_ewald_sk_sum(EwaldGeom g, int nchunk, const double *__restrict__ Cp, const double *__restrict__ Sp,
                               double *__restrict__ Ck, double *__restrict__ Sk, double *__restrict__ Fk) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.nkh) return;
    double c = 0.0, s = 0.0;
    for (int ch = 0; ch < nchunk; ch++) {
        c += Cp[(size_t)ch * g.nkh + k];
        s += Sp[(size_t)ch * g.nkh + k];
    }
    int kx, ky, kz;
    double f = 0.0;
    if (ew_decode(g, k, kx, ky, kz)) {
        const double ax = kx * g.invA[0], ay = ky * g.invA[1], az = kz * g.invA[2], k2 = ax * ax + ay * ay + az * az;
        f = 2.0 * exp(-EW_PI * EW_PI * k2 / (g.alpha * g.alpha)) / (EW_PI * k2 * g.vol);
    }
    Ck[k] = c;
    Sk[k] = s;
    Fk[k] = f;
}

// reciprocal part of the potential and of its gradient at particles [i0, i1)
__global__ __launch_bounds__(EW_T) void k_ewald_recip(EwaldGeom g, const double *__restrict__ r, int i0, int i1,
                                                      const double *__restrict__ Ck, const double *__restrict__ Sk,
                                                      const double *__restrict__ Fk, double *__restrict__ phi,
                                                      double *__restrict__ grad) {
    const int i = i0 + blockIdx.x * EW_T + threadIdx.x;
    if (i >= i1) return;
    const double xs = r[3 * (size_t)i] * g.invA[0], ys = r[3 * (size_t)i + 1] * g.invA[1], zs = r[3 * (size_t)i + 2] * g.invA[2];
    double ph = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    int k = 0;
    for (int kz = 0; kz <= g.kmax; kz++)
        for (int ky = -g.kmax; ky <= g.kmax; ky++) {
            const double base = fma((double)ky, ys, (double)kz * zs);
            for (int kx = -g.kmax; kx <= g.kmax; kx++, k++) {
                const double f = Fk[k];
                if (f == 0.0) continue; // the same for every lane
                double sn, cs;
                sincospi(2.0 * fma((double)kx, xs, base), &sn, &cs);
                const double ck = Ck[k], sk = Sk[k];
                ph = fma(f, fma(cs, ck, sn * sk), ph);
                const double t = -2.0 * EW_PI * f * fma(sn, ck, -(cs * sk));
                gx = fma(t, kx * g.invA[0], gx);
                gy = fma(t, ky * g.invA[1], gy);
                gz = fma(t, kz * g.invA[2], gz);
            }
        }
    const size_t o = (size_t)(i - i0);
    phi[o] = ph;
    grad[3 * o] = gx;
    grad[3 * o + 1] = gy;
    grad[3 * o + 2] = gz;
}

// real-space part at particles [i0, i1) from the particles of chunk blockIdx.y
__global__ __launch_bounds__(EW_T) void k_ewald_real(EwaldGeom g, const double *__restrict__ r, const double *__restrict__ q,
                                                     int i0, int i1, int per_chunk, double *__restrict__ phi_part,
                                                     double *__restrict__ grad_part) {
    __shared__ double4 tile[EW_T];
    const int i = i0 + blockIdx.x * EW_T + threadIdx.x;
    const bool live = i < i1;
    const int is = live ? i : i0;
    const double xi = r[3 * (size_t)is], yi = r[3 * (size_t)is + 1], zi = r[3 * (size_t)is + 2];
    const int j0 = blockIdx.y * per_chunk, j1 = min(g.N, j0 + per_chunk);
    const double two_a_sqrtpi = 2.0 * g.alpha / sqrt(EW_PI), a2 = g.alpha * g.alpha;
    double ph = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    for (int jb = j0; jb < j1; jb += EW_T) {
        __syncthreads();
        const int j = jb + threadIdx.x;
        tile[threadIdx.x] = j < j1 ? make_double4(r[3 * (size_t)j], r[3 * (size_t)j + 1], r[3 * (size_t)j + 2], q[j])
                                   : make_double4(0.0, 0.0, 0.0, 0.0);
        __syncthreads();
        const int cnt = min(EW_T, j1 - jb);
        for (int t = 0; t < cnt; t++) {
            const double4 p = tile[t];
            double dx = xi - p.x, dy = yi - p.y, dz = zi - p.z;
            if (!g.literal) { // nearest image; rcut <= half the shortest edge, so no other image is inside the cutoff
                dx = fma(-g.A[0], rint(dx * g.invA[0]), dx);
                dy = fma(-g.A[1], rint(dy * g.invA[1]), dy);
                dz = fma(-g.A[2], rint(dz * g.invA[2]), dz);
            }
            const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
            // r2 = 0: the particle itself (ewaldClassical.c:31 skips it by index)
            if (r2 > g.rc2 || r2 == 0.0 || (g.literal && jb + t == i)) continue;
            const double rl = sqrt(r2), ir = 1.0 / rl;
            const double e = erfc(g.alpha * rl) * ir;
            const double de = -(e + two_a_sqrtpi * exp(-a2 * r2)) * ir * ir; // (d/dr)(erfc(alpha r)/r) / r
            ph = fma(p.w, e, ph);
            const double w = p.w * de;
            gx = fma(w, dx, gx);
            gy = fma(w, dy, gy);
            gz = fma(w, dz, gz);
        }
    }
    if (live) {
        const size_t o = (size_t)blockIdx.y * (i1 - i0) + (i - i0);
        phi_part[o] = ph;
        grad_part[3 * o] = gx;
        grad_part[3 * o + 1] = gy;
        grad_part[3 * o + 2] = gz;
    }
}

__global__ __launch_bounds__(256) void k_ewald_qsum(const double *__restrict__ q, int N, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += q[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// phi, grad = real (chunks in order) + reciprocal + self and background terms; per-block energy components
// [real, reciprocal, self, background] of 1/2 q_i phi_i
__global__ __launch_bounds__(EW_T) void k_ewald_finish(EwaldGeom g, const double *__restrict__ q, int i0, int i1, int nchunk,
                                                       const double *__restrict__ phi_part, const double *__restrict__ grad_part,
                                                       const double *__restrict__ phi_rec, const double *__restrict__ grad_rec,
                                                       const double *__restrict__ qsum, double *__restrict__ phi,
                                                       double *__restrict__ grad, double *__restrict__ eparts) {
    __shared__ double red[4][EW_T];
    const int i = i0 + blockIdx.x * EW_T + threadIdx.x;
    double e[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < i1) {
        const size_t o = (size_t)(i - i0), ni = (size_t)(i1 - i0);
        double pr = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
        for (int ch = 0; ch < nchunk; ch++) {
            pr += phi_part[ch * ni + o];
            gx += grad_part[3 * (ch * ni + o)];
            gy += grad_part[3 * (ch * ni + o) + 1];
            gz += grad_part[3 * (ch * ni + o) + 2];
        }
        const double qi = q[i], pk = phi_rec[o];
        const double pself = -qi * 2.0 * g.alpha / sqrt(EW_PI), pbg = -qsum[0] * EW_PI / (g.alpha * g.alpha * g.vol);
        phi[o] = pr + pk + pself + pbg;
        if (grad) {
            grad[3 * o] = gx + grad_rec[3 * o];
            grad[3 * o + 1] = gy + grad_rec[3 * o + 1];
            grad[3 * o + 2] = gz + grad_rec[3 * o + 2];
        }
        e[0] = 0.5 * qi * pr; e[1] = 0.5 * qi * pk; e[2] = 0.5 * qi * pself; e[3] = 0.5 * qi * pbg;
    }
    for (int c = 0; c < 4; c++) red[c][threadIdx.x] = e[c];
    __syncthreads();
    for (int off = EW_T / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int c = 0; c < 4; c++) red[c][threadIdx.x] += red[c][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4) eparts[4 * (size_t)blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void k_ewald_esum(const double *__restrict__ eparts, int nblk, double *__restrict__ out) {
    __shared__ double red[4][256];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256)
        for (int c = 0; c < 4; c++) s[c] += eparts[4 * (size_t)b + c];
    for (int c = 0; c < 4; c++) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int c = 0; c < 4; c++) red[c][threadIdx.x] += red[c][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int c = 0; c < 4; c++) out[c] = red[c][0];
        out[4] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
    }
}

struct EwaldLayout {
    int nkh, sk_chunks, sk_per_chunk, real_chunks, real_per_chunk, nblk_i;
    size_t off_C, off_S, off_F, off_Cp, off_Sp, off_prec, off_grec, off_ppart, off_gpart, off_eparts, total;
};

static EwaldLayout ewald_layout(int N, int kmax, int ni) {
    EwaldLayout L;
    const int cn = 2 * kmax + 1;
    L.nkh = cn * cn * (kmax + 1);
    L.sk_chunks = std::max(1, std::min(16, (N + 4095) / 4096));
    L.sk_per_chunk = (N + L.sk_chunks - 1) / L.sk_chunks;
    L.nblk_i = (ni + EW_T - 1) / EW_T;
    L.real_chunks = std::max(1, std::min(std::min(64, 2048 / L.nblk_i), (N + 4095) / 4096));
    L.real_per_chunk = (N + L.real_chunks - 1) / L.real_chunks;
    size_t o = 8;
    L.off_C = o; o += L.nkh;
    L.off_S = o; o += L.nkh;
    L.off_F = o; o += L.nkh;
    L.off_Cp = o; o += (size_t)L.sk_chunks * L.nkh;
    L.off_Sp = o; o += (size_t)L.sk_chunks * L.nkh;
    L.off_prec = o; o += ni;
    L.off_grec = o; o += 3 * (size_t)ni;
    L.off_ppart = o; o += (size_t)L.real_chunks * ni;
    L.off_gpart = o; o += 3 * (size_t)L.real_chunks * ni;
    L.off_eparts = o; o += 4 * (size_t)L.nblk_i;
    L.total = o;
    return L;
}

} // namespace msm

using namespace msm;

extern "C" long long msm_amd_ewald_work_doubles(int N, double kcut, int ni) {
    if (N < 1 || ni < 1 || !(kcut >= 0) || kcut > 200) return -1;
    return (long long)ewald_layout(N, (int)std::ceil(kcut), ni).total;
}

extern "C" int msm_amd_ewald_device(int N, const double *d_r, const double *d_q, const double A[3], double alpha, double rcut,
                                    double kcut, int mode, int i0, int i1, double *d_phi, double *d_grad, double *d_work,
                                    void *stream, double *u5) {
    if (N < 1 || !d_r || !d_q || !A || !d_phi || !d_work || i0 < 0 || i1 > N || i0 >= i1 || !(alpha > 0) || !(rcut > 0 || (rcut == 0 && mode == MSM_AMD_EWALD_LITERAL)) ||
        !(kcut >= 0) || kcut > 200 || !(A[0] > 0 && A[1] > 0 && A[2] > 0) || (mode != MSM_AMD_EWALD_FULL && mode != MSM_AMD_EWALD_LITERAL)) {
        set_error("msm_amd_ewald_device: bad argument");
        return MSM_AMD_EINVAL;
    }
    if (mode == MSM_AMD_EWALD_FULL && rcut > 0.5 * std::min(A[0], std::min(A[1], A[2]))) {
        set_error("msm_amd_ewald_device: the full sum takes the nearest image only: rcut=%g must not exceed half the shortest cell edge",
                  rcut);
        return MSM_AMD_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    EwaldGeom g;
    for (int d = 0; d < 3; d++) { g.A[d] = A[d]; g.invA[d] = 1.0 / A[d]; }
    g.alpha = alpha; g.rc2 = rcut * rcut; g.kc2 = kcut * kcut; g.vol = A[0] * A[1] * A[2];
    g.kmax = (int)std::ceil(kcut); g.cn = 2 * g.kmax + 1;
    g.literal = mode == MSM_AMD_EWALD_LITERAL;
    g.N = N;
    const int ni = i1 - i0;
    const EwaldLayout L = ewald_layout(N, g.kmax, ni);
    g.nkh = L.nkh;
    double *W = d_work;
    const int kb = (L.nkh + EW_T - 1) / EW_T;
    hipLaunchKernelGGL(k_ewald_qsum, dim3(1), dim3(256), 0, st, d_q, N, W + 5);
    hipLaunchKernelGGL(k_ewald_sk, dim3(kb, L.sk_chunks), dim3(EW_T), 0, st, g, d_r, d_q, L.sk_per_chunk, W + L.off_Cp, W + L.off_Sp);
    hipLaunchKernelGGL(k_ewald_sk_sum, dim3(kb), dim3(EW_T), 0, st, g, L.sk_chunks, W + L.off_Cp, W + L.off_Sp, W + L.off_C,
                       W + L.off_S, W + L.off_F);
    hipLaunchKernelGGL(k_ewald_recip, dim3(L.nblk_i), dim3(EW_T), 0, st, g, d_r, i0, i1, W + L.off_C, W + L.off_S, W + L.off_F,
                       W + L.off_prec, W + L.off_grec);
    hipLaunchKernelGGL(k_ewald_real, dim3(L.nblk_i, L.real_chunks), dim3(EW_T), 0, st, g, d_r, d_q, i0, i1, L.real_per_chunk,
                       W + L.off_ppart, W + L.off_gpart);
    hipLaunchKernelGGL(k_ewald_finish, dim3(L.nblk_i), dim3(EW_T), 0, st, g, d_q, i0, i1, L.real_chunks, W + L.off_ppart,
                       W + L.off_gpart, W + L.off_prec, W + L.off_grec, W + 5, d_phi, d_grad, W + L.off_eparts);
    hipLaunchKernelGGL(k_ewald_esum, dim3(1), dim3(256), 0, st, W + L.off_eparts, L.nblk_i, W);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("Ewald launch failed: %s", hipGetErrorString(e)); return MSM_AMD_EHIP; }
    if (u5) {
        if (hipMemcpyAsync(u5, W, 5 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            set_error("Ewald: copy of the energies failed");
            return MSM_AMD_EHIP;
        }
    }
    return MSM_AMD_OK;
}

// replaces ewaldClassical(), msm/ewaldClassical.h:6, defined at msm/ewaldClassical.c:5-109: host pointers, returns the total
// energy and prints the same three lines to stderr.  A failure prints its reason and returns NaN (the reference has no
// error path).
extern "C" double ewaldClassical(double *r, double *q, int n, double Lx, double Ly, double Lz, double alpha, double cutoffDirect,
                                 double cutoffFourier) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device is visible; msm_amd has no CPU path");
        fprintf(stderr, "ewaldClassical: %s\n", msm_amd_last_error());
        return NAN;
    }
    const double A[3] = {Lx, Ly, Lz};
    const long long nw = msm_amd_ewald_work_doubles(n, cutoffFourier, n);
    double *d_r = nullptr, *d_q = nullptr, *d_phi = nullptr, *d_work = nullptr;
    double u[5] = {0, 0, 0, 0, 0};
    bool ok = n >= 1 && r && q && nw > 0;
    if (!ok) set_error("ewaldClassical: bad argument");
    ok = ok && hipMalloc((void **)&d_r, sizeof(double) * 3 * (size_t)n) == hipSuccess &&
         hipMalloc((void **)&d_q, sizeof(double) * (size_t)n) == hipSuccess &&
         hipMalloc((void **)&d_phi, sizeof(double) * (size_t)n) == hipSuccess &&
         hipMalloc((void **)&d_work, sizeof(double) * (size_t)nw) == hipSuccess;
    ok = ok && hipMemcpy(d_r, r, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_q, q, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    if (ok)
        ok = msm_amd_ewald_device(n, d_r, d_q, A, alpha, cutoffDirect, cutoffFourier, MSM_AMD_EWALD_LITERAL, 0, n, d_phi, nullptr,
                                  d_work, nullptr, u) == MSM_AMD_OK;
    hipFree(d_r); hipFree(d_q); hipFree(d_phi); hipFree(d_work);
    if (!ok) {
        if (!*msm_amd_last_error()) set_error("ewaldClassical: a HIP call failed");
        fprintf(stderr, "ewaldClassical: %s\n", msm_amd_last_error());
        return NAN;
    }
    // ewaldClassical.c:100-104
    fprintf(stderr, "\033[1;34m Classical \033[0m\n");
    fprintf(stderr, "\033[1;34m alpha = %12.4e,L=%12.4e  \033[0m\n", alpha, Lx);
    fprintf(stderr, "\033[1;34m %12.4e %12.4e %12.4e %12.4e = %12.4e \033[0m\n", u[0], u[1], u[2], u[3], u[4]);
    return u[4];
}

// replaces ewaldClassicalRunner(), msm/ewaldClassical.h:5, msm/ewaldClassical.c:111-123: read the data file, print the
// total energy with the reference's format
extern "C" void ewaldClassicalRunner(char *dataFile, double alpha, double cutoffDirect, double cutoffFourier) {
    double *r = nullptr, *q = nullptr, Lx, Ly, Lz;
    int n = 0;
    if (data_read(dataFile, &q, &r, &n, &Lx, &Ly, &Lz) != 0 || n < 1) {
        fprintf(stderr, "ewaldClassicalRunner: cannot read %s\n", dataFile ? dataFile : "(null)");
        return;
    }
    const double potential = ewaldClassical(r, q, n, Lx, Ly, Lz, alpha, cutoffDirect, cutoffFourier);
    printf("%20.15e\n", potential);
    free(r);
    free(q);
}
