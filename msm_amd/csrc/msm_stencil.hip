// Grid-kernel stencils generated on the GPU (position independent setup; the same sums as build_level / build_top of
// msm_host.cpp, i.e. msm/msmPeriodic.c:265-632 and msm/msmSimple.c:1016-1246, in the same summation order).
//
//   level l :  T[|n|]   = sum over periodic images P of g_l(|n h_l - A P|)        one thread per node of an octant
//              G        = T mirrored into all octants
//              K_l      = three 1-D passes with omega' over G                      one thread per output value
//   top     :  chi(k), F(k) = chi(k) (c_x c_y c_z)^2                               one thread per wave vector
//              K_L[n]   = Re sum_k F(k) e^{2 pi i k.n / M_L}, axis by axis          one thread per output value
//
// Every thread adds its terms in the order the host loops do and the file is compiled without FMA contraction, so
// the result differs from the host's only where the device's erf / erfc / exp round differently (a few 1e-16).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <vector>

#include "msm_host.hpp"

#pragma clang fp contract(off)

namespace msm {

namespace {

struct StLevel {
    int p, last;            // spline order; 1 on the coarsest level (kernel = splitting remainder - Ewald long-range part)
    int open;               // open boundaries: the coarsest level keeps the whole remainder gamma(r / a_l) / a_l
    double al, al1;         // cutoffs of this level and the next, 2^l a and 2^(l+1) a
    double beta, lr0;       // Ewald splitting parameter; limit of erf(beta r)/r at r = 0
    double rcut;
    double cg[8];           // gamma(rho) = sum_k cg[k] (rho^2 - 1)^k for rho < 1
    double A[3], h[3];
    int R[3];               // images -R..R
    int n[3];               // nodes of the octant per axis (amax + 1)
};

__device__ double st_gamma(double rho, const StLevel &c) {
    if (rho >= 1.0) return 1.0 / rho;
    const double y = rho * rho - 1.0;
    double s = c.cg[c.p - 1];
    for (int k = c.p - 2; k >= 0; k--) s = s * y + c.cg[k];
    return s;
}

__device__ double st_kernel(const StLevel &c, double r) {
    if (c.last && c.open) return st_gamma(r / c.al, c) / c.al;
    if (c.last) {
        if (r >= c.al) return erfc(c.beta * r) / r;
        const double lr = (r < DBL_EPSILON) ? c.lr0 : erf(c.beta * r) / r;
        return st_gamma(r / c.al, c) / c.al - lr;
    }
    if (r >= c.al1) return 0.0;
    return st_gamma(r / c.al, c) / c.al - st_gamma(r / c.al1, c) / c.al1;
}

__global__ __launch_bounds__(64) void k_st_table(StLevel c, double *__restrict__ T) {
    const int id = blockIdx.x * 64 + threadIdx.x;
    if (id >= c.n[0] * c.n[1] * c.n[2]) return;
    const int jx = id % c.n[0], jy = (id / c.n[0]) % c.n[1], jz = id / (c.n[0] * c.n[1]);
    const double x = jx * c.h[0], y = jy * c.h[1], z = jz * c.h[2];
    double g = 0.0;
    for (int pz = -c.R[2]; pz <= c.R[2]; pz++) {
        const double dz = z - c.A[2] * pz;
        if (fabs(dz) > c.rcut) continue;
        for (int py = -c.R[1]; py <= c.R[1]; py++) {
            const double dy = y - c.A[1] * py;
            if (fabs(dy) > c.rcut) continue;
            for (int px = -c.R[0]; px <= c.R[0]; px++) {
                const double dx = x - c.A[0] * px;
                if (fabs(dx) > c.rcut) continue;
                const double rr = sqrt(dx * dx + dy * dy + dz * dz);
                if (rr <= c.rcut) g += st_kernel(c, rr);
            }
        }
    }
    T[id] = g;
}

struct StExpand { int ext[3], lo[3], n[3]; }; // G[j] = T[|lo + j|]
__global__ __launch_bounds__(256) void k_st_expand(StExpand e, const double *__restrict__ T, double *__restrict__ G) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= e.ext[0] * e.ext[1] * e.ext[2]) return;
    const int jx = id % e.ext[0], jy = (id / e.ext[0]) % e.ext[1], jz = id / (e.ext[0] * e.ext[1]);
    const int ax = abs(e.lo[0] + jx), ay = abs(e.lo[1] + jy), az = abs(e.lo[2] + jz);
    G[id] = T[((size_t)az * e.n[1] + ay) * e.n[0] + ax];
}

// out[o][n][i] = sum_{k=-W..W} w[|k|] in[o][n + k + W][i]: a pass along an axis with `inner` faster and `outer` slower
// elements, n_in = n_out + 2 W
__global__ __launch_bounds__(256) void k_st_pass(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ w,
                                                 int W, int inner, int n_out, int outer) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)inner * n_out * outer) return;
    const int i = (int)(id % inner), n = (int)((id / inner) % n_out), o = (int)(id / ((long)inner * n_out));
    const double *src = in + ((size_t)o * (n_out + 2 * W) + n + W) * inner + i;
    double s = 0.0;
    for (int k = -W; k <= W; k++) s += w[abs(k)] * src[(long)k * inner];
    out[id] = s;
}

struct StChi { int kmax; double A[3], beta, vol; };
__global__ __launch_bounds__(256) void k_st_chi(StChi c, const double *__restrict__ cvx, const double *__restrict__ cvy,
                                                const double *__restrict__ cvz, double *__restrict__ chi, double *__restrict__ F) {
    const int cn = 2 * c.kmax + 1;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= cn * cn * cn) return;
    const int kx = id % cn - c.kmax, ky = (id / cn) % cn - c.kmax, kz = id / (cn * cn) - c.kmax;
    double x = 0.0, f = 0.0;
    if (kx != 0 || ky != 0 || kz != 0) {
        const double k2 = kz * kz / (c.A[2] * c.A[2]) + ky * ky / (c.A[1] * c.A[1]) + kx * kx / (c.A[0] * c.A[0]);
        x = exp(-PI * PI * k2 / (c.beta * c.beta)) / (PI * k2 * c.vol);
        const double cc = cvz[abs(kz)] * cvy[abs(ky)] * cvx[abs(kx)];
        f = x * cc * cc;
    }
    chi[id] = x;
    F[2 * (size_t)id] = f;
    F[2 * (size_t)id + 1] = 0.0;
}

// out[o][n][i] = sum_{k=0..cn-1} in[o][k][i] * root[((k - kmax) n) mod M], complex; `real_out` keeps the real part only
__global__ __launch_bounds__(256) void k_st_dft(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ root,
                                                int kmax, int M, int inner, int outer, int real_out) {
    const int cn = 2 * kmax + 1;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)inner * M * outer) return;
    const int i = (int)(id % inner), n = (int)((id / inner) % M), o = (int)(id / ((long)inner * M));
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < cn; k++) {
        long t = ((long)(k - kmax) * n) % M;
        if (t < 0) t += M;
        const double c = root[2 * t], s = root[2 * t + 1];
        const double *v = in + 2 * (((size_t)o * cn + k) * inner + i);
        sr += v[0] * c - v[1] * s;
        si += v[0] * s + v[1] * c;
    }
    if (real_out) out[id] = sr;
    else { out[2 * id] = sr; out[2 * id + 1] = si; }
}

struct DevBuf {
    double *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t n) {
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(double));
        if (e != hipSuccess) { set_error("stencil setup: hipMalloc of %zu bytes failed: %s", n * sizeof(double), hipGetErrorString(e)); return MSM_AMD_ENOMEM; }
        return MSM_AMD_OK;
    }
    int upload(const std::vector<double> &v) {
        int rc = alloc(v.size());
        if (rc) return rc;
        if (hipMemcpy(p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { set_error("stencil setup: upload failed"); return MSM_AMD_EHIP; }
        return MSM_AMD_OK;
    }
};

inline unsigned blocks(long n, int per) { return (unsigned)((n + per - 1) / per); }

int device_level(HostPlan &hp, int l, void *) {
    LevelStencil &S = hp.K[l];
    LevelGeom geo;
    level_geometry(hp, l, S, geo);
    const int W = geo.W;
    StLevel c;
    c.p = hp.d.p;
    c.last = (l == hp.d.L - 1) ? 1 : 0;
    c.open = hp.d.open_boundary ? 1 : 0;
    c.al = std::ldexp(hp.d.a, l);
    c.al1 = std::ldexp(hp.d.a, l + 1);
    c.beta = hp.d.beta;
    c.lr0 = 2.0 * hp.d.beta / std::sqrt(PI);
    c.rcut = geo.rcut;
    double cd[8];
    gamma_poly(hp.d.p, c.cg, cd);
    StExpand e;
    for (int d = 0; d < 3; d++) {
        c.A[d] = hp.d.A[d]; c.h[d] = hp.hl[l][d]; c.R[d] = geo.R[d]; c.n[d] = geo.amax[d] + 1;
        e.ext[d] = geo.ext[d]; e.lo[d] = S.first[d] - W; e.n[d] = c.n[d];
    }
    const long nT = (long)c.n[0] * c.n[1] * c.n[2], nG = (long)geo.ext[0] * geo.ext[1] * geo.ext[2];
    const long n1 = (long)geo.ext[2] * geo.ext[1] * S.size[0], n2 = (long)geo.ext[2] * S.size[1] * S.size[0];
    const long n3 = (long)S.size[2] * S.size[1] * S.size[0];
    if (nG > (1L << 28)) { set_error("stencil setup: kernel table of %ld nodes is too large", nG); return MSM_AMD_ERANGE; }
    DevBuf T, G, T1, T2, K, w;
    int rc;
    if ((rc = T.alloc(nT)) || (rc = G.alloc(nG)) || (rc = T1.alloc(n1)) || (rc = T2.alloc(n2)) || (rc = K.alloc(n3)) ||
        (rc = w.upload(hp.wprime)))
        return rc;
    hipLaunchKernelGGL(k_st_table, dim3(blocks(nT, 64)), dim3(64), 0, 0, c, T.p);
    hipLaunchKernelGGL(k_st_expand, dim3(blocks(nG, 256)), dim3(256), 0, 0, e, T.p, G.p);
    hipLaunchKernelGGL(k_st_pass, dim3(blocks(n1, 256)), dim3(256), 0, 0, G.p, T1.p, w.p, W, 1, S.size[0], geo.ext[2] * geo.ext[1]);
    hipLaunchKernelGGL(k_st_pass, dim3(blocks(n2, 256)), dim3(256), 0, 0, T1.p, T2.p, w.p, W, S.size[0], S.size[1], geo.ext[2]);
    hipLaunchKernelGGL(k_st_pass, dim3(blocks(n3, 256)), dim3(256), 0, 0, T2.p, K.p, w.p, W, S.size[0] * S.size[1], S.size[2], 1);
    S.v.assign((size_t)n3, 0.0);
    hipError_t err = hipMemcpy(S.v.data(), K.p, (size_t)n3 * sizeof(double), hipMemcpyDeviceToHost);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) { set_error("stencil setup (level %d): %s", l, hipGetErrorString(err)); return MSM_AMD_EHIP; }
    return MSM_AMD_OK;
}

int device_top(HostPlan &hp, void *) {
    if (hp.d.open_boundary) { top_open(hp); return MSM_AMD_OK; }
    top_prepare(hp);
    const int L = hp.d.L, kmax = hp.d.kmax, cn = 2 * kmax + 1;
    const int *ML = hp.Ml[L - 1];
    const long nk = (long)cn * cn * cn;
    StChi c;
    c.kmax = kmax;
    for (int d = 0; d < 3; d++) c.A[d] = hp.d.A[d];
    c.beta = hp.d.beta;
    c.vol = hp.d.A[0] * hp.d.A[1] * hp.d.A[2];
    DevBuf cv[3], root[3], chi, F, S1, S2, KL;
    int rc;
    for (int d = 0; d < 3; d++) {
        std::vector<double> ri;
        top_roots(ML[d], ri);
        if ((rc = cv[d].upload(hp.cvec[d])) || (rc = root[d].upload(ri))) return rc;
    }
    const long n1 = (long)cn * cn * ML[0], n2 = (long)cn * ML[1] * ML[0], n3 = (long)ML[2] * ML[1] * ML[0];
    if ((rc = chi.alloc(nk)) || (rc = F.alloc(2 * nk)) || (rc = S1.alloc(2 * n1)) || (rc = S2.alloc(2 * n2)) || (rc = KL.alloc(n3))) return rc;
    hipLaunchKernelGGL(k_st_chi, dim3(blocks(nk, 256)), dim3(256), 0, 0, c, cv[0].p, cv[1].p, cv[2].p, chi.p, F.p);
    hipLaunchKernelGGL(k_st_dft, dim3(blocks(n1, 256)), dim3(256), 0, 0, F.p, S1.p, root[0].p, kmax, ML[0], 1, cn * cn, 0);
    hipLaunchKernelGGL(k_st_dft, dim3(blocks(n2, 256)), dim3(256), 0, 0, S1.p, S2.p, root[1].p, kmax, ML[1], ML[0], cn, 0);
    hipLaunchKernelGGL(k_st_dft, dim3(blocks(n3, 256)), dim3(256), 0, 0, S2.p, KL.p, root[2].p, kmax, ML[2], ML[1] * ML[0], 1, 1);
    hp.chi.assign((size_t)nk, 0.0);
    hp.KL.v.assign((size_t)n3, 0.0);
    hipError_t err = hipMemcpy(hp.chi.data(), chi.p, (size_t)nk * sizeof(double), hipMemcpyDeviceToHost);
    if (err == hipSuccess) err = hipMemcpy(hp.KL.v.data(), KL.p, (size_t)n3 * sizeof(double), hipMemcpyDeviceToHost);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) { set_error("stencil setup (top level): %s", hipGetErrorString(err)); return MSM_AMD_EHIP; }
    hp.chisum = 0.0;
    for (long i = 0; i < nk; i++) hp.chisum += hp.chi[i]; // the order of the host loop (the k = 0 entry is zero)
    return MSM_AMD_OK;
}

const StencilBackend g_device_backend = {device_level, device_top, nullptr};

} // namespace

const StencilBackend *device_stencil_backend() { return &g_device_backend; }

} // namespace msm
