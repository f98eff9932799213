// Host-side plan construction (position independent, once per plan) and the
// small C helpers of the drop-in layer.
#include "msm_host.hpp"

#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace msm {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

const char *last_error_cstr() { return g_err.c_str(); }

// Binomial series of (1+y)^(-1/2) truncated after p terms, y = rho^2 - 1
// (msm/msmPeriodic.c:22-34; Horner form msm/msmSimple.c:609-638).
//   gamma(rho)        = sum_{k<p}   cg[k] y^k
//   -gamma'(rho)/rho  = sum_{k<p-1} cd[k] y^k
void gamma_poly(int p, double cg[8], double cd[8]) {
    for (int k = 0; k < 8; k++) cg[k] = cd[k] = 0.0;
    cg[0] = 1.0;
    for (int k = 1; k < p; k++) cg[k] = cg[k - 1] * (0.5 - k) / k;
    for (int k = 0; k + 1 < p; k++) cd[k] = -2.0 * (k + 1) * cg[k + 1];
}

double gamma_eval(double rho, int p) {
    if (rho >= 1.0) return 1.0 / rho;
    double cg[8], cd[8];
    gamma_poly(p, cg, cd);
    double y = rho * rho - 1.0, s = cg[p - 1];
    for (int k = p - 2; k >= 0; k--) s = s * y + cg[k];
    return s;
}

// cardinal B-spline of order k on [0,k) (msm/msmPeriodic.c:10-19)
static double bspline_rec(int k, double u) {
    if (k == 1) return (u >= 0 && u < 1) ? 1.0 : 0.0;
    return u / (double)(k - 1) * bspline_rec(k - 1, u) + (k - u) / (double)(k - 1) * bspline_rec(k - 1, u - 1.0);
}

// The centred B-spline of order p on the k-th unit interval of its support, as a polynomial in the offset f
// from the interval's left end: Phi(f + p/2 - 1 - k) = B_p(f + u), u = p - 1 - k, and
// B_p(v) = 1/(p-1)! sum_j (-1)^j C(p,j) (v - j)_+^(p-1)  (the closed pieces of msm/msmLibrary.h:120-142).
void spline_pieces(int p, double piece[6][6], double dpiece[6][6]) {
    for (int k = 0; k < 6; k++)
        for (int n = 0; n < 6; n++) piece[k][n] = dpiece[k][n] = 0.0;
    auto binom = [](int n, int k) {
        long double c = 1.0L;
        for (int i = 1; i <= k; i++) c = c * (n - k + i) / i;
        return c;
    };
    long double fact = 1.0L;
    for (int i = 2; i <= p - 1; i++) fact *= i;
    for (int k = 0; k < p; k++) {
        const int u = p - 1 - k;
        long double c[6] = {0, 0, 0, 0, 0, 0};
        for (int j = 0; j <= u; j++) {
            const long double sgn = (j & 1) ? -1.0L : 1.0L, m = (long double)(u - j);
            for (int n = 0; n <= p - 1; n++) {
                long double mp = 1.0L; // m^(p-1-n), with 0^0 = 1
                for (int e = 0; e < p - 1 - n; e++) mp *= m;
                c[n] += sgn * binom(p, j) * binom(p - 1, n) * mp;
            }
        }
        for (int n = 0; n <= p - 1; n++) piece[k][n] = (double)(c[n] / fact);
        for (int n = 0; n + 1 <= p - 1; n++) dpiece[k][n] = (double)((n + 1) * c[n + 1] / fact);
    }
}

// J_j = C(p, p/2 + j) / 2^(p-1)   (msm/msmLibrary.h:152-153)
void two_scale(int p, double J[4]) {
    for (int j = 0; j < 4; j++) J[j] = 0.0;
    for (int j = 0; j <= p / 2; j++) {
        double c = 1.0;
        int k = p / 2 + j;
        for (int i = 1; i <= k; i++) c = c * (p - k + i) / i;
        J[j] = std::ldexp(c, -(p - 1));
    }
}

// Quasi-interpolation weights omega' (msm/msmSimple.c:334-516; for p=4, mu=10
// they are the table msm/msmLibrary.h:156-170).
std::vector<double> compute_wprime(int p, int mu) {
    const int h = p / 2, n = p - 1;
    // tabulated constants of the method (msm/msmSimple.c:343-350)
    const double psi4[2] = {6.4814814814814811e-02, 9.2592592592592587e-03};
    const double psi6[4] = {-3.0804398148148150e-02, -8.9800347222222226e-03, -6.0590277777777780e-04,
                            -1.1863425925925925e-05};
    const double lead4[3] = {-3.3333333333333331e-01, 1.6666666666666665e+00, -3.3333333333333331e-01};
    const double lead6[5] = {1.7083333333333334e-01, -1.1833333333333333e+00, 3.0249999999999999e+00,
                             -1.1833333333333333e+00, 1.7083333333333334e-01};
    const double cd4[5] = {1, -4, 6, -4, 1}, cd6[7] = {1, -6, 15, -20, 15, -6, 1};
    const double *psi = (p == 4) ? psi4 : psi6, *lead = (p == 4) ? lead4 : lead6, *cdiff = (p == 4) ? cd4 : cd6;

    // spectral factor g of the symmetric band Phi(0), Phi(1), ... by fixed-point iteration
    std::vector<double> band(h), g(h, 0.0), gprev(h, 0.0);
    for (int i = 0; i < h; i++) band[i] = bspline_rec(p, i + h);
    gprev[0] = 1.0;
    for (int it = 0;; it++) {
        double tail = 0.0;
        for (int k = 1; k < h; k++) tail += gprev[k] * gprev[k];
        g[0] = std::sqrt(band[0] - tail);
        for (int j = 1; j < h; j++) {
            double s = 0.0;
            for (int i = j + 1; i < h; i++) s += gprev[i - j] * gprev[i];
            g[j] = (band[j] - s) / g[0];
        }
        double dn = 0.0, nn = 0.0;
        for (int k = 0; k < h; k++) {
            dn += (g[k] - gprev[k]) * (g[k] - gprev[k]);
            nn += g[k] * g[k];
        }
        if (std::sqrt(dn) / std::sqrt(nn) < 1e-16 || it > 1000) break;
        gprev = g;
    }
    // g2 = g * g (length p-1)
    std::vector<double> g2(n, 0.0);
    for (int i = 0; i < n; i++) {
        int lo = (i > h - 1) ? i - (h - 1) : 0, hi = (i < h - 1) ? i : h - 1;
        for (int j = lo; j <= hi; j++) g2[i] += g[j] * g[i - j];
    }
    // xi by back substitution, then the (p-1)x(p-1) Toeplitz + Hankel system
    std::vector<double> xi(n, 0.0), A(n * n, 0.0);
    for (int i = p - 3; i >= 0; i--) {
        double s = 0.0;
        for (int j = i + 1; j <= p - 3; j++) s += xi[j] * g2[j - i];
        xi[i] = (psi[i] - s) / g2[0];
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) A[i * n + j] = g2[i - j];
    for (int j = 1; j <= p - 2; j++)
        for (int i = 0; i <= p - 2 - j; i++) A[i * n + j] += g2[i + j];
    for (int j = 0; j < n - 1; j++)
        for (int i = j + 1; i < n; i++) {
            double m = -A[i * n + j] / A[j * n + j];
            for (int k = 0; k < n; k++) A[i * n + k] += m * A[j * n + k];
            xi[i] += m * xi[j];
        }
    for (int j = n - 1; j >= 0; j--) {
        xi[j] = xi[j] / A[j * n + j];
        for (int i = j - 1; i >= 0; i--) xi[i] = xi[i] - A[i * n + j] * xi[j];
    }
    // c_n by the linear recurrence, then the stencil
    std::vector<double> c(mu + p + 8, 0.0);
    for (int i = 0; i < n && i < (int)c.size(); i++) c[i] = xi[i];
    for (int i = p - 1; i < (int)c.size(); i++) {
        double s = 0.0;
        for (int k = 1; k <= p - 2; k++) s += c[i - k] * g2[k];
        c[i] = -s / g2[0];
    }
    std::vector<double> st(2 * mu + p + 2, 0.0);
    for (int i = 1; i <= p - 1; i++) st[i + mu] += lead[i - 1];
    for (int m = -mu; m <= mu; m++)
        for (int i = 1; i <= p + 1; i++) st[i - 1 + m + mu] += cdiff[i - 1] * c[std::abs(m)];
    std::vector<double> w(mu + h + 1);
    for (int i = mu + h; i <= 2 * mu + p; i++) w[i - mu - h] = st[i];
    return w;
}

static double level_kernel(const HostPlan &hp, int l, double r) {
    const int p = hp.d.p, L = hp.d.L;
    const double al = std::ldexp(hp.d.a, l), al1 = std::ldexp(hp.d.a, l + 1), beta = hp.d.beta;
    if (l == L - 1 && hp.d.open_boundary) return gamma_eval(r / al, p) / al; // the whole remainder, 1 / r beyond a_l
    if (l == L - 1) {
        if (r >= al) return std::erfc(beta * r) / r;
        const double lr = (r < DBL_EPSILON) ? 2.0 * beta / std::sqrt(PI) : std::erf(beta * r) / r;
        return gamma_eval(r / al, p) / al - lr;
    }
    if (r >= al1) return 0.0;
    return gamma_eval(r / al, p) / al - gamma_eval(r / al1, p) / al1;
}

// K_l[n] = sum_k w'_|kx| w'_|ky| w'_|kz| sum_P g_l(|(n+k) h_l - A P|)   (msm/msmPeriodic.c:413-598,
// msm/msmSimple.c:1016-1074) evaluated as: image-summed table over n+k, then three 1-D passes with w'.
void level_geometry(const HostPlan &hp, int l, LevelStencil &S, LevelGeom &G) {
    const int W = hp.d.mu + hp.d.p / 2;
    G.W = W;
    for (int d = 0; d < 3; d++) {
        // msmPeriodic.c:435-453; in literal mode the z range is taken from Mx as the reference does (:437)
        const int Md = (hp.literal && d == 2) ? hp.Ml[l][0] : hp.Ml[l][d];
        const double half = Md / 2.0;
        int lo, hi;
        if (hp.khat[d] > half) {
            lo = (int)std::floor(-half) + 1;
            hi = (int)std::floor(half);
        } else {
            lo = -hp.khat[d] + 1;
            hi = hp.khat[d] - 1;
        }
        // open boundaries: the coarsest level couples every pair of nodes, differences -(M-1)..M-1
        if (hp.d.open_boundary && l == hp.d.L - 1) { lo = -(hp.Ml[l][d] - 1); hi = hp.Ml[l][d] - 1; }
        S.first[d] = lo;
        S.size[d] = hi - lo + 1;
        G.ext[d] = S.size[d] + 2 * W;
    }
    G.rcut = (l == hp.d.L - 1) ? std::fmax(std::ldexp(hp.d.a, l), 7.5 / hp.d.beta) : std::ldexp(hp.d.a, l + 1);
    if (hp.d.open_boundary && l == hp.d.L - 1) G.rcut = 1e300;
    for (int d = 0; d < 3; d++) {
        int jm = std::max(std::abs(S.first[d]), std::abs(S.first[d] + S.size[d] - 1)) + W;
        G.R[d] = (int)std::ceil((jm * hp.hl[l][d] + G.rcut) / hp.d.A[d]);
        if (hp.literal) G.R[d] = 5; // the reference sums images -5..5 whatever the cell (msmPeriodic.c:536-538)
        if (hp.d.open_boundary) G.R[d] = 0;
        G.amax[d] = std::max(std::abs(S.first[d] - W), std::abs(S.first[d] - W + G.ext[d] - 1));
    }
}

// part = 0: the level's kernel; 1 / 2: the two terms of the coarsest level's kernel on their own, gamma(r/a_l)/a_l and
// erf(beta r)/r, summed over the same images without a support radius (they decay like 1/r) - the stencils behind the
// reference's diagnostic energies ulong_real_g1 / ulong_real_gLR (sl_part1 / sl_part2, msmPeriodic.c:555-590)
static void build_level_into(const HostPlan &hp, int l, int part, LevelStencil &S) {
    LevelGeom geo;
    level_geometry(hp, l, S, geo);
    const int W = geo.W;
    const int *ext = geo.ext, *R = geo.R, *amax = geo.amax;
    const double rcut = part ? 1e300 : geo.rcut;
    const double al_ = std::ldexp(hp.d.a, l), beta_ = hp.d.beta;
    auto kernel = [&](double r) {
        if (part == 1) return gamma_eval(r / al_, hp.d.p) / al_;
        if (part == 2) return (r < DBL_EPSILON) ? 2.0 * beta_ / std::sqrt(PI) : std::erf(beta_ * r) / r;
        return level_kernel(hp, l, r);
    };
    // The image sum is even in every coordinate (the images come in +- pairs), so it is evaluated once per
    // (|nx|, |ny|, |nz|) and copied to the other octants: an eighth of the kernel evaluations.
    std::vector<double> T((size_t)(amax[0] + 1) * (amax[1] + 1) * (amax[2] + 1));
    for (int jz = 0; jz <= amax[2]; jz++)
        for (int jy = 0; jy <= amax[1]; jy++)
            for (int jx = 0; jx <= amax[0]; jx++) {
                const double x = jx * hp.hl[l][0], y = jy * hp.hl[l][1], z = jz * hp.hl[l][2];
                double g = 0.0;
                for (int pz = -R[2]; pz <= R[2]; pz++) {
                    const double dz = z - hp.d.A[2] * pz;
                    if (std::fabs(dz) > rcut) continue;
                    for (int py = -R[1]; py <= R[1]; py++) {
                        const double dy = y - hp.d.A[1] * py;
                        if (std::fabs(dy) > rcut) continue;
                        for (int px = -R[0]; px <= R[0]; px++) {
                            const double dx = x - hp.d.A[0] * px;
                            if (std::fabs(dx) > rcut) continue;
                            const double rr = std::sqrt(dx * dx + dy * dy + dz * dz);
                            if (rr <= rcut) g += kernel(rr);
                        }
                    }
                }
                T[((size_t)jz * (amax[1] + 1) + jy) * (amax[0] + 1) + jx] = g;
            }
    std::vector<double> G((size_t)ext[0] * ext[1] * ext[2]);
    for (int jz = 0; jz < ext[2]; jz++)
        for (int jy = 0; jy < ext[1]; jy++)
            for (int jx = 0; jx < ext[0]; jx++) {
                const int ax = std::abs(S.first[0] - W + jx), ay = std::abs(S.first[1] - W + jy), az = std::abs(S.first[2] - W + jz);
                G[((size_t)jz * ext[1] + jy) * ext[0] + jx] = T[((size_t)az * (amax[1] + 1) + ay) * (amax[0] + 1) + ax];
            }
    const std::vector<double> &w = hp.wprime;
    // x pass
    std::vector<double> T1((size_t)ext[2] * ext[1] * S.size[0], 0.0);
    for (int jz = 0; jz < ext[2]; jz++)
        for (int jy = 0; jy < ext[1]; jy++)
            for (int nx = 0; nx < S.size[0]; nx++) {
                double s = 0.0;
                const double *row = &G[((size_t)jz * ext[1] + jy) * ext[0] + nx + W];
                for (int k = -W; k <= W; k++) s += w[std::abs(k)] * row[k];
                T1[((size_t)jz * ext[1] + jy) * S.size[0] + nx] = s;
            }
    // y pass
    std::vector<double> T2((size_t)ext[2] * S.size[1] * S.size[0], 0.0);
    for (int jz = 0; jz < ext[2]; jz++)
        for (int ny = 0; ny < S.size[1]; ny++)
            for (int nx = 0; nx < S.size[0]; nx++) {
                double s = 0.0;
                for (int k = -W; k <= W; k++) s += w[std::abs(k)] * T1[((size_t)jz * ext[1] + ny + k + W) * S.size[0] + nx];
                T2[((size_t)jz * S.size[1] + ny) * S.size[0] + nx] = s;
            }
    // z pass
    S.v.assign((size_t)S.size[0] * S.size[1] * S.size[2], 0.0);
    for (int nz = 0; nz < S.size[2]; nz++)
        for (int ny = 0; ny < S.size[1]; ny++)
            for (int nx = 0; nx < S.size[0]; nx++) {
                double s = 0.0;
                for (int k = -W; k <= W; k++) s += w[std::abs(k)] * T2[((size_t)(nz + k + W) * S.size[1] + ny) * S.size[0] + nx];
                S.v[((size_t)nz * S.size[1] + ny) * S.size[0] + nx] = s;
            }
}
static void build_level(HostPlan &hp, int l) { build_level_into(hp, l, 0, hp.K[l]); }

void build_literal_parts(const HostPlan &hp, LevelStencil &g1_part, LevelStencil &gLR_part) {
    build_level_into(hp, hp.d.L - 1, 1, g1_part);
    build_level_into(hp, hp.d.L - 1, 2, gLR_part);
}

// Fourier top level on the coarsest grid (msm/msmPeriodic.c:265-391, msm/msmSimple.c:1181-1246):
// KL[n] = sum_{k != 0, |k_d| <= kmax} chi(k) (c_x c_y c_z)^2 cos(2 pi sum_d k_d n_d / ML_d), n in [0,ML)^3,
// summed axis by axis with complex partial sums.
void top_prepare(HostPlan &hp) {
    const int L = hp.d.L, kmax = hp.d.kmax, p2 = hp.d.p / 2;
    const int *ML = hp.Ml[L - 1];
    const double knots4[3] = {4.0 / 6.0, 1.0 / 6.0, 0.0}, knots6[3] = {66.0 / 120.0, 26.0 / 120.0, 1.0 / 120.0};
    const double *kn = (hp.d.p == 4) ? knots4 : knots6; // Phi at integer knots (msmPeriodic.c:71-72)
    for (int d = 0; d < 3; d++) {
        hp.cvec[d].assign(kmax + 1, 0.0);
        for (int k = 0; k <= kmax; k++) {
            double c = kn[0];
            for (int m = 1; m <= p2 - 1; m++) c += 2.0 * std::cos(2 * PI * k * m / (double)ML[d]) * kn[m];
            hp.cvec[d][k] = 1.0 / c;
        }
    }
    for (int d = 0; d < 3; d++) {
        hp.KL.size[d] = ML[d];
        hp.KL.first[d] = 0;
    }
}

// e^{2 pi i t / M} for t in [0, M): k n is reduced mod M first, so every phase of the top-level sum is one of these
void top_roots(int M, std::vector<double> &re_im) {
    re_im.resize(2 * (size_t)M);
    for (int t = 0; t < M; t++) {
        const double th = 2 * PI * (double)t / (double)M;
        re_im[2 * t] = std::cos(th);
        re_im[2 * t + 1] = std::sin(th);
    }
}

// open boundaries: no Fourier level
void top_open(HostPlan &hp) {
    top_prepare(hp);
    const int cn = 2 * hp.d.kmax + 1;
    const int *ML = hp.Ml[hp.d.L - 1];
    hp.chi.assign((size_t)cn * cn * cn, 0.0);
    hp.chisum = 0.0;
    hp.KL.v.assign((size_t)ML[0] * ML[1] * ML[2], 0.0);
}

static void build_top(HostPlan &hp) {
    if (hp.d.open_boundary) { top_open(hp); return; }
    top_prepare(hp);
    const int L = hp.d.L, kmax = hp.d.kmax, cn = 2 * kmax + 1;
    const int *ML = hp.Ml[L - 1];
    const double vol = hp.d.A[0] * hp.d.A[1] * hp.d.A[2], beta = hp.d.beta;
    hp.chi.assign((size_t)cn * cn * cn, 0.0);
    hp.chisum = 0.0;
    std::vector<double> F((size_t)cn * cn * cn, 0.0);
    for (int kz = -kmax; kz <= kmax; kz++)
        for (int ky = -kmax; ky <= kmax; ky++)
            for (int kx = -kmax; kx <= kmax; kx++) {
                const size_t off = ((size_t)(kz + kmax) * cn + (ky + kmax)) * cn + (kx + kmax);
                if (kx == 0 && ky == 0 && kz == 0) continue;
                const double k2 = kz * kz / (hp.d.A[2] * hp.d.A[2]) + ky * ky / (hp.d.A[1] * hp.d.A[1]) +
                                  kx * kx / (hp.d.A[0] * hp.d.A[0]);
                const double x = std::exp(-PI * PI * k2 / (beta * beta)) / (PI * k2 * vol);
                const double c = hp.cvec[2][std::abs(kz)] * hp.cvec[1][std::abs(ky)] * hp.cvec[0][std::abs(kx)];
                hp.chi[off] = x;
                hp.chisum += x;
                F[off] = x * c * c;
            }
    typedef std::complex<double> cd;
    // e^{2 pi i t / M} for t in [0, M), per axis: k n is reduced mod M first, so every phase is one of these
    std::vector<cd> roots[3];
    for (int d = 0; d < 3; d++) {
        std::vector<double> ri;
        top_roots(ML[d], ri);
        roots[d].resize(ML[d]);
        for (int t = 0; t < ML[d]; t++) roots[d][t] = cd(ri[2 * t], ri[2 * t + 1]);
    }
    auto phase = [&](int k, int n, int d) {
        long t = ((long)k * n) % ML[d];
        if (t < 0) t += ML[d];
        return roots[d][t];
    };
    std::vector<cd> S1((size_t)cn * cn * ML[0]), S2((size_t)cn * ML[1] * ML[0]);
    for (int kz = 0; kz < cn; kz++)
        for (int ky = 0; ky < cn; ky++)
            for (int nx = 0; nx < ML[0]; nx++) {
                cd s = 0.0;
                for (int kx = 0; kx < cn; kx++) s += F[((size_t)kz * cn + ky) * cn + kx] * phase(kx - kmax, nx, 0);
                S1[((size_t)kz * cn + ky) * ML[0] + nx] = s;
            }
    for (int kz = 0; kz < cn; kz++)
        for (int ny = 0; ny < ML[1]; ny++)
            for (int nx = 0; nx < ML[0]; nx++) {
                cd s = 0.0;
                for (int ky = 0; ky < cn; ky++) s += S1[((size_t)kz * cn + ky) * ML[0] + nx] * phase(ky - kmax, ny, 1);
                S2[((size_t)kz * ML[1] + ny) * ML[0] + nx] = s;
            }
    hp.KL.v.assign((size_t)ML[0] * ML[1] * ML[2], 0.0);
    for (int nz = 0; nz < ML[2]; nz++)
        for (int ny = 0; ny < ML[1]; ny++)
            for (int nx = 0; nx < ML[0]; nx++) {
                cd s = 0.0;
                for (int kz = 0; kz < cn; kz++) s += S2[((size_t)kz * ML[1] + ny) * ML[0] + nx] * phase(kz - kmax, nz, 2);
                hp.KL.v[((size_t)nz * ML[1] + ny) * ML[0] + nx] = s.real();
            }
}

int build_host_plan(HostPlan &hp, const StencilBackend *be) {
    const msm_amd_plan_desc &d = hp.d;
    if (d.p != 4 && d.p != 6) { set_error("B-spline order p must be 4 or 6 (got %d)", d.p); return MSM_AMD_EINVAL; }
    if (d.L < 1 || d.L > LMAX) { set_error("number of grid levels L=%d out of range", d.L); return MSM_AMD_EINVAL; }
    if (d.mu < 0 || d.mu > 40) { set_error("mu=%d out of range", d.mu); return MSM_AMD_EINVAL; }
    if (d.N < 1 || !(d.a > 0) || !(d.beta > 0) || d.kmax < 0) { set_error("bad N/a/beta/kmax"); return MSM_AMD_EINVAL; }
    for (int k = 0; k < 3; k++) {
        if (d.M[k] < 1 || (d.M[k] % (1 << (d.L - 1))) != 0 || !(d.A[k] > 0)) {
            set_error("grid M[%d]=%d must be a positive multiple of 2^(L-1)=%d", k, d.M[k], 1 << (d.L - 1));
            return MSM_AMD_EINVAL;
        }
    }
    if (d.open_boundary && hp.literal) {
        set_error("open-boundary plans have no literal mode");
        return MSM_AMD_EINVAL;
    }
    if (d.open_boundary && (long)(d.M[0] >> (d.L - 1)) * (d.M[1] >> (d.L - 1)) * (d.M[2] >> (d.L - 1)) > 48L * 48 * 48) {
        // the coarsest level of an open plan couples every pair of its nodes
        set_error("open-boundary plan: coarsest grid %d x %d x %d is too large for the all-pairs top level, use more levels",
                  d.M[0] >> (d.L - 1), d.M[1] >> (d.L - 1), d.M[2] >> (d.L - 1));
        return MSM_AMD_ERANGE;
    }
    if (hp.literal && (d.L != 1 || d.p != 4 || d.mu + d.p / 2 > 13)) {
        set_error("literal mode needs L=1, p=4, mu<=11 (msmPeriodic.c:100-104, msmLibrary.c:64-66, msmLibrary.h:14)");
        return MSM_AMD_EINVAL;
    }
    for (int l = 0; l < d.L; l++)
        for (int k = 0; k < 3; k++) {
            hp.Ml[l][k] = d.M[k] >> l;
            hp.hl[l][k] = d.A[k] / hp.Ml[l][k];
        }
    for (int k = 0; k < 3; k++) hp.khat[k] = (int)std::ceil(2.0 * d.a / hp.hl[0][k]); // msmPeriodic.c:431-433
    if (hp.literal) {
        // the reference indexes its mu = 10 table whatever mu is (msmLibrary.h:156-170, msmPeriodic.c:588)
        std::vector<double> w10 = compute_wprime(d.p, 10);
        hp.wprime.assign(d.mu + d.p / 2 + 1, 0.0);
        for (size_t i = 0; i < hp.wprime.size() && i < w10.size(); i++) hp.wprime[i] = w10[i];
    } else {
        hp.wprime = compute_wprime(d.p, d.mu);
    }
    two_scale(d.p, hp.J);
    hp.K.assign(d.L, LevelStencil());
    for (int l = 0; l < d.L; l++) {
        if (!be) { build_level(hp, l); continue; }
        const int rc = be->level(hp, l, be->ctx);
        if (rc) return rc;
    }
    if (be) {
        const int rc = be->top(hp, be->ctx);
        if (rc) return rc;
    } else {
        build_top(hp);
    }
    const double vol = d.A[0] * d.A[1] * d.A[2];
    hp.csr = d.open_boundary ? 0.0 : PI / (d.beta * d.beta * vol); // msmPeriodic.c:294
    hp.gamma0 = gamma_eval(0.0, d.p);
    // sum over the images of a particle with itself inside the cutoff (msmLibrary.c:141-167 with i == j)
    hp.self_image_sum = 0.0;
    {
        const int Rx = d.open_boundary ? 0 : (int)(d.a / d.A[0]), Ry = d.open_boundary ? 0 : (int)(d.a / d.A[1]),
                  Rz = d.open_boundary ? 0 : (int)(d.a / d.A[2]);
        for (int px = -Rx; px <= Rx; px++)
            for (int py = -Ry; py <= Ry; py++)
                for (int pz = -Rz; pz <= Rz; pz++) {
                    if (px == 0 && py == 0 && pz == 0) continue;
                    const double x = d.A[0] * px, y = d.A[1] * py, z = d.A[2] * pz, r2 = x * x + y * y + z * z;
                    if (r2 < d.a * d.a) {
                        const double rl = std::sqrt(r2);
                        hp.self_image_sum += 1.0 / rl - gamma_eval(rl / d.a, d.p) / d.a;
                    }
                }
    }
    return MSM_AMD_OK;
}

// msmSimple's stencil cache files (msm/msmSimple.c:1018-1031 and :1196-1209): one value per line, %25.16e, node
// offsets mx (slowest), my, mz (fastest) over -(M-1)/2 .. M/2.  Our level-0 stencil covers the same offsets when it
// spans the period and is zero outside its support otherwise; the Fourier stencil is stored on [0, M)^3.
int write_msmsimple_cache(const HostPlan &hp, const char *dir) {
    if (hp.d.L != 1 || hp.literal || hp.d.open_boundary) {
        set_error("msmSimple's cache files describe a periodic two-level plan (L = 1, not literal)");
        return MSM_AMD_EINVAL;
    }
    const int *M = hp.d.M;
    int lo[3], hi[3];
    for (int d = 0; d < 3; d++) { lo[d] = -((M[d] - 1) / 2); hi[d] = M[d] / 2; }
    const std::string base = (dir && *dir) ? std::string(dir) + "/" : std::string();
    char name[160];
    snprintf(name, sizeof(name), "stencil_l%02d_Mx%02d_My%02d_Mz%02d_nu%02d_mu%02d.dat", 0, M[0], M[1], M[2], hp.d.p, hp.d.mu);
    FILE *fp = fopen((base + name).c_str(), "w");
    if (!fp) { set_error("cannot write %s", (base + name).c_str()); return MSM_AMD_EINVAL; }
    const LevelStencil &S = hp.K[0];
    for (int mx = lo[0]; mx <= hi[0]; mx++)
        for (int my = lo[1]; my <= hi[1]; my++)
            for (int mz = lo[2]; mz <= hi[2]; mz++) {
                const int ix = mx - S.first[0], iy = my - S.first[1], iz = mz - S.first[2];
                const bool in = ix >= 0 && ix < S.size[0] && iy >= 0 && iy < S.size[1] && iz >= 0 && iz < S.size[2];
                fprintf(fp, "%25.16e\n", in ? S.v[((size_t)iz * S.size[1] + iy) * S.size[0] + ix] : 0.0);
            }
    fclose(fp);
    snprintf(name, sizeof(name), "stencil_fourier_Mx%02d_My%02d_Mz%02d_nu%02d.dat", M[0], M[1], M[2], hp.d.p);
    fp = fopen((base + name).c_str(), "w");
    if (!fp) { set_error("cannot write %s", (base + name).c_str()); return MSM_AMD_EINVAL; }
    auto wrap = [](int m, int n) { return ((m % n) + n) % n; };
    for (int mx = lo[0]; mx <= hi[0]; mx++)
        for (int my = lo[1]; my <= hi[1]; my++)
            for (int mz = lo[2]; mz <= hi[2]; mz++)
                fprintf(fp, "%25.16e\n", hp.KL.v[((size_t)wrap(mz, M[2]) * M[1] + wrap(my, M[1])) * M[0] + wrap(mx, M[0])]);
    fclose(fp);
    return MSM_AMD_OK;
}

} // namespace msm

// ---------------------------------------------------------------------------
// drop-in helpers (C ABI)
// ---------------------------------------------------------------------------
extern "C" {

const char *msm_amd_last_error(void) { return msm::last_error_cstr(); }
const char *msm_amd_version(void) { return "msm_amd 0.1 (gfx950)"; }

struct msm_amd_hostplan {
    msm::HostPlan hp;
};

int msm_amd_hostplan_create(const msm_amd_plan_desc *plan, int literal, msm_amd_hostplan **out) {
    if (!plan || !out) { msm::set_error("hostplan_create: null argument"); return MSM_AMD_EINVAL; }
    msm_amd_hostplan *h = new msm_amd_hostplan();
    h->hp.d = *plan;
    h->hp.literal = literal != 0;
    int rc = msm::build_host_plan(h->hp);
    if (rc) { delete h; return rc; }
    *out = h;
    return MSM_AMD_OK;
}
void msm_amd_hostplan_destroy(msm_amd_hostplan *h) { delete h; }
static const msm::LevelStencil *pick_stencil(msm_amd_hostplan *h, int level) {
    if (!h || level < 0 || level > h->hp.d.L) return nullptr;
    return level == h->hp.d.L ? &h->hp.KL : &h->hp.K[level];
}
int msm_amd_hostplan_stencil_info(msm_amd_hostplan *h, int level, int size[3], int first[3]) {
    const msm::LevelStencil *S = pick_stencil(h, level);
    if (!S) return MSM_AMD_EINVAL;
    for (int k = 0; k < 3; k++) { size[k] = S->size[k]; first[k] = S->first[k]; }
    return MSM_AMD_OK;
}
int msm_amd_hostplan_copy_stencil(msm_amd_hostplan *h, int level, double *host_out) {
    const msm::LevelStencil *S = pick_stencil(h, level);
    if (!S || !host_out) return MSM_AMD_EINVAL;
    memcpy(host_out, S->v.data(), S->v.size() * sizeof(double));
    return MSM_AMD_OK;
}
int msm_amd_hostplan_copy_wprime(msm_amd_hostplan *h, double *host_out, int *len) {
    if (!h || !len) return MSM_AMD_EINVAL;
    *len = (int)h->hp.wprime.size();
    if (host_out) memcpy(host_out, h->hp.wprime.data(), h->hp.wprime.size() * sizeof(double));
    return MSM_AMD_OK;
}
int msm_amd_hostplan_scalars(msm_amd_hostplan *h, double out[8]) {
    if (!h || !out) return MSM_AMD_EINVAL;
    out[0] = h->hp.csr; out[1] = h->hp.self_image_sum; out[2] = h->hp.gamma0; out[3] = h->hp.chisum;
    for (int k = 0; k < 4; k++) out[4 + k] = h->hp.J[k];
    return MSM_AMD_OK;
}

int msm_amd_hostplan_write_msmsimple_cache(msm_amd_hostplan *h, const char *dir) {
    if (!h) return MSM_AMD_EINVAL;
    return msm::write_msmsimple_cache(h->hp, dir);
}

double mylog2(double x) { return std::log(x) / std::log(2.0); }

int minOf3Integers(int a, int b, int c) {
    int m = a <= b ? a : b;
    return m <= c ? m : c;
}

// Newton iteration for erfc(x aL)/aL = eps hL from x = eps (msm/msmLibrary.c:31-44)
double choosebetaNew(double eps, double aL, double hL) {
    double x = eps;
    int it = 1;
    while (it < 200) {
        const double f = std::erfc(x * aL) / aL - eps * hL;
        const double df = -2 * std::exp(-x * x * aL * aL) / std::sqrt(PI);
        if (!(std::fabs(f / df) / std::fabs(x) > eps)) break;
        x = x - f / df;
        it++;
    }
    if (it >= 200) fprintf(stderr, "beta didn't convergenge\n");
    return x;
}

// Newton iteration for erfc(pi x/beta) 2 beta/sqrt(pi) = eps hL from x = 0 (msm/msmLibrary.c:110-124)
double choosekmax(double eps, double beta, double hL) {
    double x = 0.0;
    int it = 1;
    while (it < 200) {
        const double f = std::erfc(PI * x / beta) * 2 * beta / std::sqrt(PI) - eps * hL;
        const double df = -4 * std::exp(-x * x * PI * PI / (beta * beta));
        if (!(std::fabs(f / df) / std::fabs(x) > eps)) break;
        x = x - f / df;
        it++;
    }
    if (it >= 200) fprintf(stderr, "kmax didn't convergenge\n");
    return x;
}

// msm/msmLibrary.c:69-108
int chooseM(double htilde, double Ax) {
    const double lo = (5.0 / 8.0) * Ax / htilde, mid = (5.0 / 6.0) * Ax / htilde, hi = (5.0 / 4.0) * Ax / htilde;
    int M = 0, e = 0;
    for (int i = 0, top = (int)std::ceil(mylog2(hi)); i <= top; i++) {
        const double t = std::pow(2, i);
        if (lo < t && t <= hi) { M = (int)t; e = i; break; }
    }
    if (std::pow(2, e) < mid) {
        M = 0;
        for (int i = 0, top = (int)std::ceil(mylog2(mid)); i <= top; i++) {
            const double t = std::pow(2, i);
            if (lo < t && t <= mid) { M = (int)(3 * std::pow(2, i - 1)); break; }
        }
    }
    if (M == 0) fprintf(stderr, "M is found to be zero\n");
    return M;
}

static int level_formula(int Mx, int My, int Mz, int N) {
    int fx = 0, fy = 0, fz = 0;
    const int M = Mx * My * Mz;
    while ((Mx % 2) == 0) { fx++; Mx /= 2; }
    while ((My % 2) == 0) { fy++; My /= 2; }
    while ((Mz % 2) == 0) { fz++; Mz /= 2; }
    const int allowed = minOf3Integers(fx, fy, fz) + 1;
    const int wanted = (int)std::ceil(1.0 + (1. / 3) * mylog2(M / std::sqrt((double)N)));
    return wanted > allowed ? allowed : wanted;
}

// msm/msmLibrary.c:46-68: the formula's value is discarded and 1 returned
int chooseL(int Mx, int My, int Mz, int N) {
    int L = level_formula(Mx, My, Mz, N);
    if (L != 1) L = 1;
    return L;
}

int msm_amd_plan_auto(int N, double Ax, double Ay, double Az, int p, int mu, double abar, int L, msm_amd_plan_desc *o) {
    if (!o || N < 1 || !(Ax > 0 && Ay > 0 && Az > 0) || !(abar > 0)) {
        msm::set_error("msm_amd_plan_auto: bad arguments");
        return MSM_AMD_EINVAL;
    }
    memset(o, 0, sizeof(*o));
    const double vol = Ax * Ay * Az, ht = std::pow(vol / N, 1.0 / 3.0);
    o->N = N; o->p = p; o->mu = mu;
    o->A[0] = Ax; o->A[1] = Ay; o->A[2] = Az;
    o->M[0] = chooseM(ht, Ax); o->M[1] = chooseM(ht, Ay); o->M[2] = chooseM(ht, Az);
    if (o->M[0] <= 0 || o->M[1] <= 0 || o->M[2] <= 0) { msm::set_error("chooseM failed"); return MSM_AMD_EINVAL; }
    if (L <= 0) {
        L = level_formula(o->M[0], o->M[1], o->M[2], N);
        if (L < 1) L = 1;
        while (L > 1 && minOf3Integers(o->M[0], o->M[1], o->M[2]) >> (L - 1) < 8) L--;
    }
    o->L = L;
    for (int k = 0; k < 3; k++)
        if (o->M[k] % (1 << (L - 1))) { msm::set_error("M=%d not divisible by 2^(L-1)", o->M[k]); return MSM_AMD_EINVAL; }
    const double hx = Ax / o->M[0];
    o->a = abar * ht;
    const double aL = std::pow(2, L) * o->a, hL = std::pow(2, L - 1) * hx;
    o->beta = choosebetaNew(1e-12, aL, hL);
    o->kmax = (int)std::ceil(std::fabs(choosekmax(1e-12, o->beta, hL)));
    o->zplane0 = 0;
    o->nzplanes = o->M[2];
    o->acc_convention = MSM_AMD_ACC_GRADIENT;
    return MSM_AMD_OK;
}

// room an open-boundary cell needs beyond a particle box of extent Ek with `levels` levels (hc = coarsest spacing):
//   coarsest level: grid charges and needed potentials reach (p/2) hc beyond the particles, plus a node of slack;
//   level L-2: they reach (p/2) h_{L-2} and its stencil 2^(L-1) a, which must not meet across the wrap;
//   the pair sum wants at least three cutoffs per axis.
static double open_room(int p, double h, double a, int levels, double Ek) {
    const double hc = std::ldexp(h, levels - 1);
    double need = std::max((p + 2) * hc, a + 2 * h); // (no pair may come within the cutoff through the wrap either)
    if (levels >= 2) need = std::max(need, (p / 2) * hc + std::ldexp(a, levels - 1) + 2 * h);
    return std::max(need, 3.05 * a - Ek);
}

int msm_amd_plan_open_holds(const msm_amd_plan_desc *d, const double lo[3], const double hi[3]) {
    if (!d || !lo || !hi || !d->open_boundary) return 0;
    for (int k = 0; k < 3; k++) {
        const double h = d->A[k] / d->M[k], E = hi[k] - lo[k];
        if (!(E >= 0)) return 0;
        // inside the cell, clear of its faces by the reach of a spline support and a node (slabs are cut without wrap-around)
        const double margin = (d->p / 2 + 1) * h;
        if (lo[k] < d->origin[k] + margin || hi[k] > d->origin[k] + d->A[k] - margin) return 0;
        if (d->A[k] - E < open_room(d->p, h, d->a, d->L, E)) return 0;
    }
    return 1;
}

int msm_amd_plan_open(int N, const double lo[3], const double hi[3], int p, int mu, double h, double a, int L,
                      msm_amd_plan_desc *o) {
    if (!o || !lo || !hi || N < 1 || !(h > 0) || !(a > 0) || (p != 4 && p != 6)) {
        msm::set_error("msm_amd_plan_open: bad arguments");
        return MSM_AMD_EINVAL;
    }
    double E[3], Emax = 0.0;
    for (int k = 0; k < 3; k++) {
        E[k] = hi[k] - lo[k];
        if (!(E[k] >= 0)) { msm::set_error("msm_amd_plan_open: hi < lo"); return MSM_AMD_EINVAL; }
        Emax = std::max(Emax, E[k]);
    }
    auto room = [&](int levels, double Ek) { return open_room(p, h, a, levels, Ek); };
    if (L <= 0) { // deepest hierarchy whose coarsest grid still has 8 nodes on the longest axis, at most 32 on it
        L = 1;
        while (L < LMAX && (Emax + room(L, Emax)) / std::ldexp(h, L - 1) > 32.0) L++;
        while (L > 1 && (Emax + room(L, Emax)) / std::ldexp(h, L - 1) < 8.0) L--;
    }
    if (L > LMAX) { msm::set_error("msm_amd_plan_open: L=%d out of range", L); return MSM_AMD_EINVAL; }
    memset(o, 0, sizeof(*o));
    o->N = N; o->p = p; o->mu = mu; o->L = L; o->kmax = 0;
    o->a = a; o->beta = 1.0;
    o->open_boundary = 1;
    o->acc_convention = MSM_AMD_ACC_GRADIENT;
    const int mult = 1 << (L - 1);
    for (int k = 0; k < 3; k++) {
        long M = (long)std::ceil((E[k] + room(L, E[k])) / h);
        M = (M + mult - 1) / mult * mult;
        if (M > 4096) { msm::set_error("msm_amd_plan_open: grid of %ld nodes per axis", M); return MSM_AMD_ERANGE; }
        o->M[k] = (int)M;
        o->A[k] = M * h;
        o->extent[k] = E[k];
        o->origin[k] = lo[k] - 0.5 * (o->A[k] - E[k]); // particles in the middle of the cell
    }
    o->zplane0 = 0;
    o->nzplanes = o->M[2];
    return MSM_AMD_OK;
}

// input format "Lx Ly Lz" / "N" / N x "x y z q" (msm/msmLibrary.c:7-29)
int data_read(char *filename, double **q, double **r, int *N, double *Lx, double *Ly, double *Lz) {
    FILE *fp = fopen(filename, "r");
    if (!fp) { msm::set_error("data_read: cannot open %s", filename); return 1; }
    int n = 0;
    if (fscanf(fp, "%lf %lf %lf\n", Lx, Ly, Lz) != 3 || fscanf(fp, "%d\n", &n) != 1 || n < 0) {
        fclose(fp);
        msm::set_error("data_read: bad header in %s", filename);
        return 1;
    }
    *N = n;
    *q = (double *)calloc(n, sizeof(double));
    *r = (double *)calloc((size_t)n * 3, sizeof(double));
    for (int i = 0; i < n; i++)
        if (fscanf(fp, "%lf %lf %lf %lf", *r + 3 * i, *r + 3 * i + 1, *r + 3 * i + 2, *q + i) != 4) break;
    fclose(fp);
    return 0;
}

// One pair, all images inside the cutoff (msm/msmLibrary.c:125-182): returns sum g0(r) and adds
// +-0.5 q_i (1/r^2 + tau'(r/a)/a^2) (A p - r_ij)/r to fshort[i] / fshort[j].
double short_range_periodic(int i, int j, double *r, double *q, int s, double a, double *fshort, double Ax, double Ay,
                            double Az) {
    (void)s;
    const double dx = r[3 * i] - r[3 * j], dy = r[3 * i + 1] - r[3 * j + 1], dz = r[3 * i + 2] - r[3 * j + 2];
    const int x0 = (int)((dx - a) / Ax), x1 = (int)((dx + a) / Ax), y0 = (int)((dy - a) / Ay), y1 = (int)((dy + a) / Ay),
              z0 = (int)((dz - a) / Az), z1 = (int)((dz + a) / Az);
    double u = 0.0;
    for (int px = x0; px <= x1; px++)
        for (int py = y0; py <= y1; py++)
            for (int pz = z0; pz <= z1; pz++) {
                if (px == 0 && py == 0 && pz == 0 && i == j) continue;
                const double ex = Ax * px - dx, ey = Ay * py - dy, ez = Az * pz - dz, r2 = ex * ex + ey * ey + ez * ez;
                if (r2 < a * a) {
                    const double rl = std::sqrt(r2), t = rl / a, t2 = t * t;
                    const double tau = (35. + t2 * (-35. + (21. - 5. * t2) * t2)) / 16.0;   // msmLibrary.h:18
                    const double taud = (t * (-35. + t2 * (42. - 15. * t2))) / 8.0;          // msmLibrary.h:19
                    const double c = 0.5 * q[i] * (1. / r2 + 1. / (a * a) * taud);
                    u += 1.0 / rl - (1. / a) * tau;
                    fshort[3 * i] += c * ex / rl; fshort[3 * i + 1] += c * ey / rl; fshort[3 * i + 2] += c * ez / rl;
                    fshort[3 * j] -= c * ex / rl; fshort[3 * j + 1] -= c * ey / rl; fshort[3 * j + 2] -= c * ez / rl;
                }
            }
    return u;
}

// doubly linked list of (int, double) records, append-only (msm/msmLibrary.c:183-281)
LinkedList *msm4g_linkedlist_new() {
    LinkedList *l = (LinkedList *)malloc(sizeof(LinkedList));
    l->head = l->tail = NULL;
    return l;
}
static void ll_append(LinkedList *list, int data, double value) {
    LinkedListElement *e = (LinkedListElement *)malloc(sizeof(LinkedListElement));
    e->data = data; e->value = value; e->next = NULL; e->prev = list->tail;
    if (list->tail) list->tail->next = e; else list->head = e;
    list->tail = e;
}
void msm4g_linkedlist_add(LinkedList *list, int data) { ll_append(list, data, 0.0); }
void msm4g_linkedlist_addvalue(LinkedList *list, int data, double value) { ll_append(list, data, value); }
int msm4g_linkedlist_search(LinkedList *list, int data) {
    int idx = 0;
    for (LinkedListElement *e = list->head; e; e = e->next, idx++)
        if (e->data == data) return idx;
    return -1;
}
double msm4g_linkedlist_getvalue(LinkedList *list, int index) {
    int idx = 0;
    for (LinkedListElement *e = list->head; e; e = e->next, idx++)
        if (idx == index) return e->value;
    return -1;
}
void msm4g_linkedlist_destroy(LinkedList *list) {
    LinkedListElement *e = list->head;
    while (e) { LinkedListElement *n = e->next; free(e); e = n; }
    free(list);
}

// Same report, same labels and formats as msm/msmLibrary.c:290-376.
void print_stats(msm_stats s) {
    printf("%30s %d\n", "N", s.N);
    printf("%30s %d %d %d\n", "Ax Ay Az", s.Ax, s.Ay, s.Az);
    printf("%30s %d\n", "L", s.L);
    printf("%30s %d\n", "mu", s.mu);
    printf("%30s %25.16e\n", "h (estimated)", s.h);
    printf("%30s %f %f %f\n", "h (calculated)", s.hx, s.hy, s.hz);
    printf("%30s %25.16e\n", "a", s.a);
    printf("%30s %25.16e\n", "aL", s.aL);
    printf("%30s %25.16e\n", "alpha", s.alpha);
    printf("%30s %25.16e\n", "beta", s.beta);
    printf("%30s %d\n", "kmax", s.kmax);
    printf("%30s %d\n", "p", s.p);
    printf("%30s %d\n", "s", s.s);
    printf("%30s %f\n", "csr", s.csr);
    printf("%30s %f\n", "chisum", s.chisum);
    printf("%30s %d %d %d\n", "chinx chiny chinz", s.chinx, s.chiny, s.chinz);
    printf("%30s %d\n", "chisize", s.chisize);
    printf("%30s ", "chi");
    for (int i = 0; i < s.chisize; i++) printf("%6.2e ", s.chi[i]);
    printf("\n");
    printf("%30s %f %f %f\n", "rminx rminy rminz", s.rminx, s.rminy, s.rminz);
    printf("%30s %f %f %f\n", "rmaxx rmaxy rmaxz", s.rmaxx, s.rmaxy, s.rmaxz);
    printf("%30s %d %d %d\n", "Mx My Mz", s.Mx, s.My, s.My); /* My twice: msmLibrary.c:316 */
    printf("%30s %d %d %d\n", "MLx MLy MLz", s.MLx, s.MLy, s.MLz);
    printf("%30s %d %d %d\n", "ML2x ML2y ML2z", s.ML2x, s.ML2y, s.ML2z);
    for (int l = 0; l <= s.L; l++) printf("%30s (l=%d) %d %d %d\n", "Mlx Mly Mlz", l, s.Mlx[l], s.Mly[l], s.Mlz[l]);
    printf("%30s %d %d %d\n", "clenx cleny clenz", s.clenx, s.cleny, s.clenz);
    const char *cn[3] = {"cvecx", "cvecy", "cvecz"};
    double *cv[3] = {s.cvecx, s.cvecy, s.cvecz};
    int cl[3] = {s.clenx, s.cleny, s.clenz};
    for (int d = 0; d < 3; d++) {
        printf("%30s ", cn[d]);
        for (int i = 0; i < cl[d]; i++) printf("%6.2e ", cv[d][i]);
        printf("\n");
    }
    printf("%30s %d x %d x %d = %d\n", "sLnx x sLny x sLnz = sLsize", s.sLnx, s.sLny, s.sLnz, s.sLsize);
    for (int l = 0; l < s.L; l++)
        printf("%30s (l=%d) %d %d %d\n", "slsizex slsizey slsizez", l, s.slsizex[l], s.slsizey[l], s.slsizez[l]);
    printf("%30s %f\n", "time total", s.cputime_total);
    printf("%30s %f\n", "time ulong", s.cputime_potential_long);
    printf("%30s %f\n", "time stencilL", s.cputime_stencilL);
    printf("%30s %f\n", "time stencill", s.cputime_stencill);
    printf("%30s %f\n", "time top_level", s.cputime_top_level);
    printf("%30s %f\n", "time restriction", s.cputime_restriction);
    printf("%30s %f\n", "time short_range", s.cputime_short_range);
    printf("%30s %f\n", "time self", s.cputime_self);
    printf("%30s %f\n", "time prolongation", s.cputime_prolongation);
    printf("%30s %f\n", "time anterpolation", s.cputime_anterpolation);
    printf("%30s %f\n", "time interpolation", s.cputime_interpolation);
    printf("%30s %f\n", "time init", s.cputime_initialization);
    printf("%30s %f\n", "time grid to grid", s.cputime_grid_to_grid_mapping);
    printf("%30s %f\n", "netcharge", s.netcharge);
    for (int l = 0; l < s.L; l++) printf("%30s (l=%d) %f\n", "netgridcharge", l, s.netgridcharge[l]);
    printf("%-20s : %25.16e\n", "ushort_real(MSM)", s.potential_short_real);
    printf("%-20s : %25.16e\n", "ushort_self(MSM)", s.potential_short_self);
    printf("%-20s : %25.16e\n", "ushort_corr(MSM)", s.potential_short_csr);
    printf("%-20s : %25.16e\n", "ushort_tota(MSM)", s.potential_short_total);
    printf("%-20s : %25.16e\n", "ulong_real(MSM)", s.potential_long_real);
    printf("%-20s : %25.16e\n", "ulong_self(MSM)", s.potential_long_self);
    printf("%-20s : %25.16e\n", "ulong_four(MSM)", s.potential_long_fourier);
    printf("%-20s : %25.16e\n", "ulong_tota(MSM)", s.potential_long_total);
    printf("%-20s : %25.16e\n", "utotal", s.potential_total);
    printf("%-20s : %25.16e\n", "(A) ushort_real", s.potential_short_real);
    printf("%-20s : %25.16e\n", "(B) ushort_self", s.potential_short_self);
    printf("%-20s : %25.16e\n", "(C) ushort_csr", s.potential_short_csr);
    printf("%-20s : %25.16e\n", "(D) ulong_real", s.potential_long_real);
    printf("%-20s : %25.16e\n", "(E) ulong_four", s.potential_long_fourier);
    printf("%-20s : %25.16e\n", "(F) ulong_self", s.potential_long_self);
    printf("%-20s : %25.16e\n", "    utotal", s.potential_total);
}

} // extern "C"
