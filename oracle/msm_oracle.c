/* CPU ORACLE — TEST INFRASTRUCTURE (see msm_oracle.h).
 * Parameter helpers and kernel pieces, restated from the reference. */
#include "msm_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* msmLibrary.c:3-5 */
double or_mylog2(double x) { return log(x) / log(2.0); }

/* msmLibrary.c:283-288 */
int or_minOf3(int a, int b, int c) {
    int m = (a <= b) ? a : b;
    return (m <= c) ? m : c;
}

/* msmLibrary.c:69-108.  Pick M = 2^i with 5/8 < M*htilde/A <= 5/4; if that
 * power of two is below 5/6 of A/htilde use 3*2^(i-1) from the band (5/8, 5/6]. */
int or_chooseM(double htilde, double A) {
    double ratio = A / htilde;
    double lo = (5.0 / 8.0) * A / htilde;
    double mid = (5.0 / 6.0) * A / htilde;
    double hi = (5.0 / 4.0) * A / htilde;
    int M = 0, kest = 0, i;
    int top = (int)ceil(or_mylog2(hi));
    (void)ratio;
    for (i = 0; i <= top; i++) {
        double t = pow(2, i);
        if (lo < t && t <= hi) {
            M = (int)t;
            kest = i;
            break;
        }
    }
    if (pow(2, kest) < mid) {
        int top2 = (int)ceil(or_mylog2(mid));
        M = 0;
        for (i = 0; i <= top2; i++) {
            double t = pow(2, i);
            if (lo < t && t <= mid) {
                M = (int)(3 * pow(2, i - 1));
                break;
            }
        }
    }
    if (M == 0) fprintf(stderr, "M is found to be zero\n");
    return M;
}

/* msmLibrary.c:46-62: the level-count formula before it is overridden */
int or_chooseL_formula(int Mx, int My, int Mz, int N, int *allowed, int *wanted) {
    int fx = 0, fy = 0, fz = 0;
    int M = Mx * My * Mz;
    int La, Lw;
    while ((Mx % 2) == 0) { fx++; Mx /= 2; }
    while ((My % 2) == 0) { fy++; My /= 2; }
    while ((Mz % 2) == 0) { fz++; Mz /= 2; }
    La = or_minOf3(fx, fy, fz) + 1;
    Lw = (int)ceil(1.0 + (1. / 3) * or_mylog2(M / sqrt(N)));
    if (allowed) *allowed = La;
    if (wanted) *wanted = Lw;
    return (Lw > La) ? La : Lw;
}

/* msmLibrary.c:63-67: whatever the formula says, the reference returns 1 */
int or_chooseL(int Mx, int My, int Mz, int N) {
    int L = or_chooseL_formula(Mx, My, Mz, N, NULL, NULL);
    if (L != 1) L = 1;
    return L;
}

/* msmLibrary.c:31-44: Newton on erfc(x a)/a = e h, start x = e */
double or_choosebetaNew(double e, double a, double h) {
    double x = e;
    int iter = 1;
    const int maxiter = 200;
    for (;;) {
        double step = (erfc(x * a) / a - e * h) / (-2 * exp(-x * x * a * a) / sqrt(OR_PI));
        if (!(fabs(step) / fabs(x) > e && iter < maxiter)) break;
        x = x - step;
        iter++;
    }
    if (iter >= maxiter) fprintf(stderr, "beta didn't convergenge\n");
    return x;
}

/* msmLibrary.c:110-124: Newton on erfc(pi x/beta) 2 beta/sqrt(pi) = e h, start x = 0 */
double or_choosekmax(double e, double beta, double h) {
    double x = 0.0;
    int iter = 1;
    const int maxiter = 200;
    for (;;) {
        double step = (erfc(OR_PI * x / beta) * 2 * beta / sqrt(OR_PI) - e * h) /
                      (-4 * exp(-x * x * OR_PI * OR_PI / (beta * beta)));
        if (!(fabs(step) / fabs(x) > e && iter < maxiter)) break;
        x = x - step;
        iter++;
    }
    if (iter >= maxiter) fprintf(stderr, "kmax didn't convergenge\n");
    return x;
}

/* msmLibrary.c:7-29 (input format: "Lx Ly Lz" / "N" / N x "x y z q") */
int or_data_read(const char *fn, double **q, double **r, int *N, double *Lx, double *Ly, double *Lz) {
    FILE *fp = fopen(fn, "r");
    int n, i;
    if (!fp) return 1;
    if (fscanf(fp, "%lf %lf %lf", Lx, Ly, Lz) != 3 || fscanf(fp, "%d", &n) != 1) {
        fclose(fp);
        return 2;
    }
    *N = n;
    *q = (double *)calloc(n, sizeof(double));
    *r = (double *)calloc(3 * (size_t)n, sizeof(double));
    for (i = 0; i < n; i++)
        if (fscanf(fp, "%lf %lf %lf %lf", *r + 3 * i, *r + 3 * i + 1, *r + 3 * i + 2, *q + i) != 4) break;
    fclose(fp);
    return (i == n) ? 0 : 3;
}

/* msmPeriodic.c:10-19: Cox-de Boor recursion, B_k supported on [0,k) */
double or_bspline(int k, double u) {
    if (k == 1) return (u >= 0 && u < 1) ? 1.0 : 0.0;
    return u / (double)(k - 1) * or_bspline(k - 1, u) + (k - u) / (double)(k - 1) * or_bspline(k - 1, u - 1.0);
}

/* msmSimple.c:2057-2059 */
double or_bspline_prime(int k, double u) { return or_bspline(k - 1, u) - or_bspline(k - 1, u - 1); }

/* The p non-zero centred B-spline values seen by a particle whose fractional
 * grid coordinate is t in [0,1): w[v] = B_p(t + p/2 - 1 - v + 1 ... ) i.e. the
 * weight of node floor(u) - (p/2-1) + v, and dw[v] its derivative w.r.t. u.
 * Built with the same recursion as or_bspline but bottom-up. */
void or_bspline_weights(int p, double t, double *w, double *dw) {
    /* node v sits at distance x_v = t + (p/2 - 1) - v from the particle; weight = B_p(x_v + p/2) */
    int v;
    for (v = 0; v < p; v++) {
        double arg = t + (p / 2 - 1) - v + p / 2;
        w[v] = or_bspline(p, arg);
        if (dw) dw[v] = or_bspline_prime(p, arg);
    }
}

/* msmPeriodic.c:22-34: gamma(rho) = sum_k C(-1/2,k)(rho^2-1)^k written with tgamma and pow */
double or_gamma_lit(double rho, int p) {
    double out = 0.0;
    int k;
    if (rho >= 1.0) return 1.0 / rho;
    {
        double rho2 = rho * rho;
        for (k = 0; k < p; k++) {
            double binom = tgamma(0.5) / (tgamma(k + 1) * tgamma(0.5 - k));
            out += binom * pow(rho2 - 1.0, k);
        }
    }
    return out;
}

/* msmSimple.c:609-621: the same polynomial in Horner form */
double or_gamma(double rho, int p) {
    double x, out = 1.0;
    int k;
    if (rho >= 1.0) return 1.0 / rho;
    x = rho * rho - 1.0;
    for (k = p - 1; k >= 1; k--) out = 1.0 + ((0.5 - k) / k) * x * out;
    return out;
}

/* msmSimple.c:624-638 */
double or_gamma_prime(double rho, int p) {
    double x, out;
    int k;
    if (rho >= 1.0) return -1.0 / (rho * rho);
    x = rho * rho - 1.0;
    out = p - 1;
    for (k = p - 1; k >= 2; k--) out = k - 1 + (0.5 - k) / k * x * out;
    return -rho * out;
}

/* msmLibrary.h:18-19 */
static double tau4(double x) { return (35. + (x * x) * (-35. + (21. - 5. * (x * x)) * (x * x))) / 16.0; }
static double tau4d(double x) { return (x * (-35. + x * x * (42. - 15. * x * x))) / 8.0; }

/* msmLibrary.c:125-182.  Sum over the periodic images of pair (i,j) inside the
 * cutoff of g0(r) = 1/r - tau(r/a)/a, and add 0.5*q_i*(1/r^2 + tau'(r/a)/a^2)*(A p - r_ij)/r
 * to fshort[i], minus the same to fshort[j] (no q_j: msmLibrary.c:168-176). */
double or_short_range_periodic(int i, int j, const double *r, const double *q, int s, double a, double *fshort,
                               double Ax, double Ay, double Az) {
    double dx = r[3 * i] - r[3 * j], dy = r[3 * i + 1] - r[3 * j + 1], dz = r[3 * i + 2] - r[3 * j + 2];
    int pxlo = (int)((dx - a) / Ax), pxhi = (int)((dx + a) / Ax);
    int pylo = (int)((dy - a) / Ay), pyhi = (int)((dy + a) / Ay);
    int pzlo = (int)((dz - a) / Az), pzhi = (int)((dz + a) / Az);
    double a2 = a * a, u = 0.0;
    int px, py, pz;
    (void)s;
    for (px = pxlo; px <= pxhi; px++)
        for (py = pylo; py <= pyhi; py++)
            for (pz = pzlo; pz <= pzhi; pz++) {
                double ex, ey, ez, r2;
                if (px == 0 && py == 0 && pz == 0 && i == j) continue;
                ex = Ax * px - dx;
                ey = Ay * py - dy;
                ez = Az * pz - dz;
                r2 = ex * ex + ey * ey + ez * ez;
                if (r2 < a2) {
                    double rl = sqrt(r2);
                    double g = 1.0 / rl - (1. / a) * tau4(rl / a);
                    double fx = 0.5 * q[i] * (1. / r2 + 1. / (a * a) * tau4d(rl / a)) * ex / rl;
                    double fy = 0.5 * q[i] * (1. / r2 + 1. / (a * a) * tau4d(rl / a)) * ey / rl;
                    double fz = 0.5 * q[i] * (1. / r2 + 1. / (a * a) * tau4d(rl / a)) * ez / rl;
                    u += g;
                    fshort[3 * i] += fx;
                    fshort[3 * i + 1] += fy;
                    fshort[3 * i + 2] += fz;
                    fshort[3 * j] += -fx;
                    fshort[3 * j + 1] += -fy;
                    fshort[3 * j + 2] += -fz;
                }
            }
    return u;
}

/* msmLibrary.h:152-153: two-scale coefficients of the centred B-spline, C(p, p/2+j)/2^(p-1) */
void or_jcoef(int p, double *J) {
    int j, i;
    for (j = 0; j <= p / 2; j++) {
        double c = 1.0; /* binomial C(p, p/2 + j) */
        int k = p / 2 + j;
        for (i = 1; i <= k; i++) c = c * (p - k + i) / i;
        J[j] = c / pow(2.0, p - 1);
    }
}

/* msmPeriodic.c:71-72: centred B-spline at the integer knots, Phi(m) = B_p(m + p/2) */
void or_phi_at_knots(int p, double *phi) {
    if (p == 4) {
        phi[0] = 4.0 / 6.0;
        phi[1] = 1. / 6.;
    } else {
        phi[0] = 66. / 120.;
        phi[1] = 26. / 120.;
        phi[2] = 1. / 120.;
    }
}

static double vnorm(const double *x, int n) {
    double s = 0.0;
    int i;
    for (i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s);
}
static double vdiff(const double *x, const double *y, int n) {
    double s = 0.0;
    int i;
    for (i = 0; i < n; i++) s += (x[i] - y[i]) * (x[i] - y[i]);
    return sqrt(s);
}

/* msmSimple.c:585-606: elimination without pivoting, then back substitution */
static void solve_dense(int n, double *A, double *y) {
    int i, j, k;
    for (j = 0; j < n - 1; j++)
        for (i = j + 1; i < n; i++) {
            double m = -A[i * n + j] / A[j * n + j];
            for (k = 0; k < n; k++) A[i * n + k] += m * A[j * n + k];
            y[i] += m * y[j];
        }
    for (j = n - 1; j >= 0; j--) {
        y[j] = y[j] / A[j * n + j];
        A[j * n + j] = 1.0;
        for (i = j - 1; i >= 0; i--) {
            y[i] = y[i] - A[i * n + j] * y[j];
            A[i * n + j] = 0.0;
        }
    }
}

/* msmSimple.c:334-516: quasi-interpolation weights omega'_0..omega'_{mu+p/2}.
 * Steps: spectral factor g of the band Phi(0),Phi(1),.. (fixed point); g' = g*g;
 * xi' from the tabulated psi' constants by back substitution; a (p-1)x(p-1)
 * Toeplitz+Hankel solve; a linear recurrence for c_n; finally the stencil
 * first-part + sum_m c_|m| (central difference)^p shifted by m. */
int or_wprime(int p, int mu, double *w) {
    /* numeric constants from msmSimple.c:343-350 */
    static const double psi4[2] = {6.4814814814814811e-02, 9.2592592592592587e-03};
    static const double psi6[4] = {-3.0804398148148150e-02, -8.9800347222222226e-03, -6.0590277777777780e-04,
                                   -1.1863425925925925e-05};
    static const double first4[3] = {-3.3333333333333331e-01, 1.6666666666666665e+00, -3.3333333333333331e-01};
    static const double first6[5] = {1.7083333333333334e-01, -1.1833333333333333e+00, 3.0249999999999999e+00,
                                     -1.1833333333333333e+00, 1.7083333333333334e-01};
    static const double diff4[5] = {1, -4, 6, -4, 1};
    static const double diff6[7] = {1, -6, 15, -20, 15, -6, 1};
    double P[8] = {0}, gold[8] = {0}, g[8] = {0}, g2[16] = {0}, xi[16] = {0}, psi[16] = {0}, Amat[64];
    double c[128], st[256];
    int h = p / 2, i, j, k, n, m, it = 0;
    if ((p != 4 && p != 6) || mu < 0 || 2 * mu + p + 1 > 256 || mu > 100) return 1;
    for (i = 0; i < h; i++) P[i] = or_bspline(p, i + h);
    gold[0] = 1;
    for (;;) {
        double v = 0.0, rel;
        for (k = 1; k < h; k++) v += gold[k] * gold[k];
        v = P[0] - v;
        if (v < 0) return 2;
        g[0] = sqrt(v);
        for (j = 1; j <= h - 1; j++) {
            double s = 0.0;
            for (i = j + 1; i <= h - 1; i++) s += gold[i - j] * gold[i];
            g[j] = (P[j] - s) / g[0];
        }
        rel = vdiff(g, gold, h) / vnorm(g, h);
        if (rel < 1e-16 || it > 1000) break;
        for (k = 0; k < h; k++) gold[k] = g[k];
        it++;
    }
    for (i = 0; i <= p - 2; i++) {
        int lo = (i > h - 1) ? i - (h - 1) : 0;
        int hi = (i < h - 1) ? i : h - 1;
        g2[i] = 0.0;
        for (j = lo; j <= hi; j++) g2[i] += g[j] * g[i - j];
    }
    for (i = 0; i < p - 2; i++) psi[i] = (p == 4) ? psi4[i] : psi6[i];
    for (i = p - 3; i >= 0; i--) {
        double s = 0.0;
        for (j = i + 1; j <= p - 3; j++) s += xi[j] * g2[j - i];
        xi[i] = (psi[i] - s) / g2[0];
    }
    n = p - 1;
    for (i = 0; i < n * n; i++) Amat[i] = 0.0;
    for (i = 0; i < n; i++)
        for (j = 0; j <= i; j++) Amat[i * n + j] = g2[i - j];
    for (j = 1; j <= p - 2; j++)
        for (i = 0; i <= p - 2 - j; i++) Amat[i * n + j] += g2[i + j];
    solve_dense(n, Amat, xi);
    for (i = 0; i < 128; i++) c[i] = 0.0;
    for (i = 0; i < n; i++) c[i] = xi[i];
    for (i = p - 1; i < 128; i++) {
        double s = 0.0;
        for (k = 1; k <= p - 2; k++) s += c[i - k] * g2[k];
        c[i] = -s / g2[0];
    }
    for (i = 0; i < 256; i++) st[i] = 0.0;
    for (i = 1; i <= p - 1; i++) st[i + mu] += (p == 4) ? first4[i - 1] : first6[i - 1];
    for (m = -mu; m <= mu; m++)
        for (i = 1; i <= p + 1; i++) st[i - 1 + m + mu] += ((p == 4) ? diff4[i - 1] : diff6[i - 1]) * c[abs(m)];
    for (i = mu + h; i <= 2 * mu + p; i++) w[i - mu - h] = st[i];
    return 0;
}
