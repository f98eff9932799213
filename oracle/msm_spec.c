/* CPU ORACLE — TEST INFRASTRUCTURE (see msm_oracle.h).
 *
 * Specification engine: the mathematics of msmSimple.c's run_msm()
 * (msmSimple.c:1943-2041: grid charges, real-space grid kernel, Fourier top
 * level, long-range forces from BOTH) written for any number of grid levels
 * with the kernel splitting of msmPeriodic.c:555-583 (a_l = 2^l a), the
 * two-scale restriction/prolongation of msmPeriodic.c:754-840,1018-1108 and
 * the stencil construction of msmPeriodic.c:413-598 / msmSimple.c:1016-1074,
 * but with ONE consistent index convention (grid node m sits at m*h, array
 * index m mod M) and O(N) algorithms (cell lists, local spline support,
 * compact stencils).  It is what the HIP path is compared with.
 */
#include "msm_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct or_spec_plan {
    or_spec_params prm;
    int Ml[OR_MAXL][3];
    double hl[OR_MAXL][3];
    int khat[3];
    double wp[128];
    int wlen;
    double J[8];
    double *K[OR_MAXL];
    int Ksize[OR_MAXL][3], Kfirst[OR_MAXL][3];
    double *KL;
    double *qg[OR_MAXL], *eg[OR_MAXL], *efour;
    int open;                 /* open-boundary plan (see or_spec_set_open_boundary) */
};

/* 0: sum every periodic image that can contribute; R > 0: images -R..R only, as the reference's
 * hard-wired -5..5 (msmPeriodic.c:536-538) */
static int g_image_range = 0;
void or_spec_set_image_range(int R) { g_image_range = R; }

/* Open boundaries (the non-periodic MSM that msm/msmLibrary.h:216 declares as msm3d and no file of the reference
 * defines; PARITY UNPINNED by any reference output, pinned by the direct sum of msm/nonperiodic.c:12-22 within the
 * method's accuracy).  Same pipeline on the same periodic arrays, with the caller's cell A a padded box around the
 * particles, wide enough that nothing below the coarsest level reaches across the wrap, and:
 *   - no image sums in any stencil, no Fourier level, no csr / self-image terms;
 *   - coarsest level: kernel gamma(r / a_l) / a_l at every distance (= 1 / r beyond a_l), stencil over every node
 *     difference -(M-1)..M-1, applied without wrap-around (node index m is position m h and nothing else).
 * Read when a plan is created. */
static int g_open = 0;
void or_spec_set_open_boundary(int on) { g_open = on; }

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

static int imod(int a, int m) {
    int r = a % m;
    return r < 0 ? r + m : r;
}

/* msmPeriodic.c:147-263 for a chosen number of levels L */
int or_spec_auto_params(int N, double Ax, double Ay, double Az, int p, int mu, double abar, int L, or_spec_params *o) {
    double vol = Ax * Ay * Az, ht = pow(vol / N, 1.0 / 3.0), hx, aL, hL;
    int d;
    memset(o, 0, sizeof(*o));
    o->N = N; o->p = p; o->mu = mu; o->L = L; o->nthreads = 0; o->eps = 1e-12;
    o->A[0] = Ax; o->A[1] = Ay; o->A[2] = Az;
    o->M[0] = or_chooseM(ht, Ax); o->M[1] = or_chooseM(ht, Ay); o->M[2] = or_chooseM(ht, Az);
    for (d = 0; d < 3; d++)
        if (o->M[d] <= 0 || (o->M[d] % (1 << (L - 1))) != 0) return 1;
    hx = Ax / o->M[0];
    o->a = abar * ht;
    aL = pow(2, L) * o->a;
    hL = pow(2, L - 1) * hx;
    o->beta = or_choosebetaNew(o->eps, aL, hL);
    o->kmax = (int)ceil(fabs(or_choosekmax(o->eps, o->beta, hL)));
    return 0;
}

/* kernel of grid level l at distance r (msmPeriodic.c:555-583, msmSimple.c:668) */
static double level_kernel(const or_spec_plan *pl, int l, double r) {
    int p = pl->prm.p, L = pl->prm.L;
    double al = ldexp(pl->prm.a, l), al1 = ldexp(pl->prm.a, l + 1);
    if (l == L - 1 && pl->open) return or_gamma(r / al, p) / al;
    if (l == L - 1) {
        if (r >= al) return erfc(pl->prm.beta * r) / r;
        if (r < DBL_EPSILON) return or_gamma(0.0, p) / al - 2.0 * pl->prm.beta / sqrt(OR_PI);
        return or_gamma(r / al, p) / al - erf(pl->prm.beta * r) / r;
    }
    if (r >= al1) return 0.0;
    return or_gamma(r / al, p) / al - or_gamma(r / al1, p) / al1;
}

static void build_level_stencil(or_spec_plan *pl, int l) {
    const or_spec_params *P = &pl->prm;
    int W = P->mu + P->p / 2, d, ext[3], lo[3], hi[3], R[3];
    int nx, ny, nz, kx, ky, kz, px, py, pz;
    double rcut, *G;
    for (d = 0; d < 3; d++) {
        double half = pl->Ml[l][d] / 2.0;
        if (pl->khat[d] > half) { lo[d] = (int)floor(-half) + 1; hi[d] = (int)floor(half); }
        else { lo[d] = -pl->khat[d] + 1; hi[d] = pl->khat[d] - 1; }
        if (pl->open && l == P->L - 1) { lo[d] = -(pl->Ml[l][d] - 1); hi[d] = pl->Ml[l][d] - 1; }
        pl->Kfirst[l][d] = lo[d];
        pl->Ksize[l][d] = hi[d] - lo[d] + 1;
        ext[d] = pl->Ksize[l][d] + 2 * W;
    }
    rcut = (l == P->L - 1) ? fmax(ldexp(P->a, l), 7.5 / P->beta) : ldexp(P->a, l + 1);
    if (pl->open && l == P->L - 1) rcut = 1e300;
    for (d = 0; d < 3; d++) {
        int jmax = (abs(lo[d]) > abs(hi[d]) ? abs(lo[d]) : abs(hi[d])) + W;
        R[d] = pl->open ? 0 : (int)ceil((jmax * pl->hl[l][d] + rcut) / P->A[d]);
        if (g_image_range > 0 && !pl->open) R[d] = g_image_range;
    }
    G = (double *)malloc(sizeof(double) * (size_t)ext[0] * ext[1] * ext[2]);
#pragma omp parallel for collapse(2) private(nx, px, py, pz)
    for (nz = 0; nz < ext[2]; nz++)
        for (ny = 0; ny < ext[1]; ny++)
            for (nx = 0; nx < ext[0]; nx++) {
                double jx = (lo[0] - W + nx) * pl->hl[l][0], jy = (lo[1] - W + ny) * pl->hl[l][1],
                       jz = (lo[2] - W + nz) * pl->hl[l][2], g = 0.0;
                for (pz = -R[2]; pz <= R[2]; pz++) {
                    double dz = jz - P->A[2] * pz;
                    if (fabs(dz) > rcut) continue;
                    for (py = -R[1]; py <= R[1]; py++) {
                        double dy = jy - P->A[1] * py;
                        if (fabs(dy) > rcut) continue;
                        for (px = -R[0]; px <= R[0]; px++) {
                            double dx = jx - P->A[0] * px, rr;
                            if (fabs(dx) > rcut) continue;
                            rr = sqrt(dx * dx + dy * dy + dz * dz);
                            if (rr > rcut) continue;
                            g += level_kernel(pl, l, rr);
                        }
                    }
                }
                G[((size_t)nz * ext[1] + ny) * ext[0] + nx] = g;
            }
    pl->K[l] = (double *)calloc((size_t)pl->Ksize[l][0] * pl->Ksize[l][1] * pl->Ksize[l][2], sizeof(double));
#pragma omp parallel for collapse(2) private(nx, kx, ky, kz)
    for (nz = 0; nz < pl->Ksize[l][2]; nz++)
        for (ny = 0; ny < pl->Ksize[l][1]; ny++)
            for (nx = 0; nx < pl->Ksize[l][0]; nx++) {
                double acc = 0.0;
                for (kz = -W; kz <= W; kz++)
                    for (ky = -W; ky <= W; ky++) {
                        double wyz = pl->wp[abs(ky)] * pl->wp[abs(kz)];
                        const double *row = G + ((size_t)(nz + kz + W) * ext[1] + (ny + ky + W)) * ext[0] + nx + W;
                        for (kx = -W; kx <= W; kx++) acc += pl->wp[abs(kx)] * wyz * row[kx];
                    }
                pl->K[l][((size_t)nz * pl->Ksize[l][1] + ny) * pl->Ksize[l][0] + nx] = acc;
            }
    free(G);
}

/* msmPeriodic.c:265-391 / msmSimple.c:1181-1246 on the coarsest grid, stored for n in [0,ML)^3 */
static void build_top_stencil(or_spec_plan *pl) {
    const or_spec_params *P = &pl->prm;
    const int *ML = pl->Ml[P->L - 1];
    int kmax = P->kmax, cn = 2 * kmax + 1, k, m, d, kx, ky, kz, nx, ny, nz;
    double phik[3] = {0, 0, 0}, vol = P->A[0] * P->A[1] * P->A[2];
    double *cv[3], *chi;
    if (pl->open) { pl->KL = (double *)calloc((size_t)ML[0] * ML[1] * ML[2], sizeof(double)); return; }
    or_phi_at_knots(P->p, phik);
    for (d = 0; d < 3; d++) {
        cv[d] = (double *)malloc(sizeof(double) * (kmax + 1));
        for (k = 0; k <= kmax; k++) {
            double c = phik[0];
            for (m = 1; m <= P->p / 2 - 1; m++) c += 2.0 * cos(2 * OR_PI * k * m / (double)ML[d]) * phik[m];
            cv[d][k] = 1. / c;
        }
    }
    chi = (double *)calloc((size_t)cn * cn * cn, sizeof(double));
    for (kz = -kmax; kz <= kmax; kz++)
        for (ky = -kmax; ky <= kmax; ky++)
            for (kx = -kmax; kx <= kmax; kx++) {
                double k2 = kx * kx / (P->A[0] * P->A[0]) + ky * ky / (P->A[1] * P->A[1]) + kz * kz / (P->A[2] * P->A[2]);
                double c = cv[0][abs(kx)] * cv[1][abs(ky)] * cv[2][abs(kz)];
                if (kx == 0 && ky == 0 && kz == 0) continue;
                chi[((size_t)(kz + kmax) * cn + (ky + kmax)) * cn + (kx + kmax)] =
                    exp(-OR_PI * OR_PI * k2 / (P->beta * P->beta)) / (OR_PI * k2 * vol) * c * c;
            }
    pl->KL = (double *)calloc((size_t)ML[0] * ML[1] * ML[2], sizeof(double));
#pragma omp parallel for collapse(2) private(nx, kx, ky, kz)
    for (nz = 0; nz < ML[2]; nz++)
        for (ny = 0; ny < ML[1]; ny++)
            for (nx = 0; nx < ML[0]; nx++) {
                double s = 0.0;
                for (kz = -kmax; kz <= kmax; kz++)
                    for (ky = -kmax; ky <= kmax; ky++)
                        for (kx = -kmax; kx <= kmax; kx++)
                            s += chi[((size_t)(kz + kmax) * cn + (ky + kmax)) * cn + (kx + kmax)] *
                                 cos(2 * OR_PI * (kx * nx / (double)ML[0] + ky * ny / (double)ML[1] + kz * nz / (double)ML[2]));
                pl->KL[((size_t)nz * ML[1] + ny) * ML[0] + nx] = s;
            }
    free(chi);
    for (d = 0; d < 3; d++) free(cv[d]);
}

or_spec_plan *or_spec_plan_create(const or_spec_params *prm) {
    or_spec_plan *pl;
    int l, d;
    if (prm->L < 1 || prm->L > OR_MAXL || (prm->p != 4 && prm->p != 6)) return NULL;
    for (d = 0; d < 3; d++)
        if (prm->M[d] < 1 || (prm->M[d] % (1 << (prm->L - 1))) != 0) return NULL;
    pl = (or_spec_plan *)calloc(1, sizeof(*pl));
    pl->prm = *prm;
    pl->open = g_open;
#ifdef _OPENMP
    if (prm->nthreads > 0) omp_set_num_threads(prm->nthreads);
#endif
    for (l = 0; l < prm->L; l++)
        for (d = 0; d < 3; d++) {
            pl->Ml[l][d] = prm->M[d] >> l;
            pl->hl[l][d] = prm->A[d] / pl->Ml[l][d];
        }
    for (d = 0; d < 3; d++) pl->khat[d] = (int)ceil(2.0 * prm->a / pl->hl[0][d]); /* msmPeriodic.c:431-433 */
    pl->wlen = prm->mu + prm->p / 2 + 1;
    if (or_wprime(prm->p, prm->mu, pl->wp) != 0) { free(pl); return NULL; }
    or_jcoef(prm->p, pl->J);
    for (l = 0; l < prm->L; l++) {
        size_t n = (size_t)pl->Ml[l][0] * pl->Ml[l][1] * pl->Ml[l][2];
        build_level_stencil(pl, l);
        pl->qg[l] = (double *)calloc(n, sizeof(double));
        pl->eg[l] = (double *)calloc(n, sizeof(double));
        if (l == prm->L - 1) pl->efour = (double *)calloc(n, sizeof(double));
    }
    build_top_stencil(pl);
    return pl;
}

void or_spec_plan_destroy(or_spec_plan *pl) {
    int l;
    if (!pl) return;
    for (l = 0; l < pl->prm.L; l++) { free(pl->K[l]); free(pl->qg[l]); free(pl->eg[l]); }
    free(pl->KL); free(pl->efour); free(pl);
}

const double *or_spec_stencil(const or_spec_plan *pl, int l, int size[3], int first[3]) {
    int d;
    for (d = 0; d < 3; d++) { size[d] = pl->Ksize[l][d]; first[d] = pl->Kfirst[l][d]; }
    return pl->K[l];
}
const double *or_spec_stencil_top(const or_spec_plan *pl, int size[3]) {
    int d;
    for (d = 0; d < 3; d++) size[d] = pl->Ml[pl->prm.L - 1][d];
    return pl->KL;
}
const double *or_spec_wprime(const or_spec_plan *pl, int *len) { *len = pl->wlen; return pl->wp; }
const double *or_spec_grid_q(const or_spec_plan *pl, int l, int dims[3]) {
    int d;
    for (d = 0; d < 3; d++) dims[d] = pl->Ml[l][d];
    return pl->qg[l];
}
const double *or_spec_grid_e(const or_spec_plan *pl, int l, int dims[3]) {
    int d;
    for (d = 0; d < 3; d++) dims[d] = pl->Ml[l][d];
    return pl->eg[l];
}
const double *or_spec_grid_efour(const or_spec_plan *pl, int dims[3]) {
    int d;
    for (d = 0; d < 3; d++) dims[d] = pl->Ml[pl->prm.L - 1][d];
    return pl->efour;
}

/* one in-cutoff pair: d = r_i - r_j - A p.  msmSimple.c:840-848 */
static inline void pair_term(double dx, double dy, double dz, double r2, double a, int p, double qj, double *u,
                             double *fw, double *fu) {
    double rl = sqrt(r2), rho = rl / a;
    double g0 = 1.0 / rl - or_gamma(rho, p) / a;
    double g0p = -1.0 / r2 - or_gamma_prime(rho, p) / (a * a);
    double s = g0p / rl;
    *u += qj * g0;
    fw[0] += qj * s * dx; fw[1] += qj * s * dy; fw[2] += qj * s * dz;
    fu[0] += s * dx; fu[1] += s * dy; fu[2] += s * dz;
}

void or_spec_pairs(const or_spec_plan *pl, const double *r, const double *q, double *acc_w, double *acc_unw,
                   double *u_real) {
    or_spec_pairs_ex(pl, r, q, acc_w, acc_unw, u_real, NULL);
}

/* same, and phi[i] = sum_j q_j sum_P g0 (the short-range potential at particle i) when phi != NULL */
void or_spec_pairs_ex(const or_spec_plan *pl, const double *r, const double *q, double *acc_w, double *acc_unw,
                      double *u_real, double *phi) {
    const or_spec_params *P = &pl->prm;
    int N = P->N, nc[3], d, i;
    double a = P->a, a2 = a * a, usum = 0.0;
    for (d = 0; d < 3; d++) nc[d] = (int)floor(P->A[d] / a);
    if (nc[0] < 3 || nc[1] < 3 || nc[2] < 3) {
        /* small box: every pair, every image inside the cutoff (msmSimple.c:684-699) */
#pragma omp parallel for reduction(+ : usum) schedule(dynamic, 16)
        for (i = 0; i < N; i++) {
            double fw[3] = {0, 0, 0}, fu[3] = {0, 0, 0}, ui = 0.0;
            int j, px, py, pz;
            for (j = 0; j < N; j++) {
                double ex = r[3 * i] - r[3 * j], ey = r[3 * i + 1] - r[3 * j + 1], ez = r[3 * i + 2] - r[3 * j + 2];
                int xlo = (int)ceil((ex - a) / P->A[0]), xhi = (int)floor((ex + a) / P->A[0]);
                int ylo = (int)ceil((ey - a) / P->A[1]), yhi = (int)floor((ey + a) / P->A[1]);
                int zlo = (int)ceil((ez - a) / P->A[2]), zhi = (int)floor((ez + a) / P->A[2]);
                for (px = xlo; px <= xhi; px++)
                    for (py = ylo; py <= yhi; py++)
                        for (pz = zlo; pz <= zhi; pz++) {
                            double dx = ex - P->A[0] * px, dy = ey - P->A[1] * py, dz = ez - P->A[2] * pz;
                            double r2 = dx * dx + dy * dy + dz * dz;
                            if (i == j && px == 0 && py == 0 && pz == 0) continue;
                            if (i == j) continue; /* self images belong to the (B) term */
                            if (r2 < a2) pair_term(dx, dy, dz, r2, a, P->p, q[j], &ui, fw, fu);
                        }
            }
            usum += 0.5 * q[i] * ui;
            if (phi) phi[i] = ui;
            for (d = 0; d < 3; d++) {
                if (acc_w) acc_w[3 * i + d] = fw[d];
                if (acc_unw) acc_unw[3 * i + d] = fu[d];
            }
        }
        *u_real = usum;
        return;
    }
    {
        int ncell = nc[0] * nc[1] * nc[2], c;
        int *cellof = (int *)malloc(sizeof(int) * N), *start = (int *)calloc(ncell + 1, sizeof(int));
        int *order = (int *)malloc(sizeof(int) * N), *fill = (int *)calloc(ncell, sizeof(int));
        double *xs = (double *)malloc(sizeof(double) * 4 * (size_t)N);
        for (i = 0; i < N; i++) {
            int ci[3];
            for (d = 0; d < 3; d++) {
                double s = r[3 * i + d] / P->A[d];
                s -= floor(s);
                ci[d] = (int)(s * nc[d]);
                if (ci[d] >= nc[d]) ci[d] = nc[d] - 1;
            }
            cellof[i] = (ci[2] * nc[1] + ci[1]) * nc[0] + ci[0];
            start[cellof[i] + 1]++;
        }
        for (c = 0; c < ncell; c++) start[c + 1] += start[c];
        for (i = 0; i < N; i++) {
            int k = start[cellof[i]] + fill[cellof[i]]++;
            order[k] = i;
            for (d = 0; d < 3; d++) {
                double s = r[3 * i + d] / P->A[d];
                s -= floor(s);
                xs[4 * (size_t)k + d] = s * P->A[d];
            }
            xs[4 * (size_t)k + 3] = q[i];
        }
#pragma omp parallel for reduction(+ : usum) schedule(dynamic, 4)
        for (c = 0; c < ncell; c++) {
            int cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
            int ii, ox, oy, oz, jj, dd;
            for (ii = start[c]; ii < start[c + 1]; ii++) {
                double xi = xs[4 * (size_t)ii], yi = xs[4 * (size_t)ii + 1], zi = xs[4 * (size_t)ii + 2];
                double fw[3] = {0, 0, 0}, fu[3] = {0, 0, 0}, ui = 0.0;
                for (oz = -1; oz <= 1; oz++)
                    for (oy = -1; oy <= 1; oy++)
                        for (ox = -1; ox <= 1; ox++) {
                            int nx = cx + ox, ny = cy + oy, nz = cz + oz, cn;
                            double sx = 0, sy = 0, sz = 0;
                            if (nx < 0) { nx += nc[0]; sx = -P->A[0]; } else if (nx >= nc[0]) { nx -= nc[0]; sx = P->A[0]; }
                            if (ny < 0) { ny += nc[1]; sy = -P->A[1]; } else if (ny >= nc[1]) { ny -= nc[1]; sy = P->A[1]; }
                            if (nz < 0) { nz += nc[2]; sz = -P->A[2]; } else if (nz >= nc[2]) { nz -= nc[2]; sz = P->A[2]; }
                            cn = (nz * nc[1] + ny) * nc[0] + nx;
                            for (jj = start[cn]; jj < start[cn + 1]; jj++) {
                                double dx = xi - (xs[4 * (size_t)jj] + sx), dy = yi - (xs[4 * (size_t)jj + 1] + sy),
                                       dz = zi - (xs[4 * (size_t)jj + 2] + sz);
                                double r2 = dx * dx + dy * dy + dz * dz;
                                if (jj == ii) continue;
                                if (r2 < a2) pair_term(dx, dy, dz, r2, a, P->p, xs[4 * (size_t)jj + 3], &ui, fw, fu);
                            }
                        }
                usum += 0.5 * xs[4 * (size_t)ii + 3] * ui;
                if (phi) phi[order[ii]] = ui;
                for (dd = 0; dd < 3; dd++) {
                    if (acc_w) acc_w[3 * order[ii] + dd] = fw[dd];
                    if (acc_unw) acc_unw[3 * order[ii] + dd] = fu[dd];
                }
            }
        }
        free(cellof); free(start); free(order); free(fill); free(xs);
        *u_real = usum;
    }
}

/* msmSimple.c:978-1012 with the local support of the spline made explicit */
void or_spec_anterpolate(const or_spec_plan *pl, const double *r, const double *q, double *grid) {
    const or_spec_params *P = &pl->prm;
    const int *M = pl->Ml[0];
    int p = P->p, i;
    size_t n = (size_t)M[0] * M[1] * M[2];
    memset(grid, 0, sizeof(double) * n);
    for (i = 0; i < P->N; i++) {
        double w[3][8];
        int base[3], d, a_, b_, c_;
        for (d = 0; d < 3; d++) {
            double s = r[3 * i + d] / P->A[d], u, fl;
            s -= floor(s);
            u = s * M[d];
            fl = floor(u);
            base[d] = (int)fl - (p / 2 - 1);
            or_bspline_weights(p, u - fl, w[d], NULL);
        }
        for (c_ = 0; c_ < p; c_++) {
            int kz = imod(base[2] + c_, M[2]);
            for (b_ = 0; b_ < p; b_++) {
                int ky = imod(base[1] + b_, M[1]);
                double wyz = w[1][b_] * w[2][c_] * q[i];
                for (a_ = 0; a_ < p; a_++)
                    grid[((size_t)kz * M[1] + ky) * M[0] + imod(base[0] + a_, M[0])] += w[0][a_] * wyz;
            }
        }
    }
}

/* msmPeriodic.c:754-838: q^{l+1}_m = sum_j J_|jx| J_|jy| J_|jz| q^l_{2m+j} */
void or_spec_restrict(const or_spec_plan *pl, int l, const double *fine, double *coarse) {
    const int *Mf = pl->Ml[l], *Mc = pl->Ml[l + 1];
    int h = pl->prm.p / 2, mx, my, mz, jx, jy, jz;
#pragma omp parallel for collapse(2) private(mx, jx, jy, jz)
    for (mz = 0; mz < Mc[2]; mz++)
        for (my = 0; my < Mc[1]; my++)
            for (mx = 0; mx < Mc[0]; mx++) {
                double s = 0.0;
                for (jz = -h; jz <= h; jz++)
                    for (jy = -h; jy <= h; jy++)
                        for (jx = -h; jx <= h; jx++)
                            s += pl->J[abs(jx)] * pl->J[abs(jy)] * pl->J[abs(jz)] *
                                 fine[((size_t)imod(2 * mz + jz, Mf[2]) * Mf[1] + imod(2 * my + jy, Mf[1])) * Mf[0] +
                                      imod(2 * mx + jx, Mf[0])];
                coarse[((size_t)mz * Mc[1] + my) * Mc[0] + mx] = s;
            }
}

/* msmPeriodic.c:1018-1105: e^l_n += sum_m J_|n-2m| e^{l+1}_m (transpose of the restriction) */
void or_spec_prolong_add(const or_spec_plan *pl, int l, const double *coarse, double *fine) {
    const int *Mf = pl->Ml[l], *Mc = pl->Ml[l + 1];
    int h = pl->prm.p / 2, nx, ny, nz, jx, jy, jz;
#pragma omp parallel for collapse(2) private(nx, jx, jy, jz)
    for (nz = 0; nz < Mf[2]; nz++)
        for (ny = 0; ny < Mf[1]; ny++)
            for (nx = 0; nx < Mf[0]; nx++) {
                double s = 0.0;
                /* n = 2m + j  with |j| <= p/2  and  m taken modulo the coarse period */
                for (jz = -h; jz <= h; jz++) {
                    if ((nz - jz) & 1) continue;
                    for (jy = -h; jy <= h; jy++) {
                        if ((ny - jy) & 1) continue;
                        for (jx = -h; jx <= h; jx++) {
                            if ((nx - jx) & 1) continue;
                            /* every m with 2m + j == n (mod Mf): m = (n-j)/2 + t*Mc, same coarse node */
                            s += pl->J[abs(jx)] * pl->J[abs(jy)] * pl->J[abs(jz)] *
                                 coarse[((size_t)imod((nz - jz) / 2, Mc[2]) * Mc[1] + imod((ny - jy) / 2, Mc[1])) * Mc[0] +
                                        imod((nx - jx) / 2, Mc[0])];
                        }
                    }
                }
                fine[((size_t)nz * Mf[1] + ny) * Mf[0] + nx] += s;
            }
}

/* e_m = sum_n K_n q_{m-n}, periodic in m-n (msmSimple.c:1080-1110) */
static void conv_periodic(const int M[3], const double *K, const int ks[3], const int kf[3], const double *q, double *e) {
    int mz, my;
#pragma omp parallel for collapse(2)
    for (mz = 0; mz < M[2]; mz++)
        for (my = 0; my < M[1]; my++) {
            double *erow = e + ((size_t)mz * M[1] + my) * M[0];
            int sz, sy, sx, mx;
            for (mx = 0; mx < M[0]; mx++) erow[mx] = 0.0;
            for (sz = 0; sz < ks[2]; sz++) {
                int nz = imod(mz - (kf[2] + sz), M[2]);
                for (sy = 0; sy < ks[1]; sy++) {
                    int ny = imod(my - (kf[1] + sy), M[1]);
                    const double *qrow = q + ((size_t)nz * M[1] + ny) * M[0];
                    const double *krow = K + ((size_t)sz * ks[1] + sy) * ks[0];
                    for (sx = 0; sx < ks[0]; sx++) {
                        double kv = krow[sx];
                        int sh = imod(-(kf[0] + sx), M[0]); /* q index = (mx + sh) mod Mx */
                        int cut = M[0] - sh;
                        for (mx = 0; mx < cut; mx++) erow[mx] += kv * qrow[mx + sh];
                        for (mx = cut; mx < M[0]; mx++) erow[mx] += kv * qrow[mx + sh - M[0]];
                    }
                }
            }
        }
}

/* e_m = sum_n K_{m-n} q_n over the nodes of the grid, no wrap-around; K holds every difference -(M-1)..M-1 */
static void conv_open(const int M[3], const double *K, const double *q, double *e) {
    int mz, my;
    const int sx = 2 * M[0] - 1, sy = 2 * M[1] - 1;
#pragma omp parallel for collapse(2)
    for (mz = 0; mz < M[2]; mz++)
        for (my = 0; my < M[1]; my++) {
            int mx, nz, ny, nx;
            for (mx = 0; mx < M[0]; mx++) {
                double s = 0.0;
                for (nz = 0; nz < M[2]; nz++)
                    for (ny = 0; ny < M[1]; ny++) {
                        const double *qrow = q + ((size_t)nz * M[1] + ny) * M[0];
                        const double *krow = K + ((size_t)(mz - nz + M[2] - 1) * sy + (my - ny + M[1] - 1)) * sx + mx + M[0] - 1;
                        for (nx = 0; nx < M[0]; nx++) s += krow[-nx] * qrow[nx];
                    }
                e[((size_t)mz * M[1] + my) * M[0] + mx] = s;
            }
        }
}

void or_spec_convolve(const or_spec_plan *pl, int l, const double *q, double *e) {
    if (pl->open && l == pl->prm.L - 1) { conv_open(pl->Ml[l], pl->K[l], q, e); return; }
    conv_periodic(pl->Ml[l], pl->K[l], pl->Ksize[l], pl->Kfirst[l], q, e);
}
void or_spec_convolve_top(const or_spec_plan *pl, const double *q, double *e) {
    int zero[3] = {0, 0, 0};
    conv_periodic(pl->Ml[pl->prm.L - 1], pl->KL, pl->Ml[pl->prm.L - 1], zero, q, e);
}

/* msmSimple.c:1317-1380: acc_i = sum_m e_m grad phi_m(r_i) */
void or_spec_interpolate(const or_spec_plan *pl, const double *r, const double *e, double *along) {
    const or_spec_params *P = &pl->prm;
    const int *M = pl->Ml[0];
    int p = P->p, i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < P->N; i++) {
        double w[3][8], dw[3][8], fx = 0, fy = 0, fz = 0;
        int base[3], d, a_, b_, c_;
        for (d = 0; d < 3; d++) {
            double s = r[3 * i + d] / P->A[d], u, fl;
            s -= floor(s);
            u = s * M[d];
            fl = floor(u);
            base[d] = (int)fl - (p / 2 - 1);
            or_bspline_weights(p, u - fl, w[d], dw[d]);
        }
        for (c_ = 0; c_ < p; c_++) {
            int kz = imod(base[2] + c_, M[2]);
            for (b_ = 0; b_ < p; b_++) {
                int ky = imod(base[1] + b_, M[1]);
                for (a_ = 0; a_ < p; a_++) {
                    double ev = e[((size_t)kz * M[1] + ky) * M[0] + imod(base[0] + a_, M[0])];
                    fx += dw[0][a_] * w[1][b_] * w[2][c_] * ev;
                    fy += w[0][a_] * dw[1][b_] * w[2][c_] * ev;
                    fz += w[0][a_] * w[1][b_] * dw[2][c_] * ev;
                }
            }
        }
        along[3 * i] = fx / pl->hl[0][0];
        along[3 * i + 1] = fy / pl->hl[0][1];
        along[3 * i + 2] = fz / pl->hl[0][2];
    }
}

int or_spec_eval(or_spec_plan *pl, const double *r, const double *q, double *acc, double *fshort_unw, double *along,
                 double *en, double *sec) {
    const or_spec_params *P = &pl->prm;
    int N = P->N, L = P->L, l, i, px, py, pz, R[3], d;
    double t0, t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tall = now_s();
    double *aw = (double *)calloc(3 * (size_t)N, sizeof(double));
    double *al = along ? along : (double *)calloc(3 * (size_t)N, sizeof(double));
    double qs = 0.0, qs2 = 0.0, psum = 0.0, a = P->a, ureal = 0.0, ufour = 0.0, utot_grid = 0.0;
    size_t n0 = (size_t)pl->Ml[0][0] * pl->Ml[0][1] * pl->Ml[0][2];
    size_t nL = (size_t)pl->Ml[L - 1][0] * pl->Ml[L - 1][1] * pl->Ml[L - 1][2];

    t0 = now_s(); or_spec_pairs(pl, r, q, aw, fshort_unw, &ureal); t[0] = now_s() - t0;
    t0 = now_s(); or_spec_anterpolate(pl, r, q, pl->qg[0]); t[1] = now_s() - t0;
    t0 = now_s(); for (l = 0; l + 1 < L; l++) or_spec_restrict(pl, l, pl->qg[l], pl->qg[l + 1]); t[2] = now_s() - t0;
    t0 = now_s(); for (l = 0; l < L; l++) or_spec_convolve(pl, l, pl->qg[l], pl->eg[l]); t[3] = now_s() - t0;
    t0 = now_s();
    or_spec_convolve_top(pl, pl->qg[L - 1], pl->efour);
    for (i = 0; i < (int)nL; i++) { ufour += 0.5 * pl->qg[L - 1][i] * pl->efour[i]; pl->eg[L - 1][i] += pl->efour[i]; }
    t[4] = now_s() - t0;
    t0 = now_s(); for (l = L - 2; l >= 0; l--) or_spec_prolong_add(pl, l, pl->eg[l + 1], pl->eg[l]); t[5] = now_s() - t0;
    t0 = now_s(); or_spec_interpolate(pl, r, pl->eg[0], al); t[6] = now_s() - t0;

    for (i = 0; i < (int)n0; i++) utot_grid += 0.5 * pl->qg[0][i] * pl->eg[0][i];
    for (i = 0; i < N; i++) { qs += q[i]; qs2 += q[i] * q[i]; }
    for (d = 0; d < 3; d++) R[d] = (int)floor(a / P->A[d]);
    for (px = -R[0]; px <= R[0]; px++)
        for (py = -R[1]; py <= R[1]; py++)
            for (pz = -R[2]; pz <= R[2]; pz++) {
                double dx = P->A[0] * px, dy = P->A[1] * py, dz = P->A[2] * pz, r2 = dx * dx + dy * dy + dz * dz;
                if (px == 0 && py == 0 && pz == 0) continue;
                if (r2 < a * a) { double rl = sqrt(r2); psum += 1.0 / rl - or_gamma(rl / a, P->p) / a; }
            }
    en[OR_U_SHORT_REAL] = ureal;
    en[OR_U_SHORT_SELF] = 0.5 * qs2 * psum;
    en[OR_U_SHORT_CSR] = pl->open ? 0.0 : -0.5 * qs * qs * OR_PI / (P->beta * P->beta * P->A[0] * P->A[1] * P->A[2]);
    en[OR_U_LONG_FOUR] = ufour;
    en[OR_U_LONG_REAL] = utot_grid - ufour;
    en[OR_U_LONG_SELF] = -0.5 * qs2 * or_gamma(0.0, P->p) / a;
    en[OR_U_TOTAL] = en[0] + en[1] + en[2] + en[3] + en[4] + en[5];
    if (acc) for (i = 0; i < 3 * N; i++) acc[i] = aw[i] + al[i];
    t[7] = now_s() - tall;
    if (sec) memcpy(sec, t, sizeof(t));
    free(aw);
    if (!along) free(al);
    return 0;
}
