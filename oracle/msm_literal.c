/* CPU ORACLE — TEST INFRASTRUCTURE (see msm_oracle.h).
 *
 * Literal mirror of msmperiodic() (msmPeriodic.c:52-1421): the same stages in
 * the same order with the same index conventions, including the ones that are
 * visibly unintended (the +0.5 coordinate shift :75, `sx = mx - nz` :893/:976,
 * stencil offsets wrapped by the stencil size :896-901, the minus sign of the
 * Fourier energy :1002, the Fourier potential dropped before interpolation
 * :1006-1011, the interpolation index offset :1136-1151, the q_i-only
 * short-range "force" msmLibrary.c:168-176).  It is the thing the golden
 * fixtures of tests/golden/msmPeriodic_*.json are compared with.
 *
 * One reorganisation (value-preserving): the image sum of the level-0 kernel
 * depends on (n+k) only, so it is tabulated once per distinct n+k instead of
 * once per (n,k) pair; the inner arithmetic and summation orders are kept.
 */
#include "msm_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double lit_g1(double r, double a, int p) { return (1.0 / a) * or_gamma_lit(r / a, p); } /* msmPeriodic.c:36 */
static double lit_gLR(double r, double beta) {                                                 /* msmPeriodic.c:42 */
    if (r < DBL_EPSILON) return 2.0 * beta / sqrt(3.141592653589793238462643);
    return erf(beta * r) / r;
}

static int wrap_range(int s, int lo, int hi, int period) { /* the do/while idiom of msmPeriodic.c:896-901 */
    if (s < lo) do { s += period; } while (s < lo);
    else if (s > hi) do { s -= period; } while (s > hi);
    return s;
}

void or_literal_free(or_literal_result *o) {
    if (!o) return;
    free(o->sl); free(o->sL); free(o->qgrd); free(o->edir); free(o->etop);
    free(o->cvecx); free(o->cvecy); free(o->cvecz); free(o->chi);
    free(o->fshort); free(o->flong); free(o->facc); free(o->rshift);
    memset(o, 0, sizeof(*o));
}

int or_literal_run(int N, const double *r_in, const double *q, int p, double abar, int mu, double Ax, double Ay,
                   double Az, or_literal_result *o) {
    const int p2 = p / 2;
    double *r, phik[3] = {0, 0, 0}, omega[16];
    double htilde, volume, hx, hy, hz, a, aL, beta, csr, hL;
    int Mx, My, Mz, L, MLx, MLy, MLz, kmax, i, j, k, m;
    int kx, ky, kz, nx, ny, nz, px, py, pz, mx, my, mz;
    memset(o, 0, sizeof(*o));
    o->N = N; o->p = p; o->mu = mu; o->abar = abar; o->Ax = Ax; o->Ay = Ay; o->Az = Az;
    if (p != 4) return 1; /* msmPeriodic.c:100-104 */
    if (mu + p2 > 13) return 2; /* OMEGA has MAX_MU=14 entries (msmLibrary.h:14,156) */

    r = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    for (i = 0; i < 3 * N; i++) r[i] = r_in[i] + 0.5; /* :75-76 */
    o->rshift = r;

    volume = Ax * Ay * Az;
    htilde = pow(volume / N, 1.0 / 3.0); /* :149 */
    Mx = or_chooseM(htilde, Ax); My = or_chooseM(htilde, Ay); Mz = or_chooseM(htilde, Az);
    hx = Ax / Mx; hy = Ay / My; hz = Az / Mz;
    L = or_chooseL(Mx, My, Mz, N); /* == 1 */
    MLx = (int)(Mx / pow(2, L - 1)); MLy = (int)(My / pow(2, L - 1)); MLz = (int)(Mz / pow(2, L - 1)); /* :184-196 */
    a = abar * htilde; /* :230 */
    aL = pow(2, L) * a;
    hL = pow(2, L - 1) * hx;
    beta = or_choosebetaNew(1e-12, aL, hL); /* :249-252 */
    kmax = (int)ceil(fabs(or_choosekmax(1e-12, beta, hL)));
    csr = OR_PI / (beta * beta * volume);
    o->L = L; o->Mx = Mx; o->My = My; o->Mz = Mz; o->htilde = htilde; o->hx = hx; o->hy = hy; o->hz = hz;
    o->a = a; o->aL = aL; o->beta = beta; o->kmax = kmax; o->csr = csr;

    /* c(k), :265-283 */
    or_phi_at_knots(p, phik);
    o->cvecx = (double *)calloc(kmax + 1, sizeof(double));
    o->cvecy = (double *)calloc(kmax + 1, sizeof(double));
    o->cvecz = (double *)calloc(kmax + 1, sizeof(double));
    for (k = 0; k < kmax + 1; k++) {
        double cx = phik[0], cy = phik[0], cz = phik[0];
        for (m = 1; m <= p2 - 1; m++) {
            cx += 2.0 * cos(2 * OR_PI * k * m / (double)MLx) * phik[m];
            cy += 2.0 * cos(2 * OR_PI * k * m / (double)MLy) * phik[m];
            cz += 2.0 * cos(2 * OR_PI * k * m / (double)MLz) * phik[m];
        }
        o->cvecx[k] = 1. / cx; o->cvecy[k] = 1. / cy; o->cvecz[k] = 1. / cz;
    }

    /* chi(k), :301-333 */
    {
        int cn = 2 * kmax + 1;
        double chisum = 0.0;
        o->chisize = cn * cn * cn;
        o->chi = (double *)calloc(o->chisize, sizeof(double));
        for (kz = 0; kz < cn; kz++) {
            int kzz = kz - kmax;
            double k2z = kzz * kzz / (Az * Az);
            for (ky = 0; ky < cn; ky++) {
                int kyy = ky - kmax;
                double k2y = k2z + kyy * kyy / (Ay * Ay);
                for (kx = 0; kx < cn; kx++) {
                    int kxx = kx - kmax, off = (kz * cn + ky) * cn + kx;
                    if (kxx == 0 && kyy == 0 && kzz == 0) o->chi[off] = 0;
                    else {
                        double k2 = k2y + kxx * kxx / (Ax * Ax);
                        o->chi[off] = exp(-OR_PI * OR_PI * k2 / (beta * beta)) / (OR_PI * k2 * volume);
                    }
                    chisum += o->chi[off];
                }
            }
        }
        o->chisum = chisum;
    }

    /* top-level stencil sL, :344-391 */
    {
        int sx = 2 * (MLx - 1) + 1, sy = 2 * (MLy - 1) + 1, sz = 2 * (MLz - 1) + 1, cn = 2 * kmax + 1;
        o->sLn[0] = sx; o->sLn[1] = sy; o->sLn[2] = sz;
        o->sL = (double *)calloc((size_t)sx * sy * sz, sizeof(double));
        for (nz = -(MLz - 1); nz <= MLz - 1; nz++)
            for (ny = -(MLy - 1); ny <= MLy - 1; ny++)
                for (nx = -(MLx - 1); nx <= MLx - 1; nx++) {
                    double sum = 0.0;
                    for (kz = -kmax; kz < kmax + 1; kz++) {
                        double cz = o->cvecz[abs(kz)], ez = kz * nz / (double)MLz;
                        for (ky = -kmax; ky < kmax + 1; ky++) {
                            double cy = cz * o->cvecy[abs(ky)], ey = ez + ky * ny / (double)MLy;
                            for (kx = -kmax; kx < kmax + 1; kx++) {
                                double cx = cy * o->cvecx[abs(kx)], ex = ey + kx * nx / (double)MLx;
                                sum += o->chi[((kz + kmax) * cn + (ky + kmax)) * cn + (kx + kmax)] * cx * cx *
                                       cos(2 * OR_PI * ex);
                            }
                        }
                    }
                    o->sL[((size_t)(nz + MLz - 1) * sy + (ny + MLy - 1)) * sx + (nx + MLx - 1)] = sum;
                }
    }

    /* level-0 stencil sl[0], :423-598 (l = 0 = L-1 branch: g1 - gLR, images -5..5) */
    {
        double khx = (int)ceil(2.0 * a / hx), khy = (int)ceil(2.0 * a / hy), khz = (int)ceil(2.0 * a / hz);
        int mlx = Mx, mly = My, mlz = Mx; /* Mlx[l+1]; `mlz = Mlx[l+1]` is the reference's own (:437) */
        double hfx = mlx / 2.0, hfy = mly / 2.0, hfz = mlz / 2.0;
        int lo[3], hi[3], sz3[3], W = mu + p2, ext[3];
        double *G, w10[16];
        if (khx > hfx) { lo[0] = (int)floor(-hfx) + 1; hi[0] = (int)floor(hfx); } else { lo[0] = (int)(-khx + 1); hi[0] = (int)(khx - 1); }
        if (khy > hfy) { lo[1] = (int)floor(-hfy) + 1; hi[1] = (int)floor(hfy); } else { lo[1] = (int)(-khy + 1); hi[1] = (int)(khy - 1); }
        if (khz > hfz) { lo[2] = (int)floor(-hfz) + 1; hi[2] = (int)floor(hfz); } else { lo[2] = (int)(-khz + 1); hi[2] = (int)(khz - 1); }
        for (i = 0; i < 3; i++) {
            sz3[i] = hi[i] - lo[i] + 1;
            o->slmin[i] = lo[i]; o->slmax[i] = hi[i]; o->slsize[i] = sz3[i];
            ext[i] = sz3[i] + 2 * W;
        }
        /* the table the reference indexes is omega' for mu = 10 whatever `mu` is (msmLibrary.h:156-170) */
        memset(omega, 0, sizeof(omega));
        or_wprime(p, 10, w10);
        for (i = 0; i <= 10 + p2; i++) omega[i] = w10[i];
        /* G[j] = sum over images of (g1 - gLR)(|j h - A P|), j = n + k */
        G = (double *)malloc(sizeof(double) * (size_t)ext[0] * ext[1] * ext[2]);
        for (nz = 0; nz < ext[2]; nz++)
            for (ny = 0; ny < ext[1]; ny++)
                for (nx = 0; nx < ext[0]; nx++) {
                    int jx = lo[0] - W + nx, jy = lo[1] - W + ny, jz = lo[2] - W + nz;
                    double g = 0.0;
                    for (pz = -5; pz <= 5; pz++)
                        for (py = -5; py <= 5; py++)
                            for (px = -5; px <= 5; px++) {
                                double r2 = pow(jx * hx - Ax * px, 2) + pow(jy * hy - Ay * py, 2) +
                                            pow(jz * hz - Az * pz, 2);
                                double rr = sqrt(r2);
                                g += lit_g1(rr, a, p) - lit_gLR(rr, beta);
                            }
                    G[((size_t)nz * ext[1] + ny) * ext[0] + nx] = g;
                }
        o->sl = (double *)calloc((size_t)sz3[0] * sz3[1] * sz3[2], sizeof(double));
        for (nz = lo[2]; nz <= hi[2]; nz++)
            for (ny = lo[1]; ny <= hi[1]; ny++)
                for (nx = lo[0]; nx <= hi[0]; nx++) {
                    double acc = 0.0;
                    for (kz = -W; kz <= W; kz++)
                        for (ky = -W; ky <= W; ky++)
                            for (kx = -W; kx <= W; kx++) {
                                double g = G[((size_t)(nz + kz - lo[2] + W) * ext[1] + (ny + ky - lo[1] + W)) * ext[0] +
                                             (nx + kx - lo[0] + W)];
                                acc += omega[abs(kx)] * omega[abs(ky)] * omega[abs(kz)] * g;
                            }
                    o->sl[((size_t)(nz - lo[2]) * sz3[1] + (ny - lo[1])) * sz3[0] + (nx - lo[0])] = acc;
                }
        free(G);
    }

    /* anterpolation, :688-717 */
    o->qgrd = (double *)calloc((size_t)Mx * My * Mz, sizeof(double));
    for (i = 0; i < N; i++) {
        double sx = (r[3 * i] - 0.0) / Ax, sy = (r[3 * i + 1] - 0.0) / Ay, sz = (r[3 * i + 2] - 0.0) / Az;
        for (mz = -Mz / 2 + 1; mz <= Mz / 2; mz++)
            for (my = -My / 2 + 1; my <= My / 2; my++)
                for (mx = -Mx / 2 + 1; mx <= Mx / 2; mx++) {
                    size_t off = ((size_t)(mz + Mz / 2 - 1) * My + (my + My / 2 - 1)) * Mx + (mx + Mx / 2 - 1);
                    for (px = -p2; px <= p2; px++) {
                        double fx = or_bspline(p, Mx * (sx - px) - mx + p2);
                        for (py = -p2; py <= p2; py++) {
                            double fy = or_bspline(p, My * (sy - py) - my + p2);
                            for (pz = -p2; pz <= p2; pz++) {
                                double fz = or_bspline(p, Mz * (sz - pz) - mz + p2);
                                o->qgrd[off] += fx * fy * fz * q[i];
                            }
                        }
                    }
                }
    }

    /* grid-to-grid, :846-919 (L = 1 so only l = 0; note sx uses nz) */
    o->edir = (double *)calloc((size_t)Mx * My * Mz, sizeof(double));
    o->etop = (double *)calloc((size_t)Mx * My * Mz, sizeof(double));
    for (mz = 0; mz < Mz; mz++)
        for (my = 0; my < My; my++)
            for (mx = 0; mx < Mx; mx++) {
                double es = 0.0;
                for (nz = 0; nz < Mz; nz++)
                    for (ny = 0; ny < My; ny++)
                        for (nx = 0; nx < Mx; nx++) {
                            int sx = wrap_range(mx - nz, o->slmin[0], o->slmax[0], o->slsize[0]);
                            int sy = wrap_range(my - ny, o->slmin[1], o->slmax[1], o->slsize[1]);
                            int sz = wrap_range(mz - nz, o->slmin[2], o->slmax[2], o->slsize[2]);
                            es += o->sl[((size_t)(sz - o->slmin[2]) * o->slsize[1] + (sy - o->slmin[1])) * o->slsize[0] +
                                        (sx - o->slmin[0])] *
                                  o->qgrd[((size_t)nz * My + ny) * Mx + nx];
                        }
                o->edir[((size_t)mz * My + my) * Mx + mx] = es / pow(2, 0);
            }

    /* top level, :936-1011 */
    {
        double four = 0.0;
        for (mz = 0; mz < MLz; mz++)
            for (my = 0; my < MLy; my++)
                for (mx = 0; mx < MLx; mx++) {
                    size_t mo = ((size_t)mz * MLy + my) * MLx + mx;
                    for (nz = 0; nz < MLz; nz++)
                        for (ny = 0; ny < MLy; ny++)
                            for (nx = 0; nx < MLx; nx++) {
                                int sx = wrap_range(mx - nz, -(MLx - 1), MLx - 1, o->sLn[0]);
                                int sy = wrap_range(my - ny, -(MLy - 1), MLy - 1, o->sLn[1]);
                                int sz = wrap_range(mz - nz, -(MLz - 1), MLz + 1, o->sLn[2]);
                                o->etop[mo] += o->qgrd[((size_t)nz * MLy + ny) * MLx + nx] *
                                               o->sL[((size_t)(sz + MLz - 1) * o->sLn[1] + (sy + MLy - 1)) * o->sLn[0] +
                                                     (sx + MLx - 1)];
                            }
                }
        for (i = 0; i < MLx * MLy * MLz; i++) four += -0.5 * o->etop[i] * o->qgrd[i];
        o->u_long_four = four;
    }

    /* interpolation of edir (the Fourier part was discarded at :1006-1011), :1116-1161 */
    o->flong = (double *)calloc(3 * (size_t)N, sizeof(double));
    for (i = 0; i < N; i++) {
        int i0 = (int)floor((r[3 * i] - 0.0) / hx) - (p2 - 1);
        int j0 = (int)floor((r[3 * i + 1] - 0.0) / hy) - (p2 - 1);
        int k0 = (int)floor((r[3 * i + 2] - 0.0) / hz) - (p2 - 1);
        double rx = r[3 * i] - hx * i0 - 0.0, ry = r[3 * i + 1] - hy * j0 - 0.0, rz = r[3 * i + 2] - hz * k0 - 0.0;
        double F[3][6], D[3][6];
        int v, a_, b_, c_;
        /* piece v of the cubic spline at (t - v), closed forms as tabulated in msmLibrary.h:120-128 */
        for (v = 0; v < p; v++) {
            double t[3] = {rx / hx - v, ry / hy - v, rz / hz - v};
            int d;
            for (d = 0; d < 3; d++) {
                double x = t[d], f, g;
                switch (v) {
                case 0: f = (1. / 6) * (2 - x) * (2 - x) * (2 - x); g = -0.5 * (2 - x) * (2 - x); break;
                case 1: f = (2. / 3) + x * x * (-1 + 0.5 * x); g = x * (-2 + 1.5 * x); break;
                case 2: f = (2. / 3) + x * x * (-1 - 0.5 * x); g = x * (-2 - 1.5 * x); break;
                default: f = (1. / 6) * (2 + x) * (2 + x) * (2 + x); g = 0.5 * (2 + x) * (2 + x); break;
                }
                F[d][v] = f;
                D[d][v] = g / (d == 0 ? hx : (d == 1 ? hy : hz));
            }
        }
        for (c_ = 0; c_ < p; c_++) {
            int kk = wrap_range(k0 + c_ + p2 - 1, 0, Mz - 1, Mz);
            for (b_ = 0; b_ < p; b_++) {
                int jj = wrap_range(j0 + b_ + p2 - 1, 0, My - 1, My);
                for (a_ = 0; a_ < p; a_++) {
                    int ii = wrap_range(i0 + a_ + p2 - 1, 0, Mx - 1, Mx);
                    double e = o->edir[((size_t)kk * My + jj) * Mx + ii];
                    o->flong[3 * i] += D[0][a_] * F[1][b_] * F[2][c_] * e;
                    o->flong[3 * i + 1] += F[0][a_] * D[1][b_] * F[2][c_] * e;
                    o->flong[3 * i + 2] += F[0][a_] * F[1][b_] * D[2][c_] * e;
                }
            }
        }
        o->flong[3 * i] *= q[i]; o->flong[3 * i + 1] *= q[i]; o->flong[3 * i + 2] *= q[i];
    }

    /* short range without cells, :1292-1300, self term :1305-1310, background :1313-1319 */
    o->fshort = (double *)calloc(3 * (size_t)N, sizeof(double));
    {
        double us = 0.0, uself = 0.0, qs = 0.0, qs2 = 0.0, ulr = 0.0, gsum = 0.0;
        for (i = 0; i < N; i++)
            for (j = 0; j < N; j++) {
                if (i == j) continue;
                us += 0.5 * q[i] * q[j] * or_short_range_periodic(i, j, r, q, p - 1, a, o->fshort, Ax, Ay, Az);
            }
        for (i = 0; i < N; i++)
            uself += 0.5 * q[i] * q[i] * or_short_range_periodic(i, i, r, q, p - 1, a, o->fshort, Ax, Ay, Az);
        for (i = 0; i < N; i++) { qs += q[i]; qs2 += q[i] * q[i]; }
        for (i = 0; i < Mx * My * Mz; i++) { ulr += 0.5 * o->edir[i] * o->qgrd[i]; gsum += o->qgrd[i]; }
        o->u_short_real = us;
        o->u_short_self = uself;
        o->u_short_csr = -0.5 * qs * qs * csr;
        o->u_long_real = ulr; /* :1334-1338 */
        o->u_long_self = -0.5 * qs2 * (1.0 / a) * (35. / 16.0); /* :1343, tau(0) */
        o->u_total = (o->u_long_four + o->u_long_real + o->u_long_self) +
                     (o->u_short_real + o->u_short_self + o->u_short_csr); /* :1349,1321,1356 */
        o->netcharge = qs;
        o->netgridcharge = gsum;
    }
    o->facc = (double *)calloc(3 * (size_t)N, sizeof(double));
    for (i = 0; i < 3 * N; i++) o->facc[i] = o->fshort[i] + o->flong[i]; /* :1362-1367 */
    return 0;
}
