/* CPU ORACLE — TEST INFRASTRUCTURE (see msm_oracle.h).
 * Ewald sums used as the accuracy denominator. */
#include "msm_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ewaldClassical.c:5-109 restated literally: real sum WITHOUT image loop or
 * minimum image (:40-48), reciprocal sum over |k| < cutoffFourier through the
 * structure factor, self term and neutralising-background term. */
double or_ewald_classical(const double *r, const double *q, int N, double Lx, double Ly, double Lz, double alpha,
                          double cutoffDirect, double cutoffFourier) {
    double ereal = 0.0, efour = 0.0, q2 = 0.0, qs = 0.0, V = Lx * Ly * Lz;
    int j, l, kx, ky, kz, kmax = (int)ceil(cutoffFourier);
    for (j = 0; j < N; j++)
        for (l = 0; l < N; l++) {
            double dx, dy, dz, d;
            if (j == l) continue;
            dx = r[3 * j] - r[3 * l]; dy = r[3 * j + 1] - r[3 * l + 1]; dz = r[3 * j + 2] - r[3 * l + 2];
            d = sqrt(dx * dx + dy * dy + dz * dz);
            if (d <= cutoffDirect) ereal += 0.5 * q[j] * q[l] * erfc(alpha * d) / d;
        }
    for (kx = -kmax; kx <= kmax; kx++)
        for (ky = -kmax; ky <= kmax; ky++)
            for (kz = -kmax; kz <= kmax; kz++) {
                double kxL = kx / Lx, kyL = ky / Ly, kzL = kz / Lz, cr = 0.0, ci = 0.0, k2, fac;
                if (kx == 0 && ky == 0 && kz == 0) continue;
                if (kx * kx + ky * ky + kz * kz >= cutoffFourier * cutoffFourier) continue;
                for (j = 0; j < N; j++) {
                    double dp = kxL * r[3 * j] + kyL * r[3 * j + 1] + kzL * r[3 * j + 2];
                    cr += q[j] * cos(2 * OR_PI * dp);
                    ci += q[j] * sin(2 * OR_PI * dp);
                }
                k2 = kxL * kxL + kyL * kyL + kzL * kzL;
                fac = exp(-OR_PI * OR_PI * k2 / (alpha * alpha)) / k2;
                efour += 0.5 * (1.0 / (OR_PI * V)) * fac * (cr * cr + ci * ci);
            }
    for (j = 0; j < N; j++) { q2 += q[j] * q[j]; qs += q[j]; }
    return ereal + efour - 0.5 * q2 * (2 * alpha / sqrt(OR_PI)) - 0.5 * qs * qs * (OR_PI / (alpha * alpha * V));
}

/* Full Ewald sum: every image within rcut in real space, every |k_d| <= kcut in
 * reciprocal space.  Returns sum_{i<nsub} 1/2 q_i phi_i + (self, background
 * shares of those i); with nsub == N that is the total energy.  acc (3*nsub,
 * may be NULL) receives grad phi at the first nsub particles in the
 * convention of msmSimple.c:846-848,1432-1436. */
double or_ewald(int N, const double *r, const double *q, double Ax, double Ay, double Az, double alpha, double rcut,
                int kcut, int nsub, double *acc, int nthreads) {
    double V = Ax * Ay * Az, etot = 0.0, qs = 0.0;
    int Rx = (int)ceil(rcut / Ax), Ry = (int)ceil(rcut / Ay), Rz = (int)ceil(rcut / Az);
    int cn = 2 * kcut + 1, nk = cn * cn * cn, i, k;
    double *Ck = (double *)calloc(nk, sizeof(double)), *Sk = (double *)calloc(nk, sizeof(double));
    double *Fk = (double *)calloc(nk, sizeof(double));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (i = 0; i < N; i++) qs += q[i];
#pragma omp parallel for schedule(dynamic, 8)
    for (k = 0; k < nk; k++) {
        int kx = k % cn - kcut, ky = (k / cn) % cn - kcut, kz = k / (cn * cn) - kcut, j;
        double kxL = kx / Ax, kyL = ky / Ay, kzL = kz / Az, k2 = kxL * kxL + kyL * kyL + kzL * kzL, c = 0.0, s = 0.0;
        if (kx == 0 && ky == 0 && kz == 0) continue;
        for (j = 0; j < N; j++) {
            double dp = 2 * OR_PI * (kxL * r[3 * j] + kyL * r[3 * j + 1] + kzL * r[3 * j + 2]);
            c += q[j] * cos(dp);
            s += q[j] * sin(dp);
        }
        Ck[k] = c; Sk[k] = s;
        Fk[k] = exp(-OR_PI * OR_PI * k2 / (alpha * alpha)) / (OR_PI * k2 * V);
    }
#pragma omp parallel for reduction(+ : etot) schedule(dynamic, 4)
    for (i = 0; i < nsub; i++) {
        double phi = 0.0, g[3] = {0, 0, 0};
        int j, px, py, pz, kk;
        for (j = 0; j < N; j++) {
            double ex = r[3 * i] - r[3 * j], ey = r[3 * i + 1] - r[3 * j + 1], ez = r[3 * i + 2] - r[3 * j + 2];
            /* bring the base difference into the primary cell so that |p| <= R covers the cutoff */
            ex -= Ax * floor(ex / Ax + 0.5); ey -= Ay * floor(ey / Ay + 0.5); ez -= Az * floor(ez / Az + 0.5);
            for (px = -Rx; px <= Rx; px++)
                for (py = -Ry; py <= Ry; py++)
                    for (pz = -Rz; pz <= Rz; pz++) {
                        double dx = ex - Ax * px, dy = ey - Ay * py, dz = ez - Az * pz, r2 = dx * dx + dy * dy + dz * dz, rl, e, de;
                        if (r2 == 0.0 || r2 > rcut * rcut) continue;
                        if (i == j && px == 0 && py == 0 && pz == 0) continue;
                        rl = sqrt(r2);
                        e = erfc(alpha * rl) / rl;
                        de = -e / rl - 2 * alpha / sqrt(OR_PI) * exp(-alpha * alpha * r2) / rl;
                        phi += q[j] * e;
                        g[0] += q[j] * de * dx / rl; g[1] += q[j] * de * dy / rl; g[2] += q[j] * de * dz / rl;
                    }
        }
        for (kk = 0; kk < nk; kk++) {
            int kx = kk % cn - kcut, ky = (kk / cn) % cn - kcut, kz = kk / (cn * cn) - kcut;
            double kxL = kx / Ax, kyL = ky / Ay, kzL = kz / Az, dp, c, s, t;
            if (Fk[kk] == 0.0) continue;
            dp = 2 * OR_PI * (kxL * r[3 * i] + kyL * r[3 * i + 1] + kzL * r[3 * i + 2]);
            c = cos(dp); s = sin(dp);
            phi += Fk[kk] * (c * Ck[kk] + s * Sk[kk]);
            t = -2 * OR_PI * Fk[kk] * (s * Ck[kk] - c * Sk[kk]);
            g[0] += t * kxL; g[1] += t * kyL; g[2] += t * kzL;
        }
        phi -= q[i] * 2 * alpha / sqrt(OR_PI);
        phi -= qs * OR_PI / (alpha * alpha * V);
        etot += 0.5 * q[i] * phi;
        if (acc) { acc[3 * i] = g[0]; acc[3 * i + 1] = g[1]; acc[3 * i + 2] = g[2]; }
    }
    free(Ck); free(Sk); free(Fk);
    return etot;
}

/* Open-boundary direct sum, the loop of msm/nonperiodic.c:12-22 (which forces q = 1 first and has no gradient):
 * returns 1/2 sum_{i != j} q_i q_j / r_ij; acc (may be NULL) receives the gradient of the potential at the first
 * nsub particles, acc_i = -sum_j q_j (r_i - r_j) / r_ij^3. */
double or_direct(int N, const double *r, const double *q, int nsub, double *acc, int nthreads) {
    double u = 0.0;
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for reduction(+ : u) schedule(static)
    for (int i = 0; i < N; i++) {
        double phi = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
        for (int j = 0; j < N; j++) {
            if (i == j) continue;
            const double dx = r[3 * i] - r[3 * j], dy = r[3 * i + 1] - r[3 * j + 1], dz = r[3 * i + 2] - r[3 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz, d = sqrt(d2);
            phi += q[j] / d;
            const double s = q[j] / (d2 * d);
            gx -= s * dx; gy -= s * dy; gz -= s * dz;
        }
        u += 0.5 * q[i] * phi;
        if (acc && i < nsub) { acc[3 * i] = gx; acc[3 * i + 1] = gy; acc[3 * i + 2] = gz; }
    }
    return u;
}
