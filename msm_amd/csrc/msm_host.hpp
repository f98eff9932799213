// Host-side pieces of the engine: parameter selection, quasi-interpolation
// weights, two-scale coefficients and stencil generation.  Position
// independent; runs once per plan.
#pragma once
#include <vector>

#include "msm_amd.h"

namespace msm {

struct LevelStencil {
    int size[3];          // extent per axis, x fastest
    int first[3];         // offset of element 0 per axis
    std::vector<double> v;
};

struct HostPlan {
    msm_amd_plan_desc d;
    bool literal = false; // reproduce the reference's index conventions (L must be 1, p must be 4)
    int Ml[LMAX][3];
    double hl[LMAX][3];
    int khat[3];
    std::vector<double> wprime;          // omega'_0 .. omega'_{mu+p/2}
    double J[4];                         // two-scale coefficients J_0 .. J_{p/2}
    std::vector<LevelStencil> K;         // one per grid level
    LevelStencil KL;                     // Fourier top level on the coarsest grid, [0,ML)^3
    std::vector<double> cvec[3];         // c(k), k = 0..kmax (msmPeriodic.c:265-283)
    std::vector<double> chi;             // chi(k) on [-kmax,kmax]^3 (msmPeriodic.c:301-333)
    double chisum = 0.0;
    double csr = 0.0;
    double self_image_sum = 0.0;         // sum_{P != 0, |AP| < a} g0(|AP|)
    double gamma0 = 0.0;                 // gamma(0)
};

// gamma(rho) and the pieces the kernels need, as polynomial coefficients in x = rho^2:
//   gamma(rho)      = sum_k cg[k] x^k            (rho < 1)
//   -gamma'(rho)/rho = sum_k cd[k] x^k           (rho < 1)
void gamma_poly(int p, double cg[8], double cd[8]);
void spline_pieces(int p, double piece[6][6], double dpiece[6][6]);
double gamma_eval(double rho, int p);

std::vector<double> compute_wprime(int p, int mu);
void two_scale(int p, double J[4]);

// fills Ml/hl/khat/wprime/J/K/KL/cvec/chi... from hp.d (+ hp.literal)
int build_host_plan(HostPlan &hp);

void set_error(const char *fmt, ...);

} // namespace msm
