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

// Ranges of the stencil of level l (S.first, S.size) and of the image-summed kernel table it is built from: the table
// spans ext = size + 2 W nodes per axis (W = mu + p/2), |node| <= amax, images -R..R, kernel support rcut.
struct LevelGeom {
    int W, ext[3], R[3], amax[3];
    double rcut;
};
void level_geometry(const HostPlan &hp, int l, LevelStencil &S, LevelGeom &G);
// c(k) per axis and the extent of the top-level stencil; roots of unity e^{2 pi i t / M} as (re, im) pairs
void top_prepare(HostPlan &hp);
void top_open(HostPlan &hp); // open boundaries: empty Fourier level
void top_roots(int M, std::vector<double> &re_im);

// An alternative implementation of the two stencil builders (the GPU one): level() fills hp.K[l], top() fills
// hp.cvec, hp.chi, hp.chisum and hp.KL.
struct StencilBackend {
    int (*level)(HostPlan &hp, int l, void *ctx);
    int (*top)(HostPlan &hp, void *ctx);
    void *ctx;
};

// fills Ml/hl/khat/wprime/J/K/KL/cvec/chi... from hp.d (+ hp.literal); stencils by `be` when given, else on the host
int build_host_plan(HostPlan &hp, const StencilBackend *be = nullptr);

// literal mode: the coarsest level's stencil split into its gamma term and its erf term (host; msmPeriodic.c:555-590)
void build_literal_parts(const HostPlan &hp, LevelStencil &g1_part, LevelStencil &gLR_part);

// stencil builders on the current HIP device (msm_stencil.hip)
const StencilBackend *device_stencil_backend();

// the two stencil cache files of msmSimple (msm/msmSimple.c:1018-1031, 1196-1209) for a periodic two-level plan
int write_msmsimple_cache(const HostPlan &hp, const char *dir);

void set_error(const char *fmt, ...);

} // namespace msm
