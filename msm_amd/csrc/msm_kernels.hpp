// Device-side data structures and launch wrappers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace msm {

// Uniform binning of the periodic cell.  A particle with fractional coordinate
// s in [0,1) falls into bin min(int(s*nb), nb-1) per axis (x fastest); the
// stored coordinate is s*scale (scale = cell edge for pair cells, = M for
// grid bins, so that for grid bins the bin index is floor(coordinate)).
struct BinGeom {
    int nb[3];
    int nbins;
    double invA[3];
    double scale[3];
};

struct PairParams {
    int nc[3];       // pair cells per axis (>= 3)
    double A[3];
    double a2;       // cutoff squared
    double inv_a;    // 1/a
    double inv_a2;   // 1/a^2
    double inv_a3;   // 1/a^3
    double cg[8];    // gamma(rho)       = sum cg[k] y^k, y = rho^2-1
    double cd[8];    // -gamma'(rho)/rho = sum cd[k] y^k
    int p;
    int convention;  // MSM_AMD_ACC_*
    int n_owned;     // particles with original index >= n_owned are halo copies: no output for them
};

struct GridGeom {
    int M[3];
};

// coordinate conventions of anterpolation / interpolation
struct SplineParams {
    int p;
    int M[3];
    double inv_h[3];   // 1/h per axis (interpolation derivative scale)
    int node_shift[3]; // array index j holds node coordinate j - node_shift (0 normally; M/2-1 in literal anterpolation)
    int drop_last[3];  // literal anterpolation on odd M leaves the last plane empty (msmPeriodic.c:697-702)
    int z0, nz;        // owned z planes [z0, z0+nz)
    int convention;
    int n_owned;
};

// ---- binning ----
void launch_bin(hipStream_t st, const double *d_r, const double *d_q, int N, const BinGeom &g, int *bin_of, int *count,
                int *start, int *fill, int *slot_idx, int *scan_tmp, int *order, double4 *ps);
// ---- short range ----
void launch_pair_cells(hipStream_t st, const double4 *ps, const int *cstart, const int *order, int ncell,
                       const PairParams &pp, double *acc, double *epart, int variant);
void launch_pair_allimages(hipStream_t st, const double *d_r, const double *d_q, int N, const PairParams &pp, double *acc,
                           double *epart, int *nepart);
// ---- grid transfer ----
void launch_anterp(hipStream_t st, const double4 *ps_grid, const int *gstart, const SplineParams &sp, double *grid);
void launch_interp(hipStream_t st, const double4 *ps_grid, const int *order, int N, const SplineParams &sp,
                   const double *egrid, int index_shift, double *acc);
// one axis of the separable restriction: out has the axis halved.  J[0..p/2]
void launch_restrict_axis(hipStream_t st, const double *in, double *out, const int Min[3], int axis, int p, const double J[4]);
// one axis of the separable prolongation: out has the axis doubled; accumulate != 0 adds into out
void launch_prolong_axis(hipStream_t st, const double *in, double *out, const int Min[3], int axis, int p, const double J[4],
                         int accumulate);
// ---- grid convolution: e[m] = sum_s K[s] q[(m - (first + s)) mod M] for planes [z0, z0+nz) ----
void launch_conv_direct(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                        const int kfirst[3], double *e, int z0, int nz, int accumulate);
// one wave per output node (small grids with stencils spanning the period)
void launch_conv_small(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                       const int kfirst[3], double *e, int z0, int nz, int accumulate);
// LDS-tiled variant; returns false if the stencil shape is not covered (caller falls back to direct)
bool launch_conv_tiled(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                       const int kfirst[3], double *e, int z0, int nz);
// the reference's dense grid-to-grid loop with its own index arithmetic (msmPeriodic.c:874-918)
void launch_conv_literal(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                         const int kfirst[3], double *e);
void launch_axpy(hipStream_t st, double *y, const double *x, size_t n);
// ---- reductions ----
// partial sums of a.b in a fixed order -> parts[0..nparts)
void launch_dot_partials(hipStream_t st, const double *a, const double *b, size_t n, double *parts, int nparts);
// parts of sum q and sum q^2 -> parts[0..nparts), parts[nparts..2 nparts)
void launch_charge_partials(hipStream_t st, const double *q, size_t n, double *parts, int nparts);
struct EnergyInputs {
    const double *pair_parts; int n_pair;       // sum = (A)
    const double *grid_parts; int n_grid;       // 1/2 q0.e0
    const double *four_parts; int n_four;       // 1/2 qL.efour
    const double *charge_parts; int n_charge;   // sum q | sum q^2
    double self_image_sum, csr, gamma0_over_a;
    int literal;                                // sign / bookkeeping of msmPeriodic.c:1002,1334-1359
};
void launch_finalize_energies(hipStream_t st, const EnergyInputs &in, double *out7);
// raw sums for a slab rank: out5 = [pair sum, q0.e0 over owned planes, qL.efour, sum q, sum q^2]
void launch_raw_sums(hipStream_t st, const EnergyInputs &in, double *out5);

} // namespace msm
