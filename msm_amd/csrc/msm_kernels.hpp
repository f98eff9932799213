// Device-side data structures and launch wrappers of the gfx950 kernels.
#pragma once
#include <functional>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace msm {

// The last kernel of a stage can signal an event by its own completion (hipExtLaunchKernelGGL's stop event) instead of
// through a marker queued behind it with hipEventRecord, which keeps the next kernel of the same stream waiting for some
// microseconds.  The engine sets tail_event before the stage; the launch that goes through MSM_LAUNCH_TAIL takes it.  A
// stage that took another path leaves it set, and the engine records the event the ordinary way.
extern thread_local hipEvent_t tail_event;
inline hipEvent_t take_tail_event() {
    hipEvent_t e = tail_event;
    tail_event = nullptr;
    return e;
}
#define MSM_LAUNCH_TAIL(kernel, grid, block, lds, st, ...)                                                  \
    do {                                                                                                   \
        hipEvent_t tail_e_ = msm::take_tail_event();                                                       \
        if (tail_e_) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, nullptr, tail_e_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                \
    } while (0)

// Uniform binning of the periodic cell.  A particle with fractional coordinate
// s in [0,1) falls into bin min(int(s*nb), nb-1) per axis (x fastest); the
// stored coordinate is s*scale (scale = cell edge for pair cells, = M for
// grid bins, so that for grid bins the bin index is floor(coordinate)).
struct BinGeom {
    int nb[3];
    int nbins;
    double invA[3];
    double scale[3];
    double origin[3]; // corner of the cell (0 for periodic plans)
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
    double cga[8];   // cg / a   (gamma(rho)/a as a polynomial in y)
    double cda[8];   // cd / a^3
    double cgr[8];   // gamma(r/a)/a          = sum cgr[n] (r^2)^n  (cga re-expanded in r^2)
    double cdr[8];   // -gamma'(r/a)/(r a^2)  = sum cdr[n] (r^2)^n
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
    // piece[k][n]: coefficient of f^n of the weight of the k-th node of a particle's support, f in [0,1) the
    // fractional grid coordinate: Phi(f + p/2 - 1 - k) = sum_n piece[k][n] f^n; dpiece = d/df of the same
    double piece[6][6];
    double dpiece[6][6];
};

// tiles of SPLINE_TILE^3 grid cells for the LDS-tiled anterpolation
#define SPLINE_TILE 8
struct TileGeom {
    int nt[3];
    int ntiles;
    int tz0 = 0; // first layer of tiles along z that a launch covers: workgroup b takes tile b of the layers tz0, tz0 + 1, ... (wrapping)
};

// ---- binning ----
void launch_bin(hipStream_t st, const double *d_r, const double *d_q, int N, const BinGeom &g, int *bin_of, int *count,
                int *start, int *arrival, int *slot_idx, int *scan_tmp, int *order, double4 *ps);
void launch_exclusive_scan(hipStream_t st, const int *in, int n, int *scan_tmp, int *out);
void launch_bin_scatter(hipStream_t st, const int *bin_of, const int *arrival, int N, const int *start, int *slot_idx);

// ---- cluster pairs ----
// Particles are sorted into columns (x, y) and by z inside a column, then cut into clusters of CL_G consecutive
// particles; a column's last cluster is padded with inert slots so that no cluster straddles two columns.
#define CL_G 8
struct ClusterGeom {
    int ncx, ncy, nzf;   // columns along x and y, fine z bins per column
    int kx, ky, kz;      // grid-aligned columns (0 if not): kx x ky level-0 grid cells per column, kz fine bins per cell
    int ntx;             // column tiles along x: columns are numbered tile by tile (8 x 8 columns per tile), see cl_col()
    int ncols, nbins;    // column numbers in use (whole tiles, some columns empty), ncols*nzf
    int max_clusters;    // capacity of the per-cluster arrays: ceil(N/CL_G) + ncols
    int list_stride;     // list entries reserved per cluster
    double invA[3], A[3];
    double inv_pitch[2]; // columns per unit length
    double pitch[2];
    double inv_dzf, dzf; // fine bins per unit length in z
    double far;          // coordinate given to padding slots
    float a2_wide;       // cutoff^2 widened to cover single-precision rounding in the box tests
    double origin[3];    // corner of the cell (0 for periodic plans): positions enter as r - origin
    int sort_blocks;     // workgroups of the two-level sort (0: the sort with one counter per fine bin is used)
    int n_owned;         // particles with original index >= n_owned are halo copies; CL_ALL_OWNED when there are none
    // A slab rank (grid-aligned columns, two-level sort): the fine bins at which ownership changes along a column, in
    // ascending order (nzf = none).  The sort starts a new cluster there (the cluster before it is padded), so that no
    // cluster mixes owned particles and halo copies; a column is given nface clusters more than its particles need.
    int nface;
    int face[2];
};
#define CL_ALL_OWNED 0x7fffffff
struct ClusterBuffers {
    int *count, *start;         // per fine bin (start has nbins + 1 entries): particles in earlier bins; with ClusterGeom::nface
                                // the slot of the bin's first particle instead (padding at the faces included)
    int *colncl, *clstart;      // clusters per column and their exclusive scan (ncols + 1; last = cluster count)
    int *order;                 // original particle index per slot, CL_DUMMY for padding
    double4 *ps;                // (x, y, z, q) per slot, absolute coordinates in [0, A)
    double4 *blo, *bhi;         // cluster bounding boxes
    float4 *fbox;               // the same rounded outwards to single precision: [2k] = low corner, [2k+1] = high corner;
                                // [2k].w = 1 if every particle of the cluster is owned (no halo copy among them), else 0
    int *list, *n0, *n1, *n2, *n3; // neighbour-cluster lists: n0 entries (self + later clusters) from the front, n2 (earlier
                                // clusters) from the middle, n1 (through a periodic image) from the back; n0 < 0: overflow.
                                // The front and the middle hold clusters whose particles are all owned; unshifted neighbours
                                // with halo copies among their particles (n3 of them) run down from just below the middle.
    // two-level sort: per (workgroup, column) counts / offsets, particles per column and their exclusive scan, and the
    // particles as they land in their column's bucket (any order inside a bucket)
    int *hist, *coltotal, *cpstart;
    double4 *bps;
    int2 *bkey;                 // (original index, fine bin inside the column) of a bucket entry
    // profiling hook (null unless msm_amd_set_pair_trace gave a buffer): per cluster slot the 100 MHz clock at the start
    // and the end of its wave in the list-walking pair kernel, the hardware id of the wave and of its XCD
    unsigned long long *trace;
    // When set, the pair kernels leave the short-range acceleration of slot s in acc_slots[s] (x, y, z; 32 bytes per slot,
    // so the 8 particles of a cluster are one contiguous 256-byte block) instead of scattering 24-byte records into the
    // caller's array by original index; the interpolation kernel, which walks the same slots, then adds the grid part
    // and does the one scatter into the caller's order.  Null: the pair kernels write the caller's array themselves.
    double4 *acc_slots;
};
void launch_cluster_sort(hipStream_t st, const double *d_r, const double *d_q, int N, const ClusterGeom &g,
                         const ClusterBuffers &b, int *bin_of, int *arrival, int *slot_idx, int *scan_tmp);
// variant 0: build lists, then one wave per cluster walks its list; variant 1: no lists, every wave enumerates and
// evaluates its candidates directly.  Clusters whose list overflowed list_stride are evaluated by the direct kernel.
// ev (optional): two events recorded immediately before and after the list-walking kernel
// max_wgs > 0 caps the workgroups of the list-walking kernel (variant 3), which then loops over its clusters
// after_lists (optional): an event recorded between the list build and the pair kernel
void launch_pair_clusters(hipStream_t st, const ClusterGeom &g, const ClusterBuffers &b, const PairParams &pp, double *acc,
                          double *epart, int variant, int n_total, hipEvent_t *ev, int max_wgs = 0, hipEvent_t after_lists = nullptr,
                          int lds_pad = 0, // lds_pad: unused LDS bytes per workgroup of the list-walking kernel (caps the waves per SIMD)
                          const std::function<void()> *between = nullptr, // called between the list build and the pair kernel: what it
                                                                          // queues on st runs there
                          int nchunks = 1, // the list-walking kernel (one-wave workgroups) in that many launches
                          const std::function<void()> *before_last = nullptr); // called before the last of several launches

// ---- neighbour structures kept over several evaluations (msm_amd_eval_reuse) ----
// The "pair set": the cluster order of the last rebuild, frozen (slots, original indices, fine-bin prefix sums), with
// positions that follow the particles; per half cluster (4 particles) the slots of the particles that were within
// cutoff + skin of one of its 4 at the rebuild, in three sections (clusters after it / before it / through a periodic
// image).  While no particle has moved more than skin / 2 since the rebuild, every pair inside the cutoff is listed.
struct ReuseBuffers {
    double4 *ps, *ref;          // current / rebuild-time position and charge per slot (ps has one inert cluster behind the last slot)
    int *order;                 // original index per slot, CL_DUMMY for padding
    double4 *blo, *bhi;         // cluster bounding boxes of the current positions (for the list-free kernel)
    float4 *fbox;
    int *start, *clstart, *colncl; // frozen copies of the sort's prefix sums
    int *list, *n0, *n1, *n2;   // cluster lists built with cutoff + skin (n0 < 0: this cluster goes to the list-free kernel)
    int list_stride;
    int *hl;                    // half lists: [cluster][half][hcap] entries (slot | shift code << 27), each section padded to 16
    int4 *hn;                   // per (cluster, half): groups of 16 entries after / before / through an image
    int hcap;
    int *need;                  // device flag: 1 = the next evaluation rebuilds (set by the displacement check, cleared by the pair kernel)
    int *counters;              // [0] clusters at the rebuild, [1] rebuilds so far
};
// one evaluation's short-range part from the pair set: displacement check and position update, (conditionally, decided
// on the device) adoption of the fresh sort in `main`, cluster lists and half lists with cutoff + skin, then the pair
// kernel over the half lists and the list-free kernel for clusters whose lists overflowed
void launch_reuse_pairs(hipStream_t st, const ClusterGeom &g, const ClusterBuffers &main, const ReuseBuffers &rs, const PairParams &pp,
                        const double *d_r, const double *d_q, double skin, double *acc, double *epart, hipEvent_t *ev);

// ---- short range ----
void launch_pair_cells(hipStream_t st, const double4 *ps, const int *cstart, const int *order, int ncell,
                       const PairParams &pp, double *acc, double *epart, int variant);
void launch_pair_allimages(hipStream_t st, const double *d_r, const double *d_q, int N, const PairParams &pp, double *acc,
                           double *epart, int *nepart);
// ---- grid transfer ----
void launch_anterp(hipStream_t st, const double4 *ps_grid, const int *gstart, const SplineParams &sp, double *grid);
// LDS-tiled form; scratch holds ntiles * (SPLINE_TILE + p - 1)^3 doubles.  Needs M >= 24 on every axis and the
// plain index convention (no node_shift / drop_last).
// Where a tile kernel finds its particles: slots of a sorted particle array in grid-cell order (runs = rows of
// cells, coordinates already in grid units) or in cluster order (runs = columns, absolute coordinates).
struct TileSrc {
    const double4 *ps;
    const int *order;
    const int *start;    // per grid cell (grid-cell order) or per fine bin (cluster order), exclusive prefix sums
    const int *clstart;  // cluster order only
    ClusterGeom cg;      // cluster order only
    int cluster_order;
    double scale[3];     // slot coordinate -> grid units
    const double4 *acc_slots; // cluster order only, may be null: the short-range accelerations per slot (ClusterBuffers::acc_slots)
};
// A/B choices of the tile kernels, kept per handle
struct SplineOpts {
    int interp_threads = 256;       // threads of a tile workgroup of the interpolation kernel (multiple of 64, at most 512)
    int interp_split = 0;           // workgroups that share the particles of a tile (0 = one)
    int interp_remap = 1;           // whole-grid evaluations: an XCD takes a contiguous run of tiles
    int anterp_cols_waves = 0;      // waves per tile workgroup of the column kernel: 4, 8, 16, or 0 = by the share of owned planes
    bool anterp_cols_order4 = false; // the column kernel at order 4 too
    bool anterp_cols = true;        // register accumulation per column (cluster order, columns up to 2 x 2 cells)
    bool anterp_cols_weights = true; // order 6: the staged records hold the particles' weights, not their offsets (same bits)
};
bool launch_anterp_tiled(hipStream_t st, const TileSrc &S, const SplineParams &sp, const TileGeom &tl, double *scratch,
                         int *tile_used, double *grid, const SplineOpts &opt);
void launch_interp_tiled(hipStream_t st, const TileSrc &S, const SplineParams &sp, const TileGeom &tl, const double *egrid,
                         double *acc, const SplineOpts &opt);
void launch_interp(hipStream_t st, const double4 *ps_grid, const int *order, int N, const SplineParams &sp,
                   const double *egrid, int index_shift, double *acc);
// one axis of the separable restriction: out has the axis halved.  J[0..p/2]
// the three axis passes of a level transfer fused (same sums, same order); fine grid Mf, coarse grid Mf/2
void launch_restrict3(hipStream_t st, const double *fine, double *coarse, const int Mf[3], int p, const double J[4]);
void launch_prolong3_add(hipStream_t st, const double *coarse, double *fine, const int Mf[3], int p, const double J[4]);
void launch_restrict_axis(hipStream_t st, const double *in, double *out, const int Min[3], int axis, int p, const double J[4]);
// one axis of the separable prolongation: out has the axis doubled; accumulate != 0 adds into out
void launch_prolong_axis(hipStream_t st, const double *in, double *out, const int Min[3], int axis, int p, const double J[4],
                         int accumulate);
struct ConvArgs {
    int M[3], ks[3], kf[3];
    int z0, nz, accumulate;
    // k_conv_multi only: the stencil is nsub^3 cubes of ks^3 taps, cube (a, b, c) starting at offset kf + ks (a, b, c),
    // stored one after the other (a fastest), each transposed [sy][sx][sz]
    int nsub = 1;
    // k_conv_multi only: apply the taps sy in [sy0, sy1) (sy1 = 0: all of them).  A thin slab gives the chip fewer tiles
    // than it has CUs; the taps of a tile are then shared out over several workgroups that write partial sums
    int sy0 = 0, sy1 = 0;
};
// up to 8 convolutions (levels, or plane ranges of a level) that share one launch
struct ConvBatch {
    int n;
    int split = 0;       // a slab of planes on level 0: tiles of 8 planes whose stencil columns are shared out over wave groups
    int sym = 0;         // in: the caller allows the mirror symmetry of the stencils along x to be used (sums in another order)
    int tile_planes = 0; // in: planes per tile of the slab kernel (15 taps: 6, 8, 9, 10, 12; 19 taps: 8, or 16 to 19 outputs per thread of the whole-grid kernel; 0: chosen per launch)
    int blk0[9];
    int ntx[8], nty[8];
    ConvArgs c[8];
    const double *q[8];
    const double *Kt[8]; // transposed stencils [sy][sx][sz]
    double *e[8];
};
// the three axis passes of a restriction in one launch, tile by tile through LDS (same bits as the separate passes)
// z0, nz: the output planes [z0, z0 + nz), modulo the grid (coarse planes of the restriction, fine planes of the prolongation)
void launch_restrict_tiled(hipStream_t st, const double *fine, double *coarse, const int Mf[3], int p, const double J[4], int z0,
                           int nz);
void launch_prolong_tiled_add(hipStream_t st, const double *coarse, double *fine, const int Mf[3], int p, const double J[4], int z0,
                              int nz);
// fine += P(mid + P(low)) for whole grids whose sizes are multiples of 4 (fine Mf, mid Mf / 2, low Mf / 4): the bits of two
// launch_prolong_tiled_add calls in `fine`; `mid` is not updated
void launch_prolong_chain_add(hipStream_t st, const double *low, const double *mid, double *fine, const int Mf[3], int p,
                              const double J[4]);
bool launch_conv_multi(hipStream_t st, ConvBatch &B);
// coarsest level + Fourier level, both full-period, as offset tables Ta (sum) and Tb (Fourier part): ea = Ta * q, eb = Tb * q
void launch_conv_small_pair(hipStream_t st, const double *q, const double *Ta, const double *Tb, const int M[3], double *ea,
                            double *eb);
// ---- grid convolution: e[m] = sum_s K[s] q[(m - (first + s)) mod M] for planes [z0, z0+nz) ----
// open boundaries: e[m] = sum_n K[m - n] q[n] over the nodes of the grid without wrap-around; K holds every node
// difference -(M-1)..M-1, x fastest
// part: conv_open_groups(M) * M[0] * M[1] * M[2] doubles of scratch (partial sums)
int conv_open_groups(const int M[3]);
void launch_conv_open(hipStream_t st, const double *q, const double *K, const int M[3], double *e, double *part);
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
// y[i] = (y[i] + a[i]) + b[i] (b may be null), the order fixed
void launch_add_parts(hipStream_t st, double *y, const double *a, const double *b, size_t n);
// ---- reductions ----
// partial sums of a.b in a fixed order -> parts[0..nparts)
void launch_dot_partials(hipStream_t st, const double *a, const double *b, size_t n, double *parts, int nparts);
// parts of sum q and sum q^2 -> parts[0..nparts), parts[nparts..2 nparts)
void launch_charge_partials(hipStream_t st, const double *q, size_t n, double *parts, int nparts);
void launch_sum_partials(hipStream_t st, const double *in, int n, double *parts, int nparts);
struct EnergyInputs {
    const double *pair_parts; int n_pair;       // sum = (A)
    const double *grid_parts; int n_grid;       // 1/2 q0.e0
    const double *four_parts; int n_four;       // 1/2 qL.efour
    const double *charge_parts; int n_charge;   // sum q | sum q^2
    double self_image_sum, csr, gamma0_over_a;
    int literal;                                // sign / bookkeeping of msmPeriodic.c:1002,1334-1359
};
// the partial sums of an evaluation in one launch: q0.e0 (g_*), qL.efour (f_*), sum q and sum q^2 (c_*, parts and
// parts + nparts), pair partials (p_*, skipped when p_parts is null)
struct EnergyPartialsArgs {
    const double *g_q, *g_e; size_t g_n; double *g_parts;
    const double *f_q, *f_e; size_t f_n; double *f_parts;
    const double *c_q; size_t c_n; double *c_parts;
    const double *p_in; int p_n; double *p_parts;
};
void launch_energy_partials(hipStream_t st, const EnergyPartialsArgs &a, int nparts);
// the same partial sums and the finishing step (launch_finalize_energies, or launch_raw_sums when raw) in one launch:
// the workgroup that ends last finishes; ticket: one zero-initialised counter in device memory, back at 0 afterwards
void launch_energy_all(hipStream_t st, const EnergyPartialsArgs &a, int nparts, const EnergyInputs &in, bool raw, double *out,
                       double *out_second, unsigned *ticket);
void launch_finalize_energies(hipStream_t st, const EnergyInputs &in, double *out7, double *out7_second = nullptr);
// raw sums for a slab rank: out5 = [pair sum, q0.e0 over owned planes, qL.efour, sum q, sum q^2]
void launch_raw_sums(hipStream_t st, const EnergyInputs &in, double *out5);

} // namespace msm
