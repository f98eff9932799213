// Handle layer + the drop-in msmperiodic()/msmPeriodicRunner() built on it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "msm_amd.h"
#include "msm_host.hpp"
#include "msm_kernels.hpp"

using namespace msm;

#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);     \
            return MSM_AMD_EHIP;                                                                      \
        }                                                                                             \
    } while (0)

enum { T_BIN = 0, T_PAIR, T_ANT, T_RES, T_CONV, T_TOP, T_PRO, T_INT, T_RED, T_TOTAL, T_COUNT };
static const int NPARTS = 256;
static const int CONV_SPLIT_MAX = 5; // shares of a tile's taps (conv_part holds the partial sums of all but the first)

struct msm_amd_handle {
    HostPlan hp;
    int device = 0;
    hipStream_t stream = nullptr;
    int N = 0, n_owned = 0, capacity = 0;
    bool small_box = false;
    int nc[3] = {0, 0, 0}, ncell = 0;
    BinGeom cellg, gridg;
    PairParams pp;
    SplineParams sp;
    // particles
    double4 *ps_cell = nullptr, *ps_grid = nullptr;
    int *order_cell = nullptr, *order_grid = nullptr, *bin_of = nullptr, *slot_idx = nullptr;
    int *ccount = nullptr, *cstart = nullptr, *gcount = nullptr, *gstart = nullptr, *arrival = nullptr;
    int *scan_tmp = nullptr;
    // grids
    std::vector<double *> gq, ge, dK, dKt;
    // stencils wider than the tiled convolution handles in one piece: cut into sub_n^3 cubes of sub_s^3 taps
    std::vector<double *> dKsub;
    std::vector<int> sub_s, sub_n;
    double *efour = nullptr, *dKL = nullptr, *tmpA = nullptr, *tmpB = nullptr;
    double *dTsum = nullptr, *dTfour = nullptr; // coarsest-level + Fourier stencils as offset tables (both span one period)
    int opt_top_fused = 1;
    LevelStencil lit_part[2]; // literal mode: gamma / erf terms of the level-0 stencil (diagnostic energies), built on first use
    double *lit_part_d[2] = {nullptr, nullptr}, *lit_part_e = nullptr; // their device copies and the potential they give (first use)
    double *open_part = nullptr; // open boundaries: partial sums of the coarsest level's all-pairs convolution
    int opt_sort_two_level = 1; // creation-time choice (MSM_AMD_SORT=0 in the environment selects the sort with global counters)
    // reductions
    double *epart = nullptr;
    int nepart = 0;
    double *parts = nullptr; // [grid NPARTS | four NPARTS | charge 2*NPARTS | pair NPARTS]
    double *d_energies = nullptr, *h_energies = nullptr;
    unsigned *energy_ticket = nullptr; // counter of launch_energy_all
    int opt_energy_fused = 0;          // 1: partial sums and finishing step in one launch (A/B).  Measured at 1M particles: 24 us against
                                       // the 18 us of the two launches (the finishing workgroup reads the partial sums of the
                                       // others past every cache), 60 us when the hand-over is a release / acquire fence pair
                                       // (each fence writes back and invalidates a chiplet's L2)
    const double *cur_r = nullptr, *cur_q = nullptr;
    // host staging for eval_host
    double *s_r = nullptr, *s_q = nullptr, *s_acc = nullptr, *s_acc2 = nullptr;
    // profiling
    bool profiling = false;
    hipEvent_t ev[T_COUNT + 1];
    hipEvent_t ev_pk[2];      // around the cluster-pair kernel alone
    bool pk_recorded = false;
    double pair_kernel_ms = -1.0;
    bool ev_made = false;
    double last_ms[T_COUNT];
    int opt_pair_variant = 8, opt_conv_tiled = 2, opt_conv_small = 1; // conv_tiled: 0 off, 1 per level, 2 levels share a launch
    // cluster-pair path of the short-range sum
    ClusterGeom clg;
    ClusterBuffers clb = {};
    bool cells_binned = false, clusters_sorted = false, grid_binned = false;
    // Whole evaluations (msm_amd_eval, msm_amd_slab_step) keep the short-range accelerations in slot order between the
    // pair kernels and the interpolation kernel, which adds the grid part and writes the caller's array once
    // (ClusterBuffers::acc_slots); the stage entry points, whose d_acc is the caller's to read after every stage, do not.
    double4 *acc_slots = nullptr;
    int opt_acc_slots = 1;      // 0: the pair kernels scatter into the caller's array themselves (A/B)
    int opt_face_cut = 1;       // slab ranks: clusters never mix owned particles and halo copies (0: cut every 8 particles, A/B)
    bool face_room = false;     // the cluster arrays were sized for the padded clusters at the faces (a handle created as a slab rank)
    bool fused_acc = false;     // set for the duration of a whole evaluation that qualifies
    // msm_amd_eval_reuse: the pair set and its lists (allocated by the first call, sized for its skin)
    ReuseBuffers rs = {};
    bool rs_made = false;
    double rs_skin = -1.0;      // skin the lists in the buffers were built with (a change makes the next call rebuild)
    double rs_skin_cap = 0.0;   // largest skin the buffers are sized for
    long rs_evals = 0;
    // LDS-tiled anterpolation
    TileGeom tiles = {};
    double *tile_scratch = nullptr;
    int *tile_used = nullptr;
    bool anterp_tiled = false;
    int opt_anterp_tiled = 1;
    SplineOpts sopt; // A/B choices of the tile kernels
    // second stream: the grid pipeline runs beside the list build and the pair kernel
    double *energies_alias = nullptr; // device-visible host slot the energies of the evaluation in flight go to
    // A handle creates at most two streams, once, both non-blocking and without a priority: stream_b (the second branch
    // of an evaluation) and stream_a (what a handle on the default stream runs the first branch of a slab step on).
    hipStream_t stream_b = nullptr;
    hipStream_t stream_a = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipEvent_t ev_energy = nullptr; // slab step: behind the all-reduce of the energy sums when that runs on the second stream
    bool energy_on_b = false;       // the last slab step left its energy sums to the second stream
    int opt_energy_on_b = 1;        // (0: the all-reduce on the first stream, where the next step queues behind it; A/B)
    hipEvent_t ev_gfork = nullptr, ev_gjoin = nullptr; // fork / join of the coarse-level branch of the grid stage (on stream_b)
    hipEvent_t pairs_after = nullptr; // set for one msm_amd_stage_pairs call: recorded between the list build and the pair kernel
    std::function<void()> pairs_between; // set for one msm_amd_stage_pairs call: called between the list build and the pair kernel
    std::function<void()> pairs_before_last; // the same, before the last launch of a pair kernel launched in chunks
    int pairs_chunks_now = 1;                // launches of the list-walking pair kernel of the msm_amd_stage_pairs call under way
    int opt_pair_chunks = 2;  // launches of the list-walking pair kernel of a slab step whose last launch waits for the second stream
                              // (the earlier ones run while the exchanges travel; 1: one launch)
    hipEvent_t ev_restricted = nullptr;  // set for one grid stage: recorded behind its restrictions
    hipEvent_t ev_stage_tail = nullptr;  // set for one whole grid stage: signalled by its last kernel (cleared when that happened)
    int opt_conv_sym = 2;       // tiled convolution: mirror columns of a stencil share their multiplications (2: and 16 outputs per thread; 0: off)
    int opt_transfer_chain = 1; // two prolongations per launch where the grids allow it
    int opt_conv_tile_planes = 0; // slab ranks: planes per tile of k_conv_split (0: chosen per launch)
    int opt_conv_groups = 1;  // slab ranks: the level-0 convolution by k_conv_split
    int opt_slab_level1 = 1;  // slab ranks: level 1 convolved on the planes under the rank's own only
    int opt_grid_streams = 0; // measured: slower (1.68 vs 1.65 ms at 1M; the chain of small kernels does not get CUs while the
                              // level-0 convolution runs), and cross-stream waits stalled with 4 processes sharing one GPU
    hipEvent_t ev_fork = nullptr, ev_pairs = nullptr, ev_join = nullptr;
    double *conv_part = nullptr; // partial sums of a level-0 convolution whose taps are shared out over workgroups (4 grids)
    int opt_conv_split = 0;      // 0: never (default: measured slower, 128 against 108 us for 18 planes of 96 x 96 in three shares);
                                 // 1: when a slab rank has few tiles; n > 1: that many shares
    int opt_pair_lds_pad = 0; // unused LDS per workgroup of the pair kernel, bytes: caps its waves per SIMD (16 KB are its own)
    int opt_pair_wpb = 0;     // A/B: waves per workgroup of the list-walking pair kernel (0: one; 2, 4)
    int opt_list_wpb = 0;     // A/B: waves per workgroup of the list build (0: four; 1)
    int pair_wgs_now = 0;     // A/B: cap on the workgroups of the pair kernel (0: one workgroup per four clusters; measured
                              // slower at 2, 3 and 4 workgroups per CU, and kernels of a second stream gain nothing from the room)
    int opt_slab_own_stream = 1; // slab step of a handle on the default stream: on streams of its own (0: as given, A/B)
    int opt_tail_events = 1;   // slab step: events signalled by the completion of a stage's last kernel instead of a marker behind it (0: A/B)
    int opt_pair_chunks_coarse = -1; // the same where only the chain of the coarser levels runs beside the pair kernel and nothing
                                     // waits (-1: four for a rank with more than a million level-0 nodes, else one)
    int opt_conv_leads = 1;    // slab step with the grid stage on the second stream: the pair kernel starts behind the restrictions (0: A/B)
    int opt_slab_overlap = -1; // slab step, what runs on a second stream beside the list build and the pair sum: 0 nothing (one
                              // stream: for several processes sharing one GPU, where cross-stream waits were seen to stall);
                              // 1 the charge exchange; 2 and the chain of the coarser levels behind it (-16 / -12 / -23 us at
                              // 8 / 4 / 2 ranks of 1M particles); 3 and the level-0 convolution and the last prolongation, i.e.
                              // the whole grid stage (a further -21 us at 8 ranks, nothing at 4, +35 us at 2, where a
                              // convolution of 54 planes competes with the pair kernel; 10M particles, 8 ranks: -1 %);
                              // 4 and the anterpolation (slower everywhere); -1: 3 for a rank that owns at most 300 000
                              // nodes of level 0 (12 or 24 planes of 96 x 96; with 32 planes of 256 x 256 the kernels of the grid
                              // stage fill the chip by themselves and 3 costs 3 %), else 2
    // parts of the grid branch on a second stream beside the pair branch: 1 all of it (slower: the level-0 convolution
    // wants the FP64 pipe the pair kernel is using), 2 the anterpolation (bound by the LDS port: -22 us), 3 and the chain
    // of the coarser levels (-35 us), 4 as 3, from the end of the list build on.  0: the stage times and the pair
    // kernel's own duration stay those of kernels that have the chip to themselves.
    int opt_overlap = 0;
    int opt_transfer_fused = 4;              // 4: restriction and prolongation tiled through LDS, one launch per level (default); 3: restriction
                                             // only; 0: separable passes; 1, 2: thread per node, slower at every size measured
    long opt_transfer_fused_max = 1L << 30;
};

static int g_mode = MSM_AMD_MODE_AUTO;
static int g_quiet = 0;

static int g_stencils = MSM_AMD_STENCILS_DEVICE;

extern "C" void msm_amd_set_stencil_backend(int where) { g_stencils = where; }
extern "C" int msm_amd_get_stencil_backend(void) {
    // the environment overrides the process-wide setting (A/B runs of unmodified programs)
    if (const char *e = getenv("MSM_AMD_STENCILS")) {
        if (!strcmp(e, "host")) return MSM_AMD_STENCILS_HOST;
        if (!strcmp(e, "gpu") || !strcmp(e, "device")) return MSM_AMD_STENCILS_DEVICE;
    }
    return g_stencils;
}
static bool g_mode_set = false;   // msm_amd_set_mode() was called: the environment is not consulted
static int g_last_call_mode = -1; // what the last msmperiodic() did: MSM_AMD_MODE_LITERAL or MSM_AMD_MODE_SCALABLE
static bool g_auto_noticed = false;
extern "C" void msm_amd_set_mode(int mode) { g_mode = mode; g_mode_set = true; }
extern "C" int msm_amd_get_mode(void) { return g_mode; }
extern "C" int msm_amd_last_call_mode(void) { return g_last_call_mode; }
// mode of a msmperiodic() call: msm_amd_set_mode() if it was called, else MSM_AMD_MODE=literal|scalable|auto from the
// environment (an unmodified driver has no other way to choose), else auto
static int effective_mode() {
    if (g_mode_set) return g_mode;
    const char *e = getenv("MSM_AMD_MODE");
    if (e && !strcmp(e, "literal")) return MSM_AMD_MODE_LITERAL;
    if (e && !strcmp(e, "scalable")) return MSM_AMD_MODE_SCALABLE;
    return g_mode;
}
extern "C" void msm_amd_set_quiet(int quiet) { g_quiet = quiet; }

extern "C" int msm_amd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <typename T>
static int dalloc(T **p, size_t n) {
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
        return MSM_AMD_ENOMEM;
    }
    return MSM_AMD_OK;
}

// Where ownership changes along a column of a slab rank's sort (ClusterGeom::face): the fine bins of the slab's two
// faces.  A periodic slab that touches the end of the cell has its lower (upper) halo at the other end: its second
// face is put in the middle of the planes it does not hold, so that no cluster reaches from one halo to the other.
static void set_slab_faces(msm_amd_handle *h) {
    const msm_amd_plan_desc &d = h->hp.d;
    ClusterGeom &cg = h->clg;
    cg.nface = 0;
    cg.face[0] = cg.face[1] = cg.nzf;
    if (!h->opt_face_cut || !h->face_room || d.nzplanes >= d.M[2] || cg.kx <= 0 || cg.kz <= 0 || cg.sort_blocks <= 0 || h->small_box) return;
    const int Mz = d.M[2], z0 = d.zplane0, z1 = d.zplane0 + d.nzplanes;
    const int mid = ((z1 + (Mz - d.nzplanes) / 2) % Mz) * cg.kz; // centre of the planes of the other ranks
    int a = z0 > 0 ? z0 * cg.kz : (d.open_boundary ? cg.nzf : mid);
    int b = z1 < Mz ? z1 * cg.kz : (d.open_boundary ? cg.nzf : mid);
    if (a > b) std::swap(a, b);
    if (a <= 0) a = cg.nzf; // (a face at the first bin cuts nothing)
    if (a > b) std::swap(a, b);
    cg.face[0] = a;
    cg.face[1] = b;
    cg.nface = 2;
}

static int create_impl(const msm_amd_plan_desc *plan, bool literal, int device, void *stream, msm_amd_handle **out) {
    if (!plan || !out) { set_error("msm_amd_create: null argument"); return MSM_AMD_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device is visible; msm_amd has no CPU path");
        return MSM_AMD_ENODEV;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return MSM_AMD_EINVAL; }
    HIPCHK(hipSetDevice(device));
    msm_amd_handle *h = new msm_amd_handle();
    h->hp.d = *plan;
    h->hp.literal = literal;
    if (h->hp.d.nzplanes <= 0) { h->hp.d.zplane0 = 0; h->hp.d.nzplanes = plan->M[2]; }
    if (h->hp.d.zplane0 < 0 || h->hp.d.zplane0 + h->hp.d.nzplanes > plan->M[2]) {
        set_error("owned z planes [%d,%d) outside the grid", h->hp.d.zplane0, h->hp.d.zplane0 + h->hp.d.nzplanes);
        delete h;
        return MSM_AMD_EINVAL;
    }
    int rc = build_host_plan(h->hp, msm_amd_get_stencil_backend() == MSM_AMD_STENCILS_DEVICE ? device_stencil_backend() : nullptr);
    if (rc) { delete h; return rc; }
    const msm_amd_plan_desc &d = h->hp.d;
    h->device = device;
    h->stream = (hipStream_t)stream;
    h->N = d.N;
    h->capacity = d.N;
    h->n_owned = (d.n_owned > 0 && d.n_owned <= d.N) ? d.n_owned : d.N;
    const int N = d.N, L = d.L;

    // pair cells: edge >= a; with fewer than 3 cells on an axis a pair can see several images -> all-images kernel
    for (int k = 0; k < 3; k++) h->nc[k] = (int)std::floor(d.A[k] / d.a);
    h->small_box = h->nc[0] < 3 || h->nc[1] < 3 || h->nc[2] < 3;
    if (h->small_box && N > 65536) {
        set_error("cutoff a=%g exceeds a third of the cell: the all-images pair kernel is limited to N<=65536", d.a);
        delete h;
        return MSM_AMD_ERANGE;
    }
    if (h->small_box && (d.nzplanes != d.M[2] || h->n_owned != N)) {
        set_error("slab decomposition needs at least 3 pair cells per axis");
        delete h;
        return MSM_AMD_EINVAL;
    }
    h->ncell = h->small_box ? 1 : h->nc[0] * h->nc[1] * h->nc[2];
    for (int k = 0; k < 3; k++) {
        h->cellg.nb[k] = h->small_box ? 1 : h->nc[k];
        h->cellg.origin[k] = h->gridg.origin[k] = d.origin[k];
        h->cellg.invA[k] = 1.0 / d.A[k];
        h->cellg.scale[k] = d.A[k];
        h->gridg.nb[k] = d.M[k];
        h->gridg.invA[k] = 1.0 / d.A[k];
        h->gridg.scale[k] = (double)d.M[k];
    }
    h->cellg.nbins = h->ncell;
    h->gridg.nbins = d.M[0] * d.M[1] * d.M[2];

    PairParams &pp = h->pp;
    for (int k = 0; k < 3; k++) { pp.nc[k] = h->nc[k]; pp.A[k] = d.A[k]; }
    pp.a2 = d.a * d.a;
    pp.inv_a = 1.0 / d.a;
    pp.inv_a2 = 1.0 / (d.a * d.a);
    pp.inv_a3 = 1.0 / (d.a * d.a * d.a);
    gamma_poly(d.p, pp.cg, pp.cd);
    for (int k = 0; k < 8; k++) { pp.cga[k] = pp.cg[k] * pp.inv_a; pp.cda[k] = pp.cd[k] * pp.inv_a3; }
    {   // sum_k c_k (t - 1)^k = sum_n t^n sum_{k>=n} c_k C(k,n) (-1)^(k-n), t = r^2 / a^2
        auto expand = [&](const double *c, int deg, double scale, double *out) {
            for (int n = 0; n < 8; n++) out[n] = 0.0;
            for (int n = 0; n <= deg; n++) {
                long double s = 0.0L;
                for (int k = n; k <= deg; k++) {
                    long double b = 1.0L;
                    for (int i = 1; i <= n; i++) b = b * (k - n + i) / i;
                    s += (long double)c[k] * b * (((k - n) & 1) ? -1.0L : 1.0L);
                }
                out[n] = (double)(s * (long double)scale * powl((long double)pp.inv_a2, n));
            }
        };
        expand(pp.cg, d.p - 1, pp.inv_a, pp.cgr);
        expand(pp.cd, d.p - 2, pp.inv_a3, pp.cdr);
    }
    pp.p = d.p;
    pp.convention = d.acc_convention;
    pp.n_owned = h->n_owned;

    SplineParams &sp = h->sp;
    sp.p = d.p;
    for (int k = 0; k < 3; k++) {
        sp.M[k] = d.M[k];
        sp.inv_h[k] = 1.0 / h->hp.hl[0][k];
        // literal anterpolation: array index j holds node j - (M/2 - 1), and an odd M leaves its last plane
        // empty (msmPeriodic.c:697-702)
        sp.node_shift[k] = literal ? d.M[k] / 2 - 1 : 0;
        sp.drop_last[k] = (literal && (d.M[k] & 1)) ? 1 : 0;
    }
    spline_pieces(d.p, sp.piece, sp.dpiece);
    sp.z0 = d.zplane0;
    sp.nz = d.nzplanes;
    sp.convention = d.acc_convention;
    sp.n_owned = h->n_owned;

    // columns of the cluster sort: pitch ~ edge of a cube holding CL_G particles at the mean density of the
    // occupied part of the cell, fine z bins of an eighth of that
    ClusterGeom &cg = h->clg;
    {
        double occ = 1.0; // fraction of the cell a slab rank's particles (owned + halo) can occupy
        if (d.nzplanes < d.M[2]) occ = std::min(1.0, (d.nzplanes + 2.0 * (d.a / h->hp.hl[0][2] + d.p / 2)) / d.M[2]);
        if (d.open_boundary) // the particles fill their bounding box, not the padded cell
            for (int k = 0; k < 3; k++)
                if (d.extent[k] > 0) occ *= std::min(1.0, std::max(d.extent[k], 0.05 * d.A[k]) / d.A[k]);
        const double vol = d.A[0] * d.A[1] * d.A[2] * occ;
        const double c = std::cbrt(CL_G * vol / std::max(N, 1));
        // One sort for everything when the level-0 grid allows it: columns of kx x ky whole grid cells (k a
        // divisor of the spline tile edge, nearest the ideal pitch) and kz fine bins per cell along z, so that a
        // tile of grid cells is a set of columns and a run of fine bins.
        cg.kx = cg.ky = cg.kz = 0;
        const bool tiled_grid = !literal && d.M[0] >= 24 && d.M[1] >= 24 && d.M[2] >= 24;
        if (tiled_grid && d.acc_convention != MSM_AMD_ACC_LITERAL) {
            auto pick = [&](double h, int M) {
                int best = 0;
                double bd = 1e300;
                for (int k = 1; k <= SPLINE_TILE; k *= 2) {
                    const double dist = std::fabs(std::log(k * h / c));
                    if (M % k == 0 && dist < bd) { bd = dist; best = k; }
                }
                return best;
            };
            cg.kx = pick(h->hp.hl[0][0], d.M[0]);
            cg.ky = pick(h->hp.hl[0][1], d.M[1]);
            const double hz = h->hp.hl[0][2];
            int kz = (int)std::floor(8.0 * hz / c + 0.5);
            kz = std::max(1, std::min(kz, 16));
            while (kz > 1 && (long)(d.M[0] / cg.kx + 7) * (d.M[1] / cg.ky + 7) * d.M[2] * kz > 6L * N + 65536) kz--;
            cg.kz = kz;
        }
        if (cg.kx) {
            cg.ncx = d.M[0] / cg.kx;
            cg.ncy = d.M[1] / cg.ky;
        } else {
            cg.ncx = std::max(1, (int)std::floor(d.A[0] / c + 0.5));
            cg.ncy = std::max(1, (int)std::floor(d.A[1] / c + 0.5));
        }
        cg.ntx = (cg.ncx + 7) / 8;
        cg.ncols = cg.ntx * ((cg.ncy + 7) / 8) * 64;
        if (cg.kx) {
            cg.nzf = d.M[2] * cg.kz;
        } else {
            long nzf = std::max(1L, (long)std::floor(8.0 * d.A[2] / c + 0.5));
            nzf = std::min(nzf, std::max(1L, (4L * N + 1024) / cg.ncols));
            cg.nzf = (int)nzf;
        }
        cg.nbins = cg.ncols * cg.nzf;
        cg.max_clusters = (N + CL_G - 1) / CL_G + cg.ncols;
        // (a slab rank starts new clusters at the faces of its slab: up to two padded clusters more per column)
        if (d.nzplanes < d.M[2] && cg.kx > 0) { cg.max_clusters += 2 * cg.ncols; h->face_room = true; }
        {
            // list length: halves for the later / earlier neighbour clusters, the tail of the second half for shifted
            // ones.  Expected neighbours of a cluster = clusters in the volume within a of a cube of its size,
            // times 1.5 for ragged boxes; each half gets 1.3 times that (half of it would do on average).
            const double vol = d.A[0] * d.A[1] * d.A[2], e = std::cbrt(8.0 * vol / N), a = d.a;
            const double vexp = e * e * e + 6.0 * e * e * a + 3.0 * M_PI * a * a * e + 4.0 / 3.0 * M_PI * a * a * a;
            const double est = 1.5 * vexp * N / (8.0 * vol);
            long half = std::max(192L, ((long)(1.3 * est) + 31) / 32 * 32);
            const long cap = (1L << 32) / std::max(1, cg.max_clusters) / 2 / 32 * 32; // lists of at most 16 GB
            half = std::max(192L, std::min(half, cap));
            cg.list_stride = (int)(2 * half);
        }
        double amax = 0;
        for (int k = 0; k < 3; k++) { cg.A[k] = d.A[k]; cg.invA[k] = 1.0 / d.A[k]; cg.origin[k] = d.origin[k]; amax = std::max(amax, d.A[k]); }
        cg.pitch[0] = d.A[0] / cg.ncx; cg.pitch[1] = d.A[1] / cg.ncy;
        cg.inv_pitch[0] = cg.ncx / d.A[0]; cg.inv_pitch[1] = cg.ncy / d.A[1];
        cg.dzf = d.A[2] / cg.nzf; cg.inv_dzf = cg.nzf / d.A[2];
        cg.far = -1.0e3 * amax;
        { const double aw = d.a + 4e-6 * amax; cg.a2_wide = (float)(aw * aw * (1.0 + 1e-6)); }
        cg.n_owned = h->n_owned >= h->N ? CL_ALL_OWNED : h->n_owned;
    }

#define DA(ptr, n)                      \
    do {                                \
        rc = dalloc(&(ptr), (size_t)(n)); \
        if (rc) { msm_amd_destroy(h); return rc; } \
    } while (0)
    // inside this function a failed HIP call must also release the handle and what it already owns
#define HIPCHK_H(call)                                                                                \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);     \
            msm_amd_destroy(h);                                                                       \
            return MSM_AMD_EHIP;                                                                      \
        }                                                                                             \
    } while (0)
    DA(h->ps_cell, N); DA(h->ps_grid, N); DA(h->order_cell, N); DA(h->order_grid, N); DA(h->bin_of, N); DA(h->slot_idx, N);
    DA(h->ccount, h->ncell); DA(h->cstart, h->ncell + 1); DA(h->arrival, N);
    DA(h->gcount, h->gridg.nbins); DA(h->gstart, h->gridg.nbins + 1);
    {
        const int nb = std::max(std::max(h->ncell, h->gridg.nbins), cg.nbins);
        DA(h->scan_tmp, nb / 2048 + 2);
    }
    if (!h->small_box) {
        ClusterBuffers &cb = h->clb;
        const size_t nslots = (size_t)cg.max_clusters * CL_G;
        // two-level sort: needs the columns (and the fine bins of a column) to fit an LDS histogram
        if (const char *e = getenv("MSM_AMD_SORT")) h->opt_sort_two_level = atoi(e) != 0; // A/B: 0 = one global counter per fine bin
        cg.sort_blocks = (h->opt_sort_two_level && cg.ncols <= 16384 && cg.nzf <= 4096) ? std::max(1, std::min(512, (N + 2047) / 2048)) : 0;
        h->clg.sort_blocks = cg.sort_blocks;
        set_slab_faces(h);
        if (cg.sort_blocks) {
            DA(cb.hist, (size_t)cg.sort_blocks * cg.ncols); DA(cb.coltotal, cg.ncols); DA(cb.cpstart, cg.ncols + 1);
            DA(cb.bps, N); DA(cb.bkey, N);
        }
        DA(cb.count, cg.nbins); DA(cb.start, cg.nbins + 1);
        DA(cb.colncl, cg.ncols + 1); DA(cb.clstart, cg.ncols + 1);
        DA(cb.order, nslots); DA(cb.ps, nslots + CL_G); // + one cluster of inert slots behind the last one
        DA(cb.blo, cg.max_clusters); DA(cb.bhi, cg.max_clusters); DA(cb.fbox, 2 * (size_t)cg.max_clusters);
        DA(cb.list, (size_t)cg.max_clusters * cg.list_stride); DA(cb.n0, cg.max_clusters); DA(cb.n1, cg.max_clusters); DA(cb.n2, cg.max_clusters); DA(cb.n3, cg.max_clusters);
        HIPCHK_H(hipMemset(cb.n3, 0, sizeof(int) * (size_t)cg.max_clusters)); // (a list build without halo copies does not write it)
        if (cg.kx > 0) DA(h->acc_slots, nslots); // (only grid-aligned columns let the interpolation walk the same slots)
    }
    h->gq.assign(L, nullptr); h->ge.assign(L, nullptr); h->dK.assign(L, nullptr); h->dKt.assign(L, nullptr);
    h->dKsub.assign(L, nullptr); h->sub_s.assign(L, 0); h->sub_n.assign(L, 0);
    for (int l = 0; l < L; l++) {
        const size_t n = (size_t)h->hp.Ml[l][0] * h->hp.Ml[l][1] * h->hp.Ml[l][2];
        DA(h->gq[l], n); DA(h->ge[l], n);
        HIPCHK_H(hipMemset(h->gq[l], 0, n * sizeof(double)));
        HIPCHK_H(hipMemset(h->ge[l], 0, n * sizeof(double)));
        const LevelStencil &S = h->hp.K[l];
        DA(h->dK[l], S.v.size()); DA(h->dKt[l], S.v.size());
        HIPCHK_H(hipMemcpy(h->dK[l], S.v.data(), S.v.size() * sizeof(double), hipMemcpyHostToDevice));
        // transposed copy [sy][sx][sz] for the tiled kernel (z taps contiguous)
        std::vector<double> t(S.v.size());
        for (int sz = 0; sz < S.size[2]; sz++)
            for (int sy = 0; sy < S.size[1]; sy++)
                for (int sx = 0; sx < S.size[0]; sx++)
                    t[((size_t)sy * S.size[0] + sx) * S.size[2] + sz] = S.v[((size_t)sz * S.size[1] + sy) * S.size[0] + sx];
        HIPCHK_H(hipMemcpy(h->dKt[l], t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
        const bool open_top = d.open_boundary && l == L - 1; // applied by k_conv_open from dK
        if (!literal && !open_top && S.size[0] == S.size[1] && S.size[1] == S.size[2] && S.size[0] > 23) {
            // smallest padded width n * s over the cube sizes the kernel is built for; ties go to fewer cubes
            const int W = S.size[0];
            int bs = 0, bn = 0;
            for (int sz = 9; sz <= 23; sz += 2) {
                const int n = (W + sz - 1) / sz;
                if (!bs || n * sz < bn * bs || (n * sz == bn * bs && n < bn)) { bs = sz; bn = n; }
            }
            std::vector<double> sub((size_t)bn * bn * bn * bs * bs * bs, 0.0);
            for (int z = 0; z < W; z++)
                for (int y = 0; y < W; y++)
                    for (int x = 0; x < W; x++) {
                        const size_t cube = ((size_t)(z / bs) * bn + y / bs) * bn + x / bs;
                        sub[cube * bs * bs * bs + ((size_t)(y % bs) * bs + x % bs) * bs + z % bs] = S.v[((size_t)z * W + y) * W + x];
                    }
            DA(h->dKsub[l], sub.size());
            HIPCHK_H(hipMemcpy(h->dKsub[l], sub.data(), sub.size() * sizeof(double), hipMemcpyHostToDevice));
            h->sub_s[l] = bs;
            h->sub_n[l] = bn;
        }
    }
    {
        const size_t nL = h->hp.KL.v.size(), n0 = (size_t)d.M[0] * d.M[1] * d.M[2];
        DA(h->efour, nL); DA(h->dKL, nL);
        {
            const LevelStencil &S = h->hp.K[L - 1];
            const int *ML = h->hp.Ml[L - 1];
            if (!literal && S.size[0] == ML[0] && S.size[1] == ML[1] && S.size[2] == ML[2] && h->hp.KL.v.size() == nL) {
                // e[m] = sum_s K[s] q[m - (first + s)]: the tap at offset o = (first + s) mod M
                std::vector<double> ts(nL), tf(nL);
                for (int oz = 0; oz < ML[2]; oz++)
                    for (int oy = 0; oy < ML[1]; oy++)
                        for (int ox = 0; ox < ML[0]; ox++) {
                            const int sx = ((ox - S.first[0]) % ML[0] + ML[0]) % ML[0], sy = ((oy - S.first[1]) % ML[1] + ML[1]) % ML[1],
                                      sz = ((oz - S.first[2]) % ML[2] + ML[2]) % ML[2];
                            const size_t o = ((size_t)oz * ML[1] + oy) * ML[0] + ox;
                            tf[o] = h->hp.KL.v[o]; // KL starts at offset 0
                            ts[o] = S.v[((size_t)sz * S.size[1] + sy) * S.size[0] + sx] + tf[o];
                        }
                DA(h->dTsum, nL); DA(h->dTfour, nL);
                HIPCHK_H(hipMemcpy(h->dTsum, ts.data(), nL * sizeof(double), hipMemcpyHostToDevice));
                HIPCHK_H(hipMemcpy(h->dTfour, tf.data(), nL * sizeof(double), hipMemcpyHostToDevice));
            }
        }
        if (d.open_boundary) DA(h->open_part, (size_t)conv_open_groups(h->hp.Ml[L - 1]) * nL);
        HIPCHK_H(hipMemset(h->efour, 0, nL * sizeof(double)));
        HIPCHK_H(hipMemcpy(h->dKL, h->hp.KL.v.data(), nL * sizeof(double), hipMemcpyHostToDevice));
        DA(h->tmpA, n0 / 2 + 1); DA(h->tmpB, n0 / 4 + 1);
    }
    h->anterp_tiled = !literal && d.M[0] >= 24 && d.M[1] >= 24 && d.M[2] >= 24;
    if (h->anterp_tiled) {
        for (int k = 0; k < 3; k++) h->tiles.nt[k] = (d.M[k] + SPLINE_TILE - 1) / SPLINE_TILE;
        h->tiles.ntiles = h->tiles.nt[0] * h->tiles.nt[1] * h->tiles.nt[2];
        const int R = SPLINE_TILE + d.p - 1;
        DA(h->tile_scratch, (size_t)h->tiles.ntiles * R * R * R);
        DA(h->tile_used, h->tiles.ntiles);
    }
    h->nepart = h->small_box ? (N + 63) / 64 : h->ncell;
    DA(h->epart, std::max(h->nepart, cg.max_clusters));
    DA(h->parts, 5 * NPARTS);
    DA(h->d_energies, MSM_AMD_U_COUNT);
    DA(h->energy_ticket, 1);
    HIPCHK_H(hipMemset(h->energy_ticket, 0, sizeof(unsigned)));
    HIPCHK_H(hipHostMalloc((void **)&h->h_energies, MSM_AMD_U_COUNT * sizeof(double), hipHostMallocDefault));
#undef HIPCHK_H
#undef DA
    for (int i = 0; i < T_COUNT; i++) h->last_ms[i] = 0.0;
    *out = h;
    return MSM_AMD_OK;
}

extern "C" int msm_amd_create(const msm_amd_plan_desc *plan, int device, void *stream, msm_amd_handle **out) {
    return create_impl(plan, false, device, stream, out);
}
// literal-mode handle (msmperiodic() in literal mode and the parity tests of that mode)
extern "C" int msm_amd_create_literal(const msm_amd_plan_desc *plan, int device, void *stream, msm_amd_handle **out) {
    return create_impl(plan, true, device, stream, out);
}

static void free_reuse(msm_amd_handle *h) {
    ReuseBuffers &rs = h->rs;
    hipFree(rs.ps); hipFree(rs.ref); hipFree(rs.order); hipFree(rs.blo); hipFree(rs.bhi); hipFree(rs.fbox);
    hipFree(rs.start); hipFree(rs.clstart); hipFree(rs.colncl); hipFree(rs.list); hipFree(rs.n0); hipFree(rs.n1); hipFree(rs.n2);
    hipFree(rs.hl); hipFree(rs.hn); hipFree(rs.need); hipFree(rs.counters);
    rs = ReuseBuffers();
    h->rs_made = false;
    h->rs_skin = -1.0;
    h->rs_skin_cap = 0.0;
}

extern "C" void msm_amd_destroy(msm_amd_handle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    hipFree(h->ps_cell); hipFree(h->ps_grid); hipFree(h->order_cell); hipFree(h->order_grid); hipFree(h->bin_of);
    hipFree(h->slot_idx); hipFree(h->ccount); hipFree(h->cstart); hipFree(h->arrival); hipFree(h->gcount); hipFree(h->gstart);
    hipFree(h->scan_tmp); hipFree(h->tile_scratch); hipFree(h->tile_used);
    {
        ClusterBuffers &cb = h->clb;
        hipFree(cb.hist); hipFree(cb.coltotal); hipFree(cb.cpstart); hipFree(cb.bps); hipFree(cb.bkey);
        hipFree(cb.count); hipFree(cb.start); hipFree(cb.colncl); hipFree(cb.clstart); hipFree(cb.order);
        hipFree(cb.ps); hipFree(cb.blo); hipFree(cb.bhi); hipFree(cb.fbox); hipFree(cb.list); hipFree(cb.n0); hipFree(cb.n1); hipFree(cb.n2); hipFree(cb.n3);
    }
    for (auto p : h->gq) hipFree(p);
    for (auto p : h->ge) hipFree(p);
    for (auto p : h->dK) hipFree(p);
    for (auto p : h->dKt) hipFree(p);
    for (auto p : h->dKsub) hipFree(p);
    hipFree(h->open_part);
    hipFree(h->efour); hipFree(h->dKL); hipFree(h->dTsum); hipFree(h->dTfour); hipFree(h->tmpA); hipFree(h->tmpB); hipFree(h->epart); hipFree(h->parts);
    hipFree(h->d_energies); hipFree(h->energy_ticket);
    if (h->h_energies) hipHostFree(h->h_energies);
    hipFree(h->s_r); hipFree(h->s_q); hipFree(h->s_acc); hipFree(h->s_acc2); hipFree(h->conv_part);
    hipFree(h->lit_part_d[0]); hipFree(h->lit_part_d[1]); hipFree(h->lit_part_e);
    hipFree(h->acc_slots);
    free_reuse(h);
    if (h->stream_b) {
        hipStreamSynchronize(h->stream_b);
        hipEventDestroy(h->ev_fork); hipEventDestroy(h->ev_pairs); hipEventDestroy(h->ev_join);
        hipEventDestroy(h->ev_gfork); hipEventDestroy(h->ev_gjoin); hipEventDestroy(h->ev_energy);
        hipStreamDestroy(h->stream_b);
    }
    if (h->stream_a) {
        hipStreamSynchronize(h->stream_a);
        hipEventDestroy(h->ev_in); hipEventDestroy(h->ev_out);
        hipStreamDestroy(h->stream_a);
    }
    if (h->ev_made) { for (int i = 0; i <= T_COUNT; i++) hipEventDestroy(h->ev[i]); hipEventDestroy(h->ev_pk[0]); hipEventDestroy(h->ev_pk[1]); }
    delete h;
}

// A slab rank evaluates a different number of particles every step (owned + halo copies); the plan's N is the
// capacity the buffers were sized for.
extern "C" int msm_amd_set_counts(msm_amd_handle *h, int n_total, int n_owned) {
    if (!h || n_total < 1 || n_total > h->capacity || n_owned < 0 || n_owned > n_total) {
        set_error("set_counts: need 1 <= n_total <= capacity (%d) and 0 <= n_owned <= n_total", h ? h->capacity : 0);
        return MSM_AMD_EINVAL;
    }
    if (h->small_box && n_owned != n_total) { set_error("set_counts: halo particles need the cell-list path"); return MSM_AMD_EINVAL; }
    h->N = n_total;
    h->n_owned = n_owned;
    h->pp.n_owned = n_owned;
    h->sp.n_owned = n_owned;
    h->clg.n_owned = n_owned >= n_total ? CL_ALL_OWNED : n_owned;
    h->rs_skin = -1.0; // (lists kept by msm_amd_eval_reuse describe another set of particles: its next call rebuilds)
    return MSM_AMD_OK;
}

extern "C" int msm_amd_set_planes(msm_amd_handle *h, int zplane0, int nzplanes) {
    if (!h || zplane0 < 0 || nzplanes < 1 || zplane0 + nzplanes > h->hp.d.M[2]) {
        set_error("set_planes: need 0 <= zplane0, 1 <= nzplanes, zplane0 + nzplanes <= %d", h ? h->hp.d.M[2] : 0);
        return MSM_AMD_EINVAL;
    }
    if (h->small_box && nzplanes != h->hp.d.M[2]) { set_error("set_planes: slab decomposition needs at least 3 pair cells per axis"); return MSM_AMD_EINVAL; }
    h->hp.d.zplane0 = zplane0;
    h->hp.d.nzplanes = nzplanes;
    h->sp.z0 = zplane0;
    h->sp.nz = nzplanes;
    set_slab_faces(h);
    return MSM_AMD_OK;
}

extern "C" void msm_amd_set_profiling(msm_amd_handle *h, int on) {
    if (!h) return;
    hipSetDevice(h->device);
    if (on && !h->ev_made) {
        for (int i = 0; i <= T_COUNT; i++) hipEventCreate(&h->ev[i]);
        hipEventCreate(&h->ev_pk[0]); hipEventCreate(&h->ev_pk[1]);
        h->ev_made = true;
    }
    h->profiling = on != 0;
}

extern "C" int msm_amd_set_option(msm_amd_handle *h, const char *key, int value) {
    if (!h || !key) return MSM_AMD_EINVAL;
    if (!strcmp(key, "pair_variant")) { h->opt_pair_variant = value; return MSM_AMD_OK; }
    if (!strcmp(key, "conv_tiled")) { h->opt_conv_tiled = value; return MSM_AMD_OK; }
    if (!strcmp(key, "conv_small")) { h->opt_conv_small = value; return MSM_AMD_OK; }
    if (!strcmp(key, "anterp_tiled")) { h->opt_anterp_tiled = value; return MSM_AMD_OK; }
    if (!strcmp(key, "interp_threads")) { if (value >= 64 && value <= 512 && value % 64 == 0) { h->sopt.interp_threads = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "conv_sym")) { h->opt_conv_sym = value; return MSM_AMD_OK; }
    if (!strcmp(key, "transfer_chain")) { h->opt_transfer_chain = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "slab_level1")) { h->opt_slab_level1 = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "conv_groups")) { h->opt_conv_groups = value; return MSM_AMD_OK; } // 2: for whole grids too (A/B)
    if (!strcmp(key, "interp_remap")) { h->sopt.interp_remap = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "interp_split")) { if (value >= 0 && value <= 16) { h->sopt.interp_split = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "anterp_cols_weights")) { h->sopt.anterp_cols_weights = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "anterp_cols_waves")) { if (value == 0 || value == 4 || value == 8 || value == 16) { h->sopt.anterp_cols_waves = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "anterp_cols_order4")) { h->sopt.anterp_cols_order4 = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "anterp_cols")) { h->sopt.anterp_cols = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "energy_fused")) { h->opt_energy_fused = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "acc_slots")) { h->opt_acc_slots = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "face_cut")) { h->opt_face_cut = value != 0; set_slab_faces(h); return MSM_AMD_OK; }
    if (!strcmp(key, "overlap")) { h->opt_overlap = value; return MSM_AMD_OK; }
    if (!strcmp(key, "slab_own_stream")) { h->opt_slab_own_stream = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "energy_on_b")) { h->opt_energy_on_b = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "pair_chunks")) { if (value >= 1 && value <= 8) { h->opt_pair_chunks = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "tail_events")) { h->opt_tail_events = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "pair_chunks_coarse")) { if (value >= -1 && value <= 8) { h->opt_pair_chunks_coarse = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "conv_leads")) { h->opt_conv_leads = value != 0; return MSM_AMD_OK; }
    if (!strcmp(key, "slab_overlap")) { h->opt_slab_overlap = value; return MSM_AMD_OK; }
    if (!strcmp(key, "conv_split")) { // shares of a level-0 tile's taps: conv_part holds the partial sums of shares 1..4
        if (value < 0 || value > 5) { set_error("conv_split: 0 (off), 1 (chosen per launch) or 2..5 shares"); return MSM_AMD_EINVAL; }
        h->opt_conv_split = value;
        return MSM_AMD_OK;
    }
    if (!strcmp(key, "conv_tile_planes")) { if (value == 0 || value == 6 || (value >= 8 && value <= 10) || value == 12 || (value >= 16 && value <= 19)) { h->opt_conv_tile_planes = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "pair_lds_pad")) { if (value >= 0 && value <= 48 * 1024) { h->opt_pair_lds_pad = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "pair_wpb")) { if (value == 0 || value == 1 || value == 2 || value == 4) { h->opt_pair_wpb = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "list_wpb")) { if (value == 0 || value == 1 || value == 4) { h->opt_list_wpb = value; return MSM_AMD_OK; } }
    if (!strcmp(key, "pair_wgs")) { h->pair_wgs_now = value; return MSM_AMD_OK; } // A/B: cap on the pair kernel's workgroups
    if (!strcmp(key, "top_fused")) { h->opt_top_fused = value; return MSM_AMD_OK; }
    if (!strcmp(key, "grid_streams")) { h->opt_grid_streams = value; return MSM_AMD_OK; }
    if (!strcmp(key, "transfer_fused")) { h->opt_transfer_fused = value; return MSM_AMD_OK; }
    if (!strcmp(key, "transfer_fused_max")) { h->opt_transfer_fused_max = value; return MSM_AMD_OK; }
    set_error("unknown option %s", key);
    return MSM_AMD_EINVAL;
}

static inline void mark(msm_amd_handle *h, int slot) {
    if (h->profiling) hipEventRecord(h->ev[slot], h->stream);
}

// the tiled spline kernels read the cluster order when the columns are grid aligned and that sort was made
static bool tiles_from_clusters(const msm_amd_handle *h) {
    return h->anterp_tiled && h->opt_anterp_tiled && h->clg.kx > 0 && !h->small_box && h->clusters_sorted;
}
static void ensure_grid_binned(msm_amd_handle *h) {
    if (h->grid_binned) return;
    launch_bin(h->stream, h->cur_r, h->cur_q, h->N, h->gridg, h->bin_of, h->gcount, h->gstart, h->arrival, h->slot_idx,
               h->scan_tmp, h->order_grid, h->ps_grid);
    h->grid_binned = true;
}
static TileSrc tile_source(msm_amd_handle *h) {
    TileSrc S;
    memset(&S, 0, sizeof(S));
    if (tiles_from_clusters(h)) {
        S.ps = h->clb.ps; S.order = h->clb.order; S.start = h->clb.start; S.clstart = h->clb.clstart;
        S.cg = h->clg; S.cluster_order = 1;
        S.acc_slots = h->fused_acc ? h->acc_slots : nullptr;
        for (int k = 0; k < 3; k++) S.scale[k] = h->hp.d.M[k] / h->hp.d.A[k];
    } else {
        ensure_grid_binned(h);
        S.ps = h->ps_grid; S.order = h->order_grid; S.start = h->gstart; S.cluster_order = 0;
        for (int k = 0; k < 3; k++) S.scale[k] = 1.0;
    }
    return S;
}

extern "C" int msm_amd_stage_bin(msm_amd_handle *h, const double *d_r, const double *d_q) {
    if (!h || !d_r || !d_q) { set_error("stage_bin: null argument"); return MSM_AMD_EINVAL; }
    h->cur_r = d_r;
    h->cur_q = d_q;
    // the sort the selected short-range kernel needs (stage_pairs redoes it if the variant is changed in between)
    h->cells_binned = false;
    h->clusters_sorted = false;
    if (!h->small_box) {
        if (h->opt_pair_variant >= 5 && h->pp.convention != MSM_AMD_ACC_LITERAL) {
            launch_cluster_sort(h->stream, d_r, d_q, h->N, h->clg, h->clb, h->bin_of, h->arrival, h->slot_idx, h->scan_tmp);
            h->clusters_sorted = true;
        } else {
            launch_bin(h->stream, d_r, d_q, h->N, h->cellg, h->bin_of, h->ccount, h->cstart, h->arrival, h->slot_idx, h->scan_tmp,
                       h->order_cell, h->ps_cell);
            h->cells_binned = true;
        }
    }
    // the grid-cell order is needed unless the tile kernels can walk the cluster order
    h->grid_binned = false;
    if (!tiles_from_clusters(h)) ensure_grid_binned(h);
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

extern "C" int msm_amd_stage_pairs(msm_amd_handle *h, double *d_acc) {
    if (!h || !d_acc || !h->cur_r) { set_error("stage_pairs: bin first"); return MSM_AMD_EINVAL; }
    if (h->small_box) {
        int n = 0;
        launch_pair_allimages(h->stream, h->cur_r, h->cur_q, h->N, h->pp, d_acc, h->epart, &n);
        h->nepart = n;
    } else if (h->opt_pair_variant >= 5 && h->pp.convention != MSM_AMD_ACC_LITERAL) {
        // cluster pairs (8: lists walked per particle, 5: per cluster, 6: list-free, 7: no one-sided energies); the unweighted sums of the literal convention exist
        // only in the cell kernels
        if (!h->clusters_sorted)
            launch_cluster_sort(h->stream, h->cur_r, h->cur_q, h->N, h->clg, h->clb, h->bin_of, h->arrival, h->slot_idx, h->scan_tmp);
        h->clusters_sorted = true;
        ClusterBuffers cb = h->clb;
        cb.acc_slots = h->fused_acc ? h->acc_slots : nullptr;
        launch_pair_clusters(h->stream, h->clg, cb, h->pp, d_acc, h->epart, h->opt_pair_variant - 5, h->N,
                             h->profiling ? h->ev_pk : nullptr, h->pair_wgs_now, h->pairs_after, h->opt_pair_lds_pad | (h->opt_pair_wpb << 20) | (h->opt_list_wpb << 24),
                             h->pairs_between ? &h->pairs_between : nullptr, h->pairs_chunks_now,
                             h->pairs_before_last ? &h->pairs_before_last : nullptr);
        h->pairs_after = nullptr;
        h->pairs_between = nullptr;
        h->pairs_before_last = nullptr;
        h->pk_recorded = h->profiling && h->opt_pair_variant != 6;
        h->nepart = h->clg.max_clusters;
    } else {
        if (!h->cells_binned)
            launch_bin(h->stream, h->cur_r, h->cur_q, h->N, h->cellg, h->bin_of, h->ccount, h->cstart, h->arrival, h->slot_idx,
                       h->scan_tmp, h->order_cell, h->ps_cell);
        h->cells_binned = true;
        launch_pair_cells(h->stream, h->ps_cell, h->cstart, h->order_cell, h->ncell, h->pp, d_acc, h->epart,
                          h->opt_pair_variant >= 5 ? 1 : h->opt_pair_variant);
        h->nepart = h->ncell;
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

extern "C" int msm_amd_stage_anterpolate(msm_amd_handle *h) {
    if (!h || !h->cur_r) { set_error("stage_anterpolate: bin first"); return MSM_AMD_EINVAL; }
    if (!(h->anterp_tiled && h->opt_anterp_tiled &&
          launch_anterp_tiled(h->stream, tile_source(h), h->sp, h->tiles, h->tile_scratch, h->tile_used, h->gq[0], h->sopt))) {
        ensure_grid_binned(h);
        launch_anterp(h->stream, h->ps_grid, h->gstart, h->sp, h->gq[0]);
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

// thread-per-node for big outputs, wave-per-node when the grid is too small to fill the chip that way
static void conv_untiled(msm_amd_handle *h, const double *q, const double *K, const int M[3], const int ks[3],
                         const int kf[3], double *e, int z0, int nz) {
    const long nout = (long)M[0] * M[1] * nz, taps = (long)ks[0] * ks[1] * ks[2];
    if (h->opt_conv_small && nout <= 65536 && taps >= 256) launch_conv_small(h->stream, q, K, M, ks, kf, e, z0, nz, 0);
    else launch_conv_direct(h->stream, q, K, M, ks, kf, e, z0, nz, 0);
}

static void convolve_level(msm_amd_handle *h, int l, const double *q, double *e, int z0, int nz) {
    const LevelStencil &S = h->hp.K[l];
    if (h->hp.literal) {
        launch_conv_literal(h->stream, q, h->dK[l], h->hp.Ml[l], S.size, S.first, e);
        return;
    }
    // the tiled kernel pays off once the grid is much larger than the stencil
    const bool big = h->hp.Ml[l][0] >= 16 && h->hp.Ml[l][1] >= 8 && nz >= 8;
    if (h->opt_conv_tiled && big && launch_conv_tiled(h->stream, q, h->dKt[l], h->hp.Ml[l], S.size, S.first, e, z0, nz)) return;
    conv_untiled(h, q, h->dK[l], h->hp.Ml[l], S.size, S.first, e, z0, nz);
}

// Level transfers l <-> l+1.  One fused kernel when the grids are small (launch-bound), three separable passes
// through the scratch grids when they are large; both give the same bits.
static bool transfer_fused(const msm_amd_handle *h, int l) {
    const long n = (long)h->hp.Ml[l][0] * h->hp.Ml[l][1] * h->hp.Ml[l][2];
    return h->opt_transfer_fused == 1 || (h->opt_transfer_fused == 2 && n <= h->opt_transfer_fused_max);
}
// zc0, nzc: coarse planes to produce (nzc <= 0: all of them); a plane range needs the tiled kernel
static void restrict_level(msm_amd_handle *h, int l, const double *fine, double *coarse, int zc0 = 0, int nzc = 0) {
    const int p = h->hp.d.p;
    int m0[3] = {h->hp.Ml[l][0], h->hp.Ml[l][1], h->hp.Ml[l][2]};
    if (nzc <= 0) { zc0 = 0; nzc = m0[2] / 2; }
    if (h->opt_transfer_fused >= 3 || nzc < m0[2] / 2) { launch_restrict_tiled(h->stream, fine, coarse, m0, p, h->hp.J, zc0, nzc); return; }
    if (transfer_fused(h, l)) { launch_restrict3(h->stream, fine, coarse, m0, p, h->hp.J); return; }
    launch_restrict_axis(h->stream, fine, h->tmpA, m0, 0, p, h->hp.J);
    int m1[3] = {m0[0] / 2, m0[1], m0[2]};
    launch_restrict_axis(h->stream, h->tmpA, h->tmpB, m1, 1, p, h->hp.J);
    int m2[3] = {m0[0] / 2, m0[1] / 2, m0[2]};
    launch_restrict_axis(h->stream, h->tmpB, coarse, m2, 2, p, h->hp.J);
}
// zf0, nzf: fine planes to add to (nzf <= 0: all of them); a plane range needs the tiled kernel
static void prolong_level_add(msm_amd_handle *h, int l, const double *coarse, double *fine, int zf0 = 0, int nzf = 0) {
    const int p = h->hp.d.p;
    if (nzf <= 0 || nzf >= h->hp.Ml[l][2]) { zf0 = 0; nzf = h->hp.Ml[l][2]; }
    if (h->opt_transfer_fused == 4 || nzf < h->hp.Ml[l][2]) {
        int m0[3] = {h->hp.Ml[l][0], h->hp.Ml[l][1], h->hp.Ml[l][2]};
        launch_prolong_tiled_add(h->stream, coarse, fine, m0, p, h->hp.J, zf0, nzf);
        return;
    }
    if (transfer_fused(h, l)) {
        int m0[3] = {h->hp.Ml[l][0], h->hp.Ml[l][1], h->hp.Ml[l][2]};
        launch_prolong3_add(h->stream, coarse, fine, m0, p, h->hp.J);
        return;
    }
    int c0[3] = {h->hp.Ml[l + 1][0], h->hp.Ml[l + 1][1], h->hp.Ml[l + 1][2]};
    launch_prolong_axis(h->stream, coarse, h->tmpB, c0, 2, p, h->hp.J, 0);
    int c1[3] = {c0[0], c0[1], c0[2] * 2};
    launch_prolong_axis(h->stream, h->tmpB, h->tmpA, c1, 1, p, h->hp.J, 0);
    int c2[3] = {c0[0], c0[1] * 2, c0[2] * 2};
    launch_prolong_axis(h->stream, h->tmpA, fine, c2, 0, p, h->hp.J, 1);
}

// prolongations from the coarsest level down to level 1; two at a time where the grids allow it (the potentials of the
// level in the middle then stay without the prolonged part, which nothing reads)
static void prolong_to_level1(msm_amd_handle *h) {
    const int L = h->hp.d.L;
    int l = L - 2;
    while (l >= 1) {
        const int *Mf = h->hp.Ml[l - (l >= 2 ? 1 : 0)];
        const bool chain = h->opt_transfer_chain && l >= 2 && !h->hp.literal && h->opt_transfer_fused >= 3 && Mf[0] % 4 == 0 &&
                           Mf[1] % 4 == 0 && Mf[2] % 4 == 0;
        if (chain) {
            int m[3] = {Mf[0], Mf[1], Mf[2]};
            launch_prolong_chain_add(h->stream, h->ge[l + 1], h->ge[l], h->ge[l - 1], m, h->hp.d.p, h->hp.J);
            l -= 2;
        } else {
            prolong_level_add(h, l, h->ge[l + 1], h->ge[l]);
            l -= 1;
        }
    }
}

namespace {
struct PlaneRange { int l, z0, nz; };
}

// convolutions of the given plane ranges on the handle's current stream: the large ones share one launch of the
// multi-level tiled kernel, the rest go one by one
static void convolve_ranges(msm_amd_handle *h, const std::vector<PlaneRange> &ranges) {
    hipStream_t st = h->stream;
    std::vector<bool> done(ranges.size(), false);
    if (!h->hp.literal && h->opt_conv_tiled >= 2) {
        ConvBatch B;
        B.n = 0;
        B.split = h->opt_conv_groups;
        B.sym = h->opt_conv_sym;
        B.tile_planes = h->opt_conv_tile_planes;
        std::vector<size_t> members;
        // A slab rank convolves a few planes of level 0: fewer tiles (16 x 8 x 16 outputs each) than the chip has CUs,
        // so the kernel lasts as long as one workgroup needs for all taps of its tile.  The taps are then shared out
        // over `split` workgroups per tile (by rows sy of the stencil), which leave partial sums that are added
        // in a fixed order afterwards.
        int split = 1;
        {
            long tiles0 = 0, entries = 0;
            bool splittable = true;
            for (const PlaneRange &r : ranges) {
                const int *M = h->hp.Ml[r.l];
                if (!(M[0] >= 16 && M[1] >= 8 && M[2] >= 8)) continue;
                entries++;
                if (r.l == 0) {
                    tiles0 += (long)((M[0] + 15) / 16) * ((M[1] + 7) / 8) * ((r.nz + 15) / 16);
                    if (h->sub_n[0]) splittable = false;
                }
            }
            const long n0 = std::count_if(ranges.begin(), ranges.end(), [](const PlaneRange &r) { return r.l == 0; });
            if (h->opt_conv_split && splittable && tiles0 > 0 && tiles0 < 200 && h->hp.d.nzplanes < h->hp.d.M[2]) {
                split = std::min(CONV_SPLIT_MAX, h->opt_conv_split > 1 ? h->opt_conv_split : (tiles0 < 100 ? 5 : 3));
                while (split > 1 && entries + n0 * (split - 1) > 8) split--;
            }
        }
        const size_t n0grid = (size_t)h->hp.d.M[0] * h->hp.d.M[1] * h->hp.d.M[2];
        if (split > 1 && !h->conv_part) {
            if (dalloc(&h->conv_part, (size_t)(CONV_SPLIT_MAX - 1) * n0grid)) { split = 1; (void)hipGetLastError(); }
        }
        // A range that runs past the last plane comes as two entries, [lo, Mz) and [0, rest): the tiled kernels take it as
        // one, [lo, lo + n) modulo Mz (two entries are two sets of tiles, and the short one fills whole tiles with a few
        // planes: 72 workgroups more for the 3 planes of rank 0 of 8, which the rank in the middle does not have).
        std::vector<PlaneRange> joined;
        std::vector<std::pair<size_t, size_t>> source; // entries of `ranges` a joined entry stands for
        for (size_t i = 0; i < ranges.size(); i++) {
            const PlaneRange &r = ranges[i];
            if (split == 1 && i + 1 < ranges.size() && ranges[i + 1].l == r.l && ranges[i + 1].z0 == 0 &&
                r.z0 + r.nz == h->hp.Ml[r.l][2] && r.nz + ranges[i + 1].nz <= h->hp.Ml[r.l][2]) {
                joined.push_back({r.l, r.z0, r.nz + ranges[i + 1].nz});
                source.push_back({i, i + 1});
                i++;
            } else {
                joined.push_back(r);
                source.push_back({i, i});
            }
        }
        for (size_t i = 0; i < joined.size() && B.n < 8; i++) {
            const PlaneRange &r = joined[i];
            const LevelStencil &S = h->hp.K[r.l];
            const int *M = h->hp.Ml[r.l];
            if (!(M[0] >= 16 && M[1] >= 8 && M[2] >= 8)) continue; // a short plane range of a large grid is fine
            const int width = h->sub_n[r.l] ? h->sub_s[r.l] : S.size[2]; // taps per axis the kernel is instantiated for
            if (B.n > 0 && width != B.c[0].ks[2]) continue;
            const int parts = (r.l == 0 && split > 1) ? split : 1;
            if (B.n + parts > 8) continue;
            for (int part = 0; part < parts; part++) {
                ConvArgs &c = B.c[B.n];
                for (int k = 0; k < 3; k++) { c.M[k] = M[k]; c.ks[k] = h->sub_n[r.l] ? width : S.size[k]; c.kf[k] = S.first[k]; }
                c.z0 = r.z0; c.nz = r.nz; c.accumulate = 0;
                c.nsub = h->sub_n[r.l] ? h->sub_n[r.l] : 1;
                c.sy0 = 0; c.sy1 = 0;
                if (parts > 1) { c.sy0 = width * part / parts; c.sy1 = width * (part + 1) / parts; }
                B.q[B.n] = h->gq[r.l]; B.Kt[B.n] = h->sub_n[r.l] ? h->dKsub[r.l] : h->dKt[r.l];
                B.e[B.n] = part == 0 ? h->ge[r.l] : h->conv_part + (size_t)(part - 1) * n0grid;
                B.n++;
            }
            members.push_back(i);
        }
        if (B.n > 0 && launch_conv_multi(st, B)) {
            for (size_t i : members) {
                done[source[i].first] = done[source[i].second] = true;
                const PlaneRange &r = joined[i];
                if (r.l == 0 && split > 1) {
                    // e = ((part 0 + part 1) + part 2) + ..., two partial grids per pass
                    const size_t plane = (size_t)h->hp.d.M[0] * h->hp.d.M[1], off = plane * r.z0, n = plane * r.nz;
                    for (int part = 1; part < split; part += 2)
                        launch_add_parts(st, h->ge[0] + off, h->conv_part + (size_t)(part - 1) * n0grid + off,
                                         part + 1 < split ? h->conv_part + (size_t)part * n0grid + off : nullptr, n);
                }
            }
        }
    }
    for (size_t i = 0; i < ranges.size(); i++)
        if (!done[i]) convolve_level(h, ranges[i].l, h->gq[ranges[i].l], h->ge[ranges[i].l], ranges[i].z0, ranges[i].nz);
}

static bool top_fused(const msm_amd_handle *h) {
    const msm_amd_plan_desc &d = h->hp.d;
    // (a slab rank of a one-level plan convolves only some planes of level 0 = the coarsest level: not fused)
    return h->opt_top_fused && h->dTsum && (d.L > 1 || d.nzplanes >= d.M[2]);
}

// Fourier (or, literal mode, full-period) top level on the coarsest grid
static void top_level(msm_amd_handle *h) {
    const int L = h->hp.d.L;
    const int *ML = h->hp.Ml[L - 1];
    const size_t nL = (size_t)ML[0] * ML[1] * ML[2];
    if (h->hp.d.open_boundary) {
        // every node of the coarsest grid against every node, no wrap-around; there is no Fourier level (efour stays 0)
        launch_conv_open(h->stream, h->gq[L - 1], h->dK[L - 1], ML, h->ge[L - 1], h->open_part);
    } else if (top_fused(h)) {
        // the coarsest level's own convolution was left out of the coarse ranges: both in one pass
        launch_conv_small_pair(h->stream, h->gq[L - 1], h->dTsum, h->dTfour, ML, h->ge[L - 1], h->efour);
    } else if (h->hp.literal) {
        // sL spans one full period on every axis (msmPeriodic.c:344-391), same index arithmetic as above
        launch_conv_literal(h->stream, h->gq[L - 1], h->dKL, ML, h->hp.KL.size, h->hp.KL.first, h->efour);
        // the reference then overwrites elng with edir (msmPeriodic.c:1006-1011): nothing is added
    } else {
        conv_untiled(h, h->gq[L - 1], h->dKL, ML, h->hp.KL.size, h->hp.KL.first, h->efour, 0, ML[2]);
        launch_axpy(h->stream, h->ge[L - 1], h->efour, nL);
    }
}

// The grid stage.  front_done: a slab rank in halo mode - level 0 holds the owned charge planes and the halo planes
// received from the z neighbours only, and grid_front() has already produced the level-1 charges; the coarser levels
// are solved by every rank, being too small to shard.
// First part of the grid stage of a slab rank in halo mode, queued on the handle's current stream: coarse planes under
// the owned ones from the owned + halo planes of level 0, then the all-gather that completes the level-1 charge grid.
static int grid_front(msm_amd_handle *h, const msm_amd_collectives *coll) {
    const msm_amd_plan_desc &d = h->hp.d;
    const size_t plane1 = (size_t)h->hp.Ml[1][0] * h->hp.Ml[1][1];
    restrict_level(h, 0, h->gq[0], h->gq[1], d.zplane0 / 2, d.nzplanes / 2);
    if (coll->all_gather(coll->ctx, h->gq[1] + plane1 * (d.zplane0 / 2), h->gq[1], plane1 * (d.nzplanes / 2), (void *)h->stream)) {
        set_error("slab step: the all-gather of the level-1 charges failed");
        return MSM_AMD_EHIP;
    }
    return MSM_AMD_OK;
}

static int ensure_stream_b(msm_amd_handle *h);
// part: 0 the whole stage; 1 only the branch of the coarser levels (restrictions, their convolutions, top level,
// prolongations down to level 1); 2 only what is left after it (level-0 convolution, last prolongation); 3 and 4: the
// two halves of 2, the level-0 convolution (which needs no coarser level) and the last prolongation
static int grid_stage(msm_amd_handle *h, bool front_done, int part = 0) {
    const msm_amd_plan_desc &d = h->hp.d;
    const int L = d.L, p = d.p;
    hipStream_t st = h->stream;
    const bool slab = d.nzplanes < d.M[2];
    // plane ranges to convolve on level 0: everything, except on a slab rank, where the potentials are needed on the
    // owned planes plus p/2 planes either side (interpolation support) and the other planes of ge[0] are not computed
    std::vector<PlaneRange> fine, coarse;
    if (d.nzplanes < d.M[2]) {
        const int hz = p / 2, Mz = d.M[2];
        int lo = d.zplane0 - hz, n = d.nzplanes + 2 * hz;
        // (the planes of ge[0] outside that range keep whatever they held: nothing reads them)
        if (n >= Mz) fine.push_back({0, 0, Mz});
        else {
            lo = ((lo % Mz) + Mz) % Mz;
            const int first = std::min(n, Mz - lo);
            fine.push_back({0, lo, first});
            if (n > first) fine.push_back({0, 0, n - first});
        }
    } else {
        fine.push_back({0, 0, d.M[2]});
    }
    // the coarsest level's convolution belongs to top_level() when it is fused with the Fourier level or open
    const bool fused_top = top_fused(h) || d.open_boundary;
    for (int l = 1; l < L; l++) {
        if (fused_top && l == L - 1) continue;
        const int M1 = h->hp.Ml[l][2];
        if (slab && l == 1 && h->opt_slab_level1) {
            // Level 1 of a slab rank: its potentials are read by the last prolongation only, for the fine planes
            // [z0 - p/2, z0 + nz + p/2), i.e. on the coarse planes below them and p/2 either side (one more for the
            // rounding of the halves).  The charges of level 1 are complete on every rank, so the rank convolves these
            // planes by itself; the other planes of ge[1] keep whatever they held (the prolongation from level 2 is
            // added to them as well, nothing reads them).  At 256^3 nodes and 8 ranks that is 26 planes of 128.
            const int flo = d.zplane0 - p / 2, fhi = d.zplane0 + d.nzplanes + p / 2 - 1;
            const int clo = (flo >= 0 ? flo / 2 : -((1 - flo) / 2)) - p / 2 - 1, chi = (fhi >= 0 ? fhi / 2 : -((1 - fhi) / 2)) + p / 2 + 1;
            const int n = chi - clo + 1;
            if (n < M1) {
                const int lo = ((clo % M1) + M1) % M1, first = std::min(n, M1 - lo);
                coarse.push_back({l, lo, first});
                if (n > first) coarse.push_back({l, 0, n - first});
                continue;
            }
        }
        coarse.push_back({l, 0, M1});
    }
    if (fused_top && L == 1) fine.clear();

    // Two branches that only meet at the last prolongation: the level-0 convolution (most of the arithmetic, fills the
    // chip) and everything on the coarser levels (a chain of small, latency-bound kernels: restrictions,
    // convolutions of levels >= 1, top level, prolongations down to level 1).  On two streams the chain runs in the
    // shadow of the large kernel; the arithmetic and its order are the same either way.
    if (part && L >= 2) {
        if (part == 1) {
            mark(h, T_RES);
            for (int l = front_done ? 1 : 0; l + 1 < L; l++) restrict_level(h, l, h->gq[l], h->gq[l + 1]);
            convolve_ranges(h, coarse);
            top_level(h);
            prolong_to_level1(h);
        } else {
            if (part != 4) {
                mark(h, T_CONV);
                convolve_ranges(h, fine);
            }
            if (part != 3) {
                mark(h, T_TOP);
                mark(h, T_PRO);
                if (slab && d.nzplanes + p < d.M[2]) prolong_level_add(h, 0, h->ge[1], h->ge[0], d.zplane0 - p / 2, d.nzplanes + p);
                else prolong_level_add(h, 0, h->ge[1], h->ge[0]);
            }
        }
        HIPCHK(hipGetLastError());
        return MSM_AMD_OK;
    }
    if (part == 1) return MSM_AMD_OK; // a one-level plan has no coarse branch
    // (A/B option, off by default: measured slower.)  The branch goes to the handle's second stream, so it is only taken
    // when the stage itself is not already running there.
    const bool two = h->opt_grid_streams && !h->hp.literal && L >= 2 && st != h->stream_b;
    if (two) { int rc = ensure_stream_b(h); if (rc) return rc; }
    mark(h, T_RES);
    if (two) {
        HIPCHK(hipEventRecord(h->ev_gfork, st));
        HIPCHK(hipStreamWaitEvent(h->stream_b, h->ev_gfork, 0));
        h->stream = h->stream_b; // the helpers below enqueue on the handle's stream
    }
    for (int l = front_done ? 1 : 0; l + 1 < L; l++) {
        // (the event behind the restrictions is signalled by the last of them itself where that is a tiled kernel)
        if (l + 2 == L && h->ev_restricted && h->opt_tail_events) msm::tail_event = h->ev_restricted;
        restrict_level(h, l, h->gq[l], h->gq[l + 1]);
    }
    if (h->ev_restricted) {
        const bool signalled = h->opt_tail_events && msm::tail_event == nullptr && L - (front_done ? 1 : 0) >= 2;
        msm::tail_event = nullptr;
        hipError_t e = signalled ? hipSuccess : hipEventRecord(h->ev_restricted, h->stream);
        h->ev_restricted = nullptr;
        if (e != hipSuccess) { set_error("hipEventRecord failed: %s", hipGetErrorString(e)); return MSM_AMD_EHIP; }
    }
    if (!two) {
        mark(h, T_CONV);
        std::vector<PlaneRange> all(fine);
        all.insert(all.end(), coarse.begin(), coarse.end());
        convolve_ranges(h, all); // one stream: the levels share a launch
        mark(h, T_TOP);
    } else {
        convolve_ranges(h, coarse);
    }
    top_level(h);
    if (!two) mark(h, T_PRO);
    prolong_to_level1(h);
    if (two) {
        hipError_t e = hipEventRecord(h->ev_gjoin, h->stream_b);
        h->stream = st;
        if (e != hipSuccess) { set_error("hipEventRecord failed: %s", hipGetErrorString(e)); return MSM_AMD_EHIP; }
        mark(h, T_CONV);
        convolve_ranges(h, fine);
        mark(h, T_TOP); // with two streams: what is left of the coarse branch once the level-0 convolution is done
        HIPCHK(hipStreamWaitEvent(st, h->ev_gjoin, 0));
        mark(h, T_PRO);
    }
    if (L >= 2) {
        // (an event behind the stage is signalled by this kernel itself where it is the tiled one)
        if (h->ev_stage_tail && h->opt_tail_events) msm::tail_event = h->ev_stage_tail;
        // a slab rank reads the potentials of its own planes and p/2 planes either side only
        if (slab && d.nzplanes + p < d.M[2]) prolong_level_add(h, 0, h->ge[1], h->ge[0], d.zplane0 - p / 2, d.nzplanes + p);
        else prolong_level_add(h, 0, h->ge[1], h->ge[0]);
        if (h->ev_stage_tail && h->opt_tail_events && msm::tail_event == nullptr) h->ev_stage_tail = nullptr; // taken
        msm::tail_event = nullptr;
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

extern "C" int msm_amd_stage_grid(msm_amd_handle *h) {
    if (!h) return MSM_AMD_EINVAL;
    return grid_stage(h, false);
}

extern "C" int msm_amd_stage_interpolate(msm_amd_handle *h, double *d_acc) {
    if (!h || !d_acc || !h->cur_r) { set_error("stage_interpolate: bin first"); return MSM_AMD_EINVAL; }
    // literal: grid index = node + p/2 - 1 (msmPeriodic.c:1136-1151)
    const int shift = h->hp.literal ? h->hp.d.p / 2 - 1 : 0;
    if (h->anterp_tiled && h->opt_anterp_tiled) {
        launch_interp_tiled(h->stream, tile_source(h), h->sp, h->tiles, h->ge[0], d_acc, h->sopt);
    } else {
        ensure_grid_binned(h);
        launch_interp(h->stream, h->ps_grid, h->order_grid, h->N, h->sp, h->ge[0], shift, d_acc);
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

static int queue_energies(msm_amd_handle *h, bool raw = false) {
    if (h->energy_on_b) { // the sums of the last slab step may still be with the second stream: they are about to be overwritten
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_energy, 0));
        h->energy_on_b = false;
    }
    const msm_amd_plan_desc &d = h->hp.d;
    const int L = d.L;
    const size_t plane = (size_t)d.M[0] * d.M[1];
    const size_t nL = (size_t)h->hp.Ml[L - 1][0] * h->hp.Ml[L - 1][1] * h->hp.Ml[L - 1][2];
    double *pg = h->parts, *pf = h->parts + NPARTS, *pc = h->parts + 2 * NPARTS;
    EnergyPartialsArgs pa;
    pa.g_q = h->gq[0] + plane * d.zplane0; pa.g_e = h->ge[0] + plane * d.zplane0; pa.g_n = plane * d.nzplanes; pa.g_parts = pg;
    pa.f_q = h->gq[L - 1]; pa.f_e = h->efour; pa.f_n = nL; pa.f_parts = pf;
    pa.c_q = h->cur_q; pa.c_n = (size_t)h->n_owned; pa.c_parts = pc;
    pa.p_in = h->epart; pa.p_n = h->nepart; pa.p_parts = nullptr;
    EnergyInputs in;
    in.pair_parts = h->epart; in.n_pair = h->nepart;
    if (h->nepart > 8 * NPARTS) { // one block sums these lists: shorten a long one first
        pa.p_parts = h->parts + 4 * NPARTS;
        in.pair_parts = h->parts + 4 * NPARTS; in.n_pair = NPARTS;
    }
    in.grid_parts = pg; in.n_grid = NPARTS;
    in.four_parts = pf; in.n_four = NPARTS;
    in.charge_parts = pc; in.n_charge = NPARTS;
    in.self_image_sum = h->hp.self_image_sum;
    in.csr = h->hp.csr;
    in.gamma0_over_a = h->hp.gamma0 / d.a;
    in.literal = h->hp.literal ? 1 : 0;
    if (h->opt_energy_fused) {
        launch_energy_all(h->stream, pa, NPARTS, in, raw, h->d_energies, raw ? nullptr : h->energies_alias, h->energy_ticket);
    } else {
        launch_energy_partials(h->stream, pa, NPARTS);
        if (raw) launch_raw_sums(h->stream, in, h->d_energies);
        else launch_finalize_energies(h->stream, in, h->d_energies, h->energies_alias);
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

// [pair sum, q0.e0 over the owned planes, qL.efour, sum q, sum q^2 (owned particles)] on the device
// (msm_amd_raw_sums_device_ptr); host_out5 != NULL additionally copies them back (synchronises).
extern "C" int msm_amd_stage_raw_sums(msm_amd_handle *h, double *host_out5) {
    if (!h || !h->cur_q) { set_error("stage_raw_sums: bin first"); return MSM_AMD_EINVAL; }
    int rc = queue_energies(h, true);
    if (rc) return rc;
    if (host_out5) {
        HIPCHK(hipMemcpyAsync(h->h_energies, h->d_energies, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(host_out5, h->h_energies, 5 * sizeof(double));
    }
    return MSM_AMD_OK;
}
extern "C" void *msm_amd_raw_sums_device_ptr(msm_amd_handle *h) { return h ? h->d_energies : nullptr; }

// (A)..(F) and the total from raw sums that were added over all slab ranks (host arithmetic, no device work)
extern "C" int msm_amd_combine_energies(msm_amd_handle *h, const double sums[5], double en[MSM_AMD_U_COUNT]) {
    if (!h || !sums || !en) return MSM_AMD_EINVAL;
    const double ulf = 0.5 * sums[2], ulr = 0.5 * sums[1] - ulf;
    en[MSM_AMD_U_SHORT_REAL] = sums[0];
    en[MSM_AMD_U_SHORT_SELF] = 0.5 * sums[4] * h->hp.self_image_sum;
    en[MSM_AMD_U_SHORT_CSR] = -0.5 * sums[3] * sums[3] * h->hp.csr;
    en[MSM_AMD_U_LONG_REAL] = ulr;
    en[MSM_AMD_U_LONG_FOUR] = ulf;
    en[MSM_AMD_U_LONG_SELF] = -0.5 * sums[4] * h->hp.gamma0 / h->hp.d.a;
    en[MSM_AMD_U_TOTAL] = (ulf + ulr + en[MSM_AMD_U_LONG_SELF]) + (sums[0] + en[MSM_AMD_U_SHORT_SELF] + en[MSM_AMD_U_SHORT_CSR]);
    return MSM_AMD_OK;
}

extern "C" int msm_amd_stage_energies(msm_amd_handle *h, double *energies) {
    if (!h || !h->cur_q) { set_error("stage_energies: bin first"); return MSM_AMD_EINVAL; }
    int rc = queue_energies(h);
    if (rc) return rc;
    if (energies) {
        HIPCHK(hipMemcpyAsync(h->h_energies, h->d_energies, MSM_AMD_U_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(energies, h->h_energies, MSM_AMD_U_COUNT * sizeof(double));
    }
    return MSM_AMD_OK;
}

// The pair branch (list build + pair kernel) is bound by VALU issue, the grid branch (anterpolation, level
// transfers, convolutions) mostly by latency and LDS, and they only meet at the interpolation, which adds to the
// accelerations the pair kernel wrote.  So after the sort the grid branch goes to a second stream:
//   A: sort -> lists -> pairs ---------------------\
//   B:      \-> anterpolation -> grids -> wait(pairs) -> interpolation -> energies -> join A
// Needs the single-sort layout (cluster order feeding the tile kernels).
static bool can_overlap(const msm_amd_handle *h) {
    return h->opt_overlap && !h->small_box && !h->hp.literal && h->clg.kx > 0 && h->anterp_tiled && h->opt_anterp_tiled &&
           h->opt_pair_variant >= 5 && h->pp.convention != MSM_AMD_ACC_LITERAL;
}

// The handle's second stream, created once.  Non-blocking and without a priority, like stream_a: every dependency
// between the streams of a handle is one of the events of this file, recorded before it is waited for, and no wait in
// the middle of an evaluation involves the default stream (which synchronises implicitly with every blocking stream of
// the process, whoever made it).  A stream priority is deliberately not offered: the hardware queues of a priority
// level come out of a pool of their own, so a priority stream changes which streams of the process share a queue, and
// a two-stream step with a high-priority second stream was seen to leave the GPU busy for good in a process that also
// held the streams of a communication library.
static int ensure_stream_b(msm_amd_handle *h) {
    if (h->stream_b) return MSM_AMD_OK;
    HIPCHK(hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&h->ev_pairs));
    HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_gfork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_gjoin, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_energy, hipEventDisableTiming));
    return MSM_AMD_OK;
}

// A whole evaluation whose pair sum and interpolation both walk the cluster order passes the short-range accelerations
// from the one to the other in slot order (see msm_amd_handle::acc_slots).
namespace {
struct FusedAcc {
    msm_amd_handle *h;
    explicit FusedAcc(msm_amd_handle *hh) : h(hh) {
        h->fused_acc = h->opt_acc_slots && h->acc_slots && !h->small_box && !h->hp.literal && h->anterp_tiled && h->opt_anterp_tiled &&
                       h->clg.kx > 0 && h->opt_pair_variant >= 5 && h->pp.convention != MSM_AMD_ACC_LITERAL;
    }
    ~FusedAcc() { h->fused_acc = false; }
};
} // namespace

extern "C" int msm_amd_eval(msm_amd_handle *h, const double *d_r, const double *d_q, double *d_acc, double *energies) {
    if (!h || !d_r || !d_q || !d_acc) { set_error("msm_amd_eval: null argument"); return MSM_AMD_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    h->pk_recorded = false;
    FusedAcc fused(h);
    const bool overlap = can_overlap(h);
    if (overlap && (rc = ensure_stream_b(h))) return rc;
    mark(h, T_BIN);
    if ((rc = msm_amd_stage_bin(h, d_r, d_q))) return rc;
    if (overlap && h->opt_overlap >= 2) {
        // The anterpolation (bound by the LDS port) beside the pair sum (bound by the FP64 pipe); 3: and the branch of
        // the coarser grid levels (latency-bound kernels); 4: as 3, but the second stream starts when the neighbour lists
        // are built (beside the list build the anterpolation takes twice its time).
        const bool coarse_too = h->opt_overlap >= 3, after_lists = h->opt_overlap == 4;
        hipStream_t a = h->stream;
        if (!after_lists) HIPCHK(hipEventRecord(h->ev_fork, a));
        mark(h, T_PAIR);
        if (after_lists) h->pairs_after = h->ev_fork;
        rc = msm_amd_stage_pairs(h, d_acc);
        const bool forked = !after_lists || h->pairs_after == nullptr; // (a path without lists did not record the event)
        h->pairs_after = nullptr;
        if (rc) return rc;
        if (!forked) HIPCHK(hipEventRecord(h->ev_fork, a));
        HIPCHK(hipEventRecord(h->ev_pairs, a));
        HIPCHK(hipStreamWaitEvent(h->stream_b, h->ev_fork, 0));
        h->stream = h->stream_b;
        mark(h, T_ANT);
        rc = msm_amd_stage_anterpolate(h);
        if (!rc && coarse_too) rc = grid_stage(h, false, 1);
        hipError_t e = hipEventRecord(h->ev_join, h->stream_b);
        h->stream = a;
        if (rc) return rc;
        if (e != hipSuccess) return MSM_AMD_EHIP;
        HIPCHK(hipStreamWaitEvent(a, h->ev_join, 0));
        if ((rc = coarse_too ? grid_stage(h, false, 2) : msm_amd_stage_grid(h))) return rc; // marks T_RES..T_PRO
        mark(h, T_INT);
        if ((rc = msm_amd_stage_interpolate(h, d_acc))) return rc;
        mark(h, T_RED);
        if ((rc = queue_energies(h))) return rc;
        mark(h, T_TOTAL);
    } else if (overlap) {
        hipStream_t a = h->stream;
        HIPCHK(hipEventRecord(h->ev_fork, a));
        HIPCHK(hipStreamWaitEvent(h->stream_b, h->ev_fork, 0));
        mark(h, T_PAIR);
        if ((rc = msm_amd_stage_pairs(h, d_acc))) return rc;
        HIPCHK(hipEventRecord(h->ev_pairs, a));
        h->stream = h->stream_b; // the stage functions below enqueue on the handle's stream
        rc = 0;
        do {
            mark(h, T_ANT);
            if ((rc = msm_amd_stage_anterpolate(h))) break;
            if ((rc = msm_amd_stage_grid(h))) break; // marks T_RES..T_PRO
            if (hipStreamWaitEvent(h->stream_b, h->ev_pairs, 0) != hipSuccess) { rc = MSM_AMD_EHIP; break; }
            mark(h, T_INT);
            if ((rc = msm_amd_stage_interpolate(h, d_acc))) break;
            mark(h, T_RED);
            if ((rc = queue_energies(h))) break;
            mark(h, T_TOTAL);
            if (hipEventRecord(h->ev_join, h->stream_b) != hipSuccess) rc = MSM_AMD_EHIP;
        } while (0);
        h->stream = a;
        if (rc) return rc;
        HIPCHK(hipStreamWaitEvent(a, h->ev_join, 0));
    } else {
        mark(h, T_PAIR);
        if ((rc = msm_amd_stage_pairs(h, d_acc))) return rc;
        mark(h, T_ANT);
        if ((rc = msm_amd_stage_anterpolate(h))) return rc;
        if ((rc = msm_amd_stage_grid(h))) return rc; // marks T_RES..T_PRO
        mark(h, T_INT);
        if ((rc = msm_amd_stage_interpolate(h, d_acc))) return rc;
        mark(h, T_RED);
        if ((rc = queue_energies(h))) return rc;
        mark(h, T_TOTAL);
    }
    if (energies) {
        HIPCHK(hipMemcpyAsync(h->h_energies, h->d_energies, MSM_AMD_U_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(energies, h->h_energies, MSM_AMD_U_COUNT * sizeof(double));
    }
    if (h->profiling) {
        HIPCHK(hipEventSynchronize(h->ev[T_TOTAL]));
        for (int i = 0; i < T_TOTAL; i++) {
            float ms = 0;
            hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]);
            h->last_ms[i] = ms;
        }
        float ms = 0;
        if (overlap) { // the pair branch ends at its own event, not where the grid branch begins
            hipEventSynchronize(h->ev_pairs);
            hipEventElapsedTime(&ms, h->ev[T_PAIR], h->ev_pairs);
            h->last_ms[T_PAIR] = ms;
        }
        hipEventElapsedTime(&ms, h->ev[T_BIN], h->ev[T_TOTAL]);
        h->last_ms[T_TOTAL] = ms;
        h->pair_kernel_ms = -1.0;
        if (h->pk_recorded && hipEventElapsedTime(&ms, h->ev_pk[0], h->ev_pk[1]) == hipSuccess) h->pair_kernel_ms = ms;
    }
    return MSM_AMD_OK;
}

// ---------------------------------------------------------------------------
// evaluations that keep their neighbour structures (no counterpart in the reference)
// ---------------------------------------------------------------------------
static int ensure_reuse(msm_amd_handle *h, double skin) {
    if (h->rs_made && skin <= h->rs_skin_cap) return MSM_AMD_OK;
    if (h->rs_made) { // a larger skin than the buffers hold: start over (synchronises)
        HIPCHK(hipStreamSynchronize(h->stream));
        free_reuse(h);
    }
    const msm_amd_plan_desc &d = h->hp.d;
    const ClusterGeom &cg = h->clg;
    ReuseBuffers &rs = h->rs;
    const size_t nslots = (size_t)cg.max_clusters * CL_G, ncl = (size_t)cg.max_clusters;
    const double vol = d.A[0] * d.A[1] * d.A[2], rho = std::max(h->capacity, 1) / vol, aw = d.a + skin;
    {   // cluster lists for cutoff + skin: the sizing rule of msm_amd_create with the wider cutoff
        const double e = std::cbrt(8.0 / rho);
        const double vexp = e * e * e + 6.0 * e * e * aw + 3.0 * M_PI * aw * aw * e + 4.0 / 3.0 * M_PI * aw * aw * aw;
        const double est = 1.5 * vexp * rho / 8.0;
        long half = std::max(192L, ((long)(1.3 * est) + 31) / 32 * 32);
        const long cap = (1L << 32) / std::max(1, cg.max_clusters) / 2 / 32 * 32;
        rs.list_stride = (int)(2 * std::max(192L, std::min(half, cap)));
    }
    // half lists: particles within cutoff + skin of one of 4 neighbouring particles, with room for fluctuations
    // (measured at the benchmark's density: 2.1 spheres' worth; sections end on whole groups of 16)
    rs.hcap = (int)(((long)(1.4 * 2.1 * 4.0 / 3.0 * M_PI * aw * aw * aw * rho) + 96 + 31) / 32 * 32);
    int rc = MSM_AMD_OK;
#define RA(ptr, n) do { if (!rc) rc = dalloc(&(ptr), (size_t)(n)); } while (0)
    RA(rs.ps, nslots + CL_G); RA(rs.ref, nslots); RA(rs.order, nslots);
    RA(rs.blo, ncl); RA(rs.bhi, ncl); RA(rs.fbox, 2 * ncl);
    RA(rs.start, (size_t)cg.nbins + 1); RA(rs.clstart, (size_t)cg.ncols + 1); RA(rs.colncl, (size_t)cg.ncols + 1);
    RA(rs.list, ncl * rs.list_stride); RA(rs.n0, ncl); RA(rs.n1, ncl); RA(rs.n2, ncl);
    RA(rs.hl, ncl * 2 * (size_t)rs.hcap); RA(rs.hn, 2 * ncl);
    RA(rs.need, 1); RA(rs.counters, 2);
#undef RA
    if (rc) { free_reuse(h); return rc; }
    h->rs_made = true;
    h->rs_skin_cap = skin;
    h->rs_skin = -1.0;
    HIPCHK(hipMemsetAsync(rs.counters, 0, 2 * sizeof(int), h->stream));
    return MSM_AMD_OK;
}

extern "C" int msm_amd_eval_reuse(msm_amd_handle *h, const double *d_r, const double *d_q, double *d_acc, double *energies,
                                  double skin) {
    if (!h || !d_r || !d_q || !d_acc) { set_error("msm_amd_eval_reuse: null argument"); return MSM_AMD_EINVAL; }
    const msm_amd_plan_desc &d = h->hp.d;
    if (!(skin >= 0.0) || !(skin <= d.a)) { set_error("msm_amd_eval_reuse: the skin must lie in [0, a]"); return MSM_AMD_EINVAL; }
    if (h->small_box || h->hp.literal || h->pp.convention == MSM_AMD_ACC_LITERAL || h->n_owned != h->N || d.nzplanes != d.M[2] ||
        d.open_boundary || h->clg.max_clusters >= (1 << 24)) {
        set_error("msm_amd_eval_reuse: needs a periodic plan of at least 3 cutoffs per cell edge that owns every particle and "
                  "every plane (not the literal mode, not a slab rank), and fewer than 2^24 clusters");
        return MSM_AMD_EINVAL;
    }
    for (int k = 0; k < 3; k++)
        if (d.A[k] < 3.0 * (d.a + skin)) { set_error("msm_amd_eval_reuse: the cell must be at least 3 (a + skin) wide"); return MSM_AMD_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = ensure_reuse(h, skin))) return rc;
    if (skin != h->rs_skin) { // first call, or a new skin: the lists are built (again) by this evaluation
        HIPCHK(hipMemsetAsync(h->rs.need, 1, sizeof(int), h->stream)); // (any non-zero value raises the flag)
        h->rs_skin = skin;
    }
    h->pk_recorded = false;
    mark(h, T_BIN);
    // the fresh sort: what the tile kernels of the grid transfer walk, and what a rebuild adopts
    const int variant = h->opt_pair_variant;
    h->opt_pair_variant = 8;
    rc = msm_amd_stage_bin(h, d_r, d_q);
    h->opt_pair_variant = variant;
    if (rc) return rc;
    mark(h, T_PAIR);
    launch_reuse_pairs(h->stream, h->clg, h->clb, h->rs, h->pp, d_r, d_q, skin, d_acc, h->epart, h->profiling ? h->ev_pk : nullptr);
    h->pk_recorded = h->profiling;
    h->nepart = h->clg.max_clusters;
    HIPCHK(hipGetLastError());
    mark(h, T_ANT);
    if ((rc = msm_amd_stage_anterpolate(h))) return rc;
    if ((rc = msm_amd_stage_grid(h))) return rc; // marks T_RES..T_PRO
    mark(h, T_INT);
    if ((rc = msm_amd_stage_interpolate(h, d_acc))) return rc;
    mark(h, T_RED);
    if ((rc = queue_energies(h))) return rc;
    mark(h, T_TOTAL);
    h->rs_evals++;
    if (energies) {
        HIPCHK(hipMemcpyAsync(h->h_energies, h->d_energies, MSM_AMD_U_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(energies, h->h_energies, MSM_AMD_U_COUNT * sizeof(double));
    }
    if (h->profiling) {
        HIPCHK(hipEventSynchronize(h->ev[T_TOTAL]));
        for (int i = 0; i < T_TOTAL; i++) {
            float ms = 0;
            hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]);
            h->last_ms[i] = ms;
        }
        float ms = 0;
        hipEventElapsedTime(&ms, h->ev[T_BIN], h->ev[T_TOTAL]);
        h->last_ms[T_TOTAL] = ms;
        h->pair_kernel_ms = -1.0;
        if (hipEventElapsedTime(&ms, h->ev_pk[0], h->ev_pk[1]) == hipSuccess) h->pair_kernel_ms = ms;
    }
    return MSM_AMD_OK;
}

extern "C" int msm_amd_reuse_stats(msm_amd_handle *h, long *evaluations, long *rebuilds) {
    if (!h) return MSM_AMD_EINVAL;
    if (evaluations) *evaluations = h->rs_evals;
    if (rebuilds) {
        *rebuilds = 0;
        if (h->rs_made) {
            int c[2] = {0, 0};
            HIPCHK(hipSetDevice(h->device));
            HIPCHK(hipMemcpyAsync(c, h->rs.counters, sizeof(c), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            *rebuilds = c[1];
        }
    }
    return MSM_AMD_OK;
}

extern "C" int msm_amd_eval_enqueue(msm_amd_handle *h, const double *d_r, const double *d_q, double *d_acc,
                                    double *host_energies) {
    if (!h) return MSM_AMD_EINVAL;
    // pinned (device-visible) host memory: the last kernel of the evaluation writes the energies there itself;
    // anything else gets a copy command behind the evaluation
    double *alias = nullptr;
    if (host_energies) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, host_energies) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
            alias = (double *)at.devicePointer;
        else
            (void)hipGetLastError();
    }
    h->energies_alias = alias;
    int rc = msm_amd_eval(h, d_r, d_q, d_acc, nullptr);
    h->energies_alias = nullptr;
    if (rc) return rc;
    if (host_energies && !alias)
        HIPCHK(hipMemcpyAsync(host_energies, h->d_energies, MSM_AMD_U_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return MSM_AMD_OK;
}

// ---------------------------------------------------------------------------
// one evaluation of a slab rank, collectives included
// ---------------------------------------------------------------------------
// planes a rank needs from each z neighbour in halo mode: the interpolation support (p/2) plus the reach of the
// level-0 stencil; 0 when the stencil spans the whole period (then only the all-gather works)
static int slab_halo_planes(const msm_amd_handle *h) {
    const msm_amd_plan_desc &d = h->hp.d;
    const int S = h->hp.K[0].size[2];
    if (S >= d.M[2] || !(S & 1)) return 0;
    return d.p / 2 + (S - 1) / 2;
}

extern "C" int msm_amd_slab_exchange_mode(msm_amd_handle *h, int world, int mode) {
    if (!h || world < 1) return MSM_AMD_EINVAL;
    const msm_amd_plan_desc &d = h->hp.d;
    const int Mz = d.M[2], nz = d.nzplanes, H = slab_halo_planes(h);
    const bool can_halo = d.L >= 2 && world >= 2 && H > 0 && nz % 2 == 0 && d.zplane0 % 2 == 0 && H <= nz && nz + 2 * H <= Mz &&
                          (long)nz * world == Mz && !h->hp.literal && !d.open_boundary;
    if (d.open_boundary) {
        if (mode == MSM_AMD_SLAB_HALO || mode == MSM_AMD_SLAB_ALLGATHER) {
            set_error("slab step: an open-boundary plan exchanges its level-0 charges by a sum over the ranks only");
            return MSM_AMD_EINVAL;
        }
        return MSM_AMD_SLAB_REDUCE;
    }
    if (mode == MSM_AMD_SLAB_HALO) {
        if (can_halo) return MSM_AMD_SLAB_HALO;
        set_error("slab step: a halo exchange needs L >= 2, an even number of owned planes (%d) and %d halo planes per side "
                  "that fit a neighbour's slab and do not meet around the period (%d planes)", nz, H, Mz);
        return MSM_AMD_EINVAL;
    }
    if (mode == MSM_AMD_SLAB_ALLGATHER) return MSM_AMD_SLAB_ALLGATHER;
    if (mode != MSM_AMD_SLAB_AUTO) { set_error("slab step: unknown exchange mode %d", mode); return MSM_AMD_EINVAL; }
    return (can_halo && 2L * H < (long)(world - 1) * nz) ? MSM_AMD_SLAB_HALO : MSM_AMD_SLAB_ALLGATHER;
}

extern "C" int msm_amd_slab_step(msm_amd_handle *h, const msm_amd_collectives *coll, const double *d_r, const double *d_q,
                                 double *d_acc, int mode) {
    if (!h || !d_r || !d_q || !d_acc) { set_error("msm_amd_slab_step: null argument"); return MSM_AMD_EINVAL; }
    const msm_amd_plan_desc &d = h->hp.d;
    const int world = coll ? coll->world : 1;
    const bool slab = d.nzplanes < d.M[2];
    if (slab && (!coll || world < 2)) { set_error("msm_amd_slab_step: a handle that owns some of the planes needs the collectives"); return MSM_AMD_EINVAL; }
    if (h->small_box || h->hp.literal || !(h->anterp_tiled && h->opt_anterp_tiled) || h->opt_pair_variant < 5 ||
        h->pp.convention == MSM_AMD_ACC_LITERAL) {
        set_error("msm_amd_slab_step: needs the cluster pair sum and the tiled grid transfer (grid of at least 24 nodes per axis, "
                  "at least 3 cutoffs per cell edge, not the literal mode)");
        return MSM_AMD_EINVAL;
    }
    int xmode = MSM_AMD_SLAB_ALLGATHER;
    if (slab) {
        xmode = msm_amd_slab_exchange_mode(h, world, mode);
        if (xmode < 0) return xmode;
        if (xmode != MSM_AMD_SLAB_REDUCE && ((long)d.nzplanes * world != d.M[2] || d.zplane0 != coll->rank * d.nzplanes)) {
            set_error("msm_amd_slab_step: rank %d of %d must own planes [%d, %d) of %d", coll->rank, world, coll->rank * (d.M[2] / world),
                      (coll->rank + 1) * (d.M[2] / world), d.M[2]);
            return MSM_AMD_EINVAL;
        }
    }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    FusedAcc fused(h);
    // Two streams: the exchange of the grid charges (and, in halo mode, the first restriction and the all-gather of
    // the level-1 charges) travels on the second one while the first builds the lists and computes the pair sum, which
    // need nothing from the other ranks.  What else follows the exchange on the second stream is opt_slab_overlap's:
    // small, latency-bound kernels pay there (the chain of the coarser levels; on a thin slab the whole grid stage),
    // large ones compete with the pair kernel for the FP64 pipe and do not.
    const int ov = h->opt_slab_overlap >= 0 ? h->opt_slab_overlap : ((long)d.nzplanes * d.M[0] * d.M[1] <= 300000 ? 3 : 2);
    const bool two = ov != 0 && slab;
    if (two && (rc = ensure_stream_b(h))) return rc;
    h->pk_recorded = false;
    // A handle on the default stream runs the two-stream step on its own two streams, ordered behind the default stream's
    // work by one event at the start and ahead of its later work by one event at the end.  The default stream
    // synchronises implicitly with every blocking stream of the process (the caller's, a communication library's), so a
    // wait of the default stream in the middle of a step would tie the step to streams this file knows nothing about;
    // between the handle's non-blocking streams the only dependencies are the events recorded here.
    struct DefaultStreamGuard {
        msm_amd_handle *h;
        hipStream_t user;
        bool on;
        ~DefaultStreamGuard() {
            if (!on) return;
            if (hipEventRecord(h->ev_out, h->stream_a) == hipSuccess) (void)hipStreamWaitEvent(user, h->ev_out, 0);
            h->stream = user;
        }
    } guard{h, h->stream, false};
    if (two && h->stream == nullptr && h->opt_slab_own_stream) {
        if (!h->stream_a) {
            HIPCHK(hipStreamCreateWithFlags(&h->stream_a, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(h->ev_in, h->stream));
        HIPCHK(hipStreamWaitEvent(h->stream_a, h->ev_in, 0));
        h->stream = h->stream_a;
        guard.on = true;
    }
    hipStream_t a = h->stream, b = two ? h->stream_b : h->stream;
    const size_t plane = (size_t)d.M[0] * d.M[1];
    bool coarse_on_b = false, wait_before_pairs = false;
    // 3: the whole grid stage behind the exchange on the second stream, 4: and the anterpolation before it (A/B)
    // 5: as 2, but the first stream convolves level 0 between the list build and the pair kernel, as soon as the exchanged
    // level-0 charges are there: whatever is queued beside the pair kernel gets the compute units its one-wave workgroups
    // leave over, which is next to nothing until the last of them has started - a level-0 convolution queued a few
    // microseconds before the pair kernel took 61 us in one step and 200 in the next; ahead of it, it takes its 48
    const bool conv_first = two && ov == 5 && d.L >= 2 && !h->profiling;
    const bool grid_on_b = two && (ov == 3 || ov == 4) && !h->profiling, anterp_on_b = grid_on_b && ov == 4;
    if ((rc = msm_amd_stage_bin(h, d_r, d_q))) return rc;
    // (the fork event is signalled by the last kernel of the anterpolation itself when that is the stage in front of it)
    if (two && !anterp_on_b && h->opt_tail_events) msm::tail_event = h->ev_fork;
    if (!anterp_on_b && (rc = msm_amd_stage_anterpolate(h))) { msm::tail_event = nullptr; return rc; }
    if (two) {
        const bool signalled = !anterp_on_b && h->opt_tail_events && msm::tail_event == nullptr;
        msm::tail_event = nullptr;
        if (!signalled) HIPCHK(hipEventRecord(h->ev_fork, a));
        HIPCHK(hipStreamWaitEvent(b, h->ev_fork, 0));
    }
    if (slab) {
        h->stream = b; // the helpers enqueue on the handle's stream
        double *q0 = h->gq[0];
        rc = anterp_on_b ? msm_amd_stage_anterpolate(h) : 0;
        if (rc) {
        } else if (xmode == MSM_AMD_SLAB_REDUCE) {
            // slabs of different thickness: clear what this rank does not own, then sum over the ranks
            const size_t below = plane * d.zplane0, above = plane * (d.M[2] - d.zplane0 - d.nzplanes);
            if ((below && hipMemsetAsync(q0, 0, below * sizeof(double), b) != hipSuccess) ||
                (above && hipMemsetAsync(q0 + plane * (d.zplane0 + d.nzplanes), 0, above * sizeof(double), b) != hipSuccess)) {
                set_error("slab step: clearing the planes of other ranks failed");
                rc = MSM_AMD_EHIP;
            } else if (coll->all_reduce_sum(coll->ctx, q0, plane * d.M[2], (void *)b)) {
                set_error("slab step: the sum of the level-0 charges over the ranks failed");
                rc = MSM_AMD_EHIP;
            }
        } else if (xmode == MSM_AMD_SLAB_ALLGATHER) {
            if (coll->all_gather(coll->ctx, q0 + plane * d.zplane0, q0, plane * d.nzplanes, (void *)b)) {
                set_error("slab step: the all-gather of the level-0 charges failed");
                rc = MSM_AMD_EHIP;
            }
        } else {
            const int H = slab_halo_planes(h), Mz = d.M[2], z0 = d.zplane0, nz = d.nzplanes;
            const int up = (coll->rank + 1) % world, down = (coll->rank + world - 1) % world;
            // every range is contiguous: H <= nz, and z0 is a multiple of nz
            if (coll->halo_exchange(coll->ctx, q0 + plane * (z0 + nz - H), q0 + plane * z0, q0 + plane * ((z0 - H + Mz) % Mz),
                                    q0 + plane * ((z0 + nz) % Mz), plane * H, up, down, (void *)b)) {
                set_error("slab step: the halo exchange of the level-0 charges failed");
                rc = MSM_AMD_EHIP;
            }
            // (the level-0 convolution needs nothing beyond this exchange)
            if (!rc && conv_first && hipEventRecord(h->ev_pairs, b) != hipSuccess) { set_error("slab step: hipEventRecord failed"); rc = MSM_AMD_EHIP; }
            if (!rc) rc = grid_front(h, coll);
        }
        if (!rc && conv_first && xmode != MSM_AMD_SLAB_HALO && hipEventRecord(h->ev_pairs, b) != hipSuccess) {
            set_error("slab step: hipEventRecord failed");
            rc = MSM_AMD_EHIP;
        }
        // 2: the chain of the coarser levels (restrictions, their convolutions, top level, prolongations down to level 1:
        // small kernels that each last as long as a launch) follows the exchange on the second stream, beside the pair
        // sum; the first stream is left with the level-0 convolution and the last prolongation
        coarse_on_b = two && (ov == 2 || conv_first) && d.L >= 2 && !h->profiling;
        // The level-0 convolution, the largest kernel of the second stream, has to be under way before the pair kernel
        // is: whatever is queued beside the pair kernel gets the compute units its one-wave workgroups leave over, next
        // to nothing until the last of them has started (a convolution that started 2 us after the pair kernel took 200 us
        // instead of 61 and the step 30 us longer, in a quarter of the steps).  The pair kernel therefore waits for the
        // restrictions in front of the convolution - as a rule they have long finished beside the list build.
        const bool conv_leads = grid_on_b && h->opt_conv_leads && d.L >= 2;
        if (conv_leads) h->ev_restricted = h->ev_pairs;
        if (grid_on_b) h->ev_stage_tail = h->ev_join;
        if (!rc && grid_on_b) rc = grid_stage(h, xmode == MSM_AMD_SLAB_HALO, 0);
        const bool join_signalled = grid_on_b && !rc && h->ev_stage_tail == nullptr;
        h->ev_stage_tail = nullptr;
        const bool lead_recorded = conv_leads && !rc && h->ev_restricted == nullptr;
        h->ev_restricted = nullptr;
        if (lead_recorded) wait_before_pairs = true;
        if (!rc && coarse_on_b) rc = grid_stage(h, xmode == MSM_AMD_SLAB_HALO, 1);
        if (!rc && two && !join_signalled && hipEventRecord(h->ev_join, b) != hipSuccess) { set_error("slab step: hipEventRecord failed"); rc = MSM_AMD_EHIP; }
        h->stream = a;
        if (rc) return rc;
    }
    int conv_rc = MSM_AMD_OK;
    if (wait_before_pairs) {
        auto wait = [h, a, &conv_rc]() {
            if (hipStreamWaitEvent(a, h->ev_pairs, 0) != hipSuccess) { set_error("slab step: hipStreamWaitEvent failed"); conv_rc = MSM_AMD_EHIP; }
        };
        // in several launches, only the last one waits: the earlier ones run while the exchanges travel
        if (h->opt_pair_chunks > 1) h->pairs_before_last = wait;
        else h->pairs_between = wait;
    }
    if (conv_first) {
        const bool halo = xmode == MSM_AMD_SLAB_HALO;
        h->pairs_between = [h, a, halo, &conv_rc]() {
            if (hipStreamWaitEvent(a, h->ev_pairs, 0) != hipSuccess) { set_error("slab step: hipStreamWaitEvent failed"); conv_rc = MSM_AMD_EHIP; return; }
            conv_rc = grid_stage(h, halo, 3);
        };
    }
    // in several launches also where nothing waits: between them the chain of the coarser levels on the second stream gets
    // the wave slots it does not get while one launch has workgroups left to start, and is done when the pair kernel is
    // (the chain of a 256^3 grid is worth four launches - a rank of 8 at 10M particles: 2.98 -> 2.87 ms - that of a 96^3 grid
    // costs more at the boundaries than it gains)
    const int chunks_coarse = h->opt_pair_chunks_coarse >= 0 ? h->opt_pair_chunks_coarse
                                                             : ((long)d.nzplanes * d.M[0] * d.M[1] > 1000000 ? 4 : 1);
    h->pairs_chunks_now = wait_before_pairs ? h->opt_pair_chunks : (coarse_on_b ? std::max(1, chunks_coarse) : 1);
    rc = msm_amd_stage_pairs(h, d_acc);
    h->pairs_chunks_now = 1;
    h->pairs_between = nullptr;
    h->pairs_before_last = nullptr;
    if (rc || (rc = conv_rc)) return rc;
    if (two) HIPCHK(hipStreamWaitEvent(a, h->ev_join, 0));
    if (!(slab && grid_on_b) && (rc = grid_stage(h, slab && xmode == MSM_AMD_SLAB_HALO, conv_first ? 4 : coarse_on_b ? 2 : 0))) return rc;
    if ((rc = msm_amd_stage_interpolate(h, d_acc))) return rc;
    const bool energy_b = coll && world > 1 && two && h->opt_energy_on_b;
    if (energy_b && h->opt_tail_events) msm::tail_event = h->ev_fork; // (signalled by the kernel that writes the sums)
    rc = queue_energies(h, true);
    const bool sums_signalled = energy_b && h->opt_tail_events && msm::tail_event == nullptr;
    msm::tail_event = nullptr;
    if (rc) return rc;
    // The five sums are added over the ranks on the second stream when the step has one: nothing of the next step waits
    // for them (its own sums are written behind its join with the second stream, which works through its queue in
    // order), only msm_amd_slab_energies does.
    h->energy_on_b = false;
    if (coll && world > 1) {
        const bool on_b = two && h->opt_energy_on_b;
        if (on_b) {
            if (!sums_signalled) HIPCHK(hipEventRecord(h->ev_fork, a));
            HIPCHK(hipStreamWaitEvent(b, h->ev_fork, 0));
        }
        if (coll->all_reduce_sum(coll->ctx, h->d_energies, 5, (void *)(on_b ? b : a))) {
            set_error("slab step: the all-reduce of the energy sums failed");
            return MSM_AMD_EHIP;
        }
        if (on_b) {
            HIPCHK(hipEventRecord(h->ev_energy, b));
            h->energy_on_b = true;
        }
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

extern "C" int msm_amd_slab_energies(msm_amd_handle *h, const msm_amd_collectives *coll, double energies[MSM_AMD_U_COUNT]) {
    if (!h || !energies) return MSM_AMD_EINVAL;
    HIPCHK(hipSetDevice(h->device));
    if (h->energy_on_b) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_energy, 0));
    HIPCHK(hipMemcpyAsync(h->h_energies, h->d_energies, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double sums[5];
    memcpy(sums, h->h_energies, sizeof(sums));
    if (coll && coll->world > 1) sums[2] /= coll->world; // the Fourier term is the same on every rank: the all-reduce added it world times
    return msm_amd_combine_energies(h, sums, energies);
}

// ---------------------------------------------------------------------------
// the collectives over RCCL (loaded on first use, so that programs that never split a system need no librccl)
// ---------------------------------------------------------------------------
namespace {
struct RcclId { char internal[128]; };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommUserRank)(void *, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} g_rccl;
std::mutex g_rccl_mu;
const int RCCL_DOUBLE = 8, RCCL_SUM = 0; // ncclFloat64, ncclSum

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return MSM_AMD_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { set_error("librccl.so could not be loaded: %s", dlerror()); return MSM_AMD_ENODEV; }
    RcclApi api;
    api.lib = lib;
#define SYM(field, name)                                                                       \
    do {                                                                                       \
        *(void **)(&api.field) = dlsym(lib, name);                                             \
        if (!api.field) { set_error("librccl.so has no %s", name); dlclose(lib); return MSM_AMD_ENODEV; } \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = api;
    return MSM_AMD_OK;
}
int rccl_check(int r, const char *what) {
    if (r == 0) return 0;
    set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return MSM_AMD_EHIP;
}
int rccl_all_gather(void *ctx, const double *send, double *recv, size_t count, void *stream) {
    return rccl_check(g_rccl.AllGather(send, recv, count, RCCL_DOUBLE, ctx, (hipStream_t)stream), "ncclAllGather");
}
int rccl_all_reduce_sum(void *ctx, double *buf, size_t count, void *stream) {
    return rccl_check(g_rccl.AllReduce(buf, buf, count, RCCL_DOUBLE, RCCL_SUM, ctx, (hipStream_t)stream), "ncclAllReduce");
}
int rccl_halo_exchange(void *ctx, const double *send_up, const double *send_down, double *recv_from_down, double *recv_from_up,
                       size_t count, int up, int down, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    // every rank posts (send up, receive from below) before (send down, receive from above): with two ranks, where both
    // neighbours are the same peer, its receives then meet the sends in order
    int r1 = g_rccl.Send(send_up, count, RCCL_DOUBLE, up, ctx, st);
    int r2 = g_rccl.Recv(recv_from_down, count, RCCL_DOUBLE, down, ctx, st);
    int r3 = g_rccl.Send(send_down, count, RCCL_DOUBLE, down, ctx, st);
    int r4 = g_rccl.Recv(recv_from_up, count, RCCL_DOUBLE, up, ctx, st);
    int r5 = g_rccl.GroupEnd();
    return rccl_check(r1 ? r1 : (r2 ? r2 : (r3 ? r3 : (r4 ? r4 : r5))), "ncclSend / ncclRecv (halo exchange)");
}
} // namespace

extern "C" int msm_amd_rccl_available(void) { return rccl_load(); }

extern "C" int msm_amd_rccl_comm_info(const msm_amd_collectives *c, int *ranks, int *rank) {
    if (!c || !c->ctx || c->all_gather != rccl_all_gather) { set_error("msm_amd_rccl_comm_info: not a table made by msm_amd_rccl_create"); return MSM_AMD_EINVAL; }
    int n = 0, r = 0, rc;
    if ((rc = rccl_check(g_rccl.CommCount(c->ctx, &n), "ncclCommCount"))) return rc;
    if ((rc = rccl_check(g_rccl.CommUserRank(c->ctx, &r), "ncclCommUserRank"))) return rc;
    if (ranks) *ranks = n;
    if (rank) *rank = r;
    return MSM_AMD_OK;
}

extern "C" int msm_amd_rccl_unique_id(char id[128]) {
    if (!id) return MSM_AMD_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    RcclId u;
    if ((rc = rccl_check(g_rccl.GetUniqueId(&u), "ncclGetUniqueId"))) return rc;
    memcpy(id, u.internal, 128);
    return MSM_AMD_OK;
}
extern "C" int msm_amd_rccl_create(const char id[128], int rank, int world, msm_amd_collectives **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { set_error("msm_amd_rccl_create: bad argument"); return MSM_AMD_EINVAL; }
    int rc = rccl_load();
    if (rc) return rc;
    RcclId u;
    memcpy(u.internal, id, 128);
    void *comm = nullptr;
    if ((rc = rccl_check(g_rccl.CommInitRank(&comm, world, u, rank), "ncclCommInitRank"))) return rc;
    msm_amd_collectives *c = new msm_amd_collectives();
    c->rank = rank; c->world = world; c->ctx = comm;
    c->all_gather = rccl_all_gather; c->all_reduce_sum = rccl_all_reduce_sum; c->halo_exchange = rccl_halo_exchange;
    *out = c;
    return MSM_AMD_OK;
}
extern "C" void msm_amd_rccl_destroy(msm_amd_collectives *c) {
    if (!c) return;
    if (c->ctx && g_rccl.CommDestroy) g_rccl.CommDestroy(c->ctx);
    delete c;
}

extern "C" int msm_amd_write_msmsimple_cache(msm_amd_handle *h, const char *dir) {
    if (!h) return MSM_AMD_EINVAL;
    return write_msmsimple_cache(h->hp, dir);
}

// profiling hook: a device buffer of 4 x 64-bit words per cluster slot (msm_amd_pair_trace_slots of them) that the
// list-walking pair kernel fills per wave with {start, end (100 MHz clock), hardware ids, kind}; NULL switches it off
extern "C" int msm_amd_set_pair_trace(msm_amd_handle *h, void *d_buffer) {
    if (!h) return MSM_AMD_EINVAL;
    h->clb.trace = (unsigned long long *)d_buffer;
    return MSM_AMD_OK;
}
extern "C" long msm_amd_pair_trace_slots(msm_amd_handle *h) { return h ? (long)h->clg.max_clusters : 0; }

extern "C" int msm_amd_last_timings(msm_amd_handle *h, double ms[10]) {
    if (!h || !ms) return MSM_AMD_EINVAL;
    for (int i = 0; i < T_COUNT; i++) ms[i] = h->last_ms[i];
    return MSM_AMD_OK;
}

extern "C" int msm_amd_last_kernel_ms(msm_amd_handle *h, const char *name, double *ms) {
    if (!h || !name || !ms) return MSM_AMD_EINVAL;
    if (!strcmp(name, "pair_kernel") && h->pk_recorded && h->ev_made) {
        // the pair kernel's own pair of events, read once they have completed (msm_amd_slab_step leaves them unread)
        float t = 0;
        if (hipEventSynchronize(h->ev_pk[1]) == hipSuccess && hipEventElapsedTime(&t, h->ev_pk[0], h->ev_pk[1]) == hipSuccess) {
            h->pair_kernel_ms = t;
            *ms = t;
            return MSM_AMD_OK;
        }
        (void)hipGetLastError();
    }
    if (!strcmp(name, "pair_kernel") && h->pair_kernel_ms >= 0.0) { *ms = h->pair_kernel_ms; return MSM_AMD_OK; }
    set_error("last_kernel_ms: no timing for %s in the last profiled evaluation", name);
    return MSM_AMD_EINVAL;
}

// Device copies of host-side particle arrays, sized for the handle's capacity: msm_amd_set_counts may raise N up to
// it after the first host-pointer evaluation.
static int ensure_staging(msm_amd_handle *h) {
    if (h->s_r) return MSM_AMD_OK;
    int rc;
    if ((rc = dalloc(&h->s_r, 3 * (size_t)h->capacity))) return rc;
    if ((rc = dalloc(&h->s_q, (size_t)h->capacity))) return rc;
    if ((rc = dalloc(&h->s_acc, 3 * (size_t)h->capacity))) return rc;
    return MSM_AMD_OK;
}
// second acceleration array of the literal drop-in path (short-range and grid parts are reported separately)
static int ensure_staging2(msm_amd_handle *h) {
    int rc = ensure_staging(h);
    if (rc || h->s_acc2) return rc;
    return dalloc(&h->s_acc2, 3 * (size_t)h->capacity);
}

extern "C" int msm_amd_eval_host(msm_amd_handle *h, const double *r, const double *q, double *acc, double *energies) {
    if (!h || !r || !q || !acc) { set_error("msm_amd_eval_host: null argument"); return MSM_AMD_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_staging(h);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->s_r, r, 3 * (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->s_q, q, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if ((rc = msm_amd_eval(h, h->s_r, h->s_q, h->s_acc, energies))) return rc;
    HIPCHK(hipMemcpyAsync(acc, h->s_acc, 3 * (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MSM_AMD_OK;
}

// ---- state access ----
static double *grid_ptr(msm_amd_handle *h, int which, int level) {
    if (!h) return nullptr;
    if (which == MSM_AMD_GRID_EFOUR) return h->efour;
    if (level < 0 || level >= h->hp.d.L) return nullptr;
    return which == MSM_AMD_GRID_Q ? h->gq[level] : (which == MSM_AMD_GRID_E ? h->ge[level] : nullptr);
}
extern "C" void *msm_amd_grid_device_ptr(msm_amd_handle *h, int which, int level) { return grid_ptr(h, which, level); }
extern "C" int msm_amd_grid_dims(msm_amd_handle *h, int level, int dims[3]) {
    if (!h || level < 0 || level >= h->hp.d.L) return MSM_AMD_EINVAL;
    for (int k = 0; k < 3; k++) dims[k] = h->hp.Ml[level][k];
    return MSM_AMD_OK;
}
static size_t grid_count(msm_amd_handle *h, int which, int level) {
    const int l = which == MSM_AMD_GRID_EFOUR ? h->hp.d.L - 1 : level;
    return (size_t)h->hp.Ml[l][0] * h->hp.Ml[l][1] * h->hp.Ml[l][2];
}
extern "C" int msm_amd_copy_grid(msm_amd_handle *h, int which, int level, double *host_out) {
    double *p = grid_ptr(h, which, level);
    if (!p || !host_out) { set_error("copy_grid: bad grid selector"); return MSM_AMD_EINVAL; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host_out, p, grid_count(h, which, level) * sizeof(double), hipMemcpyDeviceToHost));
    return MSM_AMD_OK;
}
extern "C" int msm_amd_set_grid(msm_amd_handle *h, int which, int level, const double *host_in) {
    double *p = grid_ptr(h, which, level);
    if (!p || !host_in) { set_error("set_grid: bad grid selector"); return MSM_AMD_EINVAL; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(p, host_in, grid_count(h, which, level) * sizeof(double), hipMemcpyHostToDevice));
    return MSM_AMD_OK;
}
extern "C" int msm_amd_stencil_info(msm_amd_handle *h, int level, int size[3], int first[3]) {
    if (!h || level < 0 || level > h->hp.d.L) return MSM_AMD_EINVAL;
    const LevelStencil &S = level == h->hp.d.L ? h->hp.KL : h->hp.K[level];
    for (int k = 0; k < 3; k++) { size[k] = S.size[k]; first[k] = S.first[k]; }
    return MSM_AMD_OK;
}
extern "C" int msm_amd_copy_stencil(msm_amd_handle *h, int level, double *host_out) {
    if (!h || level < 0 || level > h->hp.d.L || !host_out) return MSM_AMD_EINVAL;
    const LevelStencil &S = level == h->hp.d.L ? h->hp.KL : h->hp.K[level];
    memcpy(host_out, S.v.data(), S.v.size() * sizeof(double));
    return MSM_AMD_OK;
}
extern "C" int msm_amd_copy_wprime(msm_amd_handle *h, double *host_out, int *len) {
    if (!h || !len) return MSM_AMD_EINVAL;
    *len = (int)h->hp.wprime.size();
    if (host_out) memcpy(host_out, h->hp.wprime.data(), h->hp.wprime.size() * sizeof(double));
    return MSM_AMD_OK;
}
extern "C" int msm_amd_op_restrict(msm_amd_handle *h, int l, const double *d_fine, double *d_coarse) {
    if (!h || l < 0 || l + 1 >= h->hp.d.L) { set_error("op_restrict: level out of range"); return MSM_AMD_EINVAL; }
    restrict_level(h, l, d_fine, d_coarse);
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}
extern "C" int msm_amd_op_prolong_add(msm_amd_handle *h, int l, const double *d_coarse, double *d_fine) {
    if (!h || l < 0 || l + 1 >= h->hp.d.L) { set_error("op_prolong_add: level out of range"); return MSM_AMD_EINVAL; }
    prolong_level_add(h, l, d_coarse, d_fine);
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}
extern "C" int msm_amd_op_convolve(msm_amd_handle *h, int l, const double *d_q, double *d_e) {
    if (!h || l < 0 || l > h->hp.d.L) { set_error("op_convolve: level out of range"); return MSM_AMD_EINVAL; }
    if (l == h->hp.d.L) {
        const int *ML = h->hp.Ml[l - 1];
        conv_untiled(h, d_q, h->dKL, ML, h->hp.KL.size, h->hp.KL.first, d_e, 0, ML[2]);
    } else if (h->hp.d.open_boundary && l == h->hp.d.L - 1) {
        launch_conv_open(h->stream, d_q, h->dK[l], h->hp.Ml[l], d_e, h->open_part);
    } else {
        convolve_level(h, l, d_q, d_e, 0, h->hp.Ml[l][2]);
    }
    HIPCHK(hipGetLastError());
    return MSM_AMD_OK;
}

// ---------------------------------------------------------------------------
// drop-in entry points
// ---------------------------------------------------------------------------
static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void write_acc_file(const char *name, int N, const double *f) { // msmPeriodic.c:1369-1381
    FILE *fp = fopen(name, "w");
    if (!fp) return;
    fprintf(fp, "%d\n", N);
    for (int d = 0; d < 3; d++)
        for (int i = 0; i < N; i++) fprintf(fp, "%20.14e\n", f[3 * i + d]);
    fclose(fp);
}

// The drop-in entry points keep their last plan (stencils, every device buffer) between calls: a caller that evaluates
// the same system again and again - a time-stepping loop around msmperiodic() - builds it once.  One plan, replaced when
// a call needs a different one; the calls are serialised (the reference is not re-entrant either: msmPeriodic.c writes
// fixed file names).
namespace {
struct PlanCache {
    std::mutex mu;
    msm_amd_handle *h = nullptr;
    msm_amd_plan_desc d;
    bool literal = false;
} g_plan_cache;
}
static long g_plan_builds = 0; // plans the drop-in entry points have built so far (guarded by g_plan_cache.mu)
static int cached_plan(const msm_amd_plan_desc &d, bool literal, msm_amd_handle **out) { // g_plan_cache.mu is held
    PlanCache &c = g_plan_cache;
    if (c.h && c.literal == literal && !memcmp(&c.d, &d, sizeof(d))) { *out = c.h; return MSM_AMD_OK; }
    if (c.h) { msm_amd_destroy(c.h); c.h = nullptr; }
    const int rc = create_impl(&d, literal, 0, nullptr, out);
    if (rc) return rc;
    g_plan_builds++;
    c.h = *out; c.d = d; c.literal = literal;
    return MSM_AMD_OK;
}
static void drop_cached_plan() { // g_plan_cache.mu is held
    if (g_plan_cache.h) msm_amd_destroy(g_plan_cache.h);
    g_plan_cache.h = nullptr;
}
extern "C" long msm_amd_plan_builds(void) {
    std::lock_guard<std::mutex> lock(g_plan_cache.mu);
    return g_plan_builds;
}
namespace msm { void direct_release_staging(); }
extern "C" void msm_amd_release_cached_plan(void) {
    {
        std::lock_guard<std::mutex> lock(g_plan_cache.mu);
        drop_cached_plan();
    }
    msm::direct_release_staging(); // the device copies direct() keeps between calls
}

extern "C" msm_stats msmperiodic(int N, double *r, double *q, int p, double abar, int mu, double *uappr, double *fappr,
                                 double Ax, double Ay, double Az) {
    msm_stats S;
    memset(&S, 0, sizeof(S));
    const double t_total = now_s();
    S.N = N;
    S.p = p;
    const int mode = effective_mode();
    const bool literal = mode == MSM_AMD_MODE_LITERAL || (mode == MSM_AMD_MODE_AUTO && N <= 4096);
    g_last_call_mode = literal ? MSM_AMD_MODE_LITERAL : MSM_AMD_MODE_SCALABLE;
    if (mode == MSM_AMD_MODE_AUTO && !literal && !g_quiet && !g_auto_noticed) {
        // said once per process: the two modes return different quantities in fappr
        g_auto_noticed = true;
        fprintf(stderr, "msmperiodic: N = %d > 4096, scalable mode: fappr is the acceleration of a unit charge at r_i "
                        "(no factor q_i), r is left as it is; msm_amd_set_mode() or MSM_AMD_MODE=literal|scalable chooses\n", N);
    }
    if (literal && p != 4) { // msmPeriodic.c:100-104
        fprintf(stderr, "p must be 4\n");
        return S;
    }
    double t0 = now_s();
    if (literal)
        for (int i = 0; i < 3 * N; i++) r[i] += 0.5; // msmPeriodic.c:75-76: the caller's array is shifted
    msm_amd_plan_desc d;
    int rc = msm_amd_plan_auto(N, Ax, Ay, Az, p, mu, abar, literal ? 1 : 0, &d);
    if (rc) { fprintf(stderr, "msmperiodic: %s\n", msm_amd_last_error()); return S; }
    d.acc_convention = literal ? MSM_AMD_ACC_LITERAL : MSM_AMD_ACC_GRADIENT;
    S.cputime_initialization = now_s() - t0;
    t0 = now_s();
    std::lock_guard<std::mutex> cache_lock(g_plan_cache.mu);
    msm_amd_handle *h = nullptr;
    rc = cached_plan(d, literal, &h);
    if (rc) { fprintf(stderr, "msmperiodic: %s\n", msm_amd_last_error()); return S; }
    S.cputime_stencill = now_s() - t0; // stencil generation dominates plan creation (nothing when the plan was kept)
    const HostPlan &hp = h->hp;
    const int L = d.L;

    // parameters (msmPeriodic.c:128-335)
    S.L = L; S.mu = mu; S.h = std::pow(Ax * Ay * Az / N, 1.0 / 3.0);
    S.hx = hp.hl[0][0]; S.hy = hp.hl[0][1]; S.hz = hp.hl[0][2];
    S.a = d.a; S.aL = std::pow(2, L) * d.a; S.alpha = abar; S.beta = d.beta; S.s = p - 1; S.kmax = d.kmax;
    S.csr = hp.csr;
    S.chinx = S.chiny = S.chinz = 2 * d.kmax + 1;
    S.chisize = S.chinx * S.chiny * S.chinz;
    S.rminx = S.rminy = S.rminz = 0.0; S.rmaxx = S.rmaxy = S.rmaxz = 1.0; // msmPeriodic.c:124-133
    S.Ax = (int)Ax; S.Ay = (int)Ay; S.Az = (int)Az;
    S.volume = Ax * Ay * Az;
    S.Mx = d.M[0]; S.My = d.M[1]; S.Mz = d.M[2];
    S.Mlx[0] = d.M[0]; S.Mly[0] = d.M[1]; S.Mlz[0] = d.M[2];
    for (int l = 1; l <= L && l < LMAX; l++) { // msmPeriodic.c:184-188
        S.Mlx[l] = (int)(d.M[0] / std::pow(2, l - 1)); S.Mly[l] = (int)(d.M[1] / std::pow(2, l - 1));
        S.Mlz[l] = (int)(d.M[2] / std::pow(2, l - 1));
    }
    S.MLx = hp.Ml[L - 1][0]; S.MLy = hp.Ml[L - 1][1]; S.MLz = hp.Ml[L - 1][2];
    S.ML2x = S.MLx / 2; S.ML2y = S.MLy / 2; S.ML2z = S.MLz / 2;
    // print_stats reads clen = ML/2+1 entries of cvec although kmax+1 exist (msmLibrary.c:321-332): allocate the larger
    {
        const int need = std::max(d.kmax + 1, std::max(S.ML2x, std::max(S.ML2y, S.ML2z)) + 1);
        double **dst[3] = {&S.cvecx, &S.cvecy, &S.cvecz};
        for (int k = 0; k < 3; k++) {
            *dst[k] = (double *)calloc(need, sizeof(double));
            memcpy(*dst[k], hp.cvec[k].data(), hp.cvec[k].size() * sizeof(double));
        }
        S.clenx = S.ML2x + 1; S.cleny = S.ML2y + 1; S.clenz = S.ML2z + 1;
        S.chi = (double *)calloc(S.chisize, sizeof(double));
        memcpy(S.chi, hp.chi.data(), hp.chi.size() * sizeof(double));
        S.chisum = hp.chisum;
    }
    S.sLnx = 2 * (S.MLx - 1) + 1; S.sLny = 2 * (S.MLy - 1) + 1; S.sLnz = 2 * (S.MLz - 1) + 1;
    S.sLsize = S.sLnx * S.sLny * S.sLnz;
    for (int l = 0; l < L; l++) {
        S.slsizex[l] = hp.K[l].size[0]; S.slsizey[l] = hp.K[l].size[1]; S.slsizez[l] = hp.K[l].size[2];
    }
    if (literal && !g_quiet) { // msmPeriodic.c:600-610
        const LevelStencil &K = hp.K[0];
        printf("Stencil[%d]\n", 0);
        for (int nz = 0; nz < K.size[2]; nz++)
            for (int ny = 0; ny < K.size[1]; ny++)
                for (int nx = 0; nx < K.size[0]; nx++)
                    printf("%2d x %2d x %2d %25.16f \n", nz - 1, ny - 1, nx - 1,
                           K.v[((size_t)nz * K.size[1] + ny) * K.size[0] + nx]);
    }

    double en[MSM_AMD_U_COUNT];
    if (!literal) {
        // Scalable mode: one evaluation through the handle (its staging buffers, one acceleration array written straight
        // into the caller's fappr), stage times from the handle's events, net grid charge summed on the device.
        const bool was_profiling = h->profiling;
        msm_amd_set_profiling(h, 1);
        rc = msm_amd_eval_host(h, r, q, fappr, en);
        double gsum = 0.0;
        if (!rc) {
            std::vector<double> parts(2 * NPARTS);
            launch_charge_partials(h->stream, h->gq[0], (size_t)d.M[0] * d.M[1] * d.M[2], h->parts, NPARTS);
            if (hipMemcpy(parts.data(), h->parts, parts.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = MSM_AMD_EHIP;
            for (int i = 0; i < NPARTS; i++) gsum += parts[i];
        }
        h->profiling = was_profiling;
        if (rc) {
            if (rc == MSM_AMD_EHIP) set_error("msmperiodic: a HIP call failed");
            fprintf(stderr, "msmperiodic: %s\n", msm_amd_last_error());
            drop_cached_plan();
            return S;
        }
        const double *ms = h->last_ms;
        S.cputime_short_range = 1e-3 * (ms[T_BIN] + ms[T_PAIR]);
        S.cputime_anterpolation = 1e-3 * ms[T_ANT];
        S.cputime_restriction = 1e-3 * ms[T_RES];
        S.cputime_grid_to_grid_mapping = 1e-3 * ms[T_CONV];
        S.cputime_top_level = 1e-3 * ms[T_TOP];
        S.cputime_prolongation = 1e-3 * ms[T_PRO];
        S.cputime_interpolation = 1e-3 * ms[T_INT];
        S.cputime_potential_long = 1e-3 * ms[T_RED];
        S.potential_short_real = en[MSM_AMD_U_SHORT_REAL];
        S.potential_short_self = en[MSM_AMD_U_SHORT_SELF];
        S.potential_short_csr = en[MSM_AMD_U_SHORT_CSR];
        S.potential_short_total = en[0] + en[1] + en[2];
        S.potential_long_real = en[MSM_AMD_U_LONG_REAL];
        S.potential_long_fourier = en[MSM_AMD_U_LONG_FOUR];
        S.potential_long_self = en[MSM_AMD_U_LONG_SELF];
        S.potential_long_total = en[MSM_AMD_U_LONG_FOUR] + en[MSM_AMD_U_LONG_REAL] + en[MSM_AMD_U_LONG_SELF];
        S.potential_total = en[MSM_AMD_U_TOTAL];
        if (uappr) *uappr = S.potential_total;
        double qs = 0.0;
        for (int i = 0; i < N; i++) qs += q[i];
        S.netcharge = qs;
        S.netgridcharge[0] = gsum;
        S.cputime_total = now_s() - t_total; // the plan stays for the next call (msm_amd_release_cached_plan frees it)
        return S;
    }

    // literal mode: stage by stage, so that the two acceleration parts can be reported separately (msm.acclong, msm.accshort)
    std::vector<double> fs(3 * (size_t)N), fl(3 * (size_t)N), g0((size_t)d.M[0] * d.M[1] * d.M[2]);
    // the handle's staging buffers (kept with the plan), every HIP call checked: a failed copy must not come back as
    // energies and forces of garbage
    bool ok = ensure_staging2(h) == MSM_AMD_OK;
    double *dr = h->s_r, *dq = h->s_q, *dfs = h->s_acc, *dfl = h->s_acc2;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        set_error("msmperiodic: %s failed: %s", what, hipGetErrorString(e));
        return false;
    };
    if (ok) {
        ok = hip_ok(hipMemcpy(dr, r, 3 * (size_t)N * sizeof(double), hipMemcpyHostToDevice), "copy of r") &&
             hip_ok(hipMemcpy(dq, q, (size_t)N * sizeof(double), hipMemcpyHostToDevice), "copy of q") &&
             hip_ok(hipMemset(dfl, 0, 3 * (size_t)N * sizeof(double)), "clearing the grid part");
        auto tick = [&](double &slot, int rc_) {
            if (!rc_ && !hip_ok(hipStreamSynchronize(h->stream), "stream synchronisation")) rc_ = MSM_AMD_EHIP;
            slot = now_s() - t0;
            t0 = now_s();
            return rc_;
        };
        double tbin;
        t0 = now_s();
        ok = ok && !tick(tbin, msm_amd_stage_bin(h, dr, dq));
        ok = ok && !tick(S.cputime_anterpolation, msm_amd_stage_anterpolate(h));
        ok = ok && !tick(S.cputime_grid_to_grid_mapping, msm_amd_stage_grid(h));
        ok = ok && !tick(S.cputime_interpolation, msm_amd_stage_interpolate(h, dfl));
        ok = ok && !tick(S.cputime_short_range, msm_amd_stage_pairs(h, dfs));
        ok = ok && !tick(S.cputime_potential_long, msm_amd_stage_energies(h, en));
        ok = ok && hip_ok(hipMemcpy(fs.data(), dfs, fs.size() * sizeof(double), hipMemcpyDeviceToHost), "copy of the short-range part") &&
             hip_ok(hipMemcpy(fl.data(), dfl, fl.size() * sizeof(double), hipMemcpyDeviceToHost), "copy of the grid part") &&
             msm_amd_copy_grid(h, MSM_AMD_GRID_Q, 0, g0.data()) == MSM_AMD_OK;
    }
    if (!ok) {
        fprintf(stderr, "msmperiodic: %s\n", msm_amd_last_error());
        drop_cached_plan();
        return S;
    }
    if (literal && !g_quiet) { // msmPeriodic.c:723-733
        printf("grid charge at first level\n");
        for (size_t i = 0; i < g0.size(); i++) printf("%25.16e\n", g0[i]);
    }
    if (literal && !g_quiet) {
        // msmPeriodic.c:1334-1341: the grid energy of the gamma term and of the erf term of the level-0 kernel on their
        // own, 1/2 q0 . (sl_part{1,2} * q0) with the same grid-to-grid loop; the stencils are built on the host once
        // per plan, the two convolutions run on the GPU
        if (h->lit_part[0].v.empty()) build_literal_parts(h->hp, h->lit_part[0], h->lit_part[1]);
        const size_t n0 = g0.size();
        double part_u[2] = {0.0, 0.0};
        bool pok = true;
        for (int k = 0; k < 2 && pok; k++) {
            const LevelStencil &P = h->lit_part[k];
            std::vector<double> e(n0);
            // (device copies made once per plan and kept in the handle: a driver loop allocates nothing here)
            if (!h->lit_part_d[k])
                pok = dalloc(&h->lit_part_d[k], P.v.size()) == MSM_AMD_OK &&
                      hip_ok(hipMemcpy(h->lit_part_d[k], P.v.data(), P.v.size() * sizeof(double), hipMemcpyHostToDevice), "copy of a part stencil");
            if (pok && !h->lit_part_e) pok = dalloc(&h->lit_part_e, n0) == MSM_AMD_OK;
            if (pok) {
                launch_conv_literal(h->stream, h->gq[0], h->lit_part_d[k], h->hp.Ml[0], P.size, P.first, h->lit_part_e);
                pok = hip_ok(hipStreamSynchronize(h->stream), "part convolution") &&
                      hip_ok(hipMemcpy(e.data(), h->lit_part_e, n0 * sizeof(double), hipMemcpyDeviceToHost), "copy of a part potential");
            }
            if (pok) for (size_t i = 0; i < n0; i++) part_u[k] += 0.5 * e[i] * g0[i];
        }
        if (pok) {
            printf("%-20s : %25.16e\n", "    ulong_real_g1", part_u[0]);
            printf("%-20s : %25.16e\n", "    ulong_real_gLR", part_u[1]);
        } else {
            fprintf(stderr, "msmperiodic: %s\n", msm_amd_last_error());
        }
    }
    S.potential_short_real = en[MSM_AMD_U_SHORT_REAL];
    S.potential_short_self = en[MSM_AMD_U_SHORT_SELF];
    S.potential_short_csr = en[MSM_AMD_U_SHORT_CSR];
    S.potential_short_total = en[0] + en[1] + en[2];
    S.potential_long_real = en[MSM_AMD_U_LONG_REAL];
    S.potential_long_fourier = en[MSM_AMD_U_LONG_FOUR];
    S.potential_long_self = en[MSM_AMD_U_LONG_SELF];
    S.potential_long_total = en[MSM_AMD_U_LONG_FOUR] + en[MSM_AMD_U_LONG_REAL] + en[MSM_AMD_U_LONG_SELF];
    S.potential_total = en[MSM_AMD_U_TOTAL];
    if (uappr) *uappr = S.potential_total;
    for (size_t i = 0; i < fs.size(); i++) fappr[i] = fs[i] + fl[i]; // msmPeriodic.c:1362-1367
    if (literal && !g_quiet) {
        write_acc_file("msm.acclong", N, fl.data());
        write_acc_file("msm.accshort", N, fs.data());
    }
    {
        double qs = 0.0, gs = 0.0;
        for (int i = 0; i < N; i++) qs += q[i];
        for (size_t i = 0; i < g0.size(); i++) gs += g0[i];
        S.netcharge = qs;
        S.netgridcharge[0] = gs;
    }
    S.cputime_total = now_s() - t_total; // the plan stays for the next call (msm_amd_release_cached_plan frees it)
    return S;
}

// Open-boundary MSM (msm3d, declared at msm/msmLibrary.h:216 and defined nowhere in the reference): plan around the
// particles' bounding box, one evaluation through the handle layer, energies into the result record.
extern "C" msm_stats msm3d(int N, double *r, double *q, int p, double hgrid, double a, int mu, int s, double *uappr,
                           double *fappr) {
    msm_stats S;
    memset(&S, 0, sizeof(S));
    const double t_total = now_s();
    S.N = N; S.p = p; S.mu = mu; S.s = s;
    if (N < 1 || !r || !q || !fappr) { fprintf(stderr, "msm3d: bad arguments\n"); return S; }
    if ((p != 4 && p != 6) || s != p) { // the splitting gamma has as many terms as the spline has order (msmPeriodic.c:22-34)
        fprintf(stderr, "msm3d: p must be 4 or 6 and s must equal p\n");
        return S;
    }
    double lo[3] = {r[0], r[1], r[2]}, hi[3] = {r[0], r[1], r[2]};
    for (int i = 1; i < N; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], r[3 * (size_t)i + k]);
            hi[k] = std::max(hi[k], r[3 * (size_t)i + k]);
        }
    msm_amd_plan_desc d;
    int rc = msm_amd_plan_open(N, lo, hi, p, mu, hgrid, a, 0, &d);
    if (rc) { fprintf(stderr, "msm3d: %s\n", msm_amd_last_error()); return S; }
    double t0 = now_s();
    std::lock_guard<std::mutex> cache_lock(g_plan_cache.mu);
    msm_amd_handle *h = nullptr;
    rc = cached_plan(d, false, &h);
    if (rc) { fprintf(stderr, "msm3d: %s\n", msm_amd_last_error()); return S; }
    S.cputime_stencill = now_s() - t0;
    double en[MSM_AMD_U_COUNT];
    rc = msm_amd_eval_host(h, r, q, fappr, en);
    if (rc) { fprintf(stderr, "msm3d: %s\n", msm_amd_last_error()); drop_cached_plan(); return S; }
    const int L = d.L;
    S.L = L; S.h = hgrid; S.hx = h->hp.hl[0][0]; S.hy = h->hp.hl[0][1]; S.hz = h->hp.hl[0][2];
    S.a = a; S.aL = std::ldexp(a, L - 1); S.alpha = a / hgrid;
    S.rminx = lo[0]; S.rminy = lo[1]; S.rminz = lo[2]; S.rmaxx = hi[0]; S.rmaxy = hi[1]; S.rmaxz = hi[2];
    S.volume = d.A[0] * d.A[1] * d.A[2];
    S.Mx = d.M[0]; S.My = d.M[1]; S.Mz = d.M[2];
    for (int l = 0; l < L && l < LMAX; l++) { S.Mlx[l] = h->hp.Ml[l][0]; S.Mly[l] = h->hp.Ml[l][1]; S.Mlz[l] = h->hp.Ml[l][2]; }
    S.MLx = h->hp.Ml[L - 1][0]; S.MLy = h->hp.Ml[L - 1][1]; S.MLz = h->hp.Ml[L - 1][2];
    S.potential_short_real = en[MSM_AMD_U_SHORT_REAL];
    S.potential_short_total = en[0] + en[1] + en[2];
    S.potential_long_real = en[MSM_AMD_U_LONG_REAL];
    S.potential_long_self = en[MSM_AMD_U_LONG_SELF];
    S.potential_long_total = en[MSM_AMD_U_LONG_FOUR] + en[MSM_AMD_U_LONG_REAL] + en[MSM_AMD_U_LONG_SELF];
    S.potential_total = en[MSM_AMD_U_TOTAL];
    if (uappr) *uappr = S.potential_total;
    { double qs = 0.0; for (int i = 0; i < N; i++) qs += q[i]; S.netcharge = qs; }
    S.cputime_total = now_s() - t_total; // the plan stays for the next call (msm_amd_release_cached_plan frees it)
    return S;
}

extern "C" double msmPeriodicRunner(double *r, double *q, double *force, int n, int p, int mu, double abar, double Lx,
                                    double Ly, double Lz) {
    double potential = 0.0;
    msm_stats st = msmperiodic(n, r, q, p, abar, mu, &potential, force, Lx, Ly, Lz);
    print_stats(st);
    return st.potential_total;
}
