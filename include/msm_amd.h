/* msm_amd — MI355X (gfx950) multilevel-summation engine, C ABI.
 *
 * Two layers are exported by libmsm_amd.so:
 *
 *  (1) the drop-in layer: the symbols a driver written against the reference's
 *      msm/msmLibrary.h links to (msmperiodic, msmPeriodicRunner, chooseM, ...,
 *      print_stats, data_read, the linked-list helpers), same signatures, same
 *      msm_stats layout (960 bytes), computed on the GPU;
 *  (2) the handle layer (msm_amd_*): an explicit plan + device-resident
 *      evaluation for callers that keep particles in HBM, and per-stage entry
 *      points used by the parity tests and by the multi-GPU slab driver.
 *
 * All pointers are plain host or device pointers as stated per function; no
 * C++ or torch types cross this boundary.  Every function that can fail
 * returns 0 on success and a negative MSM_AMD_E* code otherwise;
 * msm_amd_last_error() gives the text.  There is no CPU fallback: without a
 * usable HIP device every compute entry point fails with MSM_AMD_ENODEV.
 */
#ifndef MSM_AMD_H
#define MSM_AMD_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------ */
/* (1) drop-in layer                                                    */
/* ------------------------------------------------------------------ */

#define MAXDIM 3
#define LMAX 10
#define DIMX 0
#define DIMY 1
#define DIMZ 2
#define MAX_POLY_DEGREE 6
#define MAX_MU 14
#ifndef PI
#define PI 3.14159265358979323846
#endif
#define VERYLARGENUMBER 1e30

/* The rest of what msm/msmLibrary.h gives a program that includes it (:18-20 TAUP4 / TAUP4D / ERFBETA, :113-149
 * MasterBaseFunction with the cubic and quintic B-spline pieces and their derivatives, :152-153 JN, :156-185 OMEGA):
 * generated from this library's own routines by tools/gen_tables.cpp.  The unmodified msmPeriodic.c / msmExpected.c
 * use them.  A translation unit that wants only the prototypes defines MSM_AMD_NO_TABLES first. */
#ifndef MSM_AMD_NO_TABLES
#include "msm_tables.h"
#endif

/* replaces struct Parameters, msm/msmLibrary.h:22-30 (unused by the reference itself) */
typedef struct Parameters {
    double abar;
    int mu;
    int p;
    double Ax, Ay, Az;
} Parameters;

/* replaces msm/msmLibrary.h:32-44 */
typedef struct LinkedListElement {
    int data;
    double value;
    struct LinkedListElement *next;
    struct LinkedListElement *prev;
} LinkedListElement;

typedef struct LinkedList {
    struct LinkedListElement *head;
    struct LinkedListElement *tail;
} LinkedList;

/* Result record, binary compatible with msm/msmLibrary.h:46-110 (960 bytes;
 * N at 0, h 16, beta 72, csr 96, cvecx 120, chi 168, Mx 204, Mlx 240,
 * slsizex 360, Ax 480, rminx 616, volume 664, netcharge 696,
 * cputime_initialization 784, potential_total 888, potential_long_fourier 952).
 * cvecx/cvecy/cvecz/chi are malloc'ed host arrays the caller may read (and
 * free); print_stats reads clenx.. / chisize entries of them. */
typedef struct msm_stats {
    int N, L, mu;
    double h, hx, hy, hz, a, aL, alpha, beta;
    int p, s, kmax;
    double csr;
    int chinx, chiny, chinz, chisize;
    double *cvecx, *cvecy, *cvecz;
    int clenx, cleny, clenz;
    double chisum, *chi;
    int sLsize, sLnx, sLny, sLnz;
    int M0[MAXDIM];
    int Mx, My, Mz;
    int MLx, MLy, MLz;
    int ML2x, ML2y, ML2z;
    int Mlx[LMAX], Mly[LMAX], Mlz[LMAX];
    int slsizex[LMAX], slsizey[LMAX], slsizez[LMAX];
    int Ax, Ay, Az;
    int size[LMAX][MAXDIM];
    double rminx, rminy, rminz, rmaxx, rmaxy, rmaxz;
    double volume, density;
    long int short_range_interactions;
    int storage;
    double netcharge;
    double netgridcharge[LMAX];
    double cputime_initialization, cputime_anterpolation, cputime_restriction, cputime_stencill, cputime_stencilL,
        cputime_grid_to_grid_mapping, cputime_top_level, cputime_prolongation, cputime_interpolation,
        cputime_potential_long, cputime_short_range, cputime_self, cputime_total;
    double potential_total, potential_short_total, potential_short_real, potential_short_csr, potential_short_self,
        potential_long_total, potential_long_real, potential_long_self, potential_long_fourier;
} msm_stats;

/* Periodic MSM energy + per-particle accelerations on the GPU.
 * replaces msmperiodic(), msm/msmLibrary.h:188-189, defined at msm/msmPeriodic.c:52.
 * r[3N] (AoS), q[N] host in; *uappr and fappr[3N] host out (overwritten).
 * In literal mode (see msm_amd_set_mode) the caller's r is shifted by +0.5 as
 * the reference does (msmPeriodic.c:75-76). */
msm_stats msmperiodic(int N, double *r, double *q, int p, double abar, int mu, double *uappr, double *fappr, double Ax,
                      double Ay, double Az);
/* replaces msmPeriodicRunner(), msm/msmLibrary.h:249, msm/msmPeriodic.c:1458 */
double msmPeriodicRunner(double *r, double *q, double *f, int n, int p, int mu, double abar, double Lx, double Ly,
                         double Lz);

/* Open-boundary direct sum on the GPU: *ureal = 1/2 sum_{i!=j} q_i q_j / r_ij, freal[3N] = gradient of the
 * potential at every particle (acc_i = -sum_j q_j (r_i - r_j) / r_ij^3), host pointers.
 * replaces direct(), declared at msm/msmLibrary.h:212 and defined nowhere in the reference; the energy is the
 * computation of msm/nonperiodic.c:12-22 (which forces q = 1). */
void direct(int N, double *r, double *q, double *ureal, double *freal);

/* Open-boundary MSM on the GPU: *uappr = approximation of 1/2 sum_{i!=j} q_i q_j / r_ij, fappr[3N] = gradient of the
 * potential at every particle (the sign and weighting of direct() above), grid spacing h, cutoff a, B-spline order p,
 * quasi-interpolation width mu, smoothness s of the splitting (must equal p: the splitting of msmPeriodic.c:22-34).
 * replaces msm3d(), declared at msm/msmLibrary.h:216 and defined nowhere in the reference: no reference output exists
 * for it; it is checked against direct() within the method's accuracy and against the CPU oracle to 1e-12.
 * Of the returned record N, L, mu, p, s, h, hx..hz, a, aL, the grid sizes and the potential_* fields are filled. */
msm_stats msm3d(int N, double *r, double *q, int p, double h, double a, int mu, int s, double *uappr, double *fappr);

/* Ewald sum of a periodic system on the GPU, host pointers: the total energy, with the three report lines on stderr.
 * replaces ewaldClassical(), msm/ewaldClassical.h:6, defined at msm/ewaldClassical.c:5-109 (real sum over the
 * particles as given, no images; reciprocal sum over kx^2+ky^2+kz^2 < cutoffFourier^2).  Returns NaN after
 * printing the reason if no GPU is visible or a HIP call fails. */
double ewaldClassical(double *r, double *q, int n, double Lx, double Ly, double Lz, double alpha, double cutoffDirect,
                      double cutoffFourier);
/* replaces ewaldClassicalRunner(), msm/ewaldClassical.h:5, msm/ewaldClassical.c:111-123: data file in, total on stdout */
void ewaldClassicalRunner(char *dataFile, double alpha, double cutoffDirect, double cutoffFourier);

/* parameter helpers, msm/msmLibrary.h:191-192,238-247 / msm/msmLibrary.c:3,31,46,69,110,283 */
double mylog2(double x);
double choosebetaNew(double epsilon, double aL, double hL);
double choosekmax(double epsilon, double beta, double hL);
int chooseM(double htilde, double Ax);
int chooseL(int Mx, int My, int Mz, int N);
int minOf3Integers(int a, int b, int c);
/* msm/msmLibrary.h:245 / msmLibrary.c:7; unlike the reference a missing file returns 1 instead of crashing */
int data_read(char *filename, double **q, double **r, int *N, double *Lx, double *Ly, double *Lz);
/* msm/msmLibrary.h:214 / msmLibrary.c:125 (host, one pair; the GPU pair kernels are its batched form) */
double short_range_periodic(int i, int j, double *r, double *q, int s, double a, double *fshort, double Ax, double Ay,
                            double Az);
/* msm/msmLibrary.h:215 / msmLibrary.c:290 */
void print_stats(msm_stats stats);
/* msm/msmLibrary.h:199-204 / msmLibrary.c:183-281 */
LinkedList *msm4g_linkedlist_new();
void msm4g_linkedlist_add(LinkedList *list, int data);
void msm4g_linkedlist_addvalue(LinkedList *list, int data, double value);
int msm4g_linkedlist_search(LinkedList *list, int data);
double msm4g_linkedlist_getvalue(LinkedList *list, int index);
void msm4g_linkedlist_destroy(LinkedList *list);

/* ------------------------------------------------------------------ */
/* (2) handle layer (no counterpart in the reference; it is what       */
/*     msmperiodic() is built from)                                    */
/* ------------------------------------------------------------------ */

enum {
    MSM_AMD_OK = 0,
    MSM_AMD_EINVAL = -1,  /* bad argument / unsupported parameter combination */
    MSM_AMD_ENODEV = -2,  /* no usable HIP device */
    MSM_AMD_EHIP = -3,    /* a HIP runtime call failed */
    MSM_AMD_ENOMEM = -4,
    MSM_AMD_ERANGE = -5   /* size beyond what the selected mode supports */
};

/* how msmperiodic() maps onto the engine */
enum {
    MSM_AMD_MODE_AUTO = 0,    /* literal when N <= 4096 (the only regime the reference algorithm can run), else scalable */
    MSM_AMD_MODE_LITERAL = 1, /* reproduce msmPeriodic.c stage by stage, index conventions included */
    MSM_AMD_MODE_SCALABLE = 2 /* consistent indexing, multilevel, O(N) */
};
void msm_amd_set_mode(int mode);
int msm_amd_get_mode(void);
/* Without a call of msm_amd_set_mode(), the environment variable MSM_AMD_MODE = literal | scalable | auto chooses (for
 * drivers that are built unchanged).  The two modes return different quantities in fappr: literal mode the reference's
 * (msmPeriodic.c:1158-1160 and 1364-1366: with the charge of particle i, and r shifted by +0.5 in the caller's array), scalable mode
 * the acceleration of a unit charge at r_i.  msm_amd_last_call_mode() tells which one the last msmperiodic() call of
 * the process used (MSM_AMD_MODE_LITERAL or MSM_AMD_MODE_SCALABLE; -1 before the first call); in auto mode the first
 * call that takes the scalable path says so on stderr unless msm_amd_set_quiet(1). */
int msm_amd_last_call_mode(void);
/* msmperiodic(), msmPeriodicRunner() and msm3d() keep the plan of their last call (stencils and device buffers) and
 * reuse it when the next call needs the same one, and direct() keeps its device copies of the caller's arrays; this
 * frees both. */
void msm_amd_release_cached_plan(void);
/* number of plans those three entry points have built in this process (a call that reuses the kept plan adds nothing) */
long msm_amd_plan_builds(void);
/* quiet != 0 suppresses the stencil / grid-charge dumps msmperiodic() prints in literal mode (msmPeriodic.c:600-610,723-733) */
void msm_amd_set_quiet(int quiet);

/* Where msm_amd_create builds the grid-kernel stencils K_l and K_L (the sl[] / sL of msmPeriodic.c:344-632, 99 % of
 * the reference's run time): on the GPU (default; msm_stencil.hip) or on the host.  Same sums in the same order, so
 * the two agree to a few 1e-16 relative.  Process-wide, read when a handle is created; the environment variable
 * MSM_AMD_STENCILS=host|gpu overrides it.  msm_amd_hostplan_create always uses the host. */
enum { MSM_AMD_STENCILS_HOST = 0, MSM_AMD_STENCILS_DEVICE = 1 };
void msm_amd_set_stencil_backend(int where);
int msm_amd_get_stencil_backend(void);

/* what the acceleration output contains */
enum {
    MSM_AMD_ACC_GRADIENT = 0, /* acc_i = sum_j q_j g0'(r) rhat + sum_m e_m grad phi_m(r_i)  (msmSimple.c:846,1330,1370) */
    MSM_AMD_ACC_LITERAL = 1,  /* fshort_i = sum_j (q_i+q_j)/2 g0' rhat (msmLibrary.c:168-176); flong_i = q_i * grid part (msmPeriodic.c:1158) */
    MSM_AMD_ACC_FORCE = 2     /* electrostatic force  F_i = -q_i * gradient form */
};

typedef struct msm_amd_plan_desc {
    int N;          /* particles this handle evaluates (owned + halo on a slab rank) */
    int p;          /* B-spline order, 4 or 6 */
    int mu;         /* quasi-interpolation width */
    int L;          /* number of real-space grid levels (>= 1) */
    int kmax;       /* Fourier cutoff of the top level */
    int acc_convention;
    int M[3];       /* level-0 grid */
    int zplane0;    /* first level-0 z plane owned by this rank (0 for single GPU) */
    int nzplanes;   /* number of owned level-0 z planes (M[2] for single GPU) */
    int n_owned;    /* particles [0, n_owned) are owned, the rest are halo copies; 0 means all N */
    double A[3];    /* periodic cell */
    double a;       /* short-range cutoff */
    double beta;    /* Ewald splitting of the top level */
    /* Open boundaries (the non-periodic MSM declared as msm3d at msm/msmLibrary.h:216): the cell [origin, origin + A)
     * is a padded box around the particles (msm_amd_plan_open sizes it), stencils have no image sums, the coarsest
     * level applies gamma(r / a_l) / a_l over every node difference without wrap-around, and there is no Fourier
     * level (kmax, beta unused), no csr and no self-image term.  0 = periodic (everything above). */
    int open_boundary;
    double origin[3];  /* position of the cell's corner: the engine works with r - origin (0 for periodic plans) */
    double extent[3];  /* open plans: size of the particles' bounding box (sizes the sort's columns); 0 = the whole cell */
} msm_amd_plan_desc;

/* The reference's parameter choices (msmPeriodic.c:147-263) for L levels:
 * htilde=(V/N)^(1/3), M=chooseM per axis, a=abar*htilde, beta=choosebetaNew(1e-12,2^L a,2^(L-1)hx),
 * kmax=ceil|choosekmax|.  L<=0 selects min(allowed, wanted) of msmLibrary.c:46-62 capped so that the
 * coarsest grid has at least 8 nodes per axis. */
int msm_amd_plan_auto(int N, double Ax, double Ay, double Az, int p, int mu, double abar, int L, msm_amd_plan_desc *out);

/* Open-boundary plan for particles inside the box [lo, hi]: grid spacing h on every axis, cutoff a, L levels
 * (L <= 0: as many as keep the coarsest grid at 8..32 nodes per axis).  The cell is the box plus the padding that
 * keeps every level below the coarsest clear of the wrap-around, (p/2 + 1) coarsest-grid spacings of room for the
 * spline supports, and at least 3 cutoffs per axis; M = A / h exactly, a multiple of 2^(L-1). */
int msm_amd_plan_open(int N, const double lo[3], const double hi[3], int p, int mu, double h, double a, int L,
                      msm_amd_plan_desc *out);

/* 1 when the open-boundary plan *plan still serves particles inside the box [lo, hi] (they moved): the box lies inside
 * the padded cell, (p/2 + 1) grid spacings clear of its faces, and the cell leaves the room msm_amd_plan_open would give
 * a box of this extent; 0 otherwise (make a new plan).  No counterpart in the reference. */
int msm_amd_plan_open_holds(const msm_amd_plan_desc *plan, const double lo[3], const double hi[3]);

typedef struct msm_amd_handle msm_amd_handle;

/* Builds omega', all stencils (position independent; see msm_amd_set_stencil_backend) and allocates every device buffer.
 * stream: a hipStream_t (NULL = the null stream) on which every later launch of this handle is queued. */
int msm_amd_create(const msm_amd_plan_desc *plan, int device, void *stream, msm_amd_handle **out);
/* Same, but every stage follows msmPeriodic.c's own index conventions (needs L=1, p=4, mu<=11); this is what
 * msmperiodic() runs in literal mode. */
int msm_amd_create_literal(const msm_amd_plan_desc *plan, int device, void *stream, msm_amd_handle **out);
void msm_amd_destroy(msm_amd_handle *h);

enum { MSM_AMD_U_SHORT_REAL = 0, MSM_AMD_U_SHORT_SELF, MSM_AMD_U_SHORT_CSR, MSM_AMD_U_LONG_REAL, MSM_AMD_U_LONG_FOUR,
       MSM_AMD_U_LONG_SELF, MSM_AMD_U_TOTAL, MSM_AMD_U_COUNT };

/* One energy + acceleration evaluation, everything device resident.
 * d_r[3N] AoS, d_q[N], d_acc[3N] device pointers; energies[MSM_AMD_U_COUNT] host out (may be NULL: then
 * nothing is copied back and the call does not synchronise). */
int msm_amd_eval(msm_amd_handle *h, const double *d_r, const double *d_q, double *d_acc, double *energies);
/* Same through host buffers (H2D 32 B/particle + D2H 24 B/particle included). */
int msm_amd_eval_host(msm_amd_handle *h, const double *r, const double *q, double *acc, double *energies);
/* msm_amd_eval without the host synchronisation: the evaluation and a copy of its MSM_AMD_U_COUNT energies into
 * host_energies (page-locked host memory, e.g. hipHostMalloc; one slot per evaluation in flight) are queued on the
 * handle's stream and the call returns.  The caller synchronises the stream before reading acc or the energies.
 * Lets consecutive evaluations queue back to back (a time-stepping loop that reads energies every few steps). */
int msm_amd_eval_enqueue(msm_amd_handle *h, const double *d_r, const double *d_q, double *d_acc, double *host_energies);

/* msm_amd_eval for a sequence of evaluations whose positions change little from one to the next (a time-stepping
 * loop).  No counterpart in the reference.  The handle keeps a frozen cluster order with, per group of 4 particles, the
 * list of the particles that were within a + skin of one of them when the lists were last built; an evaluation moves
 * the kept positions along with the caller's and sums over those lists (no box test, three lanes in four inside the
 * cutoff).  Whether any particle has moved more than skin / 2 since then is checked ON THE DEVICE at the start of every
 * call, and if so the same call rebuilds the lists before it uses them (the kernels of the rebuild are always queued and
 * return at once otherwise), so the result never depends on the host seeing the flag.  The result equals
 * msm_amd_eval's to rounding, not bit for bit: the sums run in the order of the last rebuild.  The first call
 * allocates (about 1.2 kB per particle at skin = 0.1 a); later calls with a skin up to that one allocate nothing.
 * Needs a periodic plan that owns every particle and plane (no slab rank, not the literal mode), 0 <= skin <= a, cell
 * edges of at least 3 (a + skin).  energies as in msm_amd_eval (NULL: queued, no synchronisation). */
int msm_amd_eval_reuse(msm_amd_handle *h, const double *d_r, const double *d_q, double *d_acc, double *energies, double skin);
/* evaluations made through msm_amd_eval_reuse and list rebuilds among them (reads a device counter: synchronises) */
int msm_amd_reuse_stats(msm_amd_handle *h, long *evaluations, long *rebuilds);

/* ---- stage entry points (device pointers unless noted) ---- */
/* bin particles; must precede pairs/anterpolate/interpolate of the same positions */
int msm_amd_stage_bin(msm_amd_handle *h, const double *d_r, const double *d_q);
/* d_acc[3N] <- short-range part (per acc_convention); partial energy kept in the handle */
int msm_amd_stage_pairs(msm_amd_handle *h, double *d_acc);
/* level-0 grid charges of planes [zplane0, zplane0+nzplanes) */
int msm_amd_stage_anterpolate(msm_amd_handle *h);
/* restriction -> per-level convolutions -> top level -> prolongation on the whole grids held by the handle */
int msm_amd_stage_grid(msm_amd_handle *h);
/* d_acc[3N] += grid part */
int msm_amd_stage_interpolate(msm_amd_handle *h, double *d_acc);
/* reductions; energies[MSM_AMD_U_COUNT] host out */
int msm_amd_stage_energies(msm_amd_handle *h, double *energies);

/* ---- slab ranks (one process per GPU; collectives are the caller's) ---- */
/* particles of the next evaluation: n_total <= plan N (the capacity), the first n_owned of them owned */
int msm_amd_set_counts(msm_amd_handle *h, int n_total, int n_owned);
/* the level-0 z planes [zplane0, zplane0 + nzplanes) the rank owns from the next evaluation on (the slabs of an
 * open-boundary plan are cut where the particles are, so they move with them); the buffers keep their size */
int msm_amd_set_planes(msm_amd_handle *h, int zplane0, int nzplanes);
/* [pair sum, q0.e0 over owned planes, qL.efour, sum q, sum q^2 over owned particles]: additive over ranks
 * (except the third, which is replicated).  Left on the device; host_out5 != NULL copies them back. */
int msm_amd_stage_raw_sums(msm_amd_handle *h, double *host_out5);
void *msm_amd_raw_sums_device_ptr(msm_amd_handle *h);
/* host arithmetic: energies from raw sums that were added over all ranks */
int msm_amd_combine_energies(msm_amd_handle *h, const double sums[5], double energies[MSM_AMD_U_COUNT]);

/* ---- one evaluation of a slab rank, collectives included (the host stays in C) ----
 * The exchange steps of the z-slab decomposition go through a small table of collectives, so that the same C code runs
 * over RCCL (msm_amd_rccl_create below: ncclAllGather / ncclSend + ncclRecv / ncclAllReduce on the handle's own
 * streams), over MPI on device pointers, or over whatever a test harness plugs in.  Every function queues its work
 * on `stream` (a hipStream_t) and returns 0 on success; buffers are device pointers of doubles.  No counterpart in the
 * reference, which is serial and contains no collective call. */
typedef struct msm_amd_collectives {
    int rank, world;
    void *ctx;
    /* recv[k * count .. (k + 1) * count) <- the `count` doubles of rank k; send may be recv + rank * count (in place) */
    int (*all_gather)(void *ctx, const double *send, double *recv, size_t count, void *stream);
    /* buf[0 .. count) <- sum over the ranks, in place */
    int (*all_reduce_sum)(void *ctx, double *buf, size_t count, void *stream);
    /* neighbour exchange of `count` doubles each way: send_up goes to rank `up` (= rank + 1, periodic), send_down to rank
     * `down`; recv_from_down receives what `down` sent up, recv_from_up what `up` sent down.  With two ranks up == down. */
    int (*halo_exchange)(void *ctx, const double *send_up, const double *send_down, double *recv_from_down,
                         double *recv_from_up, size_t count, int up, int down, void *stream);
} msm_amd_collectives;

/* RCCL implementation of the table (librccl.so is loaded on first use; MSM_AMD_ENODEV if it cannot be).
 * msm_amd_rccl_unique_id: rank 0 calls it and hands the 128 bytes to the other ranks by any means (a file, MPI_Bcast,
 * torch.distributed); every rank then calls msm_amd_rccl_create with the device it will run on current. */
int msm_amd_rccl_unique_id(char id[128]);
int msm_amd_rccl_create(const char id[128], int rank, int world, msm_amd_collectives **out);
void msm_amd_rccl_destroy(msm_amd_collectives *c);
/* 0 when librccl.so loads in this process and has every entry the table needs, else MSM_AMD_ENODEV: a probe that every
 * rank can make (and agree on) before any of them enters the collective msm_amd_rccl_create. */
int msm_amd_rccl_available(void);
/* what the communicator behind a table of msm_amd_rccl_create reports: ncclCommCount and ncclCommUserRank */
int msm_amd_rccl_comm_info(const msm_amd_collectives *c, int *ranks, int *rank);

/* How the level-0 charge planes reach the ranks that need them. */
enum {
    MSM_AMD_SLAB_AUTO = 0,      /* halo exchange when it is possible and moves less than the all-gather */
    MSM_AMD_SLAB_ALLGATHER = 1, /* every rank gets the whole level-0 charge grid (one collective; coarse levels solved redundantly) */
    MSM_AMD_SLAB_HALO = 2,      /* p/2 + (stencil half width) planes from each z neighbour, level-1 charges all-gathered;
                                   needs L >= 2, an even number of owned planes, and halos that fit the neighbours' slabs */
    MSM_AMD_SLAB_REDUCE = 3     /* open-boundary plans, whose slabs are cut where the particles are and so differ in size: every
                                   rank clears the planes it does not own and the level-0 charge grid is summed over the
                                   ranks (adding zeros is exact, so this is a gather); chosen by itself for such plans */
};
/* What msm_amd_slab_step would do for this handle and world size under `mode` (one of the two concrete values), or a
 * negative error when `mode` asks for a halo exchange the plan cannot do. */
int msm_amd_slab_exchange_mode(msm_amd_handle *h, int world, int mode);

/* One evaluation of the particles this rank holds (msm_amd_set_counts: the first n_owned of them owned, the rest halo
 * copies): sort -> anterpolation of the owned planes -> [exchange of the level-0 charges on a second stream] beside
 * [neighbour lists and pair sum on the handle's stream] -> grid levels -> interpolation -> raw sums -> all-reduce.
 * d_acc[3 n_total] receives the accelerations of the owned particles.  The call queues everything and returns; the reduced sums stay on the device until
 * msm_amd_slab_energies reads them.  coll == NULL (or world == 1 with a full-grid handle): no exchange, the handle
 * must own every plane. */
int msm_amd_slab_step(msm_amd_handle *h, const msm_amd_collectives *coll, const double *d_r, const double *d_q, double *d_acc,
                      int mode);
/* synchronises and turns the reduced sums of the last msm_amd_slab_step into (A)..(F) and the total */
int msm_amd_slab_energies(msm_amd_handle *h, const msm_amd_collectives *coll, double energies[MSM_AMD_U_COUNT]);

/* device pointers / copies of internal state (for the slab all-gather and the tests) */
enum { MSM_AMD_GRID_Q = 0, MSM_AMD_GRID_E = 1, MSM_AMD_GRID_EFOUR = 2 };
void *msm_amd_grid_device_ptr(msm_amd_handle *h, int which, int level);
int msm_amd_grid_dims(msm_amd_handle *h, int level, int dims[3]);
int msm_amd_copy_grid(msm_amd_handle *h, int which, int level, double *host_out);
int msm_amd_set_grid(msm_amd_handle *h, int which, int level, const double *host_in);
/* stencil of level l (host copy, x fastest); level == L selects the Fourier top stencil */
int msm_amd_stencil_info(msm_amd_handle *h, int level, int size[3], int first[3]);
int msm_amd_copy_stencil(msm_amd_handle *h, int level, double *host_out);
int msm_amd_copy_wprime(msm_amd_handle *h, double *host_out, int *len);
/* single grid operators on caller-supplied device arrays (parity tests of one operator) */
int msm_amd_op_restrict(msm_amd_handle *h, int level, const double *d_fine, double *d_coarse);
int msm_amd_op_prolong_add(msm_amd_handle *h, int level, const double *d_coarse, double *d_fine);
int msm_amd_op_convolve(msm_amd_handle *h, int level, const double *d_q, double *d_e);
/* per-kernel wall time of the last msm_amd_eval in milliseconds (HIP events on the handle's stream):
 * [bin, pairs, anterpolation, restriction, convolution, top, prolongation, interpolation, reductions, total] */
int msm_amd_last_timings(msm_amd_handle *h, double ms[10]);
/* duration in milliseconds of one named kernel inside the last profiled msm_amd_eval, bracketed by its own pair
 * of HIP events: "pair_kernel" (the cluster-pair kernel alone, without the sort and the list build that the
 * "pairs" stage above includes).  MSM_AMD_EINVAL for an unknown name or when that kernel did not run. */
int msm_amd_last_kernel_ms(msm_amd_handle *h, const char *name, double *ms);
/* Profiling hook with no counterpart in the reference: d_buffer is device memory of 4 x 64-bit words per cluster slot
 * (msm_amd_pair_trace_slots() slots) which the list-walking pair kernel fills, per wave, with {start, end (100 MHz
 * clock), hardware id | XCD id << 32, kind (1 empty slot, 2 halo copies only, 3 evaluated)}; NULL switches it off. */
int msm_amd_set_pair_trace(msm_amd_handle *h, void *d_buffer);
long msm_amd_pair_trace_slots(msm_amd_handle *h);
/* enable per-stage event timing (adds event records to the stream); off by default */
void msm_amd_set_profiling(msm_amd_handle *h, int on);
/* select kernel variants (for A/B benchmarking); key/value, returns MSM_AMD_EINVAL for unknown keys */
int msm_amd_set_option(msm_amd_handle *h, const char *key, int value);

/* Host-only view of a plan (no device needed): omega', J, stencils.  Used to validate the position-independent
 * setup on machines without a GPU and to export stencils. */
typedef struct msm_amd_hostplan msm_amd_hostplan;
int msm_amd_hostplan_create(const msm_amd_plan_desc *plan, int literal, msm_amd_hostplan **out);
void msm_amd_hostplan_destroy(msm_amd_hostplan *hp);
int msm_amd_hostplan_stencil_info(msm_amd_hostplan *hp, int level, int size[3], int first[3]);
int msm_amd_hostplan_copy_stencil(msm_amd_hostplan *hp, int level, double *host_out);
int msm_amd_hostplan_copy_wprime(msm_amd_hostplan *hp, double *host_out, int *len);
/* scalars: [csr, self_image_sum, gamma(0), chisum, J0, J1, J2, J3] */
int msm_amd_hostplan_scalars(msm_amd_hostplan *hp, double out[8]);

/* Writes the two stencil cache files of msmSimple into directory dir ("" or NULL: the working directory), in its own
 * names and format (msm/msmSimple.c:1018-1031: stencil_l00_Mx.._My.._Mz.._nu.._mu...dat; :1196-1209:
 * stencil_fourier_Mx.._My.._Mz.._nu...dat; one %25.16e value per line, x slowest), for a periodic plan with L = 1 -
 * msmSimple's two-level method.  An unmodified msmSimple started in that directory with the same grid, nu and mu reads
 * them instead of spending its run time on the sums.  The first form needs no GPU; the second writes what the
 * handle's stencils (GPU-built by default) hold. */
int msm_amd_hostplan_write_msmsimple_cache(msm_amd_hostplan *hp, const char *dir);
int msm_amd_write_msmsimple_cache(msm_amd_handle *h, const char *dir);

/* device-resident direct sum for particles [i0, i1) against all N (a rank of a multi-GPU run takes a slice of i
 * and the slice energies add up): d_acc[3 (i1 - i0)], d_work[msm_amd_direct_work_doubles(N)], stream = hipStream_t;
 * the slice energy is left in d_work[0] and copied to *u when u != NULL (synchronises).  No counterpart in the
 * reference beyond direct() above. */
int msm_amd_direct_work_doubles(int N);
int msm_amd_direct_device(int N, const double *d_r, const double *d_q, int i0, int i1, int acc_convention, double *d_acc,
                          double *d_work, void *stream, double *u);

/* msm_amd_direct_device for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same kernels and the same bits as
 * msm_amd_direct_device, but the per-workgroup energy partials (its d_work argument) live in a 64 KiB array inside
 * the library's code object, one copy per GPU, held until the process ends; the call makes no hipMalloc / hipFree
 * and does not synchronise.  d_u (device, 1 double) receives the slice energy; d_acc[3 (i1 - i0)] as above; the
 * device that owns the pointers and the stream must be current.  A slice i1 - i0 beyond
 * msm_amd_direct_noalloc_max_slice() (4 194 304 particles) is refused with MSM_AMD_ERANGE: split it.  Calls that
 * overlap on one GPU share the array and race; calls queued on one stream do not. */
int msm_amd_direct_noalloc_max_slice(void);
int msm_amd_direct_device_noalloc(int N, const double *d_r, const double *d_q, int i0, int i1, int acc_convention,
                                  double *d_acc, double *d_u, void *stream);

/* Device-resident Ewald sum: potential d_phi[i1 - i0] and its gradient d_grad[3 (i1 - i0)] (may be NULL) at
 * particles [i0, i1) of the N in d_r/d_q, cell edges A[3].  mode MSM_AMD_EWALD_FULL: nearest image inside rcut
 * (rcut <= half the shortest edge), every |k_d| <= ceil(kcut): the converged sum that accuracy figures are
 * quoted against (energy as msm/ewaldClassical.c:5-109 with the image handling its comments leave out, gradient
 * as tabulated by msm/ewaldHernquist.c:113-172).  MSM_AMD_EWALD_LITERAL: the sums exactly as ewaldClassical()
 * takes them.  d_work needs msm_amd_ewald_work_doubles(N, kcut, i1 - i0) doubles; on return d_work[0..4] hold
 * the shares of particles [i0, i1) in the [real, reciprocal, self, background, total] energy (1/2 q_i phi_i summed),
 * copied to u5 when u5 != NULL (synchronises the stream). */
#define MSM_AMD_EWALD_FULL 0
#define MSM_AMD_EWALD_LITERAL 1
long long msm_amd_ewald_work_doubles(int N, double kcut, int ni);
int msm_amd_ewald_device(int N, const double *d_r, const double *d_q, const double A[3], double alpha, double rcut, double kcut,
                         int mode, int i0, int i1, double *d_phi, double *d_grad, double *d_work, void *stream, double *u5);

const char *msm_amd_last_error(void);
int msm_amd_device_count(void);
const char *msm_amd_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MSM_AMD_H */
