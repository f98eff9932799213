/* CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's MSM algorithm (hkayabilisim/msm,
 * files under msm/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (msm_amd/) never does.
 *
 * Parity pin: every function here is checked against (i) the golden fixtures
 * in tests/golden/ that were produced by running the real reference
 * (oracle/gen_golden.py) and (ii) oracle/_ref/libmsmref.so when present.
 * Unpinned by any reference output: multilevel (L>1) grids and p=6 inside the
 * msmperiodic pipeline — the reference forces L=1 (msmLibrary.c:64-66) and
 * rejects p!=4 (msmPeriodic.c:100-104).  Those paths are pinned only by the
 * 2-level msmSimple fixtures (v=4/6) and by Ewald physics.
 */
#ifndef MSM_ORACLE_H
#define MSM_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define OR_PI 3.14159265358979323846 /* msmLibrary.h:15 */
#define OR_MAXL 10

/* ---- parameter helpers (msmLibrary.c) ---- */
double or_mylog2(double x);                                       /* msmLibrary.c:3 */
int or_chooseM(double htilde, double A);                          /* msmLibrary.c:69 */
int or_chooseL(int Mx, int My, int Mz, int N);                    /* msmLibrary.c:46 (forced to 1) */
int or_chooseL_formula(int Mx, int My, int Mz, int N, int *allowed, int *wanted);
double or_choosebetaNew(double eps, double aL, double hL);        /* msmLibrary.c:31 */
double or_choosekmax(double eps, double beta, double hL);         /* msmLibrary.c:110 */
int or_minOf3(int a, int b, int c);                               /* msmLibrary.c:283 */
int or_data_read(const char *fn, double **q, double **r, int *N, double *Lx, double *Ly, double *Lz);

/* ---- kernels pieces ---- */
double or_bspline(int k, double u);                               /* msmPeriodic.c:10 */
double or_bspline_prime(int k, double u);                         /* msmSimple.c:2057 */
void or_bspline_weights(int p, double t, double *w, double *dw);  /* iterative equivalent */
double or_gamma_lit(double rho, int p);                           /* msmPeriodic.c:22 */
double or_gamma(double rho, int p);                               /* msmSimple.c:609 */
double or_gamma_prime(double rho, int p);                         /* msmSimple.c:624 */
double or_short_range_periodic(int i, int j, const double *r, const double *q, int s, double a, double *fshort,
                               double Ax, double Ay, double Az);  /* msmLibrary.c:125 */
int or_wprime(int p, int mu, double *w);                          /* msmSimple.c:334; w[0..mu+p/2] */
void or_jcoef(int p, double *J);                                  /* msmLibrary.h:152; J[0..p/2] */
void or_phi_at_knots(int p, double *phi);                         /* msmPeriodic.c:71; phi[0..p/2-1] */

/* ---- literal mirror of msmperiodic() (msmPeriodic.c:52-1421) ---- */
typedef struct or_literal_result {
    int N, p, mu, L, kmax, Mx, My, Mz;
    int slmin[3], slmax[3], slsize[3];
    int sLn[3];
    int chisize, pad_;
    double abar, Ax, Ay, Az, htilde, hx, hy, hz, a, aL, beta, csr, chisum;
    double u_short_real, u_short_self, u_short_csr, u_long_real, u_long_four, u_long_self, u_total;
    double netcharge, netgridcharge;
    double *sl;     /* level-0 stencil, slsize[0]*slsize[1]*slsize[2], x fastest */
    double *sL;     /* top stencil, sLn^3 */
    double *qgrd;   /* level-0 grid charges Mx*My*Mz */
    double *edir;   /* sl (*) q  (with the reference's index quirks) */
    double *etop;   /* sL (*) q  before it is discarded (msmPeriodic.c:990) */
    double *cvecx, *cvecy, *cvecz, *chi;
    double *fshort, *flong, *facc; /* 3N each, AoS */
    double *rshift; /* positions after the +0.5 shift (msmPeriodic.c:75) */
} or_literal_result;

/* returns 0 on success, 1 if p != 4 (the reference returns early then) */
int or_literal_run(int N, const double *r, const double *q, int p, double abar, int mu, double Ax, double Ay, double Az,
                   or_literal_result *out);
void or_literal_free(or_literal_result *res);

/* ---- specification engine: same mathematics with consistent indexing,
 *      any number of grid levels, p in {4,6}, any mu, O(N) algorithms ---- */
typedef struct or_spec_params {
    int N, p, mu, L, kmax, nthreads;
    int M[3];
    double A[3];
    double a, beta, eps;
} or_spec_params;

/* the reference's parameter choices (msmPeriodic.c:147-263) generalised to L levels */
int or_spec_auto_params(int N, double Ax, double Ay, double Az, int p, int mu, double abar, int L, or_spec_params *out);

typedef struct or_spec_plan or_spec_plan;
/* image range of the stencil sums: 0 = complete (default), R = -R..R only (msmPeriodic.c:536-538 uses 5) */
void or_spec_set_image_range(int R);
/* on != 0: plans created from now on are open-boundary plans (msm3d of msm/msmLibrary.h:216; see msm_spec.c) */
void or_spec_set_open_boundary(int on);
or_spec_plan *or_spec_plan_create(const or_spec_params *prm);
void or_spec_plan_destroy(or_spec_plan *pl);
/* stencil of grid level l: values (x fastest), extents and first offset per dimension */
const double *or_spec_stencil(const or_spec_plan *pl, int l, int size[3], int first[3]);
const double *or_spec_stencil_top(const or_spec_plan *pl, int size[3]);
const double *or_spec_wprime(const or_spec_plan *pl, int *len);
/* grids of the last evaluation */
const double *or_spec_grid_q(const or_spec_plan *pl, int l, int dims[3]);
const double *or_spec_grid_e(const or_spec_plan *pl, int l, int dims[3]);
const double *or_spec_grid_efour(const or_spec_plan *pl, int dims[3]);

enum { OR_U_SHORT_REAL = 0, OR_U_SHORT_SELF, OR_U_SHORT_CSR, OR_U_LONG_REAL, OR_U_LONG_FOUR, OR_U_LONG_SELF, OR_U_TOTAL, OR_U_COUNT };

/* acc (3N): sum_j q_j g0'(r) rhat + sum_m e_m grad phi_m   (msmSimple.c:846-848,1330-1332,1370-1372)
 * fshort_unw (3N, may be NULL): the same pair sum without the q_j weight
 * along (3N, may be NULL): the grid part of acc alone
 * seconds (may be NULL): per-stage wall time [pair, anterp, restrict, conv, top, prolong, interp, total] */
int or_spec_eval(or_spec_plan *pl, const double *r, const double *q, double *acc, double *fshort_unw, double *along,
                 double *energies, double *seconds);
/* stage entry points used for stage-wise parity tests */
void or_spec_pairs(const or_spec_plan *pl, const double *r, const double *q, double *acc_w, double *acc_unw, double *u_real);
void or_spec_pairs_ex(const or_spec_plan *pl, const double *r, const double *q, double *acc_w, double *acc_unw,
                      double *u_real, double *phi);
void or_spec_anterpolate(const or_spec_plan *pl, const double *r, const double *q, double *grid);
void or_spec_restrict(const or_spec_plan *pl, int l, const double *fine, double *coarse);
void or_spec_prolong_add(const or_spec_plan *pl, int l, const double *coarse, double *fine);
void or_spec_convolve(const or_spec_plan *pl, int l, const double *q, double *e);
void or_spec_convolve_top(const or_spec_plan *pl, const double *q, double *e);
void or_spec_interpolate(const or_spec_plan *pl, const double *r, const double *e, double *along);

/* ---- Ewald sum with full image / reciprocal loops (accuracy denominator) ---- */
/* energy and accelerations (msmSimple convention) of the first nsub particles against all N */
double or_ewald(int N, const double *r, const double *q, double Ax, double Ay, double Az, double alpha, double rcut,
                int kcut, int nsub, double *acc, int nthreads);
/* literal restatement of ewaldClassical() (ewaldClassical.c:5-109): no image loop */
double or_direct(int N, const double *r, const double *q, int nsub, double *acc, int nthreads);
double or_ewald_classical(const double *r, const double *q, int N, double Lx, double Ly, double Lz, double alpha,
                          double cutoffDirect, double cutoffFourier);

#ifdef __cplusplus
}
#endif
#endif
