// gfx950 (MI355X) kernels of the MSM hot path.  All arithmetic is FP64.
//
//   binning        k_bin_*            counting sort, deterministic order inside a bin
//   short range    k_pair_cells       cell list, LDS-staged neighbour tiles (msmLibrary.c:125-182 batched)
//                  k_pair_allimages   small boxes: every pair, every image inside the cutoff
//   anterpolation  k_anterp_gather    one thread per grid node gathers its particles (msmPeriodic.c:688-717)
//   restriction    k_restrict_axis    separable, one axis per pass (msmPeriodic.c:754-838)
//   convolution    k_conv_direct / k_conv_tiled / k_conv_literal (msmPeriodic.c:846-918, 936-998)
//   prolongation   k_prolong_axis     separable (msmPeriodic.c:1018-1108)
//   interpolation  k_interp           one thread per particle (msmPeriodic.c:1116-1161)
//   reductions     k_dot_partials, k_charge_partials, k_finalize_energies (msmPeriodic.c:1305-1359)
#include "msm_kernels.hpp"
#include <algorithm>
#include "msm_device.hpp"

#include "msm_amd.h"

namespace msm {

thread_local hipEvent_t tail_event = nullptr;

// ---------------------------------------------------------------------------
// binning
// ---------------------------------------------------------------------------
__device__ __forceinline__ void frac_coords(const double *__restrict__ r, int i, const BinGeom &g, double s[3]) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double t = (r[3 * (size_t)i + d] - g.origin[d]) * g.invA[d];
        t -= floor(t);
        s[d] = t;
    }
}
__device__ __forceinline__ int bin_index(const double s[3], const BinGeom &g) {
    int b[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int v = (int)(s[d] * g.nb[d]);
        b[d] = v >= g.nb[d] ? g.nb[d] - 1 : v;
    }
    return (b[2] * g.nb[1] + b[1]) * g.nb[0] + b[0];
}

__global__ void k_bin_count(const double *__restrict__ r, int N, BinGeom g, int *__restrict__ bin_of, int *__restrict__ count,
                            int *__restrict__ arrival) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s[3];
    frac_coords(r, i, g, s);
    int b = bin_index(s, g);
    bin_of[i] = b;
    arrival[i] = atomicAdd(&count[b], 1); // place among the bin's members, in the (arbitrary) order the adds landed
}

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__global__ void k_scan_tile_sums(const int *__restrict__ in, int n, int *__restrict__ tile_sums) {
    __shared__ int red[SCAN_THREADS];
    int base = blockIdx.x * SCAN_TILE, s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int idx = base + k * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = SCAN_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = red[0];
}

// exclusive scan of tile_sums in place by one block (ntiles is small)
__global__ void k_scan_tile_offsets(int *__restrict__ tile_sums, int ntiles) {
    __shared__ int buf[SCAN_THREADS];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += SCAN_THREADS) {
        int idx = base + threadIdx.x;
        int v = idx < ntiles ? tile_sums[idx] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < SCAN_THREADS; off <<= 1) {
            int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = buf[threadIdx.x], c = carry;
        if (idx < ntiles) tile_sums[idx] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry = c + incl;
        __syncthreads();
    }
}

// out[i] = exclusive prefix of in, out[n] = total
__global__ void k_scan_apply(const int *__restrict__ in, int n, const int *__restrict__ tile_off, int *__restrict__ out) {
    __shared__ int buf[SCAN_THREADS];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    buf[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    int run = tile_off[blockIdx.x] + buf[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (base + k == n - 1) out[n] = run;
    }
}

__global__ void k_bin_scatter(const int *__restrict__ bin_of, const int *__restrict__ arrival, int N,
                              const int *__restrict__ start, int *__restrict__ slot_idx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    slot_idx[start[bin_of[i]] + arrival[i]] = i;
}

// The scatter above fills a bin in arrival order.  Rank every entry by its particle index inside
// its bin so that the stored order (hence every later summation order) is the same on every run.
__global__ void k_bin_finalize(const double *__restrict__ r, const double *__restrict__ q, int N, BinGeom g,
                               const int *__restrict__ bin_of, const int *__restrict__ start,
                               const int *__restrict__ slot_idx, int *__restrict__ order, double4 *__restrict__ ps) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    int i = slot_idx[s];
    int b = bin_of[i];
    int st = start[b], en = start[b + 1], rank = 0;
    for (int t = st; t < en; t++) rank += (slot_idx[t] < i);
    double f[3];
    frac_coords(r, i, g, f);
    int dst = st + rank;
    order[dst] = i;
    ps[dst] = make_double4(f[0] * g.scale[0], f[1] * g.scale[1], f[2] * g.scale[2], q[i]);
}

// out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total); scan_tmp holds n/SCAN_TILE + 1 ints
void launch_exclusive_scan(hipStream_t st, const int *in, int n, int *scan_tmp, int *out) {
    const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(SCAN_THREADS), 0, st, in, n, scan_tmp);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, st, scan_tmp, ntiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(SCAN_THREADS), 0, st, in, n, scan_tmp, out);
}

void launch_bin_scatter(hipStream_t st, const int *bin_of, const int *arrival, int N, const int *start, int *slot_idx) {
    const int tb = 256, gb = (N + tb - 1) / tb;
    hipLaunchKernelGGL(k_bin_scatter, dim3(gb), dim3(tb), 0, st, bin_of, arrival, N, start, slot_idx);
}

void launch_bin(hipStream_t st, const double *d_r, const double *d_q, int N, const BinGeom &g, int *bin_of, int *count,
                int *start, int *arrival, int *slot_idx, int *scan_tmp, int *order, double4 *ps) {
    const int nb = g.nbins;
    hipMemsetAsync(count, 0, sizeof(int) * (size_t)nb, st);
    const int tb = 256, gb = (N + tb - 1) / tb;
    hipLaunchKernelGGL(k_bin_count, dim3(gb), dim3(tb), 0, st, d_r, N, g, bin_of, count, arrival);
    launch_exclusive_scan(st, count, nb, scan_tmp, start);
    launch_bin_scatter(st, bin_of, arrival, N, start, slot_idx);
    hipLaunchKernelGGL(k_bin_finalize, dim3(gb), dim3(tb), 0, st, d_r, d_q, N, g, bin_of, start, slot_idx, order, ps);
}

// ---------------------------------------------------------------------------
// short range
// ---------------------------------------------------------------------------
// One in-cutoff pair.  d = r_i - r_j - A p.  g0(r) = 1/r - gamma(r/a)/a;  g0'(r)/r = -1/r^3 + D(y)/a^3 with
// D = -gamma'(rho)/rho  (msmLibrary.h:18-19 for p = 4).  Adds q_j g0 to u, q_j g0'/r d to fw and g0'/r d to fu.
template <int P, bool UNW>
__device__ __forceinline__ void pair_term(double dx, double dy, double dz, double r2, double qj, const PairParams &pp,
                                          double &u, double &fwx, double &fwy, double &fwz, double &fux, double &fuy,
                                          double &fuz) {
    const double ir = rsqrt(r2);
    const double y = r2 * pp.inv_a2 - 1.0;
    double gam = pp.cg[P - 1], dg = pp.cd[P - 2];
#pragma unroll
    for (int k = P - 2; k >= 0; k--) gam = fma(gam, y, pp.cg[k]);
#pragma unroll
    for (int k = P - 3; k >= 0; k--) dg = fma(dg, y, pp.cd[k]);
    const double g0 = ir - gam * pp.inv_a;
    const double s = dg * pp.inv_a3 - ir * ir * ir;
    u = fma(qj, g0, u);
    const double qs = qj * s;
    fwx = fma(qs, dx, fwx);
    fwy = fma(qs, dy, fwy);
    fwz = fma(qs, dz, fwz);
    if (UNW) {
        fux = fma(s, dx, fux);
        fuy = fma(s, dy, fuy);
        fuz = fma(s, dz, fuz);
    }
}

__device__ __forceinline__ void store_acc(double *__restrict__ acc, int orig, int convention, double qi, double fwx,
                                          double fwy, double fwz, double fux, double fuy, double fuz) {
    double ox, oy, oz;
    if (convention == MSM_AMD_ACC_LITERAL) { // msmLibrary.c:168-176 summed over both visits of a pair
        ox = 0.5 * (qi * fux + fwx); oy = 0.5 * (qi * fuy + fwy); oz = 0.5 * (qi * fuz + fwz);
    } else if (convention == MSM_AMD_ACC_FORCE) {
        ox = -qi * fwx; oy = -qi * fwy; oz = -qi * fwz;
    } else {
        ox = fwx; oy = fwy; oz = fwz;
    }
    acc[3 * (size_t)orig] = ox;
    acc[3 * (size_t)orig + 1] = oy;
    acc[3 * (size_t)orig + 2] = oz;
}

// sum over a workgroup, result valid in thread 0
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *red) {
    red[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int off = NT / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    return red[0];
}

#define PAIR_THREADS 128
#define PAIR_TJ 128

// One workgroup per cell (edge >= a, at least 3 cells per axis so the 27 neighbours are distinct).
// Lane = one particle of the cell; the neighbour cell's particles, already translated by the periodic
// shift of that neighbour, are staged in LDS and read by every lane at the same address (broadcast).
template <int P, bool UNW>
__global__ __launch_bounds__(PAIR_THREADS) void k_pair_cells(const double4 *__restrict__ ps, const int *__restrict__ cstart,
                                                             const int *__restrict__ order, PairParams pp,
                                                             double *__restrict__ acc, double *__restrict__ epart) {
    __shared__ double4 tile[PAIR_TJ];
    __shared__ double red[PAIR_THREADS];
    const int tid = threadIdx.x;
    const int cell = xcd_remap(blockIdx.x, gridDim.x);
    const int cx = cell % pp.nc[0], cy = (cell / pp.nc[0]) % pp.nc[1], cz = cell / (pp.nc[0] * pp.nc[1]);
    const int i0 = cstart[cell], i1 = cstart[cell + 1];
    double esum = 0.0;
    for (int ib = i0; ib < i1; ib += PAIR_THREADS) {
        const int ii = ib + tid;
        const int orig = ii < i1 ? order[ii] : 0x7fffffff;
        const bool active = orig < pp.n_owned;
        double4 pi = active ? ps[ii] : make_double4(0, 0, 0, 0);
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        for (int nb = 0; nb < 27; nb++) {
            int nx = cx + (nb % 3) - 1, ny = cy + ((nb / 3) % 3) - 1, nz = cz + (nb / 9) - 1;
            double sx = 0, sy = 0, sz = 0;
            if (nx < 0) { nx += pp.nc[0]; sx = -pp.A[0]; } else if (nx >= pp.nc[0]) { nx -= pp.nc[0]; sx = pp.A[0]; }
            if (ny < 0) { ny += pp.nc[1]; sy = -pp.A[1]; } else if (ny >= pp.nc[1]) { ny -= pp.nc[1]; sy = pp.A[1]; }
            if (nz < 0) { nz += pp.nc[2]; sz = -pp.A[2]; } else if (nz >= pp.nc[2]) { nz -= pp.nc[2]; sz = pp.A[2]; }
            const int cn = (nz * pp.nc[1] + ny) * pp.nc[0] + nx;
            const int j0 = cstart[cn], j1 = cstart[cn + 1];
            for (int jb = j0; jb < j1; jb += PAIR_TJ) {
                const int cnt = min(PAIR_TJ, j1 - jb);
                __syncthreads();
                if (tid < cnt) {
                    double4 pj = ps[jb + tid];
                    pj.x += sx; pj.y += sy; pj.z += sz;
                    tile[tid] = pj;
                }
                __syncthreads();
                if (active) {
                    for (int k = 0; k < cnt; k++) {
                        const double4 pj = tile[k];
                        const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                        const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                        if (r2 < pp.a2 && (jb + k) != ii)
                            pair_term<P, UNW>(dx, dy, dz, r2, pj.w, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                    }
                }
            }
        }
        if (active) {
            store_acc(acc, orig, pp.convention, pi.w, fwx, fwy, fwz, fux, fuy, fuz);
            esum += 0.5 * pi.w * u;
        }
    }
    const double tot = block_sum<PAIR_THREADS>(esum, red);
    if (tid == 0) epart[cell] = tot;
}

// Two-phase variant: phase 1 filters the candidates of a neighbour tile with a cheap FP64 distance test and
// appends the survivors to a per-lane list in LDS; phase 2 runs the expensive kernel only on survivors, so
// lanes no longer idle through the ~85 % of candidates that fall outside the cutoff.
#define PAIR2_THREADS 64
#define PAIR2_TJ 192
#define PAIR2_LIST 48
template <int P, bool UNW>
__global__ __launch_bounds__(PAIR2_THREADS) void k_pair_cells2(const double4 *__restrict__ ps, const int *__restrict__ cstart,
                                                               const int *__restrict__ order, PairParams pp,
                                                               double *__restrict__ acc, double *__restrict__ epart) {
    __shared__ double4 tile[PAIR2_TJ];
    __shared__ unsigned char list[PAIR2_LIST][PAIR2_THREADS]; // [entry][lane]: conflict-free per-lane columns
    __shared__ double red[PAIR2_THREADS];
    const int tid = threadIdx.x;
    const int cell = xcd_remap(blockIdx.x, gridDim.x);
    const int cx = cell % pp.nc[0], cy = (cell / pp.nc[0]) % pp.nc[1], cz = cell / (pp.nc[0] * pp.nc[1]);
    const int i0 = cstart[cell], i1 = cstart[cell + 1];
    double esum = 0.0;
    for (int ib = i0; ib < i1; ib += PAIR2_THREADS) {
        const int ii = ib + tid;
        const int orig = ii < i1 ? order[ii] : 0x7fffffff;
        const bool active = orig < pp.n_owned;
        double4 pi = active ? ps[ii] : make_double4(1e300, 1e300, 1e300, 0);
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        // the 27 neighbours form 9 rows of 3 x-adjacent cells; rows are contiguous in memory unless they wrap
        for (int nb = 0; nb < 27; nb++) {
            int nx = cx + (nb % 3) - 1, ny = cy + ((nb / 3) % 3) - 1, nz = cz + (nb / 9) - 1;
            double sx = 0, sy = 0, sz = 0;
            if (nx < 0) { nx += pp.nc[0]; sx = -pp.A[0]; } else if (nx >= pp.nc[0]) { nx -= pp.nc[0]; sx = pp.A[0]; }
            if (ny < 0) { ny += pp.nc[1]; sy = -pp.A[1]; } else if (ny >= pp.nc[1]) { ny -= pp.nc[1]; sy = pp.A[1]; }
            if (nz < 0) { nz += pp.nc[2]; sz = -pp.A[2]; } else if (nz >= pp.nc[2]) { nz -= pp.nc[2]; sz = pp.A[2]; }
            const int cn = (nz * pp.nc[1] + ny) * pp.nc[0] + nx;
            const int j0 = cstart[cn], j1 = cstart[cn + 1];
            for (int jb = j0; jb < j1; jb += PAIR2_TJ) {
                const int cnt = min(PAIR2_TJ, j1 - jb);
                __syncthreads();
                for (int t = tid; t < cnt; t += PAIR2_THREADS) {
                    double4 pj = ps[jb + t];
                    pj.x += sx; pj.y += sy; pj.z += sz;
                    tile[t] = pj;
                }
                __syncthreads();
                int k = 0;
                while (k < cnt) {
                    // phase 1: scan candidates until this lane's list is full or the tile ends
                    int n = 0;
                    for (; k < cnt && n < PAIR2_LIST; k++) {
                        const double4 pj = tile[k];
                        const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                        const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                        if (r2 < pp.a2 && (jb + k) != ii) { list[n][tid] = (unsigned char)k; n++; }
                    }
                    // phase 2: the expensive kernel on this lane's survivors
                    for (int e = 0; e < n; e++) {
                        const double4 pj = tile[list[e][tid]];
                        const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                        const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                        pair_term<P, UNW>(dx, dy, dz, r2, pj.w, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                    }
                }
            }
        }
        if (active) {
            store_acc(acc, orig, pp.convention, pi.w, fwx, fwy, fwz, fux, fuy, fuz);
            esum += 0.5 * pi.w * u;
        }
    }
    const double tot = block_sum<PAIR2_THREADS>(esum, red);
    if (tid == 0) epart[cell] = tot;
}

// Variant 3.  One wave per cell.
//  * main part: lane = one of the first 64 particles of the cell.  The 27 neighbour cells are walked as 9 rows;
//    a row of 3 x-adjacent cells is one contiguous run of the sorted array (unless it wraps in x), staged in LDS
//    in tiles of up to 256 candidates.  Phase 1 (cheap distance test, candidate broadcast to all lanes) appends
//    survivors to a per-lane list; phase 2 runs the kernel on survivors only.  Longer tiles make the per-lane
//    survivor counts more even, so fewer lanes idle in phase 2.
//  * overflow part: a cell with more than 64 particles does not get a second, mostly empty, lane-per-particle pass;
//    each extra particle is handled by the whole wave with lane = candidate and a butterfly reduction.
#define PAIR3_THREADS 64
#define PAIR3_TJ 256
#define PAIR3_LIST 64

struct RowSeg { int j0, j1; double sx, sy, sz; };

// segments (contiguous runs of the sorted array with one periodic shift) of neighbour row `row` (0..8) of a cell
__device__ __forceinline__ int row_segments(const PairParams &pp, const int *__restrict__ cstart, int cx, int cy, int cz,
                                            int row, RowSeg seg[3]) {
    int ny = cy + (row % 3) - 1, nz = cz + (row / 3) - 1;
    double sy = 0, sz = 0;
    if (ny < 0) { ny += pp.nc[1]; sy = -pp.A[1]; } else if (ny >= pp.nc[1]) { ny -= pp.nc[1]; sy = pp.A[1]; }
    if (nz < 0) { nz += pp.nc[2]; sz = -pp.A[2]; } else if (nz >= pp.nc[2]) { nz -= pp.nc[2]; sz = pp.A[2]; }
    const int base = (nz * pp.nc[1] + ny) * pp.nc[0];
    if (cx >= 1 && cx <= pp.nc[0] - 2) {
        seg[0].j0 = cstart[base + cx - 1]; seg[0].j1 = cstart[base + cx + 2];
        seg[0].sx = 0; seg[0].sy = sy; seg[0].sz = sz;
        return 1;
    }
    for (int o = -1; o <= 1; o++) {
        int nx = cx + o;
        double sx = 0;
        if (nx < 0) { nx += pp.nc[0]; sx = -pp.A[0]; } else if (nx >= pp.nc[0]) { nx -= pp.nc[0]; sx = pp.A[0]; }
        seg[o + 1].j0 = cstart[base + nx]; seg[o + 1].j1 = cstart[base + nx + 1];
        seg[o + 1].sx = sx; seg[o + 1].sy = sy; seg[o + 1].sz = sz;
    }
    return 3;
}

template <int P, bool UNW>
__global__ __launch_bounds__(PAIR3_THREADS) void k_pair_cells3(const double4 *__restrict__ ps, const int *__restrict__ cstart,
                                                               const int *__restrict__ order, PairParams pp,
                                                               double *__restrict__ acc, double *__restrict__ epart) {
    __shared__ double4 tile[PAIR3_TJ];
    __shared__ unsigned char list[PAIR3_LIST][PAIR3_THREADS];
    const int tid = threadIdx.x;
    const int cell = xcd_remap(blockIdx.x, gridDim.x);
    const int cx = cell % pp.nc[0], cy = (cell / pp.nc[0]) % pp.nc[1], cz = cell / (pp.nc[0] * pp.nc[1]);
    const int i0 = cstart[cell], i1 = cstart[cell + 1];
    const int imain = min(i1, i0 + PAIR3_THREADS);
    double esum = 0.0;
    if (i1 > i0) {
        const int ii = i0 + tid;
        const int orig = ii < imain ? order[ii] : 0x7fffffff;
        const bool active = orig < pp.n_owned;
        const double4 pi = active ? ps[ii] : make_double4(1e300, 1e300, 1e300, 0);
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        for (int row = 0; row < 9; row++) {
            RowSeg seg[3];
            const int nseg = row_segments(pp, cstart, cx, cy, cz, row, seg);
            for (int s = 0; s < nseg; s++) {
                const double xi = pi.x - seg[s].sx, yi = pi.y - seg[s].sy, zi = pi.z - seg[s].sz;
                for (int jb = seg[s].j0; jb < seg[s].j1; jb += PAIR3_TJ) {
                    const int cnt = min(PAIR3_TJ, seg[s].j1 - jb);
                    __syncthreads();
                    for (int t = tid; t < cnt; t += PAIR3_THREADS) tile[t] = ps[jb + t];
                    __syncthreads();
                    int k = 0;
                    while (k < cnt) {
                        int n = 0;
                        for (; k < cnt && n < PAIR3_LIST; k++) {
                            const double4 pj = tile[k];
                            const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
                            const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                            if (r2 < pp.a2 && (jb + k) != ii) { list[n][tid] = (unsigned char)k; n++; }
                        }
                        for (int e = 0; e < n; e++) {
                            const double4 pj = tile[list[e][tid]];
                            const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
                            const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                            pair_term<P, UNW>(dx, dy, dz, r2, pj.w, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                        }
                    }
                }
            }
        }
        if (active) {
            store_acc(acc, orig, pp.convention, pi.w, fwx, fwy, fwz, fux, fuy, fuz);
            esum += 0.5 * pi.w * u;
        }
    }
    // overflow particles: the wave works on one particle at a time, lane = candidate
    for (int ii = imain; ii < i1; ii++) {
        const int orig = order[ii];
        if (orig >= pp.n_owned) continue;
        const double4 pi = ps[ii];
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        for (int row = 0; row < 9; row++) {
            RowSeg seg[3];
            const int nseg = row_segments(pp, cstart, cx, cy, cz, row, seg);
            for (int s = 0; s < nseg; s++) {
                const double xi = pi.x - seg[s].sx, yi = pi.y - seg[s].sy, zi = pi.z - seg[s].sz;
                for (int j = seg[s].j0 + tid; j < seg[s].j1; j += PAIR3_THREADS) {
                    const double4 pj = ps[j];
                    const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
                    const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                    if (r2 < pp.a2 && j != ii) pair_term<P, UNW>(dx, dy, dz, r2, pj.w, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                }
            }
        }
        u = wave_sum(u); fwx = wave_sum(fwx); fwy = wave_sum(fwy); fwz = wave_sum(fwz);
        if (UNW) { fux = wave_sum(fux); fuy = wave_sum(fuy); fuz = wave_sum(fuz); }
        if (tid == 0) {
            store_acc(acc, orig, pp.convention, pi.w, fwx, fwy, fwz, fux, fuy, fuz);
            esum += 0.5 * pi.w * u;
        }
    }
    esum = wave_sum(esum);
    if (tid == 0) epart[cell] = esum;
}

// Variant 4.  As variant 3, but the distance filter runs in FP32 on coordinates relative to the cell origin and
// produces a per-lane hit bitmask for 64 candidates at a time: no branches, no LDS writes, every LDS read of the
// unrolled block can be in flight together.  The FP32 test uses a cutoff inflated by 1e-5 (coordinates relative
// to the cell are at most 2a in size, so the FP32 error of r^2 is below 1e-6 a^2); phase 2 recomputes the
// distance in FP64 and applies the exact cutoff, so the result does not depend on the filter's rounding.
#define PAIR4_THREADS 64
#define PAIR4_TJ 256

template <int P, bool UNW>
__global__ __launch_bounds__(PAIR4_THREADS) void k_pair_cells4(const double4 *__restrict__ ps, const int *__restrict__ cstart,
                                                               const int *__restrict__ order, PairParams pp,
                                                               double *__restrict__ acc, double *__restrict__ epart) {
    __shared__ double4 tile[PAIR4_TJ];
    __shared__ float4 tilef[PAIR4_TJ];
    const int tid = threadIdx.x;
    const int cell = xcd_remap(blockIdx.x, gridDim.x);
    const int cx = cell % pp.nc[0], cy = (cell / pp.nc[0]) % pp.nc[1], cz = cell / (pp.nc[0] * pp.nc[1]);
    const int i0 = cstart[cell], i1 = cstart[cell + 1];
    const int imain = min(i1, i0 + PAIR4_THREADS);
    const double ox = cx * (pp.A[0] / pp.nc[0]), oy = cy * (pp.A[1] / pp.nc[1]), oz = cz * (pp.A[2] / pp.nc[2]);
    const float a2f = (float)pp.a2 * 1.00001f;
    double esum = 0.0;
    if (i1 > i0) {
        const int ii = i0 + tid;
        const int orig = ii < imain ? order[ii] : 0x7fffffff;
        const bool active = orig < pp.n_owned;
        const double4 pi = active ? ps[ii] : make_double4(1e300, 1e300, 1e300, 0);
        const float fxi = active ? (float)(pi.x - ox) : 1e30f, fyi = active ? (float)(pi.y - oy) : 1e30f,
                    fzi = active ? (float)(pi.z - oz) : 1e30f;
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        for (int row = 0; row < 9; row++) {
            RowSeg seg[3];
            const int nseg = row_segments(pp, cstart, cx, cy, cz, row, seg);
#pragma unroll
            for (int s = 0; s < 3; s++) {
                if (s >= nseg) break;
                const double sx = seg[s].sx, sy = seg[s].sy, sz = seg[s].sz;
                const double xi = pi.x - sx, yi = pi.y - sy, zi = pi.z - sz;
                for (int jb = seg[s].j0; jb < seg[s].j1; jb += PAIR4_TJ) {
                    const int cnt = min(PAIR4_TJ, seg[s].j1 - jb);
                    const int cnt64 = (cnt + 63) & ~63;
                    __syncthreads();
                    for (int t = tid; t < cnt64; t += PAIR4_THREADS) {
                        if (t < cnt) {
                            const double4 pj = ps[jb + t];
                            tile[t] = pj;
                            tilef[t] = make_float4((float)(pj.x + sx - ox), (float)(pj.y + sy - oy), (float)(pj.z + sz - oz), 0.f);
                        } else {
                            tilef[t] = make_float4(-1e30f, -1e30f, -1e30f, 0.f);
                        }
                    }
                    __syncthreads();
                    for (int c0 = 0; c0 < cnt64; c0 += 64) {
                        unsigned m0 = 0u, m1 = 0u;
#pragma unroll
                        for (int k = 0; k < 32; k++) {
                            const float4 f = tilef[c0 + k];
                            const float dx = fxi - f.x, dy = fyi - f.y, dz = fzi - f.z;
                            const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                            m0 |= (r2 < a2f) ? (1u << k) : 0u;
                        }
#pragma unroll
                        for (int k = 0; k < 32; k++) {
                            const float4 f = tilef[c0 + 32 + k];
                            const float dx = fxi - f.x, dy = fyi - f.y, dz = fzi - f.z;
                            const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                            m1 |= (r2 < a2f) ? (1u << k) : 0u;
                        }
                        const int self = ii - (jb + c0); // a particle is not its own neighbour
                        if (self >= 0 && self < 32) m0 &= ~(1u << self);
                        else if (self >= 32 && self < 64) m1 &= ~(1u << (self - 32));
                        while (m0 | m1) {
                            int k;
                            if (m0) { k = __ffs(m0) - 1; m0 &= m0 - 1; }
                            else { k = 31 + __ffs(m1); m1 &= m1 - 1; }
                            const double4 pj = tile[c0 + k];
                            const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
                            const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                            if (r2 < pp.a2) pair_term<P, UNW>(dx, dy, dz, r2, pj.w, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                        }
                    }
                }
            }
        }
        if (active) {
            store_acc(acc, orig, pp.convention, pi.w, fwx, fwy, fwz, fux, fuy, fuz);
            esum += 0.5 * pi.w * u;
        }
    }
    for (int ii = imain; ii < i1; ii++) { // overflow particles: lane = candidate
        const int orig = order[ii];
        if (orig >= pp.n_owned) continue;
        const double4 pi = ps[ii];
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        for (int row = 0; row < 9; row++) {
            RowSeg seg[3];
            const int nseg = row_segments(pp, cstart, cx, cy, cz, row, seg);
#pragma unroll
            for (int s = 0; s < 3; s++) {
                if (s >= nseg) break;
                const double xi = pi.x - seg[s].sx, yi = pi.y - seg[s].sy, zi = pi.z - seg[s].sz;
                for (int j = seg[s].j0 + tid; j < seg[s].j1; j += PAIR4_THREADS) {
                    const double4 pj = ps[j];
                    const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
                    const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
                    if (r2 < pp.a2 && j != ii) pair_term<P, UNW>(dx, dy, dz, r2, pj.w, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                }
            }
        }
        u = wave_sum(u); fwx = wave_sum(fwx); fwy = wave_sum(fwy); fwz = wave_sum(fwz);
        if (UNW) { fux = wave_sum(fux); fuy = wave_sum(fuy); fuz = wave_sum(fuz); }
        if (tid == 0) {
            store_acc(acc, orig, pp.convention, pi.w, fwx, fwy, fwz, fux, fuy, fuz);
            esum += 0.5 * pi.w * u;
        }
    }
    esum = wave_sum(esum);
    if (tid == 0) epart[cell] = esum;
}

void launch_pair_cells(hipStream_t st, const double4 *ps, const int *cstart, const int *order, int ncell,
                       const PairParams &pp, double *acc, double *epart, int variant) {
    const bool unw = pp.convention == MSM_AMD_ACC_LITERAL;
#define LAUNCH_PAIR(KERN, NT)                                                                                          \
    do {                                                                                                               \
        if (pp.p == 4) {                                                                                               \
            if (unw) hipLaunchKernelGGL((KERN<4, true>), dim3(ncell), dim3(NT), 0, st, ps, cstart, order, pp, acc, epart); \
            else hipLaunchKernelGGL((KERN<4, false>), dim3(ncell), dim3(NT), 0, st, ps, cstart, order, pp, acc, epart);  \
        } else {                                                                                                       \
            if (unw) hipLaunchKernelGGL((KERN<6, true>), dim3(ncell), dim3(NT), 0, st, ps, cstart, order, pp, acc, epart); \
            else hipLaunchKernelGGL((KERN<6, false>), dim3(ncell), dim3(NT), 0, st, ps, cstart, order, pp, acc, epart);  \
        }                                                                                                              \
    } while (0)
    if (variant == 4) LAUNCH_PAIR(k_pair_cells4, PAIR4_THREADS);
    else if (variant == 3) LAUNCH_PAIR(k_pair_cells3, PAIR3_THREADS);
    else if (variant == 2) LAUNCH_PAIR(k_pair_cells2, PAIR2_THREADS);
    else LAUNCH_PAIR(k_pair_cells, PAIR_THREADS);
#undef LAUNCH_PAIR
}

// Small boxes (fewer than 3 cells per axis): thread i visits every j and every image inside the cutoff
// (msmPeriodic.c:1292-1300 + msmLibrary.c:141-178).  Self images belong to the (B) term and are skipped.
#define PAIRA_THREADS 64
template <int P, bool UNW>
__global__ __launch_bounds__(PAIRA_THREADS) void k_pair_allimages(const double *__restrict__ r, const double *__restrict__ q,
                                                                  int N, PairParams pp, double a,
                                                                  double *__restrict__ acc, double *__restrict__ epart) {
    __shared__ double red[PAIRA_THREADS];
    const int i = blockIdx.x * PAIRA_THREADS + threadIdx.x;
    double esum = 0.0;
    if (i < N) {
        const double xi = r[3 * (size_t)i], yi = r[3 * (size_t)i + 1], zi = r[3 * (size_t)i + 2], qi = q[i];
        double u = 0, fwx = 0, fwy = 0, fwz = 0, fux = 0, fuy = 0, fuz = 0;
        for (int j = 0; j < N; j++) {
            if (j == i) continue;
            const double ex = xi - r[3 * (size_t)j], ey = yi - r[3 * (size_t)j + 1], ez = zi - r[3 * (size_t)j + 2], qj = q[j];
            const int x0 = (int)ceil((ex - a) / pp.A[0]), x1 = (int)floor((ex + a) / pp.A[0]);
            const int y0 = (int)ceil((ey - a) / pp.A[1]), y1 = (int)floor((ey + a) / pp.A[1]);
            const int z0 = (int)ceil((ez - a) / pp.A[2]), z1 = (int)floor((ez + a) / pp.A[2]);
            for (int px = x0; px <= x1; px++)
                for (int py = y0; py <= y1; py++)
                    for (int pz = z0; pz <= z1; pz++) {
                        const double dx = ex - pp.A[0] * px, dy = ey - pp.A[1] * py, dz = ez - pp.A[2] * pz;
                        const double r2 = dx * dx + dy * dy + dz * dz;
                        if (r2 < pp.a2) pair_term<P, UNW>(dx, dy, dz, r2, qj, pp, u, fwx, fwy, fwz, fux, fuy, fuz);
                    }
        }
        store_acc(acc, i, pp.convention, qi, fwx, fwy, fwz, fux, fuy, fuz);
        esum = 0.5 * qi * u;
    }
    const double tot = block_sum<PAIRA_THREADS>(esum, red);
    if (threadIdx.x == 0) epart[blockIdx.x] = tot;
}

void launch_pair_allimages(hipStream_t st, const double *d_r, const double *d_q, int N, const PairParams &pp, double *acc,
                           double *epart, int *nepart) {
    const int nblk = (N + PAIRA_THREADS - 1) / PAIRA_THREADS;
    const double a = sqrt(pp.a2);
    const bool unw = pp.convention == MSM_AMD_ACC_LITERAL;
    if (pp.p == 4) {
        if (unw) hipLaunchKernelGGL((k_pair_allimages<4, true>), dim3(nblk), dim3(PAIRA_THREADS), 0, st, d_r, d_q, N, pp, a, acc, epart);
        else hipLaunchKernelGGL((k_pair_allimages<4, false>), dim3(nblk), dim3(PAIRA_THREADS), 0, st, d_r, d_q, N, pp, a, acc, epart);
    } else {
        if (unw) hipLaunchKernelGGL((k_pair_allimages<6, true>), dim3(nblk), dim3(PAIRA_THREADS), 0, st, d_r, d_q, N, pp, a, acc, epart);
        else hipLaunchKernelGGL((k_pair_allimages<6, false>), dim3(nblk), dim3(PAIRA_THREADS), 0, st, d_r, d_q, N, pp, a, acc, epart);
    }
    *nepart = nblk;
}

// ---------------------------------------------------------------------------
// B-spline nodal basis: centred spline Phi(t) = B_p(t + p/2) and its derivative
// (closed forms of the pieces: msmLibrary.h:120-142)
// ---------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ double phi_c(double t);
template <>
__device__ __forceinline__ double phi_c<4>(double t) {
    const double a = fabs(t);
    if (a >= 2.0) return 0.0;
    if (a < 1.0) return (2.0 / 3.0) + a * a * (0.5 * a - 1.0);
    const double b = 2.0 - a;
    return (1.0 / 6.0) * b * b * b;
}
template <>
__device__ __forceinline__ double phi_c<6>(double t) {
    const double a = fabs(t);
    if (a >= 3.0) return 0.0;
    if (a < 1.0) { const double a2 = a * a; return 11.0 / 20.0 + a2 * (-0.5 + a2 * (0.25 - a * (1.0 / 12.0))); }
    if (a < 2.0) return 17.0 / 40.0 + a * (5.0 / 8.0 + a * (-7.0 / 4.0 + a * (5.0 / 4.0 + a * (-3.0 / 8.0 + a * (1.0 / 24.0)))));
    const double b = 3.0 - a, b2 = b * b;
    return (1.0 / 120.0) * b2 * b2 * b;
}
template <int P>
__device__ __forceinline__ double dphi_c(double t);
template <>
__device__ __forceinline__ double dphi_c<4>(double t) {
    const double a = fabs(t);
    double d;
    if (a >= 2.0) d = 0.0;
    else if (a < 1.0) d = a * (1.5 * a - 2.0);
    else { const double b = 2.0 - a; d = -0.5 * b * b; }
    return t < 0 ? -d : d;
}
template <>
__device__ __forceinline__ double dphi_c<6>(double t) {
    const double a = fabs(t);
    double d;
    if (a >= 3.0) d = 0.0;
    else if (a < 1.0) { const double a2 = a * a; d = a * (-1.0 + a2 * (1.0 - a * (5.0 / 12.0))); }
    else if (a < 2.0) d = 5.0 / 8.0 + a * (-7.0 / 2.0 + a * (15.0 / 4.0 + a * (-3.0 / 2.0 + a * (5.0 / 24.0))));
    else { const double b = 3.0 - a, b2 = b * b; d = -(1.0 / 24.0) * b2 * b2; }
    return t < 0 ? -d : d;
}

// Gather form of q_m = sum_i q_i sum_P Phi(M(s_i - P) - m): the thread of array element j collects every
// particle stored in the p^3 grid bins around its node.  Bins are walked unwrapped so that a grid smaller
// than the spline support receives every periodic image.  Particle order inside a bin is fixed, so the sum
// is bitwise reproducible.
template <int P>
__global__ __launch_bounds__(256) void k_anterp_gather(const double4 *__restrict__ ps, const int *__restrict__ gstart,
                                                       SplineParams sp, double *__restrict__ grid) {
    const int Mx = sp.M[0], My = sp.M[1], Mz = sp.M[2];
    const size_t nplane = (size_t)Mx * My;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nplane * sp.nz) return;
    const int jx = (int)(idx % Mx), jy = (int)((idx / Mx) % My), jz = sp.z0 + (int)(idx / nplane);
    double sum = 0.0;
    const bool dropped = (sp.drop_last[0] && jx == Mx - 1) || (sp.drop_last[1] && jy == My - 1) ||
                         (sp.drop_last[2] && jz == Mz - 1);
    if (!dropped) {
        const int mx = jx - sp.node_shift[0], my = jy - sp.node_shift[1], mz = jz - sp.node_shift[2];
        for (int bz = mz - P / 2; bz < mz + P / 2; bz++) {
            const int wz = modp(bz, Mz);
            const double oz = (double)(bz - wz) - (double)mz;
            for (int by = my - P / 2; by < my + P / 2; by++) {
                const int wy = modp(by, My);
                const double oy = (double)(by - wy) - (double)my;
                for (int bx = mx - P / 2; bx < mx + P / 2; bx++) {
                    const int wx = modp(bx, Mx);
                    const double ox = (double)(bx - wx) - (double)mx;
                    const int b = (wz * My + wy) * Mx + wx;
                    const int s0 = gstart[b], s1 = gstart[b + 1];
                    for (int s = s0; s < s1; s++) {
                        const double4 pt = ps[s];
                        sum += pt.w * phi_c<P>(pt.x + ox) * phi_c<P>(pt.y + oy) * phi_c<P>(pt.z + oz);
                    }
                }
            }
        }
    }
    grid[((size_t)jz * My + jy) * Mx + jx] = sum;
}

void launch_anterp(hipStream_t st, const double4 *ps_grid, const int *gstart, const SplineParams &sp, double *grid) {
    const size_t n = (size_t)sp.M[0] * sp.M[1] * sp.nz;
    const int tb = 256;
    const int gb = (int)((n + tb - 1) / tb);
    if (sp.p == 4) hipLaunchKernelGGL((k_anterp_gather<4>), dim3(gb), dim3(tb), 0, st, ps_grid, gstart, sp, grid);
    else hipLaunchKernelGGL((k_anterp_gather<6>), dim3(gb), dim3(tb), 0, st, ps_grid, gstart, sp, grid);
}

// acc_i += sum_m e_m grad Phi_m(r_i) over the p^3 nodes around particle i (msmPeriodic.c:1116-1161,
// msmSimple.c:1317-1380).  Particles are visited in grid-bin order, so neighbouring lanes read neighbouring
// nodes; index_shift reproduces the reference's interpolation index offset when asked to.
template <int P>
__global__ __launch_bounds__(256) void k_interp(const double4 *__restrict__ ps, const int *__restrict__ order, int N,
                                                SplineParams sp, const double *__restrict__ e, int index_shift,
                                                double *__restrict__ acc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int orig = order[k];
    if (orig >= sp.n_owned) return;
    const double4 pt = ps[k];
    const int Mx = sp.M[0], My = sp.M[1], Mz = sp.M[2];
    double w[3][P], dw[3][P];
    int base[3];
    const double u[3] = {pt.x, pt.y, pt.z};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int fl = (int)floor(u[d]);
        fl = fl >= sp.M[d] ? sp.M[d] - 1 : fl;
        base[d] = fl - (P / 2 - 1);
#pragma unroll
        for (int v = 0; v < P; v++) {
            const double t = u[d] - (double)(base[d] + v);
            w[d][v] = phi_c<P>(t);
            dw[d][v] = dphi_c<P>(t);
        }
    }
    double gx = 0, gy = 0, gz = 0;
#pragma unroll
    for (int c = 0; c < P; c++) {
        const int iz = modp(base[2] + c + index_shift, Mz);
#pragma unroll
        for (int b = 0; b < P; b++) {
            const int iy = modp(base[1] + b + index_shift, My);
            const double *row = e + ((size_t)iz * My + iy) * Mx;
            const double wyz = w[1][b] * w[2][c], dyz = dw[1][b] * w[2][c], ydz = w[1][b] * dw[2][c];
#pragma unroll
            for (int a = 0; a < P; a++) {
                const double ev = row[modp(base[0] + a + index_shift, Mx)];
                gx = fma(dw[0][a] * wyz, ev, gx);
                gy = fma(w[0][a] * dyz, ev, gy);
                gz = fma(w[0][a] * ydz, ev, gz);
            }
        }
    }
    gx *= sp.inv_h[0]; gy *= sp.inv_h[1]; gz *= sp.inv_h[2];
    double f = 1.0;
    if (sp.convention == MSM_AMD_ACC_LITERAL) f = pt.w;      // msmPeriodic.c:1158-1160
    else if (sp.convention == MSM_AMD_ACC_FORCE) f = -pt.w;
    const size_t o = 3 * (size_t)orig;
    acc[o] += f * gx;
    acc[o + 1] += f * gy;
    acc[o + 2] += f * gz;
}

void launch_interp(hipStream_t st, const double4 *ps_grid, const int *order, int N, const SplineParams &sp,
                   const double *egrid, int index_shift, double *acc) {
    const int tb = 256, gb = (N + tb - 1) / tb;
    if (sp.p == 4) hipLaunchKernelGGL((k_interp<4>), dim3(gb), dim3(tb), 0, st, ps_grid, order, N, sp, egrid, index_shift, acc);
    else hipLaunchKernelGGL((k_interp<6>), dim3(gb), dim3(tb), 0, st, ps_grid, order, N, sp, egrid, index_shift, acc);
}

// ---------------------------------------------------------------------------
// restriction / prolongation, one axis per pass
// ---------------------------------------------------------------------------
struct AxisArgs {
    int Min[3], Mout[3];
    int axis, h;
    double J[4];
};

// out[.., m, ..] = sum_{|j| <= h} J_|j| in[.., (2m + j) mod Min, ..]
// (index arithmetic in 32 bits and wraps by one conditional add: these kernels do little else)
__device__ __forceinline__ int wrap_near(int v, int M) { return v < 0 ? v + M : (v >= M ? v - M : v); } // v in [-M, 2M)
__global__ __launch_bounds__(256) void k_restrict_axis(const double *__restrict__ in, double *__restrict__ out, AxisArgs a) {
    const unsigned n = (unsigned)a.Mout[0] * a.Mout[1] * a.Mout[2];
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const unsigned row = idx / (unsigned)a.Mout[0];
    int c[3] = {(int)(idx - row * a.Mout[0]), (int)(row % (unsigned)a.Mout[1]), (int)(row / (unsigned)a.Mout[1])};
    const int m = c[a.axis], Ma = a.Min[a.axis];
    const int stride = a.axis == 0 ? 1 : (a.axis == 1 ? a.Min[0] : a.Min[0] * a.Min[1]);
    c[a.axis] = 0;
    const double *base = in + ((size_t)c[2] * a.Min[1] + c[1]) * a.Min[0] + c[0];
    double s = 0.0;
    if (Ma >= a.h) {
        for (int j = -a.h; j <= a.h; j++) s = fma(a.J[j < 0 ? -j : j], base[(size_t)wrap_near(2 * m + j, Ma) * stride], s);
    } else {
        for (int j = -a.h; j <= a.h; j++) s = fma(a.J[j < 0 ? -j : j], base[(size_t)modp(2 * m + j, Ma) * stride], s);
    }
    out[idx] = s;
}

// out[.., n, ..] (+)= sum_{|j| <= h, n-j even} J_|j| in[.., ((n-j)/2) mod Min, ..]
__global__ __launch_bounds__(256) void k_prolong_axis(const double *__restrict__ in, double *__restrict__ out, AxisArgs a,
                                                      int accumulate) {
    const unsigned n = (unsigned)a.Mout[0] * a.Mout[1] * a.Mout[2];
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const unsigned row = idx / (unsigned)a.Mout[0];
    int c[3] = {(int)(idx - row * a.Mout[0]), (int)(row % (unsigned)a.Mout[1]), (int)(row / (unsigned)a.Mout[1])};
    const int nn = c[a.axis], Ma = a.Min[a.axis];
    const int stride = a.axis == 0 ? 1 : (a.axis == 1 ? a.Min[0] : a.Min[0] * a.Min[1]);
    c[a.axis] = 0;
    const double *base = in + ((size_t)c[2] * a.Min[1] + c[1]) * a.Min[0] + c[0];
    double s = 0.0;
    const bool near = Ma >= a.h; // (nn - j) / 2 then lies in [-Ma, 2 Ma)
    for (int j = -a.h; j <= a.h; j++) {
        const int t = nn - j;
        if (t & 1) continue;
        const int src = near ? wrap_near(t / 2, Ma) : modp(t / 2, Ma);
        s = fma(a.J[j < 0 ? -j : j], base[(size_t)src * stride], s);
    }
    out[idx] = accumulate ? out[idx] + s : s;
}

static AxisArgs make_axis(const int Min[3], int axis, int p, const double J[4], bool halve) {
    AxisArgs a;
    for (int d = 0; d < 3; d++) { a.Min[d] = Min[d]; a.Mout[d] = Min[d]; }
    a.Mout[axis] = halve ? Min[axis] / 2 : Min[axis] * 2;
    a.axis = axis;
    a.h = p / 2;
    for (int k = 0; k < 4; k++) a.J[k] = J[k];
    return a;
}

void launch_restrict_axis(hipStream_t st, const double *in, double *out, const int Min[3], int axis, int p, const double J[4]) {
    AxisArgs a = make_axis(Min, axis, p, J, true);
    const size_t n = (size_t)a.Mout[0] * a.Mout[1] * a.Mout[2];
    hipLaunchKernelGGL(k_restrict_axis, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, a);
}
void launch_prolong_axis(hipStream_t st, const double *in, double *out, const int Min[3], int axis, int p, const double J[4],
                         int accumulate) {
    AxisArgs a = make_axis(Min, axis, p, J, false);
    const size_t n = (size_t)a.Mout[0] * a.Mout[1] * a.Mout[2];
    hipLaunchKernelGGL(k_prolong_axis, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, a, accumulate);
}

// The three axis passes of a level transfer in one kernel (the coarse levels are far too small to need the
// intermediate grids, and every launch saved is several microseconds).  The sums are nested in the order of the
// separate passes, so the results are the same to the last bit.
struct Transfer3 {
    int Mf[3], Mc[3]; // fine and coarse grid
    int h;
    double J[4];
    int z0, nz;       // tiled kernels: output planes [z0, z0 + nz) (coarse planes of a restriction, fine planes of a
                      // prolongation), taken modulo the grid; a slab rank transfers only the planes it needs
};

// coarse[m] = sum_jz J sum_jy J sum_jx J fine[2m + j]
template <int H>
__global__ __launch_bounds__(256) void k_restrict3(const double *__restrict__ fine, double *__restrict__ coarse, Transfer3 t) {
    const size_t n = (size_t)t.Mc[0] * t.Mc[1] * t.Mc[2];
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int mx = (int)(idx % t.Mc[0]), my = (int)((idx / t.Mc[0]) % t.Mc[1]), mz = (int)(idx / ((size_t)t.Mc[0] * t.Mc[1]));
    int ix[2 * H + 1];
#pragma unroll
    for (int j = -H; j <= H; j++) ix[j + H] = modp(2 * mx + j, t.Mf[0]);
    double sz = 0.0;
    for (int jz = -H; jz <= H; jz++) {
        const size_t pz = (size_t)modp(2 * mz + jz, t.Mf[2]) * t.Mf[1];
        double sy = 0.0;
        for (int jy = -H; jy <= H; jy++) {
            const double *row = fine + (pz + modp(2 * my + jy, t.Mf[1])) * t.Mf[0];
            double sx = 0.0;
#pragma unroll
            for (int j = -H; j <= H; j++) sx = fma(t.J[j < 0 ? -j : j], row[ix[j + H]], sx);
            sy = fma(t.J[jy < 0 ? -jy : jy], sx, sy);
        }
        sz = fma(t.J[jz < 0 ? -jz : jz], sy, sz);
    }
    coarse[idx] = sz;
}

// fine[n] += sum_{jx} J sum_{jy} J sum_{jz} J coarse[(n - j) / 2], j of the parity of n on every axis
template <int H>
__global__ __launch_bounds__(256) void k_prolong3(const double *__restrict__ coarse, double *__restrict__ fine, Transfer3 t) {
    const size_t n = (size_t)t.Mf[0] * t.Mf[1] * t.Mf[2];
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int nx = (int)(idx % t.Mf[0]), ny = (int)((idx / t.Mf[0]) % t.Mf[1]), nz = (int)(idx / ((size_t)t.Mf[0] * t.Mf[1]));
    double sx = 0.0;
    for (int jx = -H; jx <= H; jx++) {
        if ((nx - jx) & 1) continue;
        const int cx = modp((nx - jx) / 2, t.Mc[0]);
        double sy = 0.0;
        for (int jy = -H; jy <= H; jy++) {
            if ((ny - jy) & 1) continue;
            const int cy = modp((ny - jy) / 2, t.Mc[1]);
            double sz = 0.0;
            for (int jz = -H; jz <= H; jz++) {
                if ((nz - jz) & 1) continue;
                const int cz = modp((nz - jz) / 2, t.Mc[2]);
                sz = fma(t.J[jz < 0 ? -jz : jz], coarse[((size_t)cz * t.Mc[1] + cy) * t.Mc[0] + cx], sz);
            }
            sy = fma(t.J[jy < 0 ? -jy : jy], sz, sy);
        }
        sx = fma(t.J[jx < 0 ? -jx : jx], sy, sx);
    }
    fine[idx] += sx;
}

static Transfer3 make_transfer(const int Mf[3], int p, const double J[4]) {
    Transfer3 t;
    for (int d = 0; d < 3; d++) { t.Mf[d] = Mf[d]; t.Mc[d] = Mf[d] / 2; }
    t.z0 = 0; t.nz = 0;
    t.h = p / 2;
    for (int k = 0; k < 4; k++) t.J[k] = J[k];
    return t;
}
void launch_restrict3(hipStream_t st, const double *fine, double *coarse, const int Mf[3], int p, const double J[4]) {
    const Transfer3 t = make_transfer(Mf, p, J);
    const size_t n = (size_t)t.Mc[0] * t.Mc[1] * t.Mc[2];
    const dim3 g((unsigned)((n + 255) / 256));
    if (p == 4) hipLaunchKernelGGL((k_restrict3<2>), g, dim3(256), 0, st, fine, coarse, t);
    else hipLaunchKernelGGL((k_restrict3<3>), g, dim3(256), 0, st, fine, coarse, t);
}
void launch_prolong3_add(hipStream_t st, const double *coarse, double *fine, const int Mf[3], int p, const double J[4]) {
    const Transfer3 t = make_transfer(Mf, p, J);
    const size_t n = (size_t)t.Mf[0] * t.Mf[1] * t.Mf[2];
    const dim3 g((unsigned)((n + 255) / 256));
    if (p == 4) hipLaunchKernelGGL((k_prolong3<2>), g, dim3(256), 0, st, coarse, fine, t);
    else hipLaunchKernelGGL((k_prolong3<3>), g, dim3(256), 0, st, coarse, fine, t);
}

// The three axis passes of a restriction in one launch, tile by tile through LDS.  Every launch in the chain of
// an evaluation costs several microseconds whatever it does, and a pass over these grids does little; a workgroup
// here takes a tile of outputs, makes the first pass from global memory (the grids sit in L2) into LDS, the second
// from LDS to LDS and the third to global memory.  The passes recompute a halo, which is cheap next to two
// launches, and every value is formed by the same operations in the same order as in the separate passes.
// (The same construction for the prolongation was measured slower than its three separate passes.)
#define TR_T 4         // coarse nodes per tile edge: small tiles, so that even a 24^3 grid gives every CU work
#define TR_THREADS 256
template <int H>
__global__ __launch_bounds__(TR_THREADS) void k_restrict_tiled(const double *__restrict__ fine, double *__restrict__ coarse, Transfer3 t) {
    constexpr int NF = 2 * TR_T - 1 + 2 * H; // fine nodes per axis under a tile of coarse nodes
    __shared__ double t1[NF][NF][TR_T];      // [z fine][y fine][x coarse]
    __shared__ double t2[NF][TR_T][TR_T];    // [z fine][y coarse][x coarse]
    const int ntx = (t.Mc[0] + TR_T - 1) / TR_T, nty = (t.Mc[1] + TR_T - 1) / TR_T;
    const int X0 = (blockIdx.x % ntx) * TR_T, Y0 = ((blockIdx.x / ntx) % nty) * TR_T;
    const int ZT = (blockIdx.x / (ntx * nty)) * TR_T, Z0 = t.z0 + ZT; // Z0 may run past the grid: planes are taken modulo
    // x: t1[zi][yi][xc] = sum_j J|j| fine[2 (X0 + xc) + j, 2 Y0 - H + yi, 2 Z0 - H + zi]
    for (int o = threadIdx.x; o < NF * NF * TR_T; o += TR_THREADS) {
        const int xc = o % TR_T, yi = (o / TR_T) % NF, zi = o / (TR_T * NF);
        const double *row = fine + ((size_t)modp(2 * Z0 - H + zi, t.Mf[2]) * t.Mf[1] + wrapm(2 * Y0 - H + yi, t.Mf[1])) * t.Mf[0];
        const int m = X0 + xc;
        double v[2 * H + 1]; // (all loads of the element before the first use)
#pragma unroll
        for (int j = -H; j <= H; j++) v[j + H] = row[wrapm(2 * m + j, t.Mf[0])];
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) s = fma(t.J[j < 0 ? -j : j], v[j + H], s);
        t1[zi][yi][xc] = s;
    }
    __syncthreads();
    // y: t2[zi][yc][xc] = sum_j J|j| t1[zi][2 yc + j + H][xc]
    for (int o = threadIdx.x; o < NF * TR_T * TR_T; o += TR_THREADS) {
        const int xc = o % TR_T, yc = (o / TR_T) % TR_T, zi = o / (TR_T * TR_T);
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) s = fma(t.J[j < 0 ? -j : j], t1[zi][2 * yc + j + H][xc], s);
        t2[zi][yc][xc] = s;
    }
    __syncthreads();
    // z: coarse[X0 + xc, Y0 + yc, Z0 + zc] = sum_j J|j| t2[2 zc + j + H][yc][xc]
    for (int o = threadIdx.x; o < TR_T * TR_T * TR_T; o += TR_THREADS) {
        const int xc = o % TR_T, yc = (o / TR_T) % TR_T, zc = o / (TR_T * TR_T);
        if (X0 + xc >= t.Mc[0] || Y0 + yc >= t.Mc[1] || ZT + zc >= t.nz) continue;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) s = fma(t.J[j < 0 ? -j : j], t2[2 * zc + j + H][yc][xc], s);
        coarse[((size_t)modp(Z0 + zc, t.Mc[2]) * t.Mc[1] + (Y0 + yc)) * t.Mc[0] + (X0 + xc)] = s;
    }
}

// The prolongation the same way: z, then y, then x (the order of the separate passes), along an axis
// out[n] = sum_{|j| <= H, n - j even} J|j| in[(n - j) / 2].  Small tiles (8^3 fine nodes): with large ones a grid
// has fewer tiles than the chip has CUs and the three dependent passes of a workgroup are not hidden by anything.
#define PR_T 8 // fine nodes per tile edge
#define PR_THREADS 256
template <int H>
__global__ __launch_bounds__(PR_THREADS) void k_prolong_tiled(const double *__restrict__ coarse, double *__restrict__ fine, Transfer3 t) {
    constexpr int TF = PR_T, NC = PR_T / 2 + 4; // coarse nodes per axis that reach a tile: X0/2 - 2 .. X0/2 + TF/2 + 1 (H <= 3)
    constexpr int NP = NC + 1;
    __shared__ double t1[TF][NC][NP]; // [z fine][y coarse][x coarse]
    __shared__ double t2[TF][TF][NP]; // [z fine][y fine][x coarse]
    const int ntx = (t.Mf[0] + TF - 1) / TF, nty = (t.Mf[1] + TF - 1) / TF;
    const int X0 = (blockIdx.x % ntx) * TF, Y0 = ((blockIdx.x / ntx) % nty) * TF;
    const int ZT = (blockIdx.x / (ntx * nty)) * TF, Z0 = t.z0 + ZT; // may be negative or run past the grid: taken modulo
    const int cx0 = X0 / 2 - 2, cy0 = Y0 / 2 - 2;
    const size_t plane = (size_t)t.Mc[0] * t.Mc[1];
    for (int o = threadIdx.x; o < TF * NC * NC; o += PR_THREADS) {
        const int xi = o % NC, yi = (o / NC) % NC, zf = o / (NC * NC);
        const double *col = coarse + (size_t)wrapm(cy0 + yi, t.Mc[1]) * t.Mc[0] + wrapm(cx0 + xi, t.Mc[0]);
        const int n = modp(Z0 + zf, t.Mf[2]); // the plane itself: its coarse neighbours are then one wrap away at most
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], col[(size_t)wrapm(u >> 1, t.Mc[2]) * plane], s);
        }
        t1[zf][yi][xi] = s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < TF * TF * NC; o += PR_THREADS) {
        const int xi = o % NC, yf = (o / NC) % TF, zf = o / (NC * TF);
        const int n = Y0 + yf;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], t1[zf][u / 2 - cy0][xi], s);
        }
        t2[zf][yf][xi] = s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < TF * TF * TF; o += PR_THREADS) {
        const int xf = o % TF, yf = (o / TF) % TF, zf = o / (TF * TF);
        if (X0 + xf >= t.Mf[0] || Y0 + yf >= t.Mf[1] || ZT + zf >= t.nz) continue;
        const int n = X0 + xf;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], t2[zf][yf][u / 2 - cx0], s);
        }
        const size_t idx = ((size_t)modp(Z0 + zf, t.Mf[2]) * t.Mf[1] + (Y0 + yf)) * t.Mf[0] + (X0 + xf);
        fine[idx] = fine[idx] + s;
    }
}
// Two prolongations in one launch: fine += P(mid + P(low)), for whole grids (fine Mf, mid Mf / 2, low Mf / 4).
// A workgroup takes a tile of 8^3 fine nodes; the 8^3 mid nodes that reach it need 8^3 low nodes, so the first
// prolongation is recomputed per tile in LDS (three passes, z y x), added to the mid values, and prolonged again from
// there.  Every value is formed by the operations of k_prolong_tiled in the same order, so `fine` gets the bits of two
// separate launches.  `mid` itself is left as it was (its nodes are in the footprints of other workgroups, which
// must see the old values): nothing reads the potentials of an intermediate level after the prolongations.
template <int H>
__global__ __launch_bounds__(PR_THREADS) void k_prolong_chain(const double *__restrict__ low, const double *__restrict__ mid,
                                                              double *__restrict__ fine, Transfer3 t) {
    constexpr int TF = PR_T, NC = PR_T / 2 + 4, NP = NC + 1; // NC mid nodes per axis reach a tile, and NC low nodes reach those
    __shared__ double a1[NC][NC][NP], a2[NC][NC][NP], m[NC][NC][NP]; // [z mid][y][x]
    __shared__ double t1[TF][NC][NP], t2[TF][TF][NP];
    const int Mm[3] = {t.Mf[0] / 2, t.Mf[1] / 2, t.Mf[2] / 2}, Ml[3] = {t.Mf[0] / 4, t.Mf[1] / 4, t.Mf[2] / 4};
    const int ntx = (t.Mf[0] + TF - 1) / TF, nty = (t.Mf[1] + TF - 1) / TF;
    const int X0 = (blockIdx.x % ntx) * TF, Y0 = ((blockIdx.x / ntx) % nty) * TF, Z0 = (blockIdx.x / (ntx * nty)) * TF;
    const int cx0 = X0 / 2 - 2, cy0 = Y0 / 2 - 2, cz0 = Z0 / 2 - 2;          // first mid node of the footprint (even)
    const int lx0 = cx0 / 2 - 2, ly0 = cy0 / 2 - 2, lz0 = cz0 / 2 - 2;       // first low node (cx0 / 2 is exact)
    const size_t lplane = (size_t)Ml[0] * Ml[1];
    // mid footprint = P(low): z
    for (int o = threadIdx.x; o < NC * NC * NC; o += PR_THREADS) {
        const int xi = o % NC, yi = (o / NC) % NC, zm = o / (NC * NC);
        const double *col = low + (size_t)wrapm(ly0 + yi, Ml[1]) * Ml[0] + wrapm(lx0 + xi, Ml[0]);
        const int n = modp(cz0 + zm, Mm[2]);
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], col[(size_t)wrapm(u >> 1, Ml[2]) * lplane], s);
        }
        a1[zm][yi][xi] = s;
    }
    __syncthreads();
    // y (unwrapped node numbers: the parity of a node survives the wrap, the grids being even)
    for (int o = threadIdx.x; o < NC * NC * NC; o += PR_THREADS) {
        const int xi = o % NC, ym = (o / NC) % NC, zm = o / (NC * NC);
        const int n = cy0 + ym;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], a1[zm][(u >> 1) - ly0][xi], s);
        }
        a2[zm][ym][xi] = s;
    }
    __syncthreads();
    // x, and the mid values themselves: m = mid + P(low)
    for (int o = threadIdx.x; o < NC * NC * NC; o += PR_THREADS) {
        const int xm = o % NC, ym = (o / NC) % NC, zm = o / (NC * NC);
        const int n = cx0 + xm;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], a2[zm][ym][(u >> 1) - lx0], s);
        }
        const size_t idx = ((size_t)modp(cz0 + zm, Mm[2]) * Mm[1] + wrapm(cy0 + ym, Mm[1])) * Mm[0] + wrapm(cx0 + xm, Mm[0]);
        m[zm][ym][xm] = mid[idx] + s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < TF * NC * NC; o += PR_THREADS) {
        const int xi = o % NC, yi = (o / NC) % NC, zf = o / (NC * NC);
        const int n = Z0 + zf;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], m[(u >> 1) - cz0][yi][xi], s);
        }
        t1[zf][yi][xi] = s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < TF * TF * NC; o += PR_THREADS) {
        const int xi = o % NC, yf = (o / NC) % TF, zf = o / (NC * TF);
        const int n = Y0 + yf;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], t1[zf][(u >> 1) - cy0][xi], s);
        }
        t2[zf][yf][xi] = s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < TF * TF * TF; o += PR_THREADS) {
        const int xf = o % TF, yf = (o / TF) % TF, zf = o / (TF * TF);
        if (X0 + xf >= t.Mf[0] || Y0 + yf >= t.Mf[1] || Z0 + zf >= t.Mf[2]) continue;
        const int n = X0 + xf;
        double s = 0.0;
#pragma unroll
        for (int j = -H; j <= H; j++) {
            const int u = n - j;
            if (u & 1) continue;
            s = fma(t.J[j < 0 ? -j : j], t2[zf][yf][(u >> 1) - cx0], s);
        }
        const size_t idx = ((size_t)(Z0 + zf) * t.Mf[1] + (Y0 + yf)) * t.Mf[0] + (X0 + xf);
        fine[idx] = fine[idx] + s;
    }
}

void launch_prolong_chain_add(hipStream_t st, const double *low, const double *mid, double *fine, const int Mf[3], int p,
                              const double J[4]) {
    Transfer3 t = make_transfer(Mf, p, J);
    const int nb = ((t.Mf[0] + PR_T - 1) / PR_T) * ((t.Mf[1] + PR_T - 1) / PR_T) * ((t.Mf[2] + PR_T - 1) / PR_T);
    if (p == 4) hipLaunchKernelGGL((k_prolong_chain<2>), dim3(nb), dim3(PR_THREADS), 0, st, low, mid, fine, t);
    else hipLaunchKernelGGL((k_prolong_chain<3>), dim3(nb), dim3(PR_THREADS), 0, st, low, mid, fine, t);
}
void launch_prolong_tiled_add(hipStream_t st, const double *coarse, double *fine, const int Mf[3], int p, const double J[4], int z0,
                              int nz) {
    Transfer3 t = make_transfer(Mf, p, J);
    t.z0 = z0; t.nz = nz; // fine planes [z0, z0 + nz) modulo Mf[2]
    const int nb = ((t.Mf[0] + PR_T - 1) / PR_T) * ((t.Mf[1] + PR_T - 1) / PR_T) * ((nz + PR_T - 1) / PR_T);
    if (p == 4) MSM_LAUNCH_TAIL((k_prolong_tiled<2>), dim3(nb), dim3(PR_THREADS), 0, st, coarse, fine, t);
    else MSM_LAUNCH_TAIL((k_prolong_tiled<3>), dim3(nb), dim3(PR_THREADS), 0, st, coarse, fine, t);
}

void launch_restrict_tiled(hipStream_t st, const double *fine, double *coarse, const int Mf[3], int p, const double J[4], int z0,
                           int nz) {
    Transfer3 t = make_transfer(Mf, p, J);
    t.z0 = z0; t.nz = nz; // coarse planes [z0, z0 + nz) modulo Mc[2]
    const int nb = ((t.Mc[0] + TR_T - 1) / TR_T) * ((t.Mc[1] + TR_T - 1) / TR_T) * ((nz + TR_T - 1) / TR_T);
    if (p == 4) MSM_LAUNCH_TAIL((k_restrict_tiled<2>), dim3(nb), dim3(TR_THREADS), 0, st, fine, coarse, t);
    else MSM_LAUNCH_TAIL((k_restrict_tiled<3>), dim3(nb), dim3(TR_THREADS), 0, st, fine, coarse, t);
}
// ---------------------------------------------------------------------------
// grid convolution
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_conv_direct(const double *__restrict__ q, const double *__restrict__ K,
                                                     double *__restrict__ e, ConvArgs c) {
    const int Mx = c.M[0], My = c.M[1], Mz = c.M[2];
    const size_t nplane = (size_t)Mx * My;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nplane * c.nz) return;
    const int mx = (int)(idx % Mx), my = (int)((idx / Mx) % My), mz = c.z0 + (int)(idx / nplane);
    double s = 0.0;
    for (int sz = 0; sz < c.ks[2]; sz++) {
        const int nz = modp(mz - (c.kf[2] + sz), Mz);
        for (int sy = 0; sy < c.ks[1]; sy++) {
            const int ny = modp(my - (c.kf[1] + sy), My);
            const double *qrow = q + ((size_t)nz * My + ny) * Mx;
            const double *krow = K + ((size_t)sz * c.ks[1] + sy) * c.ks[0];
            for (int sx = 0; sx < c.ks[0]; sx++) s = fma(krow[sx], qrow[modp(mx - (c.kf[0] + sx), Mx)], s);
        }
    }
    const size_t o = ((size_t)mz * My + my) * Mx + mx;
    e[o] = c.accumulate ? e[o] + s : s;
}

// Open boundaries, coarsest level: every node against every node, no wrap-around.
//
// A lane owns 8 outputs (mx, my0..my0+7, mz) and meets 8 input rows (ny0..ny0+7, nz) at a time: the 64 products of a
// step at a given nx need only the 15 stencil rows my - ny = -7..7, so one stencil load (neighbouring lanes = neighbouring
// mx = neighbouring addresses) feeds four FMAs and the charges are wave-uniform.  The input row groups are shared out
// over the waves of a workgroup and over gridDim.y workgroups; a workgroup adds its waves' sums in wave order and
// stores one partial per output, and k_conv_open_sum adds the partials in group order: the order of the additions is
// fixed by the launch geometry alone.
#define CONV_OPEN_R 8
__global__ __launch_bounds__(256) void k_conv_open(const double *__restrict__ q, const double *__restrict__ K,
                                                   double *__restrict__ part, int Mx, int My, int Mz) {
    __shared__ double red[3][CONV_OPEN_R][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nyt = (My + CONV_OPEN_R - 1) / CONV_OPEN_R, total = Mx * nyt * Mz;
    const int flat = min(blockIdx.x * 64 + lane, total - 1); // lanes past the end repeat the last tile
    const int mx = flat % Mx, yt = (flat / Mx) % nyt, mz = flat / (Mx * nyt), my0 = yt * CONV_OPEN_R;
    const int sx = 2 * Mx - 1, sy = 2 * My - 1;
    const int units = Mz * nyt, nsl = gridDim.y * 4, sl = blockIdx.y * 4 + wave;
    const int per = (units + nsl - 1) / nsl, u0 = min(units, sl * per), u1 = min(units, u0 + per);
    double acc[CONV_OPEN_R];
#pragma unroll
    for (int o = 0; o < CONV_OPEN_R; o++) acc[o] = 0.0;
    for (int u = u0; u < u1; u++) {
        const int nz = u / nyt, ny0 = (u - nz * nyt) * CONV_OPEN_R;
        // stencil row my - ny = my0 - ny0 + d, d = -7..7 (rows outside the stencil only meet rows outside the grid)
        const double *kplane = K + (size_t)(mz - nz + Mz - 1) * sy * sx + mx + Mx - 1;
        int koff[2 * CONV_OPEN_R - 1];
#pragma unroll
        for (int d = 0; d < 2 * CONV_OPEN_R - 1; d++)
            koff[d] = min(max(my0 - ny0 + d - (CONV_OPEN_R - 1) + My - 1, 0), sy - 1) * sx;
        const double *qrow = q + ((size_t)nz * My + ny0) * Mx;
        const int nrows = min(CONV_OPEN_R, My - ny0); // input rows of this group inside the grid (wave-uniform)
        for (int nx = 0; nx < Mx; nx++) {
            double w[2 * CONV_OPEN_R - 1], qv[CONV_OPEN_R];
#pragma unroll
            for (int d = 0; d < 2 * CONV_OPEN_R - 1; d++) w[d] = kplane[koff[d] - nx];
#pragma unroll
            for (int i = 0; i < CONV_OPEN_R; i++) qv[i] = i < nrows ? qrow[(size_t)i * Mx + nx] : 0.0;
#pragma unroll
            for (int i = 0; i < CONV_OPEN_R; i++)
#pragma unroll
                for (int o = 0; o < CONV_OPEN_R; o++) acc[o] = fma(w[o - i + CONV_OPEN_R - 1], qv[i], acc[o]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int o = 0; o < CONV_OPEN_R; o++) red[wave - 1][o][lane] = acc[o];
    }
    __syncthreads();
    if (wave == 0 && blockIdx.x * 64 + lane < total) {
#pragma unroll
        for (int o = 0; o < CONV_OPEN_R; o++) {
            const double t = ((acc[o] + red[0][o][lane]) + red[1][o][lane]) + red[2][o][lane];
            if (my0 + o < My) part[(size_t)blockIdx.y * Mx * My * Mz + ((size_t)mz * My + my0 + o) * Mx + mx] = t;
        }
    }
}

__global__ __launch_bounds__(256) void k_conv_open_sum(const double *__restrict__ part, double *__restrict__ e, int n, int groups) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double t = part[i];
    for (int g = 1; g < groups; g++) t += part[(size_t)g * n + i];
    e[i] = t;
}

int conv_open_groups(const int M[3]) {
    const int nyt = (M[1] + CONV_OPEN_R - 1) / CONV_OPEN_R;
    const long tiles = ((long)M[0] * nyt * M[2] + 63) / 64, units = (long)M[2] * nyt;
    long g = (1024 + tiles - 1) / tiles;           // about four workgroups per CU
    g = std::min(g, (units + 3) / 4);              // at least one row group per wave
    return (int)std::max(1L, std::min(g, 64L));
}

void launch_conv_open(hipStream_t st, const double *q, const double *K, const int M[3], double *e, double *part) {
    const int n = M[0] * M[1] * M[2], groups = conv_open_groups(M);
    const int nyt = (M[1] + CONV_OPEN_R - 1) / CONV_OPEN_R;
    const unsigned tiles = (unsigned)(((long)M[0] * nyt * M[2] + 63) / 64);
    hipLaunchKernelGGL(k_conv_open, dim3(tiles, groups), dim3(256), 0, st, q, K, part, M[0], M[1], M[2]);
    hipLaunchKernelGGL(k_conv_open_sum, dim3((n + 255) / 256), dim3(256), 0, st, part, e, n, groups);
}

void launch_conv_direct(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                        const int kfirst[3], double *e, int z0, int nz, int accumulate) {
    ConvArgs c;
    for (int d = 0; d < 3; d++) { c.M[d] = M[d]; c.ks[d] = ksize[d]; c.kf[d] = kfirst[d]; }
    c.z0 = z0; c.nz = nz; c.accumulate = accumulate;
    const size_t n = (size_t)M[0] * M[1] * nz;
    hipLaunchKernelGGL(k_conv_direct, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, K, e, c);
}

// Small grids (the coarsest levels and the Fourier top level, a few thousand nodes with a stencil that spans
// the whole period): one wave per output node, lanes stride over the taps, butterfly reduction.  A thread per
// node would leave almost the whole chip idle and serialise thousands of dependent loads.
__global__ __launch_bounds__(256) void k_conv_small(const double *__restrict__ q, const double *__restrict__ K,
                                                    double *__restrict__ e, ConvArgs c) {
    const int Mx = c.M[0], My = c.M[1], Mz = c.M[2];
    const int nout = Mx * My * c.nz;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nout) return;
    const int mx = w % Mx, my = (w / Mx) % My, mz = c.z0 + w / (Mx * My);
    const int T = c.ks[0] * c.ks[1] * c.ks[2];
    double s = 0.0;
    for (int t = lane; t < T; t += 64) {
        const int sx = t % c.ks[0], sy = (t / c.ks[0]) % c.ks[1], sz = t / (c.ks[0] * c.ks[1]);
        const int nx = modp(mx - (c.kf[0] + sx), Mx), ny = modp(my - (c.kf[1] + sy), My), nz = modp(mz - (c.kf[2] + sz), Mz);
        s = fma(K[t], q[((size_t)nz * My + ny) * Mx + nx], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        const size_t o = ((size_t)mz * My + my) * Mx + mx;
        e[o] = c.accumulate ? e[o] + s : s;
    }
}

// The coarsest grid level and the Fourier level in one pass.  Both stencils span exactly one period there, so
// each is a table over the offset (m - n) mod M; Ta holds their sum, Tb the Fourier part alone (its product with
// the charges is an energy term of its own):  ea[m] = sum_n Ta[m - n] q[n],  eb[m] = sum_n Tb[m - n] q[n].
__global__ __launch_bounds__(256) void k_conv_small_pair(const double *__restrict__ q, const double *__restrict__ Ta,
                                                         const double *__restrict__ Tb, double *__restrict__ ea,
                                                         double *__restrict__ eb, int Mx, int My, int Mz) {
    // one workgroup per output node (a few thousand of them, a few thousand taps each): the kernel is over when the
    // slowest node is, so the taps of a node are spread over 256 lanes rather than 64
    __shared__ double red[2][4];
    const int nn = Mx * My * Mz;
    const int w = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mx = w % Mx, my = (w / Mx) % My, mz = w / (Mx * My);
    double sa = 0.0, sb = 0.0;
    for (int t = threadIdx.x; t < nn; t += 256) {
        const int nx = t % Mx, ny = (t / Mx) % My, nz = t / (Mx * My);
        int ox = mx - nx, oy = my - ny, oz = mz - nz; // each in (-M, M)
        ox += ox < 0 ? Mx : 0; oy += oy < 0 ? My : 0; oz += oz < 0 ? Mz : 0;
        const int o = (oz * My + oy) * Mx + ox;
        const double v = q[t];
        sa = fma(Ta[o], v, sa);
        sb = fma(Tb[o], v, sb);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_xor(sa, off);
        sb += __shfl_xor(sb, off);
    }
    if (lane == 0) { red[0][wave] = sa; red[1][wave] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ea[w] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        eb[w] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

void launch_conv_small_pair(hipStream_t st, const double *q, const double *Ta, const double *Tb, const int M[3], double *ea,
                            double *eb) {
    const int nn = M[0] * M[1] * M[2];
    hipLaunchKernelGGL(k_conv_small_pair, dim3(nn), dim3(256), 0, st, q, Ta, Tb, ea, eb, M[0], M[1], M[2]);
}

void launch_conv_small(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                       const int kfirst[3], double *e, int z0, int nz, int accumulate) {
    ConvArgs c;
    for (int d = 0; d < 3; d++) { c.M[d] = M[d]; c.ks[d] = ksize[d]; c.kf[d] = kfirst[d]; }
    c.z0 = z0; c.nz = nz; c.accumulate = accumulate;
    const int nout = M[0] * M[1] * nz;
    hipLaunchKernelGGL(k_conv_small, dim3((nout + 3) / 4), dim3(256), 0, st, q, K, e, c);
}

// LDS-tiled convolution.  A workgroup owns TX x TY x TZ outputs and stages the
// (TX+sx-1)(TY+sy-1)(TZ+SZ-1) inputs they touch in LDS once (periodic wrap applied while staging).
// A thread owns R consecutive outputs along z at one (x,y): for each (sx,sy) it reads a window of R+SZ-1
// inputs (lanes are consecutive in x, so every LDS read is a conflict-free row segment) and feeds R*SZ FMAs.
// The SZ coefficients of that (sx,sy) are contiguous in the transposed stencil Kt[sy][sx][sz] and uniform
// across the wave, so they arrive through scalar loads and cost no LDS bandwidth.
template <int SZ, int R, int TX, int TY, int TZ>
__global__ __launch_bounds__(TX *TY *(TZ / R)) void k_conv_tiled(const double *__restrict__ q, const double *__restrict__ Kt,
                                                                double *__restrict__ e, ConvArgs c, int ntx, int nty) {
    extern __shared__ double lds[];
    const int Mx = c.M[0], My = c.M[1], Mz = c.M[2];
    const int sxn = c.ks[0], syn = c.ks[1];
    const int LX = TX + sxn - 1, LY = TY + syn - 1, LZ = TZ + SZ - 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = bid % ntx, by = (bid / ntx) % nty, bz = bid / (ntx * nty);
    const int X0 = bx * TX, Y0 = by * TY, Z0 = c.z0 + bz * TZ;
    const int ox = X0 - (c.kf[0] + sxn - 1), oy = Y0 - (c.kf[1] + syn - 1), oz = Z0 - (c.kf[2] + SZ - 1);
    const int nthreads = TX * TY * (TZ / R);
    for (int t = threadIdx.x; t < LX * LY * LZ; t += nthreads) {
        const int lx = t % LX, ly = (t / LX) % LY, lz = t / (LX * LY);
        lds[t] = q[((size_t)modp(oz + lz, Mz) * My + modp(oy + ly, My)) * Mx + modp(ox + lx, Mx)];
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tg = threadIdx.x / (TX * TY);
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0;
    for (int sy = 0; sy < syn; sy++) {
        for (int sx = 0; sx < sxn; sx++) {
            const double *col = lds + ((size_t)(tg * R) * LY + (ty + syn - 1 - sy)) * LX + (tx + sxn - 1 - sx);
            double v[R + SZ - 1];
#pragma unroll
            for (int w = 0; w < R + SZ - 1; w++) v[w] = col[(size_t)w * LY * LX];
            const double *kk = Kt + ((size_t)sy * sxn + sx) * SZ;
#pragma unroll
            for (int sz = 0; sz < SZ; sz++) {
                const double kv = kk[sz];
#pragma unroll
                for (int r = 0; r < R; r++) acc[r] = fma(kv, v[r + SZ - 1 - sz], acc[r]);
            }
        }
    }
    const int mx = X0 + tx, my = Y0 + ty;
    if (mx < Mx && my < My) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int mz = Z0 + tg * R + r;
            // (a range may run past the last plane and go on at plane 0: z0 + nz > Mz)
            if (mz < c.z0 + c.nz) e[((size_t)(mz >= Mz ? mz - Mz : mz) * My + my) * Mx + mx] = acc[r];
        }
    }
}

template <int SZ, int R, int TX, int TY, int TZ>
static bool conv_tiled_launch(hipStream_t st, const double *q, const double *Kt, double *e, const ConvArgs &c) {
    const int LX = TX + c.ks[0] - 1, LY = TY + c.ks[1] - 1, LZ = TZ + SZ - 1;
    const size_t bytes = sizeof(double) * (size_t)LX * LY * LZ;
    if (bytes > 160 * 1024) return false;
    const int ntx = (c.M[0] + TX - 1) / TX, nty = (c.M[1] + TY - 1) / TY, ntz = (c.nz + TZ - 1) / TZ;
    auto kern = k_conv_tiled<SZ, R, TX, TY, TZ>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(ntx * nty * ntz), dim3(TX * TY * (TZ / R)), bytes, st, q, Kt, e, c, ntx, nty);
    return true;
}

bool launch_conv_tiled(hipStream_t st, const double *q, const double *Kt, const int M[3], const int ksize[3],
                       const int kfirst[3], double *e, int z0, int nz) {
    ConvArgs c;
    for (int d = 0; d < 3; d++) { c.M[d] = M[d]; c.ks[d] = ksize[d]; c.kf[d] = kfirst[d]; }
    c.z0 = z0; c.nz = nz; c.accumulate = 0;
    switch (ksize[2]) {
#define CASE_SZ(S)                                                                             \
    case S:                                                                                    \
        if (conv_tiled_launch<S, 4, 16, 8, 8>(st, q, Kt, e, c)) return true;                   \
        return conv_tiled_launch<S, 4, 16, 4, 8>(st, q, Kt, e, c);
        CASE_SZ(7)
        CASE_SZ(9)
        CASE_SZ(11)
        CASE_SZ(13)
        CASE_SZ(15)
        CASE_SZ(17)
        CASE_SZ(19)
#undef CASE_SZ
    default:
        return false;
    }
}

// Several levels in one launch.  The convolutions of the grid levels are independent of each other and only
// the finest one is large enough to fill the chip, so they share a launch: block -> (level, tile).  A thread
// owns R consecutive outputs along z; for every (sx, sy) it reads a window of R + SZ - 1 inputs from the LDS
// tile and feeds R * SZ FMAs whose stencil taps are wave-uniform (scalar loads).  With one wave per SIMD (the
// tile takes almost all of the LDS) nothing else hides latency, so the window and the taps of the next
// (sx, sy) are requested before the FMAs of the current one are issued.
typedef __attribute__((address_space(4))) double ctap_t; // stencil taps: constant for the kernel, uniform per wave

// SYM: the stencil is mirror symmetric along x (K[sx] = K[SZ - 1 - sx]: a level stencil is, being a function of the
// node distance |s_x h_x| filtered by the symmetric omega', whenever its first offset is -(SZ - 1) / 2).  The windows
// of the two mirror columns are then added first and multiplied by the shared taps once: 22 additions + 120 FMAs for
// two columns instead of 240 FMAs.  The sums are formed in a different order than without the flag (same to rounding).
template <int SZ, int R, int TX, int TY, int TZ, int CH, bool SYM = false>
__global__ __launch_bounds__(TX *TY *(TZ / R)) void k_conv_multi(ConvBatch B) {
    extern __shared__ double lds[];
    int lev = 0;
    while (lev + 1 < B.n && (int)blockIdx.x >= B.blk0[lev + 1]) lev++;
    const ConvArgs c = B.c[lev];
    const double *__restrict__ q = B.q[lev];
    double *__restrict__ e = B.e[lev];
    const int ntx = B.ntx[lev], nty = B.nty[lev];
    const int Mx = c.M[0], My = c.M[1], Mz = c.M[2];
    // LDS tile [z][x][y], y fastest, row length LY = 2 mod 4 doubles: the 16 x 2 (x, y) lanes of a half wave then
    // fall on 64 different banks, so a window read costs its two LDS cycles and no more.  The tile holds the y
    // rows of CH consecutive sy at a time (CH = SZ when the whole halo fits the LDS; wider stencils restage).
    constexpr int CHR = CH < SZ ? CH : SZ;
    constexpr int LX = TX + SZ - 1, LY = ((TY + CHR - 1 + 1) / 4) * 4 + 2, LZ = TZ + SZ - 1;
    static_assert(LY >= TY + CHR - 1, "row padding");
    static_assert(LX <= 64, "a row is staged by half a wave in at most two steps");
    const int bid = blockIdx.x - B.blk0[lev];
    const int bx = bid % ntx, by = (bid / ntx) % nty, bz = bid / (ntx * nty);
    const int X0 = bx * TX, Y0 = by * TY, Z0 = c.z0 + bz * TZ;
    constexpr int NT = TX * TY * (TZ / R);
    const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) % TY, tg = threadIdx.x / (TX * TY);
    constexpr int W = R + SZ - 1;
    constexpr int plane = LX * LY;
    const int half = threadIdx.x >> 5, hl = threadIdx.x & 31;
    int txm = tx;
    if (SYM) asm volatile("" : "+v"(txm));
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0;
    // a stencil wider than the kernel's SZ comes as nsub^3 cubes of SZ^3 taps, applied one after the other
    const int ncube = c.nsub * c.nsub * c.nsub;
    const int sy_lo = c.sy0, sy_hi = c.sy1 > 0 ? c.sy1 : SZ; // the share of the taps this launch entry applies
    // Staging of a pass (cube, c0): half a wave per z plane of the tile, lanes along x, one y row after the other.
    // The wrapped y of a row is the same for every lane (scalar arithmetic), the wrapped z is found once per plane.
    // PRE (workgroups of two waves, 16 outputs per thread: one wave per SIMD, registers to spare, and nothing else
    // on the SIMD to hide a wait): the rows of the next pass are requested before the FMAs of the current one and go
    // to the LDS after them.
    constexpr int NH = NT / 32, PLN = (LZ + NH - 1) / NH, ROWS = TY + CHR - 1, NCOL = LX > 32 ? 2 : 1;
    constexpr bool PRE = CHR < SZ && R >= 16 && PLN * ROWS * NCOL <= 96;
    double pre[PRE ? PLN * ROWS * NCOL : 1];
    auto fetch = [&](int cube, int c0) {
        const int ca = cube % c.nsub, cb = (cube / c.nsub) % c.nsub, cc = cube / (c.nsub * c.nsub);
        const int ox = X0 - (c.kf[0] + ca * SZ + SZ - 1), oy = Y0 - (c.kf[1] + cb * SZ + SZ - 1), oz = Z0 - (c.kf[2] + cc * SZ + SZ - 1);
        const int gx0 = modp(ox + hl, Mx), gx1 = modp(ox + hl + 32, Mx);
        const int nch = min(CHR, sy_hi - c0);
        const int yw0 = modp(oy + SZ - c0 - nch, My);
        // every wrapped index is a valid one, also for the lanes, planes and rows beyond the tile's halo: the loads
        // need no guard (a guarded load would have to wait for the value it may have to keep), the stores have one
#pragma unroll
        for (int i = 0; i < PLN; i++) {
            const double *pl = q + (size_t)modp(oz + half + i * NH, Mz) * ((size_t)My * Mx);
            int yw = yw0;
#pragma unroll
            for (int k = 0; k < ROWS; k++) {
                const double *row = pl + (size_t)yw * Mx;
                pre[(i * ROWS + k) * NCOL] = row[gx0];
                if (NCOL == 2) pre[(i * ROWS + k) * NCOL + NCOL - 1] = row[gx1];
                yw = yw + 1 == My ? 0 : yw + 1;
            }
        }
    };
    auto put = [&](int rows) {
#pragma unroll
        for (int i = 0; i < PLN; i++) {
            const int lz = half + i * NH;
            if (lz < LZ) {
                double *lrow = lds + lz * plane + hl * LY;
#pragma unroll
                for (int k = 0; k < ROWS; k++) {
                    if (k < rows) {
                        if (hl < LX) lrow[k] = pre[(i * ROWS + k) * NCOL];
                        if (NCOL == 2 && hl + 32 < LX) lrow[32 * LY + k] = pre[(i * ROWS + k) * NCOL + NCOL - 1];
                    }
                }
            }
        }
    };
    if (PRE) fetch(0, sy_lo);
    for (int cube = 0; cube < ncube; cube++) {
    const int ca = cube % c.nsub, cb = (cube / c.nsub) % c.nsub, cc = cube / (c.nsub * c.nsub);
    const int ox = X0 - (c.kf[0] + ca * SZ + SZ - 1), oy = Y0 - (c.kf[1] + cb * SZ + SZ - 1), oz = Z0 - (c.kf[2] + cc * SZ + SZ - 1);
    // the taps are the same for every lane: read through the constant address space they come as scalar loads into
    // SGPRs and leave the vector memory path to the staging loads
    const ctap_t *Kt = (const ctap_t *)(B.Kt[lev] + (size_t)cube * (SZ * SZ * SZ));
    const int gx0 = modp(ox + hl, Mx), gx1 = modp(ox + hl + 32, Mx);
    for (int c0 = sy_lo; c0 < sy_hi; c0 += CHR) {
        const int nch = min(CHR, sy_hi - c0); // this pass applies the taps sy in [c0, c0 + nch)
        const int ys = SZ - c0 - nch;      // first y row (of the full halo) they touch
        const int rows = TY + nch - 1;
        if (c0 > sy_lo || cube > 0) __syncthreads();
        if (PRE) {
            put(rows);
        } else {
            // the loads of a batch of rows are issued before the first store
            constexpr int BATCH = ROWS <= 12 ? ROWS : 8;
            const int yw0 = modp(oy + ys, My);
            for (int lz = half; lz < LZ; lz += NH) {
                const double *pl = q + (size_t)modp(oz + lz, Mz) * ((size_t)My * Mx);
                double *lrow = lds + lz * plane + hl * LY;
                int yw = yw0;
                for (int b = 0; b < rows; b += BATCH) {
                    double v0[BATCH], v1[BATCH];
#pragma unroll
                    for (int k = 0; k < BATCH; k++) {
                        if (b + k < rows) {
                            const double *row = pl + (size_t)yw * Mx;
                            if (hl < LX) v0[k] = row[gx0];
                            if (LX > 32 && hl + 32 < LX) v1[k] = row[gx1];
                        }
                        yw = yw + 1 == My ? 0 : yw + 1;
                    }
#pragma unroll
                    for (int k = 0; k < BATCH; k++) {
                        if (b + k < rows) {
                            if (hl < LX) lrow[b + k] = v0[k];
                            if (LX > 32 && hl + 32 < LX) lrow[32 * LY + b + k] = v1[k];
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (PRE) { // the rows of the pass after this one
            if (c0 + CHR < sy_hi) fetch(cube, c0 + CHR);
            else if (cube + 1 < ncube) fetch(cube + 1, sy_lo);
        }
        // window of tap (sx, sy): colc - (sx * LY + sy)
        const double *colc = lds + (tg * R) * plane + (tx + SZ - 1) * LY + (ty + c0 + nch - 1);
        if (SYM) {
            constexpr int NP = SZ / 2; // steps 0 .. NP - 1: columns j and SZ - 1 - j together; step NP: the middle column
            double va[W], vb[W], kc[SZ];
            // The mirror column is addressed from a base of its own (txm is tx behind an empty asm).  The two columns
            // of a step are an even number of doubles apart; read by one two-address instruction, which is what the
            // compiler makes of a constant distance, 16 lanes x 2 addresses are served together and fall on the same 16
            // of the 32 bank pairs (counters: 2-way conflicts on every such read, 45 % of the LDS cycles).
            const double *colm = lds + (tg * R) * plane + (txm + SZ - 1) * LY + (ty + c0 + nch - 1);
            {
                const double *ca = colc - c0, *cb = colm - ((SZ - 1) * LY + c0);
                const ctap_t *kk = Kt + (size_t)c0 * SZ * SZ;
#pragma unroll
                for (int w = 0; w < W; w++) { va[w] = ca[w * plane]; vb[w] = cb[w * plane]; }
#pragma unroll
                for (int z = 0; z < SZ; z++) kc[z] = kk[z];
            }
            for (int sy = c0; sy < c0 + nch; sy++) {
                const bool more = sy + 1 < c0 + nch;
#pragma unroll
                for (int j = 0; j <= NP; j++) {
                    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the windows and taps requested a step ago
                    __builtin_amdgcn_sched_barrier(0);
                    double sm[W], kn[SZ];
#pragma unroll
                    for (int w = 0; w < W; w++) sm[w] = j < NP ? va[w] + vb[w] : va[w];
                    {
                        // the next step's requests (after the last step of a pass there is nothing to fetch: any
                        // valid address will do)
                        const bool last = j == NP;
                        const int nsy = last ? (more ? sy + 1 : sy) : sy;
                        constexpr int dummy = 0;
                        const int nj = last ? dummy : j + 1;
                        const double *ca = colc - (nj * LY + nsy), *cb = colm - ((SZ - 1 - nj) * LY + nsy);
                        const ctap_t *kk = Kt + (size_t)(nsy * SZ + nj) * SZ;
#pragma unroll
                        for (int w = 0; w < W; w++) va[w] = ca[w * plane];
                        if (nj < NP) {
#pragma unroll
                            for (int w = 0; w < W; w++) vb[w] = cb[w * plane];
                        }
#pragma unroll
                        for (int z = 0; z < SZ; z++) kn[z] = kk[z];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int z = 0; z < SZ; z++)
#pragma unroll
                        for (int r = 0; r < R; r++) acc[r] = fma(kc[z], sm[r + SZ - 1 - z], acc[r]);
#pragma unroll
                    for (int z = 0; z < SZ; z++) kc[z] = kn[z];
                }
            }
            continue;
        }
        double vc[W], kc[SZ]; // window and taps of the current (sx, sy)
        {
            const double *col = colc - c0;
            const ctap_t *kk = Kt + (size_t)c0 * SZ * SZ;
#pragma unroll
            for (int w = 0; w < W; w++) vc[w] = col[w * plane];
#pragma unroll
            for (int z = 0; z < SZ; z++) kc[z] = kk[z];
        }
        for (int sy = c0; sy < c0 + nch; sy++) {
            const bool more = sy + 1 < c0 + nch;
#pragma unroll
            for (int sx = 0; sx < SZ; sx++) {
                // The window (LDS) and the taps (wave-uniform loads) of this tap were requested a whole tap ago:
                // wait for them now, before the next requests go out, so that those have the length of the FMA
                // block below to arrive.
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
                double vn[W], kn[SZ];
                {
                    const bool last = sx == SZ - 1;
                    // after the last tap of a pass there is nothing to fetch: any valid address will do
                    const double *col = last ? colc - (more ? sy + 1 : sy) : colc - ((sx + 1) * LY + sy);
                    const ctap_t *kk = Kt + (size_t)(last ? (more ? (sy + 1) * SZ : sy * SZ) : sy * SZ + sx + 1) * SZ;
                    constexpr int WH = (W + 1) / 2;
                    const double *colb = col + WH * plane;
#pragma unroll
                    for (int w = 0; w < W; w++) vn[w] = w < WH ? col[w * plane] : colb[(w - WH) * plane];
#pragma unroll
                    for (int z = 0; z < SZ; z++) kn[z] = kk[z];
                }
                __builtin_amdgcn_sched_barrier(0); // keep the requests above ahead of the FMAs below
#pragma unroll
                for (int z = 0; z < SZ; z++)
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r] = fma(kc[z], vc[r + SZ - 1 - z], acc[r]);
#pragma unroll
                for (int w = 0; w < W; w++) vc[w] = vn[w];
#pragma unroll
                for (int z = 0; z < SZ; z++) kc[z] = kn[z];
            }
        }
    }
    }
    const int mx = X0 + tx, my = Y0 + ty;
    if (mx < Mx && my < My) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int mz = Z0 + tg * R + r;
            // (a range may run past the last plane and go on at plane 0: z0 + nz > Mz)
            if (mz < c.z0 + c.nz) e[((size_t)(mz >= Mz ? mz - Mz : mz) * My + my) * Mx + mx] = acc[r];
        }
    }
}

template <int SZ, int R, int TX, int TY, int TZ, int CH, bool SYM = false>
static bool conv_multi_launch(hipStream_t st, ConvBatch &B) {
    constexpr int CHR = CH < SZ ? CH : SZ;
    constexpr int LX = TX + SZ - 1, LY = ((TY + CHR - 1 + 1) / 4) * 4 + 2, LZ = TZ + SZ - 1;
    constexpr size_t bytes = sizeof(double) * (size_t)LX * LY * LZ;
    static_assert(bytes <= 160 * 1024, "tile exceeds the LDS");
    int nblk = 0;
    for (int l = 0; l < B.n; l++) {
        const ConvArgs &c = B.c[l];
        if (c.ks[0] != SZ || c.ks[1] != SZ || c.ks[2] != SZ || c.nsub < 1) return false;
        B.ntx[l] = (c.M[0] + TX - 1) / TX;
        B.nty[l] = (c.M[1] + TY - 1) / TY;
        B.blk0[l] = nblk;
        nblk += B.ntx[l] * B.nty[l] * ((c.nz + TZ - 1) / TZ);
    }
    B.blk0[B.n] = nblk;
    if (nblk == 0) return false;
    auto kern = k_conv_multi<SZ, R, TX, TY, TZ, CH, SYM>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(TX * TY * (TZ / R)), bytes, st, B);
    return true;
}

// The same convolution for a rank that owns a slab: few planes of level 0, so fewer 16 x 8 x 16 tiles than the chip
// has CUs and a launch as long as one tile takes.  Tiles of 16 x 8 x 8 outputs, and G groups of two waves to a tile
// that share the stencil's columns sx between them (same LDS tile, a G-th of the taps each): partial sums meet in
// LDS at the end and are added in the order of the groups.  Four waves per SIMD hide what k_conv_multi hides by
// prefetching, so the loop is the plain one.

template <int SZ, int G, int CH, bool SYM = false, int R = 8>
__global__ __launch_bounds__(128 * G, 4) void k_conv_split(ConvBatch B) {
    constexpr int TX = 16, TY = 8, TZ = R;
    extern __shared__ double lds[];
    int lev = 0;
    while (lev + 1 < B.n && (int)blockIdx.x >= B.blk0[lev + 1]) lev++;
    const ConvArgs c = B.c[lev];
    const double *__restrict__ q = B.q[lev];
    double *__restrict__ e = B.e[lev];
    const int ntx = B.ntx[lev], nty = B.nty[lev];
    const int Mx = c.M[0], My = c.M[1], Mz = c.M[2];
    constexpr int CHR = CH < SZ ? CH : SZ;
    constexpr int LX = TX + SZ - 1, LY = ((TY + CHR - 1 + 1) / 4) * 4 + 2, LZ = TZ + SZ - 1;
    static_assert(LX <= 64, "a row is staged by half a wave in at most two steps");
    const int bid = blockIdx.x - B.blk0[lev];
    const int bx = bid % ntx, by = (bid / ntx) % nty, bz = bid / (ntx * nty);
    const int X0 = bx * TX, Y0 = by * TY, Z0 = c.z0 + bz * TZ;
    constexpr int NT = 128 * G;
    const int t128 = threadIdx.x & 127, tx = t128 % TX, ty = t128 / TX;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
    constexpr int SXG = (SZ + G - 1) / G;
    const int sx0 = grp * SXG, nsx = min(SXG, SZ - sx0);
    constexpr int W = R + SZ - 1;
    constexpr int plane = LX * LY;
    const int half = threadIdx.x >> 5, hl = threadIdx.x & 31;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0;
    const int ncube = c.nsub * c.nsub * c.nsub;
    const int sy_lo = c.sy0, sy_hi = c.sy1 > 0 ? c.sy1 : SZ;
    for (int cube = 0; cube < ncube; cube++) {
        const int ca = cube % c.nsub, cb = (cube / c.nsub) % c.nsub, cc = cube / (c.nsub * c.nsub);
        const int ox = X0 - (c.kf[0] + ca * SZ + SZ - 1), oy = Y0 - (c.kf[1] + cb * SZ + SZ - 1), oz = Z0 - (c.kf[2] + cc * SZ + SZ - 1);
        const ctap_t *Kt = (const ctap_t *)(B.Kt[lev] + (size_t)cube * (SZ * SZ * SZ));
        const int gx0 = modp(ox + hl, Mx), gx1 = modp(ox + hl + 32, Mx);
        for (int c0 = sy_lo; c0 < sy_hi; c0 += CHR) {
            const int nch = min(CHR, sy_hi - c0);
            const int ys = SZ - c0 - nch;
            const int rows = TY + nch - 1;
            if (c0 > sy_lo || cube > 0) __syncthreads();
            {
                // staging as in k_conv_multi: half a wave per z plane, lanes along x, the rows one after the other
                constexpr int NH = NT / 32, BATCH = TY + CHR - 1 <= 12 ? TY + CHR - 1 : 8;
                const int yw0 = modp(oy + ys, My);
                for (int lz = half; lz < LZ; lz += NH) {
                    const double *pl = q + (size_t)modp(oz + lz, Mz) * ((size_t)My * Mx);
                    double *lrow = lds + lz * plane + hl * LY;
                    int yw = yw0;
                    for (int b = 0; b < rows; b += BATCH) {
                        double v0[BATCH], v1[LX > 32 ? BATCH : 1];
#pragma unroll
                        for (int k = 0; k < BATCH; k++) {
                            if (b + k < rows && hl < LX) v0[k] = pl[(size_t)yw * Mx + gx0];
                            if (LX > 32 && b + k < rows && hl + 32 < LX) v1[LX > 32 ? k : 0] = pl[(size_t)yw * Mx + gx1];
                            yw = yw + 1 == My ? 0 : yw + 1;
                        }
#pragma unroll
                        for (int k = 0; k < BATCH; k++) {
                            if (b + k < rows && hl < LX) lrow[b + k] = v0[k];
                            if (LX > 32 && b + k < rows && hl + 32 < LX) lrow[32 * LY + b + k] = v1[LX > 32 ? k : 0];
                        }
                    }
                }
            }
            __syncthreads();
            const double *colc = lds + (tx + SZ - 1) * LY + (ty + c0 + nch - 1);
            if (SYM) {
                // mirror columns j and SZ - 1 - j of a centred stencil together (see k_conv_multi): SZ / 2 pairs and the
                // middle column, shared out over the groups
                constexpr int NP = SZ / 2, SPG = (NP + 1 + G - 1) / G;
                for (int sy = c0; sy < c0 + nch; sy++) {
#pragma unroll
                    for (int s = 0; s < SPG; s++) {
                        const int j = grp * SPG + s;
                        if (j <= NP) {
                            const double *ca = colc - (j * LY + sy), *cb = colc - ((SZ - 1 - j) * LY + sy);
                            const ctap_t *kk = Kt + (size_t)(sy * SZ + j) * SZ;
                            double sm[W], kc[SZ];
#pragma unroll
                            for (int w = 0; w < W; w++) sm[w] = ca[w * plane];
                            if (j < NP) {
#pragma unroll
                                for (int w = 0; w < W; w++) sm[w] += cb[w * plane];
                            }
#pragma unroll
                            for (int z = 0; z < SZ; z++) kc[z] = kk[z];
#pragma unroll
                            for (int z = 0; z < SZ; z++)
#pragma unroll
                                for (int r = 0; r < R; r++) acc[r] = fma(kc[z], sm[r + SZ - 1 - z], acc[r]);
                        }
                    }
                }
                continue;
            }
            for (int sy = c0; sy < c0 + nch; sy++) {
#pragma unroll
                for (int s = 0; s < SXG; s++) {
                    if (s < nsx) {
                        const int sx = sx0 + s;
                        const double *col = colc - (sx * LY + sy);
                        const ctap_t *kk = Kt + (size_t)(sy * SZ + sx) * SZ;
                        double vc[W], kc[SZ];
#pragma unroll
                        for (int w = 0; w < W; w++) vc[w] = col[w * plane];
#pragma unroll
                        for (int z = 0; z < SZ; z++) kc[z] = kk[z];
#pragma unroll
                        for (int z = 0; z < SZ; z++)
#pragma unroll
                            for (int r = 0; r < R; r++) acc[r] = fma(kc[z], vc[r + SZ - 1 - z], acc[r]);
                    }
                }
            }
        }
    }
    // partial sums of the groups 1 .. G-1 through LDS (the tile is no longer needed), added in that order
    __syncthreads();
    if (grp > 0) {
#pragma unroll
        for (int r = 0; r < R; r++) lds[((grp - 1) * R + r) * 128 + t128] = acc[r];
    }
    __syncthreads();
    if (grp == 0) {
        const int mx = X0 + tx, my = Y0 + ty;
#pragma unroll
        for (int r = 0; r < R; r++) {
            double v = acc[r];
#pragma unroll
            for (int g = 1; g < G; g++) v += lds[((g - 1) * R + r) * 128 + t128];
            const int mz = Z0 + r;
            if (mx < Mx && my < My && mz < c.z0 + c.nz) e[((size_t)(mz >= Mz ? mz - Mz : mz) * My + my) * Mx + mx] = v;
        }
    }
}

template <int SZ, int G, int CH, bool SYM = false, int R = 8>
static bool conv_split_launch(hipStream_t st, ConvBatch &B) {
    constexpr int TX = 16, TY = 8, TZ = R, CHR = CH < SZ ? CH : SZ;
    constexpr int LX = TX + SZ - 1, LY = ((TY + CHR - 1 + 1) / 4) * 4 + 2, LZ = TZ + SZ - 1;
    constexpr size_t bytes = sizeof(double) * (size_t)LX * LY * LZ;
    static_assert(bytes <= (R == 8 ? 80 : 160) * 1024, "two workgroups of the 8-plane tile to a CU, one of the taller ones");
    static_assert(bytes >= sizeof(double) * (G - 1) * R * 128, "the partial sums fit the tile");
    int nblk = 0;
    for (int l = 0; l < B.n; l++) {
        const ConvArgs &c = B.c[l];
        if (c.ks[0] != SZ || c.ks[1] != SZ || c.ks[2] != SZ || c.nsub < 1) return false;
        B.ntx[l] = (c.M[0] + TX - 1) / TX;
        B.nty[l] = (c.M[1] + TY - 1) / TY;
        B.blk0[l] = nblk;
        nblk += B.ntx[l] * B.nty[l] * ((c.nz + TZ - 1) / TZ);
    }
    B.blk0[B.n] = nblk;
    if (nblk == 0) return false;
    auto kern = k_conv_split<SZ, G, CH, SYM, R>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(128 * G), bytes, st, B);
    return true;
}

// tile 16 x 8 x 16 outputs, 8 per thread; CH = rows of sy per staging pass, the most the 160 KB of LDS hold
bool launch_conv_multi(hipStream_t st, ConvBatch &B) {
    if (B.n < 1) return false;
    bool cubes = false; // (stencils that come as several cubes stay with k_conv_multi, which the tests exercise with them)
    for (int l = 0; l < B.n; l++) cubes = cubes || B.c[l].nsub != 1;
    // mirror symmetry along x can be used when every stencil of the launch is one cube centred on the node
    bool centred = !cubes;
    for (int l = 0; l < B.n; l++) centred = centred && (B.c[l].ks[0] & 1) && B.c[l].kf[0] == -(B.c[l].ks[0] / 2);
    if (!centred) B.sym = 0;
    if (!cubes && (B.split > 1 || (B.split && B.c[0].nz < B.c[0].M[2]))) { // a slab of planes (or on request)
        if (B.c[0].ks[2] == 15 && B.sym) {
            // Planes per tile.  Up to 256 workgroups run one to a CU and the launch lasts as long as one of them:
            // 12 + 4.5 us per plane of the tile, measured (the window reads from LDS and the FMAs take equally long and
            // do not overlap).  Beyond that some CUs run two at a time, which takes 1.4 times one of them if their LDS
            // tiles fit side by side and twice otherwise.  18 planes of 96 x 96 nodes (a rank of 8) are two layers of 9
            // planes, 198 workgroups with the coarser levels and 52 us, against three layers of 8, 288 workgroups and
            // 64 us; 18 planes of level 0 alone (the coarser levels on another stream) are three layers of 6.
            const int heights[5] = {8, 6, 9, 10, 12};
            int best = 8;
            double best_cost = 0;
            for (int k = 0; k < 5; k++) {
                const int r = heights[k];
                long n = 0;
                for (int l = 0; l < B.n; l++)
                    n += (long)((B.c[l].M[0] + 15) / 16) * ((B.c[l].M[1] + 7) / 8) * ((B.c[l].nz + r - 1) / r);
                const bool two_fit = 2 * 30 * 14 * (r + 14) * 8 <= 160 * 1024;
                const double one = 12.0 + 4.5 * r;
                const double cost = n <= 256 ? one : (two_fit ? 1.4 * (double)((n + 511) / 512) : (double)((n + 255) / 256)) * one;
                if (k == 0 || cost < 0.9 * best_cost) { best = r; best_cost = cost; }
            }
            if (B.tile_planes) best = B.tile_planes; // A/B
            if (best == 6 && conv_split_launch<15, 4, 5, true, 6>(st, B)) return true;
            if (best == 9 && conv_split_launch<15, 4, 5, true, 9>(st, B)) return true;
            if (best == 10 && conv_split_launch<15, 4, 5, true, 10>(st, B)) return true;
            if (best == 12 && conv_split_launch<15, 4, 5, true, 12>(st, B)) return true;
        }
        if (B.c[0].ks[2] == 15 && B.sym && conv_split_launch<15, 4, 5, true>(st, B)) return true;
        if (B.c[0].ks[2] == 15 && conv_split_launch<15, 4, 5>(st, B)) return true;
        if (B.c[0].ks[2] == 19 && B.sym == 2) {
            // A thick slab of a large grid (10M particles: 38 planes of 256 x 256 nodes on a rank of 8) has tiles enough
            // to fill the chip with the whole-grid kernel, which needs half the time per plane (16+ outputs per thread:
            // half the window reads per multiplication) but comes in tiles of 32 planes: 38 planes would cost 64.  With
            // 17, 18 or 19 outputs per thread a tile is 34, 36 or 38 planes tall (the LDS holds up to 38 + 18 planes of
            // 34 x 10 doubles), and the height that wastes the fewest planes is taken: 38 for 38, 36 for the 70 planes
            // of a rank of 4, 34 for the 134 of a rank of 2.
            const int nz = B.c[0].nz;
            const long per_layer = (long)((B.c[0].M[0] + 15) / 16) * ((B.c[0].M[1] + 7) / 8);
            int pick = 0;
            long best = 2L * 8 * ((nz + 7) / 8); // the split kernel: tiles of 8 planes at twice the time per plane
            for (int r = 16; r <= 19; r++) {
                const long layers = (nz + 2 * r - 1) / (2 * r);
                if ((per_layer * layers >= 256 || B.tile_planes == r) && 2 * r * layers < best) { best = 2 * r * layers; pick = r; }
            }
            if (B.tile_planes >= 16 && B.tile_planes <= 19) pick = B.tile_planes; // A/B
            if (B.tile_planes == 8) pick = 0;
            if (pick == 16 && conv_multi_launch<19, 16, 16, 8, 32, 3, true>(st, B)) return true;
            if (pick == 17 && conv_multi_launch<19, 17, 16, 8, 34, 3, true>(st, B)) return true;
            if (pick == 18 && conv_multi_launch<19, 18, 16, 8, 36, 3, true>(st, B)) return true;
            if (pick == 19 && conv_multi_launch<19, 19, 16, 8, 38, 3, true>(st, B)) return true;
        }
        if (B.c[0].ks[2] == 19 && B.sym && conv_split_launch<19, 4, 3, true>(st, B)) return true;
        if (B.c[0].ks[2] == 19 && conv_split_launch<19, 4, 3>(st, B)) return true;
    }
    switch (B.c[0].ks[2]) {
    case 7: return conv_multi_launch<7, 8, 16, 8, 16, 7>(st, B);
    case 9: return conv_multi_launch<9, 8, 16, 8, 16, 9>(st, B);
    case 11: return conv_multi_launch<11, 8, 16, 8, 16, 11>(st, B);
    case 13: return conv_multi_launch<13, 8, 16, 8, 16, 13>(st, B);
    case 15:
        // Mirror columns together.  With 8 outputs per thread that changes nothing (157 against 154 us: the kernel is bound
        // by its window reads from LDS, 22 per column, not by the FMAs); with 16 outputs per thread - workgroups of two
        // waves, one wave per SIMD, the next pass's rows prefetched into registers - the reads per FMA halve: 121 us.
        if (B.sym == 2) return conv_multi_launch<15, 16, 16, 8, 16, 3, true>(st, B);
        if (B.sym) return conv_multi_launch<15, 8, 16, 8, 16, 3, true>(st, B);
        return conv_multi_launch<15, 8, 16, 8, 16, 3>(st, B); // 30 x 10 x 30: two workgroups per CU (2 % faster than the whole halo at once)
    case 17: return conv_multi_launch<17, 8, 16, 8, 16, 11>(st, B); // 32 x 18 x 32 doubles
    case 19:
        if (B.sym == 2) return conv_multi_launch<19, 16, 16, 8, 32, 3, true>(st, B); // 34 x 10 x 50
        return conv_multi_launch<19, 8, 16, 8, 16, 7>(st, B);  // 34 x 14 x 34
    case 21: return conv_multi_launch<21, 8, 16, 8, 16, 7>(st, B);  // 36 x 14 x 36
    case 23: return conv_multi_launch<23, 8, 16, 8, 16, 7>(st, B);  // 38 x 14 x 38
    default:
        return false;
    }
}

// The reference's grid-to-grid mapping as written (msmPeriodic.c:874-918, 962-998): dense double loop over
// all node pairs, stencil offset (mx - nz, my - ny, mz - nz) wrapped into [first, first+size-1] by the
// stencil size.  Used in literal mode only.
__global__ __launch_bounds__(64) void k_conv_literal(const double *__restrict__ q, const double *__restrict__ K,
                                                     double *__restrict__ e, ConvArgs c) {
    const int Mx = c.M[0], My = c.M[1], Mz = c.M[2];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Mx * My * Mz) return;
    const int mx = idx % Mx, my = (idx / Mx) % My, mz = idx / (Mx * My);
    double s = 0.0;
    for (int nz = 0; nz < Mz; nz++) {
        int ox = mx - nz, oz = mz - nz;
        const int lox = c.kf[0], hix = c.kf[0] + c.ks[0] - 1, loz = c.kf[2], hiz = c.kf[2] + c.ks[2] - 1;
        while (ox < lox) ox += c.ks[0];
        while (ox > hix) ox -= c.ks[0];
        while (oz < loz) oz += c.ks[2];
        while (oz > hiz) oz -= c.ks[2];
        for (int ny = 0; ny < My; ny++) {
            int oy = my - ny;
            const int loy = c.kf[1], hiy = c.kf[1] + c.ks[1] - 1;
            while (oy < loy) oy += c.ks[1];
            while (oy > hiy) oy -= c.ks[1];
            const double kv = K[((size_t)(oz - loz) * c.ks[1] + (oy - loy)) * c.ks[0] + (ox - lox)];
            const double *qrow = q + ((size_t)nz * My + ny) * Mx;
            for (int nx = 0; nx < Mx; nx++) s = fma(kv, qrow[nx], s);
        }
    }
    e[idx] = s;
}

void launch_conv_literal(hipStream_t st, const double *q, const double *K, const int M[3], const int ksize[3],
                         const int kfirst[3], double *e) {
    ConvArgs c;
    for (int d = 0; d < 3; d++) { c.M[d] = M[d]; c.ks[d] = ksize[d]; c.kf[d] = kfirst[d]; }
    c.z0 = 0; c.nz = M[2]; c.accumulate = 0;
    const int n = M[0] * M[1] * M[2];
    hipLaunchKernelGGL(k_conv_literal, dim3((n + 63) / 64), dim3(64), 0, st, q, K, e, c);
}

__global__ void k_axpy(double *__restrict__ y, const double *__restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
__global__ void k_add_parts(double *__restrict__ y, const double *__restrict__ a, const double *__restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = y[i] + a[i];
    if (b) v += b[i];
    y[i] = v;
}
void launch_add_parts(hipStream_t st, double *y, const double *a, const double *b, size_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_add_parts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, a, b, n);
}
void launch_axpy(hipStream_t st, double *y, const double *x, size_t n) {
    hipLaunchKernelGGL(k_axpy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, n);
}

// ---------------------------------------------------------------------------
// reductions (fixed summation trees: results do not depend on scheduling)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dot_partials(const double *__restrict__ a, const double *__restrict__ b, size_t n,
                                                      double *__restrict__ parts) {
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s = fma(a[i], b[i], s);
    const double t = block_sum<256>(s, red);
    if (threadIdx.x == 0) parts[blockIdx.x] = t;
}
void launch_dot_partials(hipStream_t st, const double *a, const double *b, size_t n, double *parts, int nparts) {
    hipLaunchKernelGGL(k_dot_partials, dim3(nparts), dim3(256), 0, st, a, b, n, parts);
}

__global__ __launch_bounds__(256) void k_charge_partials(const double *__restrict__ q, size_t n, double *__restrict__ parts,
                                                         int nparts) {
    __shared__ double red[256];
    double s = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = q[i];
        s += v;
        s2 = fma(v, v, s2);
    }
    const double t = block_sum<256>(s, red);
    __syncthreads();
    const double t2 = block_sum<256>(s2, red);
    if (threadIdx.x == 0) {
        parts[blockIdx.x] = t;
        parts[nparts + blockIdx.x] = t2;
    }
}
void launch_charge_partials(hipStream_t st, const double *q, size_t n, double *parts, int nparts) {
    hipLaunchKernelGGL(k_charge_partials, dim3(nparts), dim3(256), 0, st, q, n, parts, nparts);
}

// parts[b] = sum of block b's contiguous share of in[0..n): shortens a long list of partial sums in a fixed order
__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ in, int n, double *__restrict__ parts) {
    __shared__ double red[256];
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    double s = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += in[i];
    const double t = block_sum<256>(s, red);
    if (threadIdx.x == 0) parts[blockIdx.x] = t;
}
void launch_sum_partials(hipStream_t st, const double *in, int n, double *parts, int nparts) {
    hipLaunchKernelGGL(k_sum_partials, dim3(nparts), dim3(256), 0, st, in, n, parts);
}

// COHERENT: the values were left by other workgroups of the same launch with device-scope atomic stores; they are read
// with device-scope atomic loads, which no cache of this compute unit or chiplet serves
template <bool COHERENT = false>
__device__ double sum_array(const double *__restrict__ p, int n, double *red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256)
        s += COHERENT ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i];
    const double t = block_sum<256>(s, red);
    __syncthreads();
    return t;
}

// The four partial reductions of an evaluation in one launch (each a few microseconds of work, and every launch in
// the chain costs about as much again): blockIdx.y picks the reduction, blockIdx.x the part; the sums are formed
// exactly as by k_dot_partials, k_charge_partials and k_sum_partials with gridDim.x parts.
__global__ __launch_bounds__(256) void k_energy_partials(EnergyPartialsArgs a) {
    __shared__ double red[256];
    const int part = blockIdx.x, np = gridDim.x;
    if (blockIdx.y <= 1) {
        const double *__restrict__ x = blockIdx.y == 0 ? a.g_q : a.f_q;
        const double *__restrict__ y = blockIdx.y == 0 ? a.g_e : a.f_e;
        const size_t n = blockIdx.y == 0 ? a.g_n : a.f_n;
        double s = 0.0;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < n; i += (size_t)np * 256) s = fma(x[i], y[i], s);
        const double t = block_sum<256>(s, red);
        if (threadIdx.x == 0) (blockIdx.y == 0 ? a.g_parts : a.f_parts)[part] = t;
    } else if (blockIdx.y == 2) {
        double s = 0.0, s2 = 0.0;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < a.c_n; i += (size_t)np * 256) {
            const double v = a.c_q[i];
            s += v;
            s2 = fma(v, v, s2);
        }
        const double t = block_sum<256>(s, red);
        __syncthreads();
        const double t2 = block_sum<256>(s2, red);
        if (threadIdx.x == 0) {
            a.c_parts[part] = t;
            a.c_parts[np + part] = t2;
        }
    } else {
        if (!a.p_parts) return; // a short list of pair partials is summed by the finishing kernel itself
        const int per = (a.p_n + np - 1) / np;
        const int lo = part * per, hi = min(a.p_n, lo + per);
        double s = 0.0;
        for (int i = lo + threadIdx.x; i < hi; i += 256) s += a.p_in[i];
        const double t = block_sum<256>(s, red);
        if (threadIdx.x == 0) a.p_parts[part] = t;
    }
}
void launch_energy_partials(hipStream_t st, const EnergyPartialsArgs &a, int nparts) {
    hipLaunchKernelGGL(k_energy_partials, dim3(nparts, 4), dim3(256), 0, st, a);
}

// (A)..(F) and the total (msmPeriodic.c:1305-1359; msmSimple.c:1968-1969) from the partial sums, by one workgroup
template <bool COHERENT = false>
__device__ __forceinline__ void finalize_energies(const EnergyInputs &in, double *__restrict__ out, double *__restrict__ out2, double *red,
                                                  bool pair_parts_coherent = false) {
    const double ua = (COHERENT && pair_parts_coherent) ? sum_array<true>(in.pair_parts, in.n_pair, red) : sum_array(in.pair_parts, in.n_pair, red);
    const double qe = sum_array<COHERENT>(in.grid_parts, in.n_grid, red);
    const double qf = sum_array<COHERENT>(in.four_parts, in.n_four, red);
    const double qs = sum_array<COHERENT>(in.charge_parts, in.n_charge, red);
    const double qs2 = sum_array<COHERENT>(in.charge_parts + in.n_charge, in.n_charge, red);
    if (threadIdx.x == 0) {
        double ulr, ulf;
        if (in.literal) { // msmPeriodic.c:1002 (minus sign), :1334-1338 (edir only)
            ulr = 0.5 * qe;
            ulf = -0.5 * qf;
        } else {
            ulf = 0.5 * qf;
            ulr = 0.5 * qe - ulf;
        }
        out[MSM_AMD_U_SHORT_REAL] = ua;
        out[MSM_AMD_U_SHORT_SELF] = 0.5 * qs2 * in.self_image_sum;
        out[MSM_AMD_U_SHORT_CSR] = -0.5 * qs * qs * in.csr;
        out[MSM_AMD_U_LONG_REAL] = ulr;
        out[MSM_AMD_U_LONG_FOUR] = ulf;
        out[MSM_AMD_U_LONG_SELF] = -0.5 * qs2 * in.gamma0_over_a;
        out[MSM_AMD_U_TOTAL] = (ulf + ulr + out[MSM_AMD_U_LONG_SELF]) +
                               (ua + out[MSM_AMD_U_SHORT_SELF] + out[MSM_AMD_U_SHORT_CSR]);
        if (out2) // a second copy, e.g. straight into device-visible host memory (saves a copy command)
            for (int k = 0; k < MSM_AMD_U_COUNT; k++) out2[k] = out[k];
    }
}
// raw sums for a slab rank: [pair sum, q0.e0 over owned planes, qL.efour, sum q, sum q^2]
template <bool COHERENT = false>
__device__ __forceinline__ void raw_sums(const EnergyInputs &in, double *__restrict__ out, double *red, bool pair_parts_coherent = false) {
    const double ua = (COHERENT && pair_parts_coherent) ? sum_array<true>(in.pair_parts, in.n_pair, red) : sum_array(in.pair_parts, in.n_pair, red);
    const double qe = sum_array<COHERENT>(in.grid_parts, in.n_grid, red);
    const double qf = sum_array<COHERENT>(in.four_parts, in.n_four, red);
    const double qs = sum_array<COHERENT>(in.charge_parts, in.n_charge, red);
    const double qs2 = sum_array<COHERENT>(in.charge_parts + in.n_charge, in.n_charge, red);
    if (threadIdx.x == 0) { out[0] = ua; out[1] = qe; out[2] = qf; out[3] = qs; out[4] = qs2; }
}
__global__ __launch_bounds__(256) void k_finalize_energies(EnergyInputs in, double *__restrict__ out, double *__restrict__ out2) {
    __shared__ double red[256];
    finalize_energies(in, out, out2, red);
}
__global__ __launch_bounds__(256) void k_raw_sums(EnergyInputs in, double *__restrict__ out) {
    __shared__ double red[256];
    raw_sums(in, out, red);
}

// The partial reductions and the finishing step in ONE launch (a kernel boundary in the dependent chain of an
// evaluation costs about as much as either of them): every workgroup leaves its partial sum and takes a ticket; the
// workgroup that draws the last ticket - whichever it is - finishes from the partial sums of all, which it adds in
// their fixed order, so the result does not depend on who finishes.  Nothing waits for anything: a workgroup that is
// not the last simply ends.  The partial sums travel as device-scope atomics (an exchange whose return the thread waits
// for before it takes its ticket; atomic loads on the finishing side), which are performed where every chiplet sees
// them: a release / acquire fence instead writes back and invalidates the whole L2 of the chiplet, once per workgroup,
// and made this kernel three times slower than the two launches it replaces.  The ticket counter returns to 0 for the
// next evaluation (evaluations of one handle are ordered on its stream).
__device__ __forceinline__ void publish(double *p, double v) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_energy_all(EnergyPartialsArgs a, EnergyInputs in, int raw, double *__restrict__ out,
                                                    double *__restrict__ out2, unsigned *__restrict__ ticket) {
    __shared__ double red[256];
    __shared__ int last;
    const int part = blockIdx.x, np = gridDim.x;
    if (blockIdx.y <= 1) {
        const double *__restrict__ x = blockIdx.y == 0 ? a.g_q : a.f_q;
        const double *__restrict__ y = blockIdx.y == 0 ? a.g_e : a.f_e;
        const size_t n = blockIdx.y == 0 ? a.g_n : a.f_n;
        double s = 0.0;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < n; i += (size_t)np * 256) s = fma(x[i], y[i], s);
        const double t = block_sum<256>(s, red);
        if (threadIdx.x == 0) publish((blockIdx.y == 0 ? a.g_parts : a.f_parts) + part, t);
    } else if (blockIdx.y == 2) {
        double s = 0.0, s2 = 0.0;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < a.c_n; i += (size_t)np * 256) {
            const double v = a.c_q[i];
            s += v;
            s2 = fma(v, v, s2);
        }
        const double t = block_sum<256>(s, red);
        __syncthreads();
        const double t2 = block_sum<256>(s2, red);
        if (threadIdx.x == 0) {
            publish(a.c_parts + part, t);
            publish(a.c_parts + np + part, t2);
        }
    } else if (a.p_parts) { // (a short list of pair partials is summed by the finishing step itself)
        const int per = (a.p_n + np - 1) / np;
        const int lo = part * per, hi = min(a.p_n, lo + per);
        double s = 0.0;
        for (int i = lo + threadIdx.x; i < hi; i += 256) s += a.p_in[i];
        const double t = block_sum<256>(s, red);
        if (threadIdx.x == 0) publish(a.p_parts + part, t);
    }
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the exchanges above have returned: they are performed
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x * gridDim.y - 1;
    }
    __syncthreads();
    if (!last) return;
    if (raw) raw_sums<true>(in, out, red, a.p_parts != nullptr);
    else finalize_energies<true>(in, out, out2, red, a.p_parts != nullptr);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_energy_all(hipStream_t st, const EnergyPartialsArgs &a, int nparts, const EnergyInputs &in, bool raw, double *out,
                       double *out_second, unsigned *ticket) {
    hipLaunchKernelGGL(k_energy_all, dim3(nparts, 4), dim3(256), 0, st, a, in, raw ? 1 : 0, out, out_second, ticket);
}
void launch_raw_sums(hipStream_t st, const EnergyInputs &in, double *out5) {
    MSM_LAUNCH_TAIL(k_raw_sums, dim3(1), dim3(256), 0, st, in, out5);
}
void launch_finalize_energies(hipStream_t st, const EnergyInputs &in, double *out7, double *out7_second) {
    hipLaunchKernelGGL(k_finalize_energies, dim3(1), dim3(256), 0, st, in, out7, out7_second);
}

} // namespace msm
