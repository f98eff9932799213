// Tiled anterpolation (gfx950): particle charges onto the finest grid, q_m = sum_i q_i Phi(s_i - m)
// (msmPeriodic.c:688-717, msmSimple.c anterpolation), without global atomics and bitwise reproducible.
//
// The grid cells are cut into tiles of 8^3.  One workgroup spreads the particles of one tile into a private
// (8 + p - 1)^3 block of nodes in LDS: each of its four waves owns a copy of the block and walks its share of
// the tile's particles one at a time, the lanes being the nodes of the particle's support (all 4^3 of them for
// p = 4; the 6 x 6 nodes of an xy plane, six planes in turn, for p = 6).  The weights of a lane are polynomials
// in the particle's fractional coordinate whose coefficients the lane keeps in registers, so there is no
// branching.  The four copies are added in a fixed order and stored; a second kernel adds, for every node, the
// blocks of the (at most 27) tiles that reach it, again in a fixed order.
#include "msm_kernels.hpp"
#include "msm_device.hpp"

#include "msm_amd.h"

#include <algorithm>

namespace msm {

#define AT SPLINE_TILE

__device__ __forceinline__ double readlane_d(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

template <int P>
__device__ __forceinline__ double piece_eval(const double *c, double f) {
    double w = c[P - 1];
#pragma unroll
    for (int n = P - 2; n >= 0; n--) w = fma(w, f, c[n]);
    return w;
}

// The particles of a tile are a few runs of consecutive slots of a sorted particle array: in grid-cell order
// one run per (y, z) row of cells; in cluster order (columns of kx x ky cells, sorted by z) one run per column.
__device__ __forceinline__ int tile_runs(const TileSrc &S, int nxc, int nyc, int nzc) {
    return S.cluster_order ? (nxc / S.cg.kx) * (nyc / S.cg.ky) : nyc * nzc;
}
__device__ __forceinline__ void tile_run(const TileSrc &S, const SplineParams &sp, int x0, int y0, int z0, int nxc, int nyc,
                                         int nzc, int t, int &first, int &cnt) {
    if (S.cluster_order) {
        const int ncx = nxc / S.cg.kx;
        const int col = cl_col(S.cg, x0 / S.cg.kx + t % ncx, y0 / S.cg.ky + t / ncx);
        const int b = col * S.cg.nzf;
        const int colp0 = S.start[b], p0 = S.start[b + z0 * S.cg.kz], p1 = S.start[b + (z0 + nzc) * S.cg.kz];
        first = CL_G * S.clstart[col] + (p0 - colp0);
        cnt = p1 - p0;
    } else {
        const int b0 = ((z0 + t / nyc) * sp.M[1] + (y0 + t % nyc)) * sp.M[0] + x0;
        first = S.start[b0];
        cnt = S.start[b0 + nxc] - first;
    }
}

// cell of a coordinate (grid units), kept inside the tile whatever the last bit of the product did
__device__ __forceinline__ int tile_cell(double u, int c0, int n) { return min(max((int)u, c0), c0 + n - 1); }

// Fire-and-forget addition in LDS (ds_add_f64): nothing comes back, so consecutive particles do not wait for
// each other's round trip.  The block a wave adds into is private to it and LDS serves a wave's requests in
// program order, so every node still receives its terms in a fixed order.
__device__ __forceinline__ void lds_add(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// tile of a workgroup when a launch covers some layers of tiles only (TileGeom::tz0)
__device__ __forceinline__ int launched_tile(const TileGeom &tl, int b) {
    const int per = tl.nt[0] * tl.nt[1];
    const int layer = b / per;
    return ((tl.tz0 + layer) % tl.nt[2]) * per + (b - layer * per);
}

// The z cells [clo, chi] of a tile (cells z0 .. z0 + nzc - 1 of the grid) whose particles can reach a plane the rank owns:
// a particle in cell c adds to the planes c - p/2 + 1 .. c + p/2, so the cells sp.z0 - p/2 .. sp.z0 + sp.nz - 1 + p/2 - 1
// count, and one more either side for a particle whose cell and slab were rounded differently.  (A rank holds halo
// copies out to the cutoff of the pair sum, which is wider.)  Cells that do not count are skipped: their particles add
// to planes of other ranks only, so the sums of the owned planes keep their terms and their order.  When the cells that
// count are not one run inside the tile (the range meets itself around the period), all are kept.
__device__ __forceinline__ void anterp_cells(const SplineParams &sp, int z0, int nzc, int &clo, int &chi) {
    clo = 0;
    chi = nzc - 1;
    const int Mz = sp.M[2], len = sp.nz + sp.p + 1;
    if (sp.nz >= Mz || len >= Mz) return;
    const int lo = modp(sp.z0 - sp.p / 2 - 1, Mz);
    int first = -1, lastc = -1, nin = 0;
    for (int c = 0; c < nzc; c++) {
        const bool in = modp(z0 + c - lo, Mz) < len;
        if (in) { if (first < 0) first = c; lastc = c; nin++; }
    }
    if (nin == 0) { clo = 0; chi = -1; return; }
    if (lastc - first + 1 != nin) return;
    clo = first;
    chi = lastc;
}

template <int P>
__global__ __launch_bounds__(256) void k_anterp_tiles(TileSrc S, SplineParams sp, TileGeom tl, double *__restrict__ scratch,
                                                      int *__restrict__ tile_used) {
    constexpr int R = AT + P - 1, R3 = R * R * R;
    extern __shared__ double lds[]; // 4 copies of the node block
    __shared__ int any_particle;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *__restrict__ mine = lds + wave * R3;
    const int tile = launched_tile(tl, blockIdx.x);
    const int tx = tile % tl.nt[0], ty = (tile / tl.nt[0]) % tl.nt[1], tz = tile / (tl.nt[0] * tl.nt[1]);
    const int Mx = sp.M[0], My = sp.M[1], Mz = sp.M[2];
    const int x0 = tx * AT, y0 = ty * AT, z0 = tz * AT;
    const int nxc = min(AT, Mx - x0), nyc = min(AT, My - y0), nzc = min(AT, Mz - z0);
    const int nruns = tile_runs(S, nxc, nyc, nzc);
    // cluster order: only the z cells whose particles reach the rank's planes (runs are columns, sorted by z)
    int zr0 = z0, nzr = nzc;
    if (S.cluster_order) {
        int clo, chi;
        anterp_cells(sp, z0, nzc, clo, chi);
        zr0 = z0 + clo;
        nzr = chi - clo + 1;
    }
    // an empty tile (most tiles of a slab rank) is only flagged; the combine kernel then skips its block
    if (threadIdx.x == 0) any_particle = 0;
    __syncthreads();
    if ((int)threadIdx.x < nruns && nzr > 0) {
        int first, cnt;
        tile_run(S, sp, x0, y0, zr0, nxc, nyc, nzr, threadIdx.x, first, cnt);
        if (cnt > 0) any_particle = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_used[tile] = any_particle;
    if (!any_particle) return;
    for (int t = threadIdx.x; t < 4 * R3; t += 256) lds[t] = 0.0;
    __syncthreads();
    // the nodes this lane adds to, relative to the first node of a particle's support
    const int a = P == 4 ? (lane & 3) : lane % 6, b = P == 4 ? ((lane >> 2) & 3) : (lane / 6) % 6, c = P == 4 ? (lane >> 4) : 0;
    const bool active = P == 4 || lane < 36;
    double cx[P], cy[P], cz[P];
#pragma unroll
    for (int n = 0; n < P; n++) {
        cx[n] = sp.piece[a][n];
        cy[n] = sp.piece[b][n];
        cz[n] = sp.piece[c][n];
    }
    const int lane_off = (c * R + b) * R + a;
    for (int run = wave; run < nruns; run += 4) {
        int s0, n;
        tile_run(S, sp, x0, y0, zr0, nxc, nyc, nzr, run, s0, n);
        for (int sb = 0; sb < n; sb += 64) {
            const int cnt = min(64, n - sb);
            // every lane prepares one particle of the batch: cell, offset inside the cell, and (p = 6) the six
            // z weights times the charge; the batch is then spread one particle at a time
            const double4 ptl = lane < cnt ? S.ps[s0 + sb + lane] : make_double4(0, 0, 0, 0);
            const double ux = ptl.x * S.scale[0], uy = ptl.y * S.scale[1], uz = ptl.z * S.scale[2];
            const int celx = tile_cell(ux, x0, nxc), cely = tile_cell(uy, y0, nyc), celz = tile_cell(uz, z0, nzc);
            const double lfx = ux - (double)celx, lfy = uy - (double)cely, lfz = uz - (double)celz;
            const int lbase = ((celz - z0) * R + (cely - y0)) * R + (celx - x0);
            double lwz[P];
            if (P == 6) {
#pragma unroll
                for (int cc = 0; cc < P; cc++) lwz[cc] = ptl.w * piece_eval<P>(sp.piece[cc], lfz);
            }
            for (int k = 0; k < cnt; k++) {
                const double fx = readlane_d(lfx, k), fy = readlane_d(lfy, k);
                const int base = __builtin_amdgcn_readlane(lbase, k) + lane_off;
                const double wxy = piece_eval<P>(cx, fx) * piece_eval<P>(cy, fy);
                if (P == 4) {
                    const double fz = readlane_d(lfz, k), pq = readlane_d(ptl.w, k);
                    lds_add(mine + base, pq * wxy * piece_eval<P>(cz, fz));
                } else {
                    double wz[P];
#pragma unroll
                    for (int cc = 0; cc < P; cc++) wz[cc] = readlane_d(lwz[cc], k);
                    if (active) {
#pragma unroll
                        for (int cc = 0; cc < P; cc++) lds_add(mine + base + cc * R * R, wxy * wz[cc]);
                    }
                }
            }
        }
    }
    __syncthreads();
    double *__restrict__ out = scratch + (size_t)tile * R3;
    for (int t = threadIdx.x; t < R3; t += 256) out[t] = ((lds[t] + lds[R3 + t]) + lds[2 * R3 + t]) + lds[3 * R3 + t];
}

// The same spreading for particles in cluster order, accumulated in registers.  The particles of a tile come as
// runs, one per column of KX x KY cells, sorted by z.  All particles of a column reach the same
// (KX + P - 1) x (KY + P - 1) nodes in x and y: these are the lanes, each with the whole z extent of the tile's node
// block (8 + P - 1 sums) in registers.  A particle in cell (dx, dy, cz) of its column adds w_x w_y w_z q to P
// consecutive sums starting at cz; dx and dy only select which polynomial piece a lane evaluates (zero outside
// the particle's support), and the walk visits the z cells in turn so that the register indices are fixed at
// compile time.  The LDS block is touched once per column instead of P times per particle.
// WTS: the staged record of a particle holds its weights themselves - the P z weights times the charge, and the x and y
// weights of all FX and FY nodes of the column's footprint (zero outside the particle's support) - instead of the
// offsets from which every lane evaluated its own two polynomial pieces: the lane that stages a particle evaluates the
// 3 P pieces once, a lane of the walk reads "its" x and y weight (two 8-byte reads, 7 distinct addresses each) and is left
// with one multiplication and P additions per particle.  Same arithmetic, same bits; a third of the vector
// instructions and four fifths of the LDS bytes per particle.  Records are twice as long, so a batch is 32 particles.
template <int P, int KX, int KY, bool WTS = false>
__global__ __launch_bounds__(1024) void k_anterp_cols(TileSrc S, SplineParams sp, TileGeom tl, double *__restrict__ scratch,
                                                     int *__restrict__ tile_used) {
    constexpr int R = AT + P - 1, R3 = R * R * R, FX = KX + P - 1, FY = KY + P - 1;
    constexpr int MAXRUNS = (AT / KX) * (AT / KY);
    static_assert(FX * FY <= 64, "one lane per (x, y) node of a column's support");
    extern __shared__ __align__(16) double lds[]; // the node block of the tile, then one staging area per wave
    __shared__ int sbnd[MAXRUNS][AT + 1]; // per run: the slots at which the z cells of the tile begin
    __shared__ int any_particle;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = launched_tile(tl, blockIdx.x);
    const int tx = tile % tl.nt[0], ty = (tile / tl.nt[0]) % tl.nt[1], tz = tile / (tl.nt[0] * tl.nt[1]);
    const int x0 = tx * AT, y0 = ty * AT, z0 = tz * AT;
    const int nxc = min(AT, sp.M[0] - x0), nyc = min(AT, sp.M[1] - y0), nzc = min(AT, sp.M[2] - z0);
    const int ncxt = nxc / KX, nruns = ncxt * (nyc / KY);
    const int kz = S.cg.kz, nzf = S.cg.nzf;
    if (threadIdx.x == 0) any_particle = 0;
    // all the bookkeeping of the tile's runs up front, one (run, z cell) per thread: the waves then never wait for
    // a chain of dependent index loads between two runs
    const int nw = blockDim.x >> 6; // waves of the workgroup: 4 to 16, one run each at a time
    int clo, chi; // z cells of the tile whose particles reach the rank's planes: the others get empty ranges
    anterp_cells(sp, z0, nzc, clo, chi);
    for (int t = threadIdx.x; t < nruns * (AT + 1); t += blockDim.x) {
        const int run = t / (AT + 1), c = min(max(t % (AT + 1), clo), chi + 1);
        const int col = cl_col(S.cg, x0 / KX + run % ncxt, y0 / KY + run / ncxt), bb = col * nzf;
        // slot of a particle = CL_G * (first cluster of its column) + its rank inside the column
        sbnd[run][t % (AT + 1)] = CL_G * S.clstart[col] - S.start[bb] + S.start[bb + (z0 + min(c, nzc)) * kz];
    }
    __syncthreads();
    if ((int)threadIdx.x < nruns && sbnd[threadIdx.x][AT] > sbnd[threadIdx.x][0]) any_particle = 1;
    __syncthreads();
    if (threadIdx.x == 0) tile_used[tile] = any_particle;
    if (!any_particle) return;
    for (int t = threadIdx.x; t < R3; t += blockDim.x) lds[t] = 0.0;
    __syncthreads();
    // doubles per staged particle record (even: 16-byte reads); a wave's staging area holds 64 records of offsets or 32 of weights
    constexpr int REC = WTS ? ((P + FX + FY) + 1) / 2 * 2 : ((2 + P + 1) + 1) / 2 * 2;
    constexpr int STAGE = WTS ? 32 * REC : 64 * REC;
    static_assert(STAGE <= 64 * 10, "the launcher reserves 640 doubles per wave");
    double *__restrict__ stage = lds + ((R3 + 1) / 2 * 2) + wave * 640;
    const int a = lane % FX, b = lane / FX;
    const bool active = lane < FX * FY;
    double cx[KX][P], cy[KY][P];
#pragma unroll
    for (int d = 0; d < KX; d++)
#pragma unroll
        for (int n = 0; n < P; n++) cx[d][n] = (a - d >= 0 && a - d < P) ? sp.piece[a - d][n] : 0.0;
#pragma unroll
    for (int d = 0; d < KY; d++)
#pragma unroll
        for (int n = 0; n < P; n++) cy[d][n] = (b - d >= 0 && b - d < P) ? sp.piece[b - d][n] : 0.0;
    const double4 none = make_double4(0, 0, 0, 0);
    // the first 64 particles of a run are requested while the run before it is being spread
    double4 pre = none;
    if (wave < nruns && sbnd[wave][0] + lane < sbnd[wave][AT]) pre = S.ps[sbnd[wave][0] + lane];
    // The waves take one run each at a time and then add their sums to the block one wave after the other, so
    // that every node receives its terms in a fixed order from a single copy of the block (a copy per wave would
    // leave room for two workgroups per CU only).
    for (int run0 = 0; run0 < nruns; run0 += nw) {
        const int run = min(run0 + wave, nruns - 1); // a wave without a run of its own repeats the last one, unused
        const bool mine = run0 + wave < nruns;
        const int colx = x0 / KX + run % ncxt, coly = y0 / KY + run / ncxt;
        int bnd[AT + 1];
#pragma unroll
        for (int c = 0; c <= AT; c++) bnd[c] = mine ? sbnd[run][c] : 0;
        const double4 first64 = pre;
        pre = none;
        if (run + nw < nruns && sbnd[run + nw][0] + lane < sbnd[run + nw][AT]) pre = S.ps[sbnd[run + nw][0] + lane];
        double acc[R];
#pragma unroll
        for (int z = 0; z < R; z++) acc[z] = 0.0;
        for (int sb = bnd[0]; sb < bnd[AT]; sb += 64) {
            const int cnt = min(64, bnd[AT] - sb);
            const int slot = sb + lane;
            double4 ptl = first64;
            if (sb > bnd[0]) ptl = lane < cnt ? S.ps[slot] : none;
            int cz = 0;
#pragma unroll
            for (int c = 1; c < AT; c++) cz += slot >= bnd[c];
            const double ux = ptl.x * S.scale[0], uy = ptl.y * S.scale[1], uz = ptl.z * S.scale[2];
            // cell inside the column, kept there whatever the last bit of the product did; the z cell is the one the
            // sort put the particle in
            const int dx = min(max((int)ux - colx * KX, 0), KX - 1), dy = min(max((int)uy - coly * KY, 0), KY - 1);
            const double lfx = ux - (double)(colx * KX + dx), lfy = uy - (double)(coly * KY + dy), lfz = uz - (double)(z0 + cz);
            if constexpr (WTS) {
                const int ai = min(a, FX - 1), bi = min(b, FY - 1); // (lanes beyond the footprint read a valid word, unused)
                // every lane evaluates the weights of its own particle once (registers); the two halves of the wave then
                // stage their records one after the other
                double rw[REC];
                {
                    // the spline is even: node P-1-n at offset f weighs what node n weighs at 1 - f
                    const double lgz = 1.0 - lfz;
#pragma unroll
                    for (int n = 0; n < P / 2; n++) {
                        rw[n] = ptl.w * piece_eval<P>(sp.piece[n], lfz);
                        rw[P - 1 - n] = ptl.w * piece_eval<P>(sp.piece[n], lgz);
                    }
                    double vx[P], vy[P];
#pragma unroll
                    for (int n = 0; n < P; n++) {
                        // (exactly the pieces the other variant evaluates: piece n at the offset itself)
                        vx[n] = piece_eval<P>(sp.piece[n], lfx);
                        vy[n] = piece_eval<P>(sp.piece[n], lfy);
                    }
#pragma unroll
                    for (int q = 0; q < FX; q++) {
                        double v = 0.0;
#pragma unroll
                        for (int d = 0; d < KX; d++)
                            if (q - d >= 0 && q - d < P) v = dx == d ? vx[q - d] : v;
                        rw[P + q] = v;
                    }
#pragma unroll
                    for (int q = 0; q < FY; q++) {
                        double v = 0.0;
#pragma unroll
                        for (int d = 0; d < KY; d++)
                            if (q - d >= 0 && q - d < P) v = dy == d ? vy[q - d] : v;
                        rw[P + FX + q] = v;
                    }
#pragma unroll
                    for (int q = P + FX + FY; q < REC; q++) rw[q] = 0.0;
                }
                for (int half = 0; half < 2 && 32 * half < cnt; half++) {
                    __builtin_amdgcn_wave_barrier();
                    if ((lane >> 5) == half) {
                        double2 *rec_mine = (double2 *)(stage + (lane & 31) * REC);
#pragma unroll
                        for (int q = 0; q < REC / 2; q++) rec_mine[q] = make_double2(rw[2 * q], rw[2 * q + 1]);
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int c = 0; c < AT; c++) {
                        const int k0 = max(max(bnd[c], sb) - sb, 32 * half), k1 = min(min(bnd[c + 1], sb + cnt) - sb, 32 * half + 32);
                        for (int k = k0; k < k1; k++) {
                            const double *rec = stage + (k - 32 * half) * REC;
                            double wz[P];
#pragma unroll
                            for (int cc = 0; cc < P; cc++) wz[cc] = rec[cc];
                            const double wxy = rec[P + ai] * rec[P + FX + bi];
#pragma unroll
                            for (int cc = 0; cc < P; cc++) acc[c + cc] = fma(wxy, wz[cc], acc[c + cc]);
                        }
                    }
                }
                continue;
            }
            // Every lane leaves what the others need of its particle (offsets inside the cell, the P z weights times
            // the charge, the cell inside the column) in the wave's staging area; the walk below reads one record per
            // particle with all lanes on the same address (broadcast reads), which costs far fewer instructions than
            // fetching the values lane by lane with v_readlane.
            __builtin_amdgcn_wave_barrier();
            double *rec_mine = stage + lane * REC;
            rec_mine[0] = lfx;
            rec_mine[1] = lfy;
            {   // the spline is even: node P-1-cc at offset f weighs what node cc weighs at 1 - f (half the coefficient table)
                const double lgz = 1.0 - lfz;
#pragma unroll
                for (int cc = 0; cc < P / 2; cc++) {
                    rec_mine[2 + cc] = ptl.w * piece_eval<P>(sp.piece[cc], lfz);
                    rec_mine[2 + P - 1 - cc] = ptl.w * piece_eval<P>(sp.piece[cc], lgz);
                }
            }
            rec_mine[2 + P] = __hiloint2double(0, dx + KX * dy);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < AT; c++) {
                const int k0 = max(bnd[c], sb) - sb, k1 = min(bnd[c + 1], sb + cnt) - sb;
                for (int k = k0; k < k1; k++) {
                    const double *rec = stage + k * REC;
                    const double fx = rec[0], fy = rec[1];
                    const int dxy = __builtin_amdgcn_readfirstlane(__double2loint(rec[2 + P]));
                    double wz[P];
#pragma unroll
                    for (int cc = 0; cc < P; cc++) wz[cc] = rec[2 + cc];
                    double wx, wy;
                    if (KX == 1 || dxy % KX == 0) wx = piece_eval<P>(cx[0], fx);
                    else wx = piece_eval<P>(cx[KX - 1], fx);
                    if (KY == 1 || dxy / KX == 0) wy = piece_eval<P>(cy[0], fy);
                    else wy = piece_eval<P>(cy[KY - 1], fy);
                    const double wxy = wx * wy;
#pragma unroll
                    for (int cc = 0; cc < P; cc++) acc[c + cc] = fma(wxy, wz[cc], acc[c + cc]);
                }
            }
        }
        for (int w = 0; w < nw; w++) {
            if (w == wave && active && bnd[AT] > bnd[0]) {
                double *dst = lds + ((coly * KY - y0) + b) * R + (colx * KX - x0) + a;
#pragma unroll
                for (int z = 0; z < R; z++) dst[z * R * R] += acc[z];
            }
            __syncthreads();
        }
    }
    double *__restrict__ out = scratch + (size_t)tile * R3;
    for (int t = threadIdx.x; t < R3; t += blockDim.x) out[t] = lds[t];
}

// (tile, index inside the tile's node block) pairs that hold contributions to node x of one axis
template <int P>
__device__ __forceinline__ int axis_sources(int x, int M, int *ti, int *li) {
    int n = 0, prev = -(1 << 30);
#pragma unroll
    for (int cc = x - P / 2; cc <= x + P / 2 - 1; cc++) { // unwrapped cells whose particles reach node x
        const int k = cc < 0 ? -1 : (cc >= M ? 1 : 0);
        const int tw = (cc - k * M) / AT;
        const int key = tw * 4 + (k + 1);
        if (key != prev && n < 3) {
            ti[n] = tw;
            li[n] = (x - k * M) - (AT * tw - P / 2 + 1);
            n++;
            prev = key;
        }
    }
    return n;
}

template <int P>
__global__ __launch_bounds__(256) void k_anterp_combine(const double *__restrict__ scratch, const int *__restrict__ tile_used,
                                                        SplineParams sp, TileGeom tl, double *__restrict__ grid) {
    constexpr int R = AT + P - 1, R3 = R * R * R;
    const int Mx = sp.M[0], My = sp.M[1];
    const unsigned nplane = (unsigned)Mx * My;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; // 32-bit index arithmetic: a level-0 grid has < 2^32 nodes
    if (idx >= nplane * sp.nz) return;
    const unsigned row = idx / (unsigned)Mx;
    const int jx = (int)(idx - row * Mx), jy = (int)(row % (unsigned)My), jz = sp.z0 + (int)(row / (unsigned)My);
    int tix[3], lix[3], tiy[3], liy[3], tiz[3], liz[3];
    const int nx = axis_sources<P>(jx, Mx, tix, lix), ny = axis_sources<P>(jy, My, tiy, liy), nz = axis_sources<P>(jz, sp.M[2], tiz, liz);
    double sum = 0.0;
    if (nx <= 2 && ny <= 2 && nz <= 2) {
        // the usual case, at most two blocks per axis: the flags of the (up to) eight tiles are requested together and
        // then their values, two round trips to L2 instead of up to sixteen; same terms, same order
        size_t at[8];
        int used[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int kx = k & 1, ky = (k >> 1) & 1, kz = k >> 2;
            const bool have = kx < nx && ky < ny && kz < nz;
            const size_t t = ((size_t)tiz[kz < nz ? kz : 0] * tl.nt[1] + tiy[ky < ny ? ky : 0]) * tl.nt[0] + tix[kx < nx ? kx : 0];
            at[k] = t * R3 + (liz[kz < nz ? kz : 0] * R + liy[ky < ny ? ky : 0]) * R + lix[kx < nx ? kx : 0];
            used[k] = have ? tile_used[t] : 0;
        }
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = used[k] ? scratch[at[k]] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (used[k]) sum += v[k];
    } else {
        for (int kz = 0; kz < nz; kz++)
            for (int ky = 0; ky < ny; ky++)
                for (int kx = 0; kx < nx; kx++) {
                    const size_t t = ((size_t)tiz[kz] * tl.nt[1] + tiy[ky]) * tl.nt[0] + tix[kx];
                    if (tile_used[t]) sum += scratch[t * R3 + (liz[kz] * R + liy[ky]) * R + lix[kx]];
                }
    }
    grid[((size_t)jz * My + jy) * Mx + jx] = sum;
}

// Interpolation of the level-0 potentials back to the particles (msmPeriodic.c:1116-1161, msmSimple.c:1317-1380),
// tile by tile: the workgroup of a tile stages the (8 + p - 1)^3 block of potentials its particles touch in LDS
// (periodic wrap applied while staging), then every thread takes one particle of the tile and sums its p^3
// nodes out of LDS.  acc_i += f sum_m e_m grad Phi_m(r_i), f by the acceleration convention.
template <int P>
__global__ __launch_bounds__(512) void k_interp_tiles(TileSrc S, SplineParams sp, TileGeom tl, const double *__restrict__ e,
                                                      double *__restrict__ acc, int remap) {
    constexpr int R = AT + P - 1, R3 = R * R * R;
    // Pitches of the block in LDS.  A read instruction takes, for every lane, the same node offset from the lane's own
    // base (z cell, y cell, x cell of its particle); the particles of a wave sit in one or two columns of 2 x 2 cells,
    // over the 8 z cells of the tile.  With a row pitch of 14 doubles and a plane pitch of 4 mod 32 the bases
    // {0, 1} + 14 {0, 1} + 4 {0..7} (mod 32 doubles = the 64 banks) are all different: lanes of one column never
    // collide, where the dense pitches 13 and 169 left 40 % of the LDS cycles to bank conflicts.
    constexpr int P1 = 14, P2 = ((R * P1 - 4 + 31) / 32) * 32 + 4;
    static_assert(P1 >= R && P2 >= R * P1, "pitches");
    __shared__ double blk[R * P2];
    __shared__ int rowstart[AT * AT + 1], rowbase[AT * AT];
    // Whole grid: an XCD gets a contiguous run of tiles, whose blocks of potentials overlap.  A rank that owns a slab
    // has its particles in a contiguous run of tiles: those are dealt out over the XCDs one by one instead.
    const int tile = launched_tile(tl, remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x);
    const int tx = tile % tl.nt[0], ty = (tile / tl.nt[0]) % tl.nt[1], tz = tile / (tl.nt[0] * tl.nt[1]);
    const int Mx = sp.M[0], My = sp.M[1], Mz = sp.M[2];
    const int x0 = tx * AT, y0 = ty * AT, z0 = tz * AT;
    const int nxc = min(AT, Mx - x0), nyc = min(AT, My - y0), nzc = min(AT, Mz - z0);
    const int nrows = tile_runs(S, nxc, nyc, nzc);
    if (threadIdx.x < 64) { // runs of the tile: first slot and running count (one wave, shuffle scan)
        const int t = threadIdx.x;
        int cnt = 0;
        if (t < nrows) {
            int s0;
            tile_run(S, sp, x0, y0, z0, nxc, nyc, nzc, t, s0, cnt);
            rowbase[t] = s0;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (t >= off) incl += v;
        }
        if (t < nrows) rowstart[t + 1] = incl;
        if (t == 0) rowstart[0] = 0;
    }
    __syncthreads();
    const int total = rowstart[nrows];
    if (total == 0) return;
    // the tile's particles may be shared between gridDim.y workgroups (a rank that owns a thin slab has few tiles)
    const int share = (total + gridDim.y - 1) / gridDim.y, idx_end = min(total, ((int)blockIdx.y + 1) * share);
    if ((int)blockIdx.y * share >= total) return;
    // staged four loads at a time per thread: one trip to L2 per batch instead of one per node
    for (int t0 = threadIdx.x; t0 < R3; t0 += 4 * blockDim.x) {
        double v[4];
        int at[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = min(t0 + k * (int)blockDim.x, R3 - 1);
            const int lx = t % R, ly = (t / R) % R, lz = t / (R * R);
            const int gx = modp(x0 - (P / 2 - 1) + lx, Mx), gy = modp(y0 - (P / 2 - 1) + ly, My), gz = modp(z0 - (P / 2 - 1) + lz, Mz);
            at[k] = lz * P2 + ly * P1 + lx;
            v[k] = e[((size_t)gz * My + gy) * Mx + gx];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) blk[at[k]] = v[k]; // a clamped index stores the last node's value again
    }
    __syncthreads();
    for (int idx = blockIdx.y * share + threadIdx.x; idx < idx_end; idx += blockDim.x) {
        int lo = 0, hi = nrows; // row with rowstart[row] <= idx < rowstart[row+1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rowstart[mid] <= idx) lo = mid; else hi = mid;
        }
        const int s = rowbase[lo] + (idx - rowstart[lo]);
        const int orig = S.order[s];
        const double4 pt = S.ps[s]; // asked for together with the index (padding slots hold a far-away particle)
        if (orig >= sp.n_owned) continue;
        const double ux = pt.x * S.scale[0], uy = pt.y * S.scale[1], uz = pt.z * S.scale[2];
        const int cellx = tile_cell(ux, x0, nxc), ry = tile_cell(uy, y0, nyc) - y0, rz = tile_cell(uz, z0, nzc) - z0;
        const double f[3] = {ux - (double)cellx, uy - (double)(y0 + ry), uz - (double)(z0 + rz)};
        // x and y weights (the z weights are made plane by plane).  The spline is even, so the weight of node P-1-v at
        // offset f is the weight of node v at 1 - f (and its derivative the negative): half of the coefficient table
        // serves all nodes, which keeps the table in scalar registers instead of spilling it.
        double w[2][P], dw[2][P];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const double g = 1.0 - f[d];
#pragma unroll
            for (int v = 0; v < P / 2; v++) {
                w[d][v] = piece_eval<P>(sp.piece[v], f[d]);
                w[d][P - 1 - v] = piece_eval<P>(sp.piece[v], g);
                dw[d][v] = piece_eval<P - 1>(sp.dpiece[v], f[d]);
                dw[d][P - 1 - v] = -piece_eval<P - 1>(sp.dpiece[v], g);
            }
        }
        const double *__restrict__ b0 = blk + rz * P2 + ry * P1 + (cellx - x0);
        double gx = 0, gy = 0, gz = 0;
#pragma unroll 1
        for (int c = 0; c < P; c++) {
            const double wz = piece_eval<P>(sp.piece[c], f[2]), dwz = piece_eval<P - 1>(sp.dpiece[c], f[2]);
            double px = 0, py = 0, pv = 0; // plane sums: d/dx, d/dy, value
#pragma unroll
            for (int b = 0; b < P; b++) {
                const double *row = b0 + c * P2 + b * P1;
                double sv = 0, sd = 0; // sum_a w_a e, sum_a dw_a e along x
#pragma unroll
                for (int a = 0; a < P; a++) {
                    const double ev = row[a];
                    sv = fma(w[0][a], ev, sv);
                    sd = fma(dw[0][a], ev, sd);
                }
                px = fma(w[1][b], sd, px);
                py = fma(dw[1][b], sv, py);
                pv = fma(w[1][b], sv, pv);
            }
            gx = fma(wz, px, gx);
            gy = fma(wz, py, gy);
            gz = fma(dwz, pv, gz);
        }
        gx *= sp.inv_h[0]; gy *= sp.inv_h[1]; gz *= sp.inv_h[2];
        double fac = 1.0;
        if (sp.convention == MSM_AMD_ACC_LITERAL) fac = pt.w;
        else if (sp.convention == MSM_AMD_ACC_FORCE) fac = -pt.w;
        const size_t o = 3 * (size_t)orig;
        if (S.acc_slots) { // the short-range part waits in slot order: one scatter for the sum (the same two terms, the same sum)
            const double4 sh = S.acc_slots[s];
            acc[o] = sh.x + fac * gx;
            acc[o + 1] = sh.y + fac * gy;
            acc[o + 2] = sh.z + fac * gz;
        } else {
            acc[o] += fac * gx;
            acc[o + 1] += fac * gy;
            acc[o + 2] += fac * gz;
        }
    }
}

// The layers of tiles a slab rank has to visit: those with a cell within `margin` cells of its planes [z0, z0 + nz).
// (The other tiles hold halo copies that reach none of the rank's planes, or nothing: launched, their workgroups would only
// find that out after a chain of index loads, which at 256^3 nodes is most of the kernel's time.)  A range that wraps
// around a grid whose size is not a multiple of the tile edge keeps all layers.
static TileGeom launched_layers(const TileGeom &tl, const SplineParams &sp, int margin, int &count) {
    TileGeom t = tl;
    t.tz0 = 0;
    count = tl.ntiles;
    if (sp.nz >= sp.M[2]) return t;
    const int lo = sp.z0 - margin, hi = sp.z0 + sp.nz - 1 + margin;
    if ((lo < 0 || hi >= sp.M[2]) && sp.M[2] % AT != 0) return t;
    const int a = lo >= 0 ? lo / AT : -((-lo + AT - 1) / AT), b = hi / AT; // floor division
    if (b - a + 1 >= tl.nt[2]) return t;
    t.tz0 = ((a % tl.nt[2]) + tl.nt[2]) % tl.nt[2];
    count = (b - a + 1) * tl.nt[0] * tl.nt[1];
    return t;
}

void launch_interp_tiled(hipStream_t st, const TileSrc &S, const SplineParams &sp, const TileGeom &tl_all, const double *egrid,
                         double *acc, const SplineOpts &opt) {
    // owned particles only; one cell of margin for a particle whose cell and slab were rounded differently
    int ntl;
    const TileGeom tl = launched_layers(tl_all, sp, 1, ntl);
    const int ns = opt.interp_split ? opt.interp_split : 1;
    const bool slab = sp.nz < sp.M[2];
    // a slab rank has few busy tiles: more threads to each
    const int nthreads = slab ? std::max(512, opt.interp_threads) : opt.interp_threads;
    const int remap = !slab && opt.interp_remap;
    if (sp.p == 4) hipLaunchKernelGGL((k_interp_tiles<4>), dim3(ntl, ns), dim3(nthreads), 0, st, S, sp, tl, egrid, acc, remap);
    else hipLaunchKernelGGL((k_interp_tiles<6>), dim3(ntl, ns), dim3(nthreads), 0, st, S, sp, tl, egrid, acc, remap);
}

bool launch_anterp_tiled(hipStream_t st, const TileSrc &S, const SplineParams &sp, const TileGeom &tl_all, double *scratch,
                         int *tile_used, double *grid, const SplineOpts &opt) {
    // a particle reaches p/2 planes below and above its cell (and one cell of margin as in the interpolation); the
    // combine kernel reads the blocks of exactly these tiles
    int ntl;
    const TileGeom tl = launched_layers(tl_all, sp, sp.p / 2 + 1, ntl);
    const size_t n = (size_t)sp.M[0] * sp.M[1] * sp.nz;
    const int gb = (int)((n + 255) / 256);
    static bool attr4 = false, attr6 = false;
    if ((sp.p == 6 || opt.anterp_cols_order4) && S.cluster_order && S.cg.kx >= 1 && S.cg.kx <= 2 && S.cg.ky >= 1 && S.cg.ky <= 2 && opt.anterp_cols) {
        // order 6, columns of at most 2 x 2 cells: sums kept in registers (at order 4 a particle's 64 nodes are one LDS
        // update of a full wave, which is faster)
        const int R = AT + sp.p - 1;
        const size_t bytes = sizeof(double) * ((size_t)(R * R * R + 1) / 2 * 2 + (size_t)16 * 64 * 10); // the most any launch asks for
        static bool attr[2][2][2] = {};
        const void *kern = nullptr;
        void (*k)(TileSrc, SplineParams, TileGeom, double *, int *) = nullptr;
        const bool wts = opt.anterp_cols_weights && sp.p == 6;
#define PICK(P, KX, KY) if (sp.p == P && S.cg.kx == KX && S.cg.ky == KY) k = k_anterp_cols<P, KX, KY>;
        PICK(4, 1, 1) PICK(4, 1, 2) PICK(4, 2, 1) PICK(4, 2, 2) PICK(6, 1, 1) PICK(6, 1, 2) PICK(6, 2, 1) PICK(6, 2, 2)
#undef PICK
#define PICK(KX, KY) if (wts && S.cg.kx == KX && S.cg.ky == KY) k = k_anterp_cols<6, KX, KY, true>;
        PICK(1, 1) PICK(1, 2) PICK(2, 1) PICK(2, 2)
#undef PICK
        kern = (const void *)k;
        static bool attr_w[2][2] = {};
        bool &done = wts ? attr_w[S.cg.kx - 1][S.cg.ky - 1] : attr[sp.p == 6][S.cg.kx - 1][S.cg.ky - 1];
        if (k && (done || hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess)) {
            done = true;
            // 4 waves per tile fill the chip when every tile has particles; a rank that owns a thin slab has few busy
            // tiles and finishes sooner with 8 (more runs at a time, at the price of more flush phases)
            const int nw = opt.anterp_cols_waves ? opt.anterp_cols_waves : (2 * sp.nz <= sp.M[2] ? 8 : 4);
            const int rec = (2 + sp.p + 1 + 1) / 2 * 2;
            const size_t dyn = sizeof(double) * ((size_t)(R * R * R + 1) / 2 * 2 + (size_t)nw * 64 * rec);
            hipLaunchKernelGGL(k, dim3(ntl), dim3(64 * nw), dyn, st, S, sp, tl, scratch, tile_used);
            if (sp.p == 4) MSM_LAUNCH_TAIL((k_anterp_combine<4>), dim3(gb), dim3(256), 0, st, scratch, tile_used, sp, tl, grid);
            else MSM_LAUNCH_TAIL((k_anterp_combine<6>), dim3(gb), dim3(256), 0, st, scratch, tile_used, sp, tl, grid);
            return true;
        }
        (void)hipGetLastError();
    }
    if (sp.p == 4) {
        const size_t bytes = 4 * sizeof(double) * (AT + 3) * (AT + 3) * (AT + 3);
        auto kern = k_anterp_tiles<4>;
        if (!attr4 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr4 = true;
        hipLaunchKernelGGL(kern, dim3(ntl), dim3(256), bytes, st, S, sp, tl, scratch, tile_used);
        MSM_LAUNCH_TAIL((k_anterp_combine<4>), dim3(gb), dim3(256), 0, st, scratch, tile_used, sp, tl, grid);
    } else {
        const size_t bytes = 4 * sizeof(double) * (AT + 5) * (AT + 5) * (AT + 5);
        auto kern = k_anterp_tiles<6>;
        if (!attr6 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr6 = true;
        hipLaunchKernelGGL(kern, dim3(ntl), dim3(256), bytes, st, S, sp, tl, scratch, tile_used);
        MSM_LAUNCH_TAIL((k_anterp_combine<6>), dim3(gb), dim3(256), 0, st, scratch, tile_used, sp, tl, grid);
    }
    return true;
}

} // namespace msm
