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

template <int P>
__global__ __launch_bounds__(256) void k_anterp_tiles(const double4 *__restrict__ ps, const int *__restrict__ gstart,
                                                      SplineParams sp, TileGeom tl, double *__restrict__ scratch,
                                                      int *__restrict__ tile_used) {
    constexpr int R = AT + P - 1, R3 = R * R * R;
    extern __shared__ double lds[]; // 4 copies of the node block
    __shared__ int any_particle;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *__restrict__ mine = lds + wave * R3;
    const int tile = blockIdx.x;
    const int tx = tile % tl.nt[0], ty = (tile / tl.nt[0]) % tl.nt[1], tz = tile / (tl.nt[0] * tl.nt[1]);
    const int Mx = sp.M[0], My = sp.M[1], Mz = sp.M[2];
    const int x0 = tx * AT, y0 = ty * AT, z0 = tz * AT;
    const int nxc = min(AT, Mx - x0), nyc = min(AT, My - y0), nzc = min(AT, Mz - z0);
    // an empty tile (most tiles of a slab rank) is only flagged; the combine kernel then skips its block
    if (threadIdx.x == 0) any_particle = 0;
    __syncthreads();
    if ((int)threadIdx.x < nyc * nzc) {
        const int b0 = ((z0 + (int)threadIdx.x / nyc) * My + (y0 + (int)threadIdx.x % nyc)) * Mx + x0;
        if (gstart[b0 + nxc] > gstart[b0]) any_particle = 1;
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
    for (int row = wave; row < nyc * nzc; row += 4) {
        const int ry = row % nyc, rz = row / nyc;
        const int b0 = ((z0 + rz) * My + (y0 + ry)) * Mx + x0;
        const int s0 = gstart[b0], s1 = gstart[b0 + nxc];
        for (int sb = s0; sb < s1; sb += 64) {
            const int cnt = min(64, s1 - sb);
            const double4 ptl = lane < cnt ? ps[sb + lane] : make_double4(0, 0, 0, 0);
            for (int k = 0; k < cnt; k++) {
                const double px = readlane_d(ptl.x, k), py = readlane_d(ptl.y, k), pz = readlane_d(ptl.z, k), pq = readlane_d(ptl.w, k);
                const int cellx = min((int)px, Mx - 1);
                const double fx = px - (double)cellx, fy = py - (double)(y0 + ry), fz = pz - (double)(z0 + rz);
                const int base = (rz * R + ry) * R + (cellx - x0) + lane_off;
                const double wxy = pq * piece_eval<P>(cx, fx) * piece_eval<P>(cy, fy);
                if (P == 4) {
                    const double w = wxy * piece_eval<P>(cz, fz);
                    mine[base] += w;
                } else if (active) {
                    double wz[P];
#pragma unroll
                    for (int cc = 0; cc < P; cc++) wz[cc] = piece_eval<P>(sp.piece[cc], fz);
                    double v[P];
#pragma unroll
                    for (int cc = 0; cc < P; cc++) v[cc] = mine[base + cc * R * R];
#pragma unroll
                    for (int cc = 0; cc < P; cc++) mine[base + cc * R * R] = fma(wxy, wz[cc], v[cc]);
                }
            }
        }
    }
    __syncthreads();
    double *__restrict__ out = scratch + (size_t)tile * R3;
    for (int t = threadIdx.x; t < R3; t += 256) out[t] = ((lds[t] + lds[R3 + t]) + lds[2 * R3 + t]) + lds[3 * R3 + t];
}

// (tile, index inside the tile's node block) pairs that hold contributions to node x of one axis
template <int P>
__device__ __forceinline__ int axis_sources(int x, int M, int *ti, int *li) {
    int n = 0, prev = -1 << 30;
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
    const size_t nplane = (size_t)Mx * My;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nplane * sp.nz) return;
    const int jx = (int)(idx % Mx), jy = (int)((idx / Mx) % My), jz = sp.z0 + (int)(idx / nplane);
    int tix[3], lix[3], tiy[3], liy[3], tiz[3], liz[3];
    const int nx = axis_sources<P>(jx, Mx, tix, lix), ny = axis_sources<P>(jy, My, tiy, liy), nz = axis_sources<P>(jz, sp.M[2], tiz, liz);
    double sum = 0.0;
    for (int kz = 0; kz < nz; kz++)
        for (int ky = 0; ky < ny; ky++)
            for (int kx = 0; kx < nx; kx++) {
                const size_t t = ((size_t)tiz[kz] * tl.nt[1] + tiy[ky]) * tl.nt[0] + tix[kx];
                if (tile_used[t]) sum += scratch[t * R3 + (liz[kz] * R + liy[ky]) * R + lix[kx]];
            }
    grid[((size_t)jz * My + jy) * Mx + jx] = sum;
}

// Interpolation of the level-0 potentials back to the particles (msmPeriodic.c:1116-1161, msmSimple.c:1317-1380),
// tile by tile: the workgroup of a tile stages the (8 + p - 1)^3 block of potentials its particles touch in LDS
// (periodic wrap applied while staging), then every thread takes one particle of the tile and sums its p^3
// nodes out of LDS.  acc_i += f sum_m e_m grad Phi_m(r_i), f by the acceleration convention.
template <int P>
__global__ __launch_bounds__(256) void k_interp_tiles(const double4 *__restrict__ ps, const int *__restrict__ order,
                                                      const int *__restrict__ gstart, SplineParams sp, TileGeom tl,
                                                      const double *__restrict__ e, double *__restrict__ acc) {
    constexpr int R = AT + P - 1, R3 = R * R * R;
    __shared__ double blk[R3];
    __shared__ int rowstart[AT * AT + 1], rowbase[AT * AT];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tl.nt[0], ty = (tile / tl.nt[0]) % tl.nt[1], tz = tile / (tl.nt[0] * tl.nt[1]);
    const int Mx = sp.M[0], My = sp.M[1], Mz = sp.M[2];
    const int x0 = tx * AT, y0 = ty * AT, z0 = tz * AT;
    const int nxc = min(AT, Mx - x0), nyc = min(AT, My - y0), nzc = min(AT, Mz - z0);
    const int nrows = nyc * nzc;
    if (threadIdx.x < 64) { // rows of the tile: first particle and running count (one wave, shuffle scan)
        const int t = threadIdx.x;
        int cnt = 0;
        if (t < nrows) {
            const int b0 = ((z0 + t / nyc) * My + (y0 + t % nyc)) * Mx + x0;
            const int s0 = gstart[b0];
            rowbase[t] = s0;
            cnt = gstart[b0 + nxc] - s0;
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
    for (int t = threadIdx.x; t < R3; t += 256) {
        const int lx = t % R, ly = (t / R) % R, lz = t / (R * R);
        const int gx = modp(x0 - (P / 2 - 1) + lx, Mx), gy = modp(y0 - (P / 2 - 1) + ly, My), gz = modp(z0 - (P / 2 - 1) + lz, Mz);
        blk[t] = e[((size_t)gz * My + gy) * Mx + gx];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        int lo = 0, hi = nrows; // row with rowstart[row] <= idx < rowstart[row+1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rowstart[mid] <= idx) lo = mid; else hi = mid;
        }
        const int s = rowbase[lo] + (idx - rowstart[lo]);
        const int orig = order[s];
        if (orig >= sp.n_owned) continue;
        const double4 pt = ps[s];
        const int ry = lo % nyc, rz = lo / nyc;
        const int cellx = min((int)pt.x, Mx - 1);
        const double f[3] = {pt.x - (double)cellx, pt.y - (double)(y0 + ry), pt.z - (double)(z0 + rz)};
        double w[2][P], dw[2][P]; // x and y weights; the z weights are made plane by plane
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int v = 0; v < P; v++) {
                w[d][v] = piece_eval<P>(sp.piece[v], f[d]);
                dw[d][v] = piece_eval<P - 1>(sp.dpiece[v], f[d]);
            }
        const double *__restrict__ b0 = blk + (rz * R + ry) * R + (cellx - x0);
        double gx = 0, gy = 0, gz = 0;
#pragma unroll 1
        for (int c = 0; c < P; c++) {
            const double wz = piece_eval<P>(sp.piece[c], f[2]), dwz = piece_eval<P - 1>(sp.dpiece[c], f[2]);
            double px = 0, py = 0, pv = 0; // plane sums: d/dx, d/dy, value
#pragma unroll
            for (int b = 0; b < P; b++) {
                const double *row = b0 + (c * R + b) * R;
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
        acc[o] += fac * gx;
        acc[o + 1] += fac * gy;
        acc[o + 2] += fac * gz;
    }
}

void launch_interp_tiled(hipStream_t st, const double4 *ps_grid, const int *order, const int *gstart, const SplineParams &sp,
                         const TileGeom &tl, const double *egrid, double *acc) {
    if (sp.p == 4) hipLaunchKernelGGL((k_interp_tiles<4>), dim3(tl.ntiles), dim3(256), 0, st, ps_grid, order, gstart, sp, tl, egrid, acc);
    else hipLaunchKernelGGL((k_interp_tiles<6>), dim3(tl.ntiles), dim3(256), 0, st, ps_grid, order, gstart, sp, tl, egrid, acc);
}

bool launch_anterp_tiled(hipStream_t st, const double4 *ps_grid, const int *gstart, const SplineParams &sp, const TileGeom &tl,
                         double *scratch, int *tile_used, double *grid) {
    const size_t n = (size_t)sp.M[0] * sp.M[1] * sp.nz;
    const int gb = (int)((n + 255) / 256);
    static bool attr4 = false, attr6 = false;
    if (sp.p == 4) {
        const size_t bytes = 4 * sizeof(double) * (AT + 3) * (AT + 3) * (AT + 3);
        auto kern = k_anterp_tiles<4>;
        if (!attr4 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr4 = true;
        hipLaunchKernelGGL(kern, dim3(tl.ntiles), dim3(256), bytes, st, ps_grid, gstart, sp, tl, scratch, tile_used);
        hipLaunchKernelGGL((k_anterp_combine<4>), dim3(gb), dim3(256), 0, st, scratch, tile_used, sp, tl, grid);
    } else {
        const size_t bytes = 4 * sizeof(double) * (AT + 5) * (AT + 5) * (AT + 5);
        auto kern = k_anterp_tiles<6>;
        if (!attr6 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr6 = true;
        hipLaunchKernelGGL(kern, dim3(tl.ntiles), dim3(256), bytes, st, ps_grid, gstart, sp, tl, scratch, tile_used);
        hipLaunchKernelGGL((k_anterp_combine<6>), dim3(gb), dim3(256), 0, st, scratch, tile_used, sp, tl, grid);
    }
    return true;
}

} // namespace msm
