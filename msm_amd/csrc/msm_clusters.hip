// Short-range pair sum over cluster pairs (gfx950).
//
// The cell-list kernel makes every lane walk every particle of the 27 neighbouring cells, and only one
// candidate in six lies inside the cutoff.  Here the particles are sorted into (x, y) columns and by z inside
// a column, and cut into clusters of CL_G = 8 consecutive particles, which are roughly cubic.  A wave owns
// one cluster i and visits only the clusters j whose bounding box comes within the cutoff of i's box; its 64
// lanes are the 8 x 8 particle pairs (i, j) of one cluster pair, so every lane does the same branch-free
// arithmetic, the j particles come straight from L1/L2 (256 contiguous bytes per cluster pair) and each lane
// keeps a private partial sum that is reduced over j once at the end.  The summation order is fixed by the
// sort, so results are the same on every run.  Same mathematics as msmLibrary.c:125-182.
#include <type_traits>
#include "msm_kernels.hpp"
#include "msm_device.hpp"

#include "msm_amd.h"

#include <algorithm>

namespace msm {

#define CL_NOSHIFT 13          // code (px+1) + 3 (py+1) + 9 (pz+1) of the zero image shift
#define CL_DUMMY 0x7fffffff
#define CL_CODE_SHIFT 27
#define CL_JMASK ((1 << CL_CODE_SHIFT) - 1)

// ---------------------------------------------------------------------------
// sort into columns / fine z bins, cut into clusters, bounding boxes
// ---------------------------------------------------------------------------
__global__ void k_cl_count(const double *__restrict__ r, int N, ClusterGeom g, int *__restrict__ bin_of, int *__restrict__ count,
                           int *__restrict__ arrival) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double t = (r[3 * (size_t)i + d] - g.origin[d]) * g.invA[d];
        s[d] = t - floor(t);
    }
    // grid-aligned columns are made of whole level-0 grid cells, so that a tile of cells is a set of columns
    const int cx = g.kx ? min((int)(s[0] * (g.ncx * g.kx)), g.ncx * g.kx - 1) / g.kx : min((int)(s[0] * g.ncx), g.ncx - 1);
    const int cy = g.ky ? min((int)(s[1] * (g.ncy * g.ky)), g.ncy * g.ky - 1) / g.ky : min((int)(s[1] * g.ncy), g.ncy - 1);
    const int zf = min((int)(s[2] * g.nzf), g.nzf - 1);
    const int b = cl_col(g, cx, cy) * g.nzf + zf;
    bin_of[i] = b;
    arrival[i] = atomicAdd(&count[b], 1);
}

__global__ void k_cl_colcount(const int *__restrict__ start, ClusterGeom g, int *__restrict__ colncl) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= g.ncols) return;
    const int n = start[(col + 1) * g.nzf] - start[col * g.nzf];
    colncl[col] = (n + CL_G - 1) / CL_G;
}

// Clusters per column and their exclusive prefix sums (clstart[ncols] = number of clusters) by one workgroup: a
// few thousand columns, so one launch instead of a counting kernel and a three-kernel scan.
__global__ __launch_bounds__(1024) void k_cl_colscan(const int *__restrict__ start, ClusterGeom g, int *__restrict__ colncl,
                                                     int *__restrict__ clstart) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < g.ncols; base += 1024) {
        const int col = base + threadIdx.x;
        int v = 0;
        if (col < g.ncols) {
            v = (start[(col + 1) * g.nzf] - start[col * g.nzf] + CL_G - 1) / CL_G;
            colncl[col] = v;
        }
        int incl = v; // inclusive scan inside the wave, then over the 16 wave totals
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (col < g.ncols) clstart[col] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) clstart[g.ncols] = carry;
}

// Slot of a particle: its column's first slot + its rank inside the column (fine bins in z order, original
// index inside a fine bin, so the layout does not depend on the order the atomics of the scatter arrived in).
__global__ void k_cl_finalize(const double *__restrict__ r, const double *__restrict__ q, int N, ClusterGeom g,
                              const int *__restrict__ bin_of, const int *__restrict__ start, const int *__restrict__ clstart,
                              const int *__restrict__ slot_idx, int *__restrict__ order, double4 *__restrict__ ps) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int i = slot_idx[s];
    const int b = bin_of[i];
    const int st = start[b], en = start[b + 1];
    int rank = 0;
    for (int t = st; t < en; t++) rank += (slot_idx[t] < i);
    const int col = b / g.nzf;
    const int dst = CL_G * clstart[col] + (st - start[col * g.nzf]) + rank;
    double f[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double t = (r[3 * (size_t)i + d] - g.origin[d]) * g.invA[d];
        f[d] = (t - floor(t)) * g.A[d];
    }
    order[dst] = i;
    ps[dst] = make_double4(f[0], f[1], f[2], q[i]);
}

// One thread per cluster: pad the tail of a column's last cluster with inert slots and box the real members.
__global__ void k_cl_bbox(ClusterGeom g, ClusterBuffers b) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncl = b.clstart[g.ncols];
    // one cluster of inert slots behind the last possible cluster: what a list walk reads past the end of a list
    if (k < CL_G) b.ps[(size_t)CL_G * g.max_clusters + k] = make_double4(g.far, g.far, g.far, 0.0);
    if (k >= ncl) return;
    int lo = 0, hi = g.ncols; // column with clstart[col] <= k < clstart[col+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.clstart[mid] <= k) lo = mid; else hi = mid;
    }
    const int col = lo;
    const int ncolp = b.start[(col + 1) * g.nzf] - b.start[col * g.nzf];
    const int nin = min(CL_G, ncolp - CL_G * (k - b.clstart[col]));
    double4 *p = b.ps + (size_t)CL_G * k;
    double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
    bool all_owned = true;
    for (int m = 0; m < nin; m++) {
        const double4 v = p[m];
        l[0] = fmin(l[0], v.x); h[0] = fmax(h[0], v.x);
        l[1] = fmin(l[1], v.y); h[1] = fmax(h[1], v.y);
        l[2] = fmin(l[2], v.z); h[2] = fmax(h[2], v.z);
        if (g.n_owned != CL_ALL_OWNED) all_owned = all_owned && b.order[(size_t)CL_G * k + m] < g.n_owned;
    }
    for (int m = nin; m < CL_G; m++) {
        p[m] = make_double4(g.far, g.far, g.far, 0.0);
        b.order[(size_t)CL_G * k + m] = CL_DUMMY;
    }
    b.blo[k] = make_double4(l[0], l[1], l[2], 0.0);
    b.bhi[k] = make_double4(h[0], h[1], h[2], 0.0);
    b.fbox[2 * (size_t)k] = make_float4(__double2float_rd(l[0]), __double2float_rd(l[1]), __double2float_rd(l[2]), all_owned ? 1.f : 0.f);
    b.fbox[2 * (size_t)k + 1] = make_float4(__double2float_ru(h[0]), __double2float_ru(h[1]), __double2float_ru(h[2]), 0.f);
}

// ---------------------------------------------------------------------------
// The same sort in two levels, without one global counter per fine bin.  (Device-wide atomics are resolved beyond the
// per-XCD L2 and a million of them take 40 us; the finishing pass of that sort gathers every particle again.)
//   1. every workgroup counts the particles of its share per column in LDS;
//   2. per column, the counts are scanned over the workgroups; columns are scanned over the grid;
//   3. every workgroup drops its particles into the buckets of their columns (position inside a bucket: an LDS
//      counter, i.e. arbitrary);
//   4. one workgroup per column ranks the bucket by (fine z bin, original index) - the order the other sort
//      produces, so everything downstream sees the same bits - writes the slots, the padding, the prefix sums
//      over the fine bins and the bounding boxes of the column's clusters.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int cl_bin_of(const double *__restrict__ r, int i, const ClusterGeom &g) {
    double s[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double t = (r[3 * (size_t)i + d] - g.origin[d]) * g.invA[d];
        s[d] = t - floor(t);
    }
    const int cx = g.kx ? min((int)(s[0] * (g.ncx * g.kx)), g.ncx * g.kx - 1) / g.kx : min((int)(s[0] * g.ncx), g.ncx - 1);
    const int cy = g.ky ? min((int)(s[1] * (g.ncy * g.ky)), g.ncy * g.ky - 1) / g.ky : min((int)(s[1] * g.ncy), g.ncy - 1);
    const int zf = min((int)(s[2] * g.nzf), g.nzf - 1);
    return cl_col(g, cx, cy) * g.nzf + zf;
}

__global__ __launch_bounds__(256) void k_cs_hist(const double *__restrict__ r, int N, ClusterGeom g, int per_block,
                                                 int *__restrict__ bin_of, int *__restrict__ hist) {
    extern __shared__ int lh[];
    for (int c = threadIdx.x; c < g.ncols; c += 256) lh[c] = 0;
    __syncthreads();
    const int i0 = blockIdx.x * per_block, i1 = min(N, i0 + per_block);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const int b = cl_bin_of(r, i, g);
        bin_of[i] = b;
        atomicAdd(&lh[b / g.nzf], 1);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < g.ncols; c += 256) hist[(size_t)blockIdx.x * g.ncols + c] = lh[c];
}

// hist[blk][col] -> particles of the column in earlier workgroups; coltotal[col].  64 columns per workgroup (coalesced
// rows of hist), the workgroups of the histogram cut into 16 segments that are summed side by side.
__global__ __launch_bounds__(1024) void k_cs_blockscan(int *__restrict__ hist, int nblk, int ncols, int *__restrict__ coltotal) {
    __shared__ int seg[16][64];
    const int cx = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int per = (nblk + 15) / 16, k0 = min(nblk, s * per), k1 = min(nblk, k0 + per);
    // (eight loads in flight at a time: the loop is a chain of L2 round trips otherwise, and in the second pass the
    // compiler cannot move a load of hist above a store to it by itself)
    constexpr int B = 8;
    int sum = 0;
    if (c < ncols)
        for (int k = k0; k < k1; k += B) {
            int v[B];
#pragma unroll
            for (int j = 0; j < B; j++) v[j] = k + j < k1 ? hist[(size_t)(k + j) * ncols + c] : 0;
#pragma unroll
            for (int j = 0; j < B; j++) sum += v[j];
        }
    seg[s][cx] = sum;
    __syncthreads();
    if (c >= ncols) return;
    int run = 0;
    for (int t = 0; t < s; t++) run += seg[t][cx];
    for (int k = k0; k < k1; k += B) {
        int v[B];
#pragma unroll
        for (int j = 0; j < B; j++) v[j] = k + j < k1 ? hist[(size_t)(k + j) * ncols + c] : 0;
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (k + j < k1) hist[(size_t)(k + j) * ncols + c] = run;
            run += v[j];
        }
    }
    if (s == 15) coltotal[c] = run;
}

// exclusive scans over the columns of the particle counts (cpstart) and of the cluster counts (clstart), one workgroup
__global__ __launch_bounds__(1024) void k_cs_colscan(const int *__restrict__ coltotal, ClusterGeom g, int *__restrict__ colncl,
                                                     int *__restrict__ clstart, int *__restrict__ cpstart) {
    __shared__ int wsum[2][16];
    __shared__ int carry[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int base = 0; base < g.ncols; base += 1024) {
        const int col = base + threadIdx.x;
        int vp = 0, vc = 0;
        if (col < g.ncols) {
            vp = coltotal[col];
            vc = (vp + CL_G - 1) / CL_G + g.nface; // (room for the padding at the faces of a slab)
            colncl[col] = vc;
        }
        int ip = vp, ic = vc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int tp = __shfl_up(ip, off), tc = __shfl_up(ic, off);
            if (lane >= off) { ip += tp; ic += tc; }
        }
        if (lane == 63) { wsum[0][wave] = ip; wsum[1][wave] = ic; }
        __syncthreads();
        int bp = carry[0], bc = carry[1];
        for (int w = 0; w < wave; w++) { bp += wsum[0][w]; bc += wsum[1][w]; }
        if (col < g.ncols) { cpstart[col] = bp + ip - vp; clstart[col] = bc + ic - vc; }
        __syncthreads();
        if (threadIdx.x == 1023) { carry[0] = bp + ip; carry[1] = bc + ic; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { cpstart[g.ncols] = carry[0]; clstart[g.ncols] = carry[1]; }
}

__global__ __launch_bounds__(256) void k_cs_scatter(const double *__restrict__ r, const double *__restrict__ q, int N, ClusterGeom g,
                                                    int per_block, const int *__restrict__ bin_of, const int *__restrict__ hist,
                                                    const int *__restrict__ cpstart, double4 *__restrict__ bps,
                                                    int2 *__restrict__ bkey) {
    extern __shared__ int cursor[];
    for (int c = threadIdx.x; c < g.ncols; c += 256) cursor[c] = cpstart[c] + hist[(size_t)blockIdx.x * g.ncols + c];
    __syncthreads();
    const int i0 = blockIdx.x * per_block, i1 = min(N, i0 + per_block);
    // eight particles per thread at a time: their loads travel together (two waves per SIMD hide little by themselves)
    constexpr int B = 8;
    for (int ib = i0 + threadIdx.x; ib < i1; ib += B * 256) {
        int bin[B];
        double x[B][3], qq[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const int i = min(ib + k * 256, i1 - 1); // (a clamped index repeats the last particle's loads; not used)
            bin[k] = bin_of[i];
            qq[k] = q[i];
#pragma unroll
            for (int d = 0; d < 3; d++) x[k][d] = r[3 * (size_t)i + d];
        }
#pragma unroll
        for (int k = 0; k < B; k++) {
            const int i = ib + k * 256;
            if (i < i1) {
                const int col = bin[k] / g.nzf;
                const int pos = atomicAdd(&cursor[col], 1);
                double f[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    double t = (x[k][d] - g.origin[d]) * g.invA[d];
                    f[d] = (t - floor(t)) * g.A[d];
                }
                bps[pos] = make_double4(f[0], f[1], f[2], qq[k]);
                bkey[pos] = make_int2(i, bin[k] - col * g.nzf); // (original index, fine bin inside the column)
            }
        }
    }
}

#define CS_CHUNK 1024
#define CS_CAP 3072
__global__ __launch_bounds__(256) void k_cs_colsort(ClusterGeom g, ClusterBuffers b) {
    extern __shared__ int zh[];                       // particles per fine bin of the column; zs = zh + nzf: their exclusive scan
    __shared__ union {
        unsigned long long keys[CS_CHUNK];            // (fine bin, original index) of a chunk of a long bucket
        int member[CS_CAP];                           // original indices, grouped by fine bin (any order inside a bin)
    } u;
    __shared__ int wtot[4];
    __shared__ int carry;
    int *zs = zh + g.nzf;
    const int col = blockIdx.x;
    const int p0 = b.cpstart[col], n = b.cpstart[col + 1] - p0, c0 = b.clstart[col], ncl = b.colncl[col];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int z = threadIdx.x; z < g.nzf; z += 256) zh[z] = 0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) atomicAdd(&zh[b.bkey[p0 + e].y], 1);
    __syncthreads();
    // zs[z] = particles of the column in fine bins below z; start[col * nzf + z] = p0 + zs[z]
    for (int base = 0; base < g.nzf; base += 256) {
        const int z = base + threadIdx.x;
        const int v = z < g.nzf ? zh[z] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; w++) before += wtot[w];
        if (z < g.nzf) {
            zs[z] = before + incl - v;
            if (!g.nface) b.start[(size_t)col * g.nzf + z] = p0 + before + incl - v;
        }
        __syncthreads();
        if (threadIdx.x == 255) carry = before + incl;
        __syncthreads();
    }
    if (!g.nface && col == g.ncols - 1 && threadIdx.x == 0) b.start[(size_t)g.ncols * g.nzf] = p0 + n;
    // A slab rank: a new cluster starts where ownership changes along the column.  pad0 / pad1 inert slots complete the
    // cluster before the first / second face; a particle of rank r in fine bin z then sits in slot r + fpad(z) of the
    // column, and start[] holds slots instead of particle counts (which is all its readers ask of it: they take
    // differences inside a column and divide by the cluster size).
    int pad0 = 0, pad1 = 0, f0 = g.nzf, f1 = g.nzf, r0 = n, r1 = n; // r0, r1: particles below the faces
    if (g.nface) {
        f0 = g.face[0]; f1 = g.face[1];
        r0 = f0 < g.nzf ? zs[f0] : n;
        r1 = f1 < g.nzf ? zs[f1] : n;
        pad0 = f0 < g.nzf ? (CL_G - r0 % CL_G) % CL_G : 0;
        pad1 = f1 < g.nzf ? (CL_G - (r1 - r0) % CL_G) % CL_G : 0;
        for (int z = threadIdx.x; z < g.nzf; z += 256)
            b.start[(size_t)col * g.nzf + z] = CL_G * c0 + zs[z] + (z >= f0 ? pad0 : 0) + (z >= f1 ? pad1 : 0);
        if (col == g.ncols - 1 && threadIdx.x == 0) b.start[(size_t)g.ncols * g.nzf] = CL_G * (c0 + ncl);
    }
    auto fpad = [&](int z) { return (z >= f0 ? pad0 : 0) + (z >= f1 ? pad1 : 0); };
    // rank of a particle inside the column: particles in lower fine bins + particles of its bin with a lower original index
    if (n <= CS_CAP) {
        for (int z = threadIdx.x; z < g.nzf; z += 256) zh[z] = 0;
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += 256) {
            const int2 k = b.bkey[p0 + e];
            u.member[zs[k.y] + atomicAdd(&zh[k.y], 1)] = k.x;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += 256) {
            const int2 k = b.bkey[p0 + e];
            const int z = k.y, idx = k.x;
            const int m0 = zs[z], m1 = m0 + zh[z];
            int rank = m0;
            for (int m = m0; m < m1; m++) rank += u.member[m] < idx;
            const size_t dst = (size_t)CL_G * c0 + rank + fpad(z);
            b.ps[dst] = b.bps[p0 + e];
            b.order[dst] = idx;
        }
    } else {
        // a bucket too long for the LDS copy: rank among the keys of the whole bucket, a chunk at a time
        for (int e0 = 0; e0 < n; e0 += 256) {
            const int e = e0 + threadIdx.x;
            unsigned long long key = 0;
            if (e < n) { const int2 k = b.bkey[p0 + e]; key = ((unsigned long long)(unsigned)k.y << 32) | (unsigned)k.x; }
            int rank = 0;
            for (int j0 = 0; j0 < n; j0 += CS_CHUNK) {
                const int m = min(CS_CHUNK, n - j0);
                __syncthreads();
                for (int j = threadIdx.x; j < m; j += 256)
                    { const int2 k = b.bkey[p0 + j0 + j]; u.keys[j] = ((unsigned long long)(unsigned)k.y << 32) | (unsigned)k.x; }
                __syncthreads();
                if (e < n)
                    for (int j = 0; j < m; j++) rank += u.keys[j] < key;
            }
            if (e < n) {
                const size_t dst = (size_t)CL_G * c0 + rank + fpad((int)(key >> 32));
                b.ps[dst] = b.bps[p0 + e];
                b.order[dst] = (int)(unsigned)key;
            }
        }
    }
    // padding: behind the column's last particle up to the end of its clusters, and in front of the faces of a slab
    for (int m = n + pad0 + pad1 + threadIdx.x; m < CL_G * ncl; m += 256) {
        b.ps[(size_t)CL_G * c0 + m] = make_double4(g.far, g.far, g.far, 0.0);
        b.order[(size_t)CL_G * c0 + m] = CL_DUMMY;
    }
    if ((int)threadIdx.x < pad0 + pad1) {
        const int m = (int)threadIdx.x < pad0 ? r0 + (int)threadIdx.x : r1 + pad0 + ((int)threadIdx.x - pad0);
        b.ps[(size_t)CL_G * c0 + m] = make_double4(g.far, g.far, g.far, 0.0);
        b.order[(size_t)CL_G * c0 + m] = CL_DUMMY;
    }
    if (col == 0 && threadIdx.x < CL_G) // one cluster of inert slots behind the last possible cluster (see k_cl_bbox)
        b.ps[(size_t)CL_G * g.max_clusters + threadIdx.x] = make_double4(g.far, g.far, g.far, 0.0);
    __syncthreads();
    // bounding boxes of the column's clusters (the slots were written by this workgroup: visible after the barriers):
    // eight lanes to a cluster, one slot each, so that a cluster's eight loads travel together; minima and maxima are
    // exact, whatever the order they are taken in
    for (int kb = 0; kb < ncl; kb += 256 / CL_G) {
        const int k = kb + (int)threadIdx.x / CL_G, m = (int)threadIdx.x % CL_G;
        const int nin = g.nface ? CL_G : min(CL_G, n - CL_G * k); // (with padding inside the column: whoever is no padding)
        const size_t slot = (size_t)CL_G * (c0 + k) + m;
        bool in = k < ncl && m < nin, own = true;
        if (in && (g.nface || g.n_owned != CL_ALL_OWNED)) {
            const int o = b.order[slot];
            if (g.nface && o == CL_DUMMY) in = false;
            else if (g.n_owned != CL_ALL_OWNED) own = o < g.n_owned;
        }
        double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
        if (in) {
            const double4 v = b.ps[slot];
            l[0] = h[0] = v.x; l[1] = h[1] = v.y; l[2] = h[2] = v.z;
        }
#pragma unroll
        for (int off = 1; off < CL_G; off <<= 1)
#pragma unroll
            for (int d = 0; d < 3; d++) {
                l[d] = fmin(l[d], __shfl_xor(l[d], off));
                h[d] = fmax(h[d], __shfl_xor(h[d], off));
            }
        const unsigned long long owned = __ballot(own);
        const bool all_owned = ((owned >> (lane & ~(CL_G - 1))) & ((1ull << CL_G) - 1)) == (1ull << CL_G) - 1;
        if (k < ncl && m == 0) {
            const int kk = c0 + k;
            b.blo[kk] = make_double4(l[0], l[1], l[2], 0.0);
            b.bhi[kk] = make_double4(h[0], h[1], h[2], 0.0);
            b.fbox[2 * (size_t)kk] = make_float4(__double2float_rd(l[0]), __double2float_rd(l[1]), __double2float_rd(l[2]), all_owned ? 1.f : 0.f);
            b.fbox[2 * (size_t)kk + 1] = make_float4(__double2float_ru(h[0]), __double2float_ru(h[1]), __double2float_ru(h[2]), 0.f);
        }
    }
}

static void launch_cluster_sort2(hipStream_t st, const double *d_r, const double *d_q, int N, const ClusterGeom &g,
                                 const ClusterBuffers &b, int *bin_of) {
    const int nblk = g.sort_blocks, per_block = (N + nblk - 1) / nblk;
    const size_t lds_cols = sizeof(int) * (size_t)g.ncols;
    hipLaunchKernelGGL(k_cs_hist, dim3(nblk), dim3(256), lds_cols, st, d_r, N, g, per_block, bin_of, b.hist);
    hipLaunchKernelGGL(k_cs_blockscan, dim3((g.ncols + 63) / 64), dim3(1024), 0, st, b.hist, nblk, g.ncols, b.coltotal);
    hipLaunchKernelGGL(k_cs_colscan, dim3(1), dim3(1024), 0, st, b.coltotal, g, b.colncl, b.clstart, b.cpstart);
    hipLaunchKernelGGL(k_cs_scatter, dim3(nblk), dim3(256), lds_cols, st, d_r, d_q, N, g, per_block, bin_of, b.hist, b.cpstart,
                       b.bps, b.bkey);
    hipLaunchKernelGGL(k_cs_colsort, dim3(g.ncols), dim3(256), 2 * sizeof(int) * (size_t)g.nzf, st, g, b);
}

void launch_cluster_sort(hipStream_t st, const double *d_r, const double *d_q, int N, const ClusterGeom &g,
                         const ClusterBuffers &b, int *bin_of, int *arrival, int *slot_idx, int *scan_tmp) {
    if (g.sort_blocks > 0) {
        launch_cluster_sort2(st, d_r, d_q, N, g, b, bin_of);
        return;
    }
    hipMemsetAsync(b.count, 0, sizeof(int) * (size_t)g.nbins, st);
    const int tb = 256, gb = (N + tb - 1) / tb;
    hipLaunchKernelGGL(k_cl_count, dim3(gb), dim3(tb), 0, st, d_r, N, g, bin_of, b.count, arrival);
    launch_exclusive_scan(st, b.count, g.nbins, scan_tmp, b.start);
    if (g.ncols <= (1 << 18)) {
        hipLaunchKernelGGL(k_cl_colscan, dim3(1), dim3(1024), 0, st, b.start, g, b.colncl, b.clstart);
    } else {
        hipLaunchKernelGGL(k_cl_colcount, dim3((g.ncols + tb - 1) / tb), dim3(tb), 0, st, b.start, g, b.colncl);
        launch_exclusive_scan(st, b.colncl, g.ncols, scan_tmp, b.clstart);
    }
    launch_bin_scatter(st, bin_of, arrival, N, b.start, slot_idx);
    hipLaunchKernelGGL(k_cl_finalize, dim3(gb), dim3(tb), 0, st, d_r, d_q, N, g, bin_of, b.start, b.clstart, slot_idx, b.order,
                       b.ps);
    hipLaunchKernelGGL(k_cl_bbox, dim3((g.max_clusters + tb - 1) / tb), dim3(tb), 0, st, g, b);
}

// ---------------------------------------------------------------------------
// candidate enumeration: every (cluster j, image shift) whose box comes within a of cluster i's box
// ---------------------------------------------------------------------------
struct IBox {
    double lox, loy, loz, hix, hiy, hiz;
};

// a lane's walk over one column of the extended (unwrapped) column grid
struct ColCursor {
    int col;      // wrapped column index
    int sxy;      // (px+1) + 3 (py+1): periods the column was wrapped by
    float ilx, ihx, ily, ihy; // the i box along x and y, shifted back by the column's image shift
    int zf0, zf1; // extended fine-bin range to visit
    int k;        // z period (-1, 0, 1) of the current candidate
    // all three z periods at once (cl_ranges): first cluster and number of clusters of each, none = 0
    int b0, b1, b2, n0, n1, n2;
};

__device__ __forceinline__ bool cl_column(const ClusterGeom &g, double a2, const IBox &ib, int exi, int eyi, ColCursor &c) {
    const double eps = 1e-6; // slack, in bins, for the different roundings of the sort and of this walk
    const double gx = fmax(0.0, fmax(ib.lox - (exi + 1 + eps) * g.pitch[0], (exi - eps) * g.pitch[0] - ib.hix));
    const double gy = fmax(0.0, fmax(ib.loy - (eyi + 1 + eps) * g.pitch[1], (eyi - eps) * g.pitch[1] - ib.hiy));
    const double rem = a2 - gx * gx - gy * gy;
    if (rem <= 0.0) return false;
    // reach along z, rounded up (single precision root, widened by more than its error)
    const double az = (double)(__builtin_amdgcn_sqrtf((float)rem) * 1.00001f);
    // the cell is at least 3 cutoffs wide, so a column is wrapped by at most one period
    const int px = exi < 0 ? -1 : (exi >= g.ncx ? 1 : 0), py = eyi < 0 ? -1 : (eyi >= g.ncy ? 1 : 0);
    const int cx = exi - px * g.ncx, cy = eyi - py * g.ncy;
    if (cx < 0 || cx >= g.ncx || cy < 0 || cy >= g.ncy) return false;
    c.col = cl_col(g, cx, cy);
    c.sxy = (px + 1) + 3 * (py + 1);
    const float sx = (float)(px * g.A[0]), sy = (float)(py * g.A[1]);
    c.ilx = __double2float_rd(ib.lox) - sx; c.ihx = __double2float_ru(ib.hix) - sx;
    c.ily = __double2float_rd(ib.loy) - sy; c.ihy = __double2float_ru(ib.hiy) - sy;
    c.zf0 = (int)floor((ib.loz - az) * g.inv_dzf - eps);
    c.zf1 = (int)floor((ib.hiz + az) * g.inv_dzf + eps);
    c.k = 0;
    return true;
}

// The cluster ranges of the three z periods of a column, computed once: the walk over a column then needs no
// control flow of its own (candidate t of the lane is found from the three lengths by two comparisons).
__device__ __forceinline__ void cl_ranges(const ClusterGeom &g, const int *__restrict__ start, const int *__restrict__ clstart,
                                          ColCursor &c) {
    const int base = c.col * g.nzf;
    const int colp0 = start[base], c0 = clstart[c.col];
    int bb[3], nn[3];
#pragma unroll
    for (int k = -1; k <= 1; k++) {
        const int off = k * g.nzf;
        const int lo = max(c.zf0, off) - off, hi = min(c.zf1, off + g.nzf - 1) - off;
        bb[k + 1] = 0;
        nn[k + 1] = 0;
        if (lo <= hi) {
            const int p0 = start[base + lo], p1 = start[base + hi + 1];
            if (p1 > p0) {
                bb[k + 1] = c0 + (p0 - colp0) / CL_G;
                nn[k + 1] = (p1 - 1 - colp0) / CL_G - (p0 - colp0) / CL_G + 1;
            }
        }
    }
    c.b0 = bb[0]; c.b1 = bb[1]; c.b2 = bb[2];
    c.n0 = nn[0]; c.n1 = nn[1]; c.n2 = nn[2];
}

// Calls emit(keep, jcl, code, jfull) with all 64 lanes converged; each lane brings its own candidate (jfull: every
// particle of cluster j is owned).
template <class Emit>
__device__ __forceinline__ void cl_enumerate(const ClusterGeom &g, const ClusterBuffers &b, double a, double a2, const IBox &ib,
                                             int lane, Emit &&emit) {
    const double eps = 1e-6;
    const int ex0 = (int)floor((ib.lox - a) * g.inv_pitch[0] - eps), ex1 = (int)floor((ib.hix + a) * g.inv_pitch[0] + eps);
    const int ey0 = (int)floor((ib.loy - a) * g.inv_pitch[1] - eps), ey1 = (int)floor((ib.hiy + a) * g.inv_pitch[1] + eps);
    const float ilz = __double2float_rd(ib.loz), ihz = __double2float_ru(ib.hiz);
    // Lanes cover the columns of the extended column grid, row by row, `per` columns at a time with 64 / per lanes to a
    // column, which take its candidates in turn: a footprint of 36 columns is walked as 32 columns with two lanes each
    // and 4 columns with sixteen lanes each, in little more than half the steps of one lane per column.
    const int nx = ex1 - ex0 + 1, ncol = nx * (ey1 - ey0 + 1);
    for (int c0 = 0, per = 64; c0 < ncol; c0 += per) {
        const int rem = ncol - c0;
        per = rem > 48 ? 64 : (rem > 16 ? 32 : (rem > 8 ? 16 : (rem > 4 ? 8 : 4)));
        const int nsub = 64 / per, sub = lane / per, lcol = lane - sub * per;
        ColCursor c;
        const int ci = c0 + lcol, row = ci / nx;
        const int exi = ex0 + (ci - row * nx), eyi = ey0 + row;
        int total = 0;
        if (ci < ncol && cl_column(g, a2, ib, exi, eyi, c)) {
            cl_ranges(g, b.start, b.clstart, c);
            total = c.n0 + c.n1 + c.n2;
        }
        for (int t = sub; __ballot(t < total); t += nsub) {
            const bool live = t < total;
            int jcl = 0, code = CL_NOSHIFT;
            bool keep = false, jfull = false;
            if (live) {
                const int seg = (t >= c.n0) + (t >= c.n0 + c.n1);
                jcl = seg == 0 ? c.b0 + t : (seg == 1 ? c.b1 + (t - c.n0) : c.b2 + (t - c.n0 - c.n1));
                c.k = seg - 1;
                code = c.sxy + 9 * seg;
                // single-precision boxes against the single-precision i box shifted back by the image shift;
                // the widened cutoff keeps the test conservative
                const float4 jl = b.fbox[2 * (size_t)jcl], jh = b.fbox[2 * (size_t)jcl + 1];
                const float szf = (float)(c.k * g.A[2]);
                const float gx = fmaxf(0.f, fmaxf(jl.x - c.ihx, c.ilx - jh.x));
                const float gy = fmaxf(0.f, fmaxf(jl.y - c.ihy, c.ily - jh.y));
                const float gz = fmaxf(0.f, fmaxf(jl.z - (ihz - szf), (ilz - szf) - jh.z));
                keep = gx * gx + gy * gy + gz * gz < g.a2_wide;
                jfull = jl.w != 0.f;
            }
            emit(keep, jcl, code, jfull);
        }
    }
}

__device__ __forceinline__ int lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// One wave per cluster: the list of its neighbour clusters.  Entry 0 is the cluster itself; unshifted
// neighbours follow from the front, neighbours seen through a periodic image are stored from the back with
// their shift code in the top bits.
// SLAB: the handle holds halo copies (only then can a neighbour be anything but fully owned; without them the fourth
// section stays empty and its bookkeeping is left out of the kernel)
template <bool SLAB>
__global__ __launch_bounds__(256) void k_cl_list(ClusterGeom g, ClusterBuffers b, double a, double a2, int n_owned,
                                                 const int *__restrict__ only_if) {
    if (only_if && *only_if == 0) return; // (msm_amd_eval_reuse: lists are rebuilt only when the device-side check says so)
    const int lane = threadIdx.x & 63;
    const int icl = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int ncl = b.clstart[g.ncols];
    if (icl >= ncl) return;
    // a cluster of halo copies only produces no output: no list needed (the pair kernels skip it the same way)
    if (!__ballot(lane < CL_G && b.order[(size_t)CL_G * icl + lane] < n_owned)) {
        if (lane == 0) { b.n0[icl] = 0; b.n1[icl] = 0; b.n2[icl] = 0; if (SLAB) b.n3[icl] = 0; }
        return;
    }
    const double4 il = b.blo[icl], ih = b.bhi[icl];
    const IBox ib = {il.x, il.y, il.z, ih.x, ih.y, ih.z};
    const int S = g.list_stride, MID = S / 2;
    int *__restrict__ L = b.list + (size_t)icl * S;
    // "up" (j > i, from the front, after the cluster itself) and "down" (j < i, from the middle): unshifted neighbours
    // whose particles are all owned; "other": unshifted neighbours with halo copies among their particles (down from
    // just below the middle; none without a slab decomposition); shifted (from the back)
    int nu = 1, nd = 0, ns = 0, no = 0;
    bool overflow = false;
    if (lane == 0) L[0] = icl;
    cl_enumerate(g, b, a, a2, ib, lane, [&](bool keep, int jcl, int code, bool jfull) {
        if (jcl == icl && code == CL_NOSHIFT) keep = false;
        const bool kn = keep && code == CL_NOSHIFT;
        if (!SLAB) jfull = true;
        const bool ku = kn && jfull && jcl > icl, kd = kn && jfull && jcl < icl, ko = SLAB && kn && !jfull;
        const bool ks = keep && code != CL_NOSHIFT;
        const unsigned long long mu = __ballot(ku), md = __ballot(kd), ms = __ballot(ks), mo = SLAB ? __ballot(ko) : 0ull;
        const int cu = __popcll(mu), cd = __popcll(md), cs = __popcll(ms), co = SLAB ? __popcll(mo) : 0;
        if (nu + cu + no + co > MID || MID + nd + cd + ns + cs > S) overflow = true;
        else {
            if (ku) L[nu + lane_rank(mu)] = jcl;
            if (kd) L[MID + nd + lane_rank(md)] = jcl;
            if (SLAB && ko) L[MID - 1 - (no + lane_rank(mo))] = jcl;
            if (ks) L[S - 1 - (ns + lane_rank(ms))] = (int)((unsigned)jcl | ((unsigned)code << CL_CODE_SHIFT));
        }
        nu += cu;
        nd += cd;
        ns += cs;
        no += co;
    });
    if (lane == 0) {
        b.n0[icl] = overflow ? -1 : nu;
        b.n1[icl] = ns;
        b.n2[icl] = nd;
        if (SLAB) b.n3[icl] = no;
    }
}

// ---------------------------------------------------------------------------
// pair arithmetic
// ---------------------------------------------------------------------------
// 1/sqrt(x) to double precision: hardware seed (v_rsq_f64: largest relative error 2^-24.2 over 2^26 arguments,
// tools/micro/rsq_accuracy.hip) and one third-order correction, error ~ (5/16) e^3 (measured 2.7e-16; a second-order
// correction is one instruction shorter and leaves 4.3e-15, one-signed).
__device__ __forceinline__ double rsqrt_seeded(double x) {
#ifdef MSM_RSQ_F32
    const double y0 = (double)__builtin_amdgcn_rsqf((float)x);
#else
    const double y0 = __builtin_amdgcn_rsq(x);
#endif
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// One (i, j) pair, branch-free: a pair outside the cutoff runs the same arithmetic with its charge set to 0.
// g0(r) = 1/r - gamma(r/a)/a,  g0'(r)/r = -1/r^3 + D(y)/a^3  (msmLibrary.h:18-19 for p = 4).
template <int P, bool SELF, bool SKIP = false>
__device__ __forceinline__ void cl_slot(double xi, double yi, double zi, const double4 &pj, const PairParams &pp, bool other,
                                        double &u, double &fx, double &fy, double &fz) {
    const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
    double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
    bool in = r2 < pp.a2;
    if (SELF) {
        in = in && other;
        r2 = in ? r2 : pp.a2;
    }
    if (SKIP && !__ballot(in)) return; // no pair of this cluster pair is inside the cutoff
    // charge of a pair outside the cutoff: high word cleared, i.e. 0 or a denormal below 1e-308 whose products
    // vanish against any in-range term (one select instead of two)
    const double qm = __hiloint2double(in ? __double2hiint(pj.w) : 0, __double2loint(pj.w));
    const double ir = rsqrt_seeded(r2);
    // gamma(r/a)/a and -gamma'(r/a)/(r a^2) as polynomials in r^2 itself
    double gam = pp.cgr[P - 1], dg = pp.cdr[P - 2];
#pragma unroll
    for (int k = P - 2; k >= 0; k--) gam = fma(gam, r2, pp.cgr[k]);
#pragma unroll
    for (int k = P - 3; k >= 0; k--) dg = fma(dg, r2, pp.cdr[k]);
    const double g0 = ir - gam;
    const double s = fma(-(ir * ir), ir, dg);
    u = fma(qm, g0, u);
    const double qs = qm * s;
    fx = fma(qs, dx, fx);
    fy = fma(qs, dy, fy);
    fz = fma(qs, dz, fz);
}

// the same pair without the energy: a pair (i, j) listed from both sides needs its energy only once
template <int P>
__device__ __forceinline__ void cl_slot_force(double xi, double yi, double zi, const double4 &pj, const PairParams &pp, double &fx,
                                              double &fy, double &fz) {
    const double dx = xi - pj.x, dy = yi - pj.y, dz = zi - pj.z;
    const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
    const bool in = r2 < pp.a2;
    const double qm = __hiloint2double(in ? __double2hiint(pj.w) : 0, __double2loint(pj.w));
    const double ir = rsqrt_seeded(r2);
    double dg = pp.cdr[P - 2];
#pragma unroll
    for (int k = P - 3; k >= 0; k--) dg = fma(dg, r2, pp.cdr[k]);
    const double s = fma(-(ir * ir), ir, dg);
    const double qs = qm * s;
    fx = fma(qs, dx, fx);
    fy = fma(qs, dy, fy);
    fz = fma(qs, dz, fz);
}

struct LaneI {
    double x, y, z, q;
    int orig;
    bool owned;
};

__device__ __forceinline__ LaneI load_lane_i(const ClusterGeom &g, const ClusterBuffers &b, const PairParams &pp, int icl, int ii) {
    LaneI L;
    const double4 pi = b.ps[(size_t)CL_G * icl + ii];
    L.orig = b.order[(size_t)CL_G * icl + ii];
    L.owned = L.orig < pp.n_owned;
    // a padding slot as i: move it away from the padding slots it would meet as j (distance 0)
    const bool dummy = L.orig == CL_DUMMY;
    L.x = dummy ? -2.0 * g.far : pi.x;
    L.y = pi.y;
    L.z = pi.z;
    L.q = pi.w;
    return L;
}

// sum the 8 lanes that share a particle i (consecutive lanes), then write its acceleration and energy share
__device__ __forceinline__ void cl_finish(const LaneI &I, const PairParams &pp, int jj, int icl, double u, double fx, double fy,
                                          double fz, double *__restrict__ acc, double *__restrict__ epart, int lane,
                                          double4 *__restrict__ acc_slots = nullptr) {
#pragma unroll
    for (int off = 1; off < CL_G; off <<= 1) {
        u += __shfl_xor(u, off);
        fx += __shfl_xor(fx, off);
        fy += __shfl_xor(fy, off);
        fz += __shfl_xor(fz, off);
    }
    double e = 0.0;
    if (jj == 0 && I.owned) {
        const double w = pp.convention == MSM_AMD_ACC_FORCE ? -I.q : 1.0;
        if (acc_slots) { // by slot: lane (ii, 0) holds particle ii of the cluster
            acc_slots[(size_t)CL_G * icl + (lane >> 3)] = make_double4(w * fx, w * fy, w * fz, 0.0);
        } else {
            acc[3 * (size_t)I.orig] = w * fx;
            acc[3 * (size_t)I.orig + 1] = w * fy;
            acc[3 * (size_t)I.orig + 2] = w * fz;
        }
        e = 0.5 * I.q * u;
    }
    e = wave_sum(e);
    if (lane == 0) epart[icl] = e;
}

__device__ __forceinline__ void shift_of_code(const ClusterGeom &g, int code, double &sx, double &sy, double &sz) {
    sx = (code % 3 - 1) * g.A[0];
    sy = ((code / 3) % 3 - 1) * g.A[1];
    sz = (code / 9 - 1) * g.A[2];
}

// One wave per cluster i, lane = (ii, jj) = (lane / 8, lane % 8); walks the list built by k_cl_list.
// HALF (every particle is owned, i.e. no slab decomposition): the energy of an unshifted pair is taken, whole, where
// j's cluster comes after i's ("up" entries) and skipped where it comes before ("down" entries, forces only: a
// quarter fewer instructions); the cluster's own pairs and pairs through a periodic image count half from either
// side.  Otherwise every listed pair counts half (a slab rank sees the other half on the neighbour's side).
template <int P, bool HALF, int U>
__global__ __launch_bounds__(256) void k_pair_clusters(ClusterGeom g, ClusterBuffers b, PairParams pp, double *__restrict__ acc,
                                                       double *__restrict__ epart) {
    const int lane = threadIdx.x & 63, ii = lane >> 3, jj = lane & 7;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int icl = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));
    if (icl >= g.max_clusters) return;
    const int ncl = b.clstart[g.ncols];
    if (icl >= ncl) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    const int nu = b.n0[icl], ns = b.n1[icl], nd = b.n2[icl];
    if (nu < 0) return; // list overflow: left to the list-free kernel
    const LaneI I = load_lane_i(g, b, pp, icl, ii);
    if (!__ballot(I.owned)) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    const int S = g.list_stride, MID = S / 2;
    const int *__restrict__ L = b.list + (size_t)icl * S;
    const double4 *__restrict__ pj = b.ps + jj;
    double u1 = 0, uh = 0, fx = 0, fy = 0, fz = 0; // u1: pairs counted whole, uh: pairs counted half
    cl_slot<P, true>(I.x, I.y, I.z, pj[(size_t)CL_G * icl], pp, ii != jj, uh, fx, fy, fz);
    int e = 1;
    for (; e + U <= nu; e += U) {
        double4 qv[U];
#pragma unroll
        for (int k = 0; k < U; k++) qv[k] = pj[(size_t)CL_G * L[e + k]];
#pragma unroll
        for (int k = 0; k < U; k++) cl_slot<P, false>(I.x, I.y, I.z, qv[k], pp, true, HALF ? u1 : uh, fx, fy, fz);
    }
    for (; e < nu; e++) cl_slot<P, false>(I.x, I.y, I.z, pj[(size_t)CL_G * L[e]], pp, true, HALF ? u1 : uh, fx, fy, fz);
    const int *__restrict__ Ld = L + MID;
    e = 0;
    for (; e + U <= nd; e += U) {
        double4 qv[U];
#pragma unroll
        for (int k = 0; k < U; k++) qv[k] = pj[(size_t)CL_G * Ld[e + k]];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (HALF) cl_slot_force<P>(I.x, I.y, I.z, qv[k], pp, fx, fy, fz);
            else cl_slot<P, false>(I.x, I.y, I.z, qv[k], pp, true, uh, fx, fy, fz);
        }
    }
    for (; e < nd; e++) {
        const double4 q = pj[(size_t)CL_G * Ld[e]];
        if (HALF) cl_slot_force<P>(I.x, I.y, I.z, q, pp, fx, fy, fz);
        else cl_slot<P, false>(I.x, I.y, I.z, q, pp, true, uh, fx, fy, fz);
    }
    if (!HALF) { // neighbours with halo copies among their particles (a slab rank only)
        const int no = min(max(b.n3[icl], 0), MID - 1); // (never beyond the half of the list it shares)
        for (e = 0; e < no; e++) cl_slot<P, false>(I.x, I.y, I.z, pj[(size_t)CL_G * L[MID - 1 - e]], pp, true, uh, fx, fy, fz);
    }
    for (e = 0; e < ns; e++) {
        const int ent = L[S - 1 - e];
        double sx, sy, sz;
        shift_of_code(g, (int)((unsigned)ent >> CL_CODE_SHIFT), sx, sy, sz);
        cl_slot<P, false>(I.x - sx, I.y - sy, I.z - sz, pj[(size_t)CL_G * (ent & CL_JMASK)], pp, true, uh, fx, fy, fz);
    }
    cl_finish(I, pp, jj, icl, 2.0 * u1 + uh, fx, fy, fz, acc, epart, lane, b.acc_slots);
}

// The list walked at particle granularity.  A listed cluster j has, on average, only a third of its particles
// within the cutoff of some particle of i, so the wave first tests the 64 particles of 8 listed clusters against
// the bounding box of i (one particle per lane), compacts the survivors into a queue in LDS, and runs the pair
// arithmetic on groups of 8 survivors (lane = (ii, jj), jj picks the survivor): about 40 % fewer pair slots for
// one cheap test per listed particle.  The box test runs in single precision against a cutoff and a box widened by
// more than the rounding of the converted coordinates, so it can never drop a pair the arithmetic would have kept.  The queue is private
// to the wave; survivors keep list order, so the sum order is fixed.  Energy bookkeeping as in k_pair_clusters.
#define CLQ_CAP 128 // queue entries per wave: fewer than 64 pending + 64 new
#define CLQ_U 4 // groups per unrolled step of the pair arithmetic (2..4 and 4..6 waves per SIMD all time the same)
template <int P, bool HALF, int WPE = 1, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, WPE) void k_pair_clusters_filtered(ClusterGeom g, ClusterBuffers b, PairParams pp,
                                                                double *__restrict__ acc, double *__restrict__ epart,
                                                                int chunk, int nchunks) {
    __shared__ double4 queue[WPB][CLQ_CAP];
    const int lane = threadIdx.x & 63, ii = lane >> 3, jj = lane & 7;
    // A workgroup takes a run of consecutive clusters (WPB per step, one per wave).  Launched with one workgroup per
    // step this is the plain one-wave-per-cluster kernel; launched with a few workgroups per CU (WPB = 4 only) it leaves
    // room on every CU for the kernels of another stream, at the same speed.  Runs are handed out so that an XCD gets
    // neighbouring clusters.  Workgroups of one wave are the default: the waves of a four-wave workgroup are placed
    // together, one per SIMD whatever the SIMDs hold, and the per-wave clocks (msm_amd_set_pair_trace) showed SIMDs
    // ending up with 10 to 24 evaluated waves on a slab rank and at most three resident where four fit (the 4-wave
    // instantiation of order 6 also needs 129 registers, one too many for four); single waves go where there is room:
    // 0.773 -> 0.705 ms for 1M particles, 148 -> 129 us on a rank of 8.
    // The launch covers the cluster slots the buffers can hold; the clusters there are come first, and a slab rank, which
    // is created with headroom, has 6 % fewer: the runs are dealt out over the live steps only (an XCD would else get an
    // eighth of the slots, and the last one the empty ones), the steps behind them keep their own number.
    // The kernel may be launched in several chunks, one after the other (one-wave workgroups only): chunk c takes the
    // c-th share of the clusters there are, the last one also the empty slots behind them.  Between two chunks the
    // kernels of another stream find free wave slots, which they do not while one launch has workgroups left to start.
    const int nsteps = (g.max_clusters + WPB - 1) / WPB;
    const int ncl = b.clstart[g.ncols];
    const int nlive_all = (ncl + WPB - 1) / WPB;
    const int lo = nchunks > 1 ? (int)((long long)nlive_all * chunk / nchunks) : 0;
    const int hi = nchunks > 1 && chunk + 1 < nchunks ? (int)((long long)nlive_all * (chunk + 1) / nchunks) : nsteps;
    const int per = nchunks > 1 ? 1 : (nsteps + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nlive = min(hi, nlive_all) - lo;
    int step0 = xcd_remap(blockIdx.x, gridDim.x) * per;
    if (per == 1) step0 = lo + ((int)blockIdx.x < nlive ? xcd_remap(blockIdx.x, nlive) : (int)blockIdx.x);
    const int step1 = min(hi, step0 + per);
    for (int step = step0; step < step1; step++) [&] {
    const int icl = __builtin_amdgcn_readfirstlane(step * WPB + (threadIdx.x >> 6));
    if (icl >= g.max_clusters) return;
    const unsigned long long t_start = b.trace ? wall_clock64() : 0ull;
    auto trace = [&](unsigned long long kind) {
        if (b.trace && lane == 0) {
            unsigned long long *t = b.trace + 4 * (size_t)icl;
            t[0] = t_start;
            t[1] = wall_clock64();
            t[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            t[3] = kind;
        }
    };
    if (icl >= ncl) {
        if (lane == 0) epart[icl] = 0.0;
        trace(1);
        return;
    }
    const int nu = b.n0[icl], ns = b.n1[icl], nd = b.n2[icl];
    if (nu < 0) return; // list overflow: left to the list-free kernel
    const LaneI I = load_lane_i(g, b, pp, icl, ii);
    if (!__ballot(I.owned)) {
        if (lane == 0) epart[icl] = 0.0;
        trace(2);
        return;
    }
    const int S = g.list_stride, MID = S / 2;
    const int *__restrict__ L = b.list + (size_t)icl * S;
    double4 *Q = queue[threadIdx.x >> 6];
    const double4 il = b.blo[icl], ih = b.bhi[icl];
    const double4 nowhere = make_double4(g.far, g.far, g.far, 0.0);
    int tail = 0; // entries waiting in the queue (wave-uniform)
    double u1 = 0, uh = 0, fx = 0, fy = 0, fz = 0;
    cl_slot<P, true>(I.x, I.y, I.z, b.ps[(size_t)CL_G * icl + jj], pp, ii != jj, uh, fx, fy, fz);

    // the 64 particles of 8 list entries, one per lane: survivors of the box test are appended to the queue
    // (box as centre and half extent)
    const double bcx = 0.5 * (il.x + ih.x), bcy = 0.5 * (il.y + ih.y), bcz = 0.5 * (il.z + ih.z);
    const double bhx = 0.5 * (ih.x - il.x) + 1e-13 * g.A[0], bhy = 0.5 * (ih.y - il.y) + 1e-13 * g.A[1],
                 bhz = 0.5 * (ih.z - il.z) + 1e-13 * g.A[2];
    const float fcx = (float)bcx, fcy = (float)bcy, fcz = (float)bcz;
    const float fhx = (float)bhx * 1.000001f + 1e-6f * (float)g.A[0], fhy = (float)bhy * 1.000001f + 1e-6f * (float)g.A[1],
                fhz = (float)bhz * 1.000001f + 1e-6f * (float)g.A[2];
    auto filter = [&](unsigned ent, double4 p, bool shifted) {
        if (shifted) {
            double sx, sy, sz;
            shift_of_code(g, (int)(ent >> CL_CODE_SHIFT), sx, sy, sz);
            p.x += sx; p.y += sy; p.z += sz;
        }
        // in single precision (half the issue cost of the double-precision form: 3 % of the kernel), against a cutoff and
        // half extents widened by more than the rounding of the converted coordinates (at most 2e-7 of the cell edge)
        const float dxf = fmaxf(fabsf((float)p.x - fcx) - fhx, 0.f);
        const float dyf = fmaxf(fabsf((float)p.y - fcy) - fhy, 0.f);
        const float dzf = fmaxf(fabsf((float)p.z - fcz) - fhz, 0.f);
        const bool keep = fmaf(dxf, dxf, fmaf(dyf, dyf, dzf * dzf)) < g.a2_wide;
        const unsigned long long m = __ballot(keep);
        const int pos = tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep) Q[pos] = p;
        tail += __popcll(m);
        __builtin_amdgcn_wave_barrier();
    };
    // pair arithmetic on the complete groups of 8 in the queue (and, with `final`, on the incomplete last one)
    auto drain = [&](auto force_only, double &ua, bool final) {
        constexpr bool FO = decltype(force_only)::value;
        const double4 *qr = Q + jj; // group k of the queue, as seen by this lane: qr[8 k]
        int done = 0;
        for (; done + CLQ_U * CL_G <= tail; done += CLQ_U * CL_G) {
            double4 qv[CLQ_U];
#pragma unroll
            for (int k = 0; k < CLQ_U; k++) qv[k] = qr[done + CL_G * k];
#pragma unroll
            for (int k = 0; k < CLQ_U; k++) {
                if (FO) cl_slot_force<P>(I.x, I.y, I.z, qv[k], pp, fx, fy, fz);
                else cl_slot<P, false>(I.x, I.y, I.z, qv[k], pp, true, ua, fx, fy, fz);
            }
        }
        for (; done + CL_G <= tail; done += CL_G) {
            const double4 q = qr[done];
            if (FO) cl_slot_force<P>(I.x, I.y, I.z, q, pp, fx, fy, fz);
            else cl_slot<P, false>(I.x, I.y, I.z, q, pp, true, ua, fx, fy, fz);
        }
        const int rem = tail - done;
        if (final) {
            if (rem > 0) {
                const double4 q = jj < rem ? qr[done] : nowhere;
                if (FO) cl_slot_force<P>(I.x, I.y, I.z, q, pp, fx, fy, fz);
                else cl_slot<P, false>(I.x, I.y, I.z, q, pp, true, ua, fx, fy, fz);
            }
            tail = 0;
        } else {
            // the incomplete last group moves to the front (done >= 64 > rem: source and destination are disjoint)
            if (lane < rem) Q[lane] = Q[done + lane];
            tail = rem;
        }
        __builtin_amdgcn_wave_barrier();
    };
    using Yes = std::true_type;
    using No = std::false_type;

    // one part of the list: n entries at base[0], base[dir], ...  Entries are fetched 64 at a time (one per lane)
    // and handed out 8 per step (past the end of the list: the cluster of inert slots, which fails the box test);
    // the particles of the next step are requested before the current ones are tested.
    const int sub = lane >> 3;
    const char *__restrict__ psb = (const char *)b.ps + 32 * jj;
    auto particle = [&](unsigned ent, bool shifted) { // byte offsets stay below 2^32: checked by the launcher
        return *(const double4 *)(psb + ((shifted ? ent & (unsigned)CL_JMASK : ent) << 8));
    };
    auto walk = [&](const int *__restrict__ base, int dir, int n, bool shifted, auto force_only, double &ua) {
        const unsigned pad = (unsigned)g.max_clusters | (shifted ? (unsigned)CL_NOSHIFT << CL_CODE_SHIFT : 0u);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const unsigned mine = c0 + lane < n ? (unsigned)base[dir * (c0 + lane)] : pad;
            const int steps = (min(64, n - c0) + CL_G - 1) / CL_G;
            unsigned ea = __shfl(mine, sub);
            double4 pa = particle(ea, shifted);
            for (int k = 0; k < steps; k += 2) {
                const unsigned eb = __shfl(mine, CL_G * (k + 1) + sub); // lane index taken mod 64
                const double4 pb = particle(eb, shifted);
                filter(ea, pa, shifted);
                if (tail >= 64) drain(force_only, ua, false);
                ea = __shfl(mine, CL_G * (k + 2) + sub);
                pa = particle(ea, shifted);
                if (k + 1 < steps) {
                    filter(eb, pb, shifted);
                    if (tail >= 64) drain(force_only, ua, false);
                }
            }
        }
        drain(force_only, ua, true);
    };
    // A slab rank (not HALF) keeps the one-sided bookkeeping for the cluster pairs that are all its own: a cluster whose
    // particles are all owned takes the pairs with such clusters after it whole and those with such clusters before it
    // not at all (forces only), as on one GPU; pairs with a cluster that holds halo copies count half from here (the
    // owner of the copies, or that cluster's own wave, brings the other half), and so does everything of a cluster with
    // halo copies among its own particles.
    const bool ifull = HALF || __builtin_amdgcn_readfirstlane((int)(b.fbox[2 * (size_t)icl].w != 0.f));
    walk(L + 1, 1, nu - 1, false, No{}, ifull ? u1 : uh);
    if (ifull) walk(L + MID, 1, nd, false, Yes{}, uh);
    else walk(L + MID, 1, nd, false, No{}, uh);
    if (!HALF) walk(L + MID - 1, -1, min(max(b.n3[icl], 0), MID - 1), false, No{}, uh); // (never beyond its half of the list)
    walk(L + S - 1, -1, ns, true, No{}, uh);
    cl_finish(I, pp, jj, icl, 2.0 * u1 + uh, fx, fy, fz, acc, epart, lane, b.acc_slots);
    trace(3);
    }();
}

// The list walked at particle granularity, with the survivors sorted into two queues.  Of the particles that come
// within the cutoff of i's bounding box only 44 % lie within the cutoff of a given particle of i, so more than half of
// the lanes of k_pair_clusters_filtered evaluate pairs that contribute nothing.  Here the 64 listed particles of a
// step are tested against the 8 particles of i themselves - squared distances in packed single precision, two
// particles of i per instruction, against a cutoff widened by more than the rounding of the single-precision
// coordinates, so that no pair the double-precision arithmetic would keep is lost - and a particle is kept for the
// lower half of i (its first four particles: the cluster is sorted along z) and / or for the upper half only if it
// is within the cutoff of a particle of that half.  Each half has its own queue, evaluated with lane = (i of the
// half, one of 16 queued particles): 22 % fewer particles survive the test, and a survivor is paired only with the
// half (or halves) it is near.  The exact cutoff test of the pair arithmetic is unchanged, so are the energy
// bookkeeping and the fixed order of every sum.
#define CLS_W 16 // queued particles per evaluated group
#define CLS_U 2  // groups per unrolled step of the pair arithmetic
typedef float v2f __attribute__((ext_vector_type(2)));
template <int P, bool HALF>
__global__ __launch_bounds__(256, 4) void k_pair_clusters_split(ClusterGeom g, ClusterBuffers b, PairParams pp, float a2w,
                                                             double *__restrict__ acc, double *__restrict__ epart) {
    // a queue entry is stored as two 16-byte halves, (x, y) and (z, q), in separate arrays: 16 consecutive entries of
    // either array cover the 64 LDS banks exactly once, which is what a 16-lane group of a ds_read_b128 touches
    __shared__ double2 queue[4][2][2][CLQ_CAP];
    const int lane = threadIdx.x & 63, hh = lane >> 4, jj = lane & (CLS_W - 1);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int icl = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));
    if (icl >= g.max_clusters) return;
    const int ncl = b.clstart[g.ncols];
    if (icl >= ncl) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    const int nu = b.n0[icl], ns = b.n1[icl], nd = b.n2[icl];
    if (nu < 0) return; // list overflow: left to the list-free kernel
    // this lane's particle of the lower half (A) and of the upper half (B)
    const LaneI IA = load_lane_i(g, b, pp, icl, hh), IB = load_lane_i(g, b, pp, icl, CL_G / 2 + hh);
    if (!__ballot(IA.owned || IB.owned)) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    const int S = g.list_stride, MID = S / 2;
    const int *__restrict__ L = b.list + (size_t)icl * S;
    double2 *QA = queue[threadIdx.x >> 6][0][0], *QB = queue[threadIdx.x >> 6][1][0]; // (z, q) halves follow at + CLQ_CAP
    const double4 nowhere = make_double4(g.far, g.far, g.far, 0.0);
    // the 8 particles of i in single precision, two to a register pair (same address in every lane)
    v2f ix[4], iy[4], iz[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double4 p0 = b.ps[(size_t)CL_G * icl + 2 * k], p1 = b.ps[(size_t)CL_G * icl + 2 * k + 1];
        // a padding slot as i keeps away from the padding slots it would meet as j (as load_lane_i does)
        const float x0 = b.order[(size_t)CL_G * icl + 2 * k] == CL_DUMMY ? (float)(-2.0 * g.far) : (float)p0.x;
        const float x1 = b.order[(size_t)CL_G * icl + 2 * k + 1] == CL_DUMMY ? (float)(-2.0 * g.far) : (float)p1.x;
        ix[k] = (v2f){x0, x1};
        iy[k] = (v2f){(float)p0.y, (float)p1.y};
        iz[k] = (v2f){(float)p0.z, (float)p1.z};
    }
    int tailA = 0, tailB = 0; // entries waiting in the queues (wave-uniform)
    double u1A = 0, uhA = 0, fxA = 0, fyA = 0, fzA = 0, u1B = 0, uhB = 0, fxB = 0, fyB = 0, fzB = 0;
    {   // the cluster's own pairs: particle jj of the cluster against this lane's two particles
        const double4 own = b.ps[(size_t)CL_G * icl + (jj & (CL_G - 1))];
        cl_slot<P, true>(IA.x, IA.y, IA.z, own, pp, jj < CL_G && jj != hh, uhA, fxA, fyA, fzA);
        cl_slot<P, true>(IB.x, IB.y, IB.z, own, pp, jj < CL_G && jj != CL_G / 2 + hh, uhB, fxB, fyB, fzB);
    }

    auto filter = [&](unsigned ent, double4 p, bool shifted) {
        if (shifted) {
            double sx, sy, sz;
            shift_of_code(g, (int)(ent >> CL_CODE_SHIFT), sx, sy, sz);
            p.x += sx; p.y += sy; p.z += sz;
        }
        const float xf = (float)p.x, yf = (float)p.y, zf = (float)p.z;
        const v2f X = (v2f){xf, xf}, Y = (v2f){yf, yf}, Z = (v2f){zf, zf};
        v2f d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const v2f dx = X - ix[k], dy = Y - iy[k], dz = Z - iz[k];
            d[k] = dx * dx + dy * dy + dz * dz;
        }
        const bool keepA = fminf(fminf(d[0].x, d[0].y), fminf(d[1].x, d[1].y)) < a2w;
        const bool keepB = fminf(fminf(d[2].x, d[2].y), fminf(d[3].x, d[3].y)) < a2w;
        const unsigned long long mA = __ballot(keepA), mB = __ballot(keepB);
        if (keepA) {
            const int pos = tailA + lane_rank(mA);
            QA[pos] = make_double2(p.x, p.y);
            QA[CLQ_CAP + pos] = make_double2(p.z, p.w);
        }
        if (keepB) {
            const int pos = tailB + lane_rank(mB);
            QB[pos] = make_double2(p.x, p.y);
            QB[CLQ_CAP + pos] = make_double2(p.z, p.w);
        }
        tailA += __popcll(mA);
        tailB += __popcll(mB);
        __builtin_amdgcn_wave_barrier();
    };
    // pair arithmetic on the complete groups of CLS_W in a queue (and, with `final`, on the incomplete last one)
    auto drain = [&](auto force_only, double2 *Q, int &tail, const LaneI &I, double &ua, double &fx, double &fy, double &fz,
                     bool final) {
        constexpr bool FO = decltype(force_only)::value;
        const double2 *qr = Q + jj;
        auto entry = [&](int at) {
            const double2 xy = qr[at], zq = qr[CLQ_CAP + at];
            return make_double4(xy.x, xy.y, zq.x, zq.y);
        };
        int done = 0;
        for (; done + CLS_U * CLS_W <= tail; done += CLS_U * CLS_W) {
            double4 qv[CLS_U];
#pragma unroll
            for (int k = 0; k < CLS_U; k++) qv[k] = entry(done + CLS_W * k);
#pragma unroll
            for (int k = 0; k < CLS_U; k++) {
                if (FO) cl_slot_force<P>(I.x, I.y, I.z, qv[k], pp, fx, fy, fz);
                else cl_slot<P, false>(I.x, I.y, I.z, qv[k], pp, true, ua, fx, fy, fz);
            }
        }
        for (; done + CLS_W <= tail; done += CLS_W) {
            const double4 q = entry(done);
            if (FO) cl_slot_force<P>(I.x, I.y, I.z, q, pp, fx, fy, fz);
            else cl_slot<P, false>(I.x, I.y, I.z, q, pp, true, ua, fx, fy, fz);
        }
        const int rem = tail - done;
        if (final) {
            if (rem > 0) {
                const double4 q = jj < rem ? entry(done) : nowhere;
                if (FO) cl_slot_force<P>(I.x, I.y, I.z, q, pp, fx, fy, fz);
                else cl_slot<P, false>(I.x, I.y, I.z, q, pp, true, ua, fx, fy, fz);
            }
            tail = 0;
        } else {
            // the incomplete last group moves to the front (done >= 49 > rem: source and destination are disjoint)
            if (lane < rem) { Q[lane] = Q[done + lane]; Q[CLQ_CAP + lane] = Q[CLQ_CAP + done + lane]; }
            tail = rem;
        }
        __builtin_amdgcn_wave_barrier();
    };
    using Yes = std::true_type;
    using No = std::false_type;

    // one part of the list, as in k_pair_clusters_filtered: entries fetched 64 at a time, handed out 8 per step (lane
    // (e, s) loads particle s of entry e), the particles of the next step requested before the current ones are tested
    const int sub = lane >> 3;
    const char *__restrict__ psb = (const char *)b.ps + 32 * (lane & 7);
    auto particle = [&](unsigned ent, bool shifted) { // byte offsets stay below 2^32: checked by the launcher
        return *(const double4 *)(psb + ((shifted ? ent & (unsigned)CL_JMASK : ent) << 8));
    };
    auto walk = [&](const int *__restrict__ base, int dir, int n, bool shifted, auto force_only, double &uaA, double &uaB) {
        const unsigned pad = (unsigned)g.max_clusters | (shifted ? (unsigned)CL_NOSHIFT << CL_CODE_SHIFT : 0u);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const unsigned mine = c0 + lane < n ? (unsigned)base[dir * (c0 + lane)] : pad;
            const int steps = (min(64, n - c0) + CL_G - 1) / CL_G;
            unsigned ea = __shfl(mine, sub);
            double4 pa = particle(ea, shifted);
            for (int k = 0; k < steps; k++) {
                const unsigned en = __shfl(mine, CL_G * (k + 1) + sub); // lane index taken mod 64; past the end: padding
                const double4 pn = particle(en, shifted);
                filter(ea, pa, shifted);
                if (tailA >= 64) drain(force_only, QA, tailA, IA, uaA, fxA, fyA, fzA, false);
                if (tailB >= 64) drain(force_only, QB, tailB, IB, uaB, fxB, fyB, fzB, false);
                ea = en;
                pa = pn;
            }
        }
        drain(force_only, QA, tailA, IA, uaA, fxA, fyA, fzA, true);
        drain(force_only, QB, tailB, IB, uaB, fxB, fyB, fzB, true);
    };
    walk(L + 1, 1, nu - 1, false, No{}, HALF ? u1A : uhA, HALF ? u1B : uhB);
    if (HALF) walk(L + MID, 1, nd, false, Yes{}, uhA, uhB);
    else walk(L + MID, 1, nd, false, No{}, uhA, uhB);
    if (!HALF) walk(L + MID - 1, -1, min(max(b.n3[icl], 0), MID - 1), false, No{}, uhA, uhB); // neighbours with halo copies (a slab rank only)
    walk(L + S - 1, -1, ns, true, No{}, uhA, uhB);

    // sum the CLS_W lanes that share a particle i, write its acceleration and energy share
    double e = 0.0;
    auto finish = [&](const LaneI &I, int member, double u, double fx, double fy, double fz) {
#pragma unroll
        for (int off = 1; off < CLS_W; off <<= 1) {
            u += __shfl_xor(u, off);
            fx += __shfl_xor(fx, off);
            fy += __shfl_xor(fy, off);
            fz += __shfl_xor(fz, off);
        }
        if (jj == 0 && I.owned) {
            const double w = pp.convention == MSM_AMD_ACC_FORCE ? -I.q : 1.0;
            if (b.acc_slots) {
                b.acc_slots[(size_t)CL_G * icl + member] = make_double4(w * fx, w * fy, w * fz, 0.0);
            } else {
                acc[3 * (size_t)I.orig] = w * fx;
                acc[3 * (size_t)I.orig + 1] = w * fy;
                acc[3 * (size_t)I.orig + 2] = w * fz;
            }
            e += 0.5 * I.q * u;
        }
    };
    finish(IA, hh, 2.0 * u1A + uhA, fxA, fyA, fzA);
    finish(IB, CL_G / 2 + hh, 2.0 * u1B + uhB, fxB, fyB, fzB);
    e = wave_sum(e);
    if (lane == 0) epart[icl] = e;
}

// The same sum for one cluster without a list: the wave enumerates its candidates and evaluates each surviving
// cluster pair on the spot.
template <int P>
__device__ __forceinline__ void direct_one_cluster(const ClusterGeom &g, const ClusterBuffers &b, const PairParams &pp, double a,
                                                   double *__restrict__ acc, double *__restrict__ epart, int icl, int lane,
                                                   bool half, bool onesided = false, double a2_enum = 0.0) {
    const int ii = lane >> 3, jj = lane & 7;
    const LaneI I = load_lane_i(g, b, pp, icl, ii);
    if (!__ballot(I.owned)) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    const double4 il = b.blo[icl], ih = b.bhi[icl];
    const IBox ib = {il.x, il.y, il.z, ih.x, ih.y, ih.z};
    const double4 *__restrict__ pj = b.ps + jj;
    // energy bookkeeping as in k_pair_clusters: with `half`, an unshifted pair counts whole on the side whose own
    // cluster comes first, and not at all on the other side
    // onesided (a slab rank whose other clusters are evaluated by k_pair_clusters_filtered): the same for the pairs of
    // two clusters whose particles are all owned, half for the others
    double u1 = 0, uh = 0, fx = 0, fy = 0, fz = 0;
    const bool ifull = onesided && b.fbox[2 * (size_t)icl].w != 0.f;
    // (a, a2_enum: reach of the walk over the sort's bins - the cutoff, or more when the particles have moved since the
    // sort; the pair arithmetic applies the cutoff itself)
    cl_enumerate(g, b, a, a2_enum > 0.0 ? a2_enum : pp.a2, ib, lane, [&](bool keep, int jcl, int code, bool jfull_lane) {
        const int jfull_bits = (int)jfull_lane;
        unsigned long long m = __ballot(keep);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int j = __builtin_amdgcn_readlane(jcl, l), cd = __builtin_amdgcn_readlane(code, l);
            const bool one = half || (ifull && __builtin_amdgcn_readlane(jfull_bits, l) != 0);
            const double4 q = pj[(size_t)CL_G * j];
            if (cd == CL_NOSHIFT) {
                if (j == icl) cl_slot<P, true>(I.x, I.y, I.z, q, pp, ii != jj, uh, fx, fy, fz);
                else {
                    // (the sum the pair's energy goes to is picked by value: a reference picked at run time would
                    // put all three sums, and whatever is captured with them, into scratch memory)
                    double ut = 0.0;
                    cl_slot<P, false>(I.x, I.y, I.z, q, pp, true, ut, fx, fy, fz);
                    uh += one ? 0.0 : ut;
                    u1 += one && j > icl ? ut : 0.0;
                }
            } else {
                double sx, sy, sz;
                shift_of_code(g, cd, sx, sy, sz);
                cl_slot<P, false>(I.x - sx, I.y - sy, I.z - sz, q, pp, true, uh, fx, fy, fz);
            }
        }
    });
    cl_finish(I, pp, jj, icl, 2.0 * u1 + uh, fx, fy, fz, acc, epart, lane, b.acc_slots);
}

// every cluster list-free (A/B variant)
template <int P>
__global__ __launch_bounds__(256) void k_pair_clusters_direct(ClusterGeom g, ClusterBuffers b, PairParams pp, double a,
                                                              double *__restrict__ acc, double *__restrict__ epart, int half) {
    const int lane = threadIdx.x & 63;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int icl = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));
    if (icl >= g.max_clusters) return;
    if (icl >= b.clstart[g.ncols]) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    direct_one_cluster<P>(g, b, pp, a, acc, epart, icl, lane, half != 0);
}

// only the clusters whose list overflowed (n0 < 0): a small fixed grid scans the list lengths, 64 per wave step,
// so that the usual case (none) costs a few microseconds
template <int P>
__global__ __launch_bounds__(256) void k_pair_clusters_overflow(ClusterGeom g, ClusterBuffers b, PairParams pp, double a,
                                                                double *__restrict__ acc, double *__restrict__ epart,
                                                                int half, int onesided, double a2_enum) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int ncl = b.clstart[g.ncols];
    for (int base = wave * 64; base < ncl; base += nwaves * 64) {
        unsigned long long m = __ballot(base + lane < ncl && b.n0[base + lane] < 0);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            direct_one_cluster<P>(g, b, pp, a, acc, epart, base + l, lane, half != 0, onesided != 0, a2_enum);
        }
    }
}

// ---------------------------------------------------------------------------
// neighbour structures kept over several evaluations (msm_amd_eval_reuse; every particle owned)
// ---------------------------------------------------------------------------
#define HL_PAD_CODE ((unsigned)CL_NOSHIFT << CL_CODE_SHIFT)

// Every slot of the pair set follows its particle: new position = the caller's position wrapped into the cell exactly as
// the sort wraps it, then the periodic image of that nearest the slot's position at the rebuild (so a caller that wraps
// its coordinates moves nothing, a particle that crosses a face of the cell stays next to its neighbours, and the
// coordinates are bit for bit those a fresh sort would hold, up to that whole image), charge as given now.  A particle that
// has moved more than skin / 2 raises the flag that makes the kernels behind this one rebuild.  The 8 lanes of a cluster
// also renew its bounding box (read by the list-free kernel only).
__global__ __launch_bounds__(256) void k_reuse_follow(ClusterGeom g, ReuseBuffers rs, const double *__restrict__ r,
                                                      const double *__restrict__ q, double half_skin2) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncl = rs.counters[0];
    if (s >= CL_G * ncl) return; // (whole clusters: 8 consecutive lanes leave together)
    const int i = rs.order[s];
    double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
    bool moved = false;
    if (i != CL_DUMMY) {
        const double4 ref = rs.ref[s];
        const double refc[3] = {ref.x, ref.y, ref.z};
        double nw[3], d2 = 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double t = (r[3 * (size_t)i + d] - g.origin[d]) * g.invA[d];
            const double w = (t - floor(t)) * g.A[d];
            nw[d] = w - g.A[d] * rint((w - refc[d]) * g.invA[d]);
            const double dd = nw[d] - refc[d];
            d2 = fma(dd, dd, d2);
            l[d] = h[d] = nw[d];
        }
        moved = !(d2 <= half_skin2); // (a NaN position counts as moved)
        rs.ps[s] = make_double4(nw[0], nw[1], nw[2], q[i]);
    }
    if (__ballot(moved) && (threadIdx.x & 63) == 0) atomicOr(rs.need, 1);
#pragma unroll
    for (int off = 1; off < CL_G; off <<= 1)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            l[d] = fmin(l[d], __shfl_xor(l[d], off));
            h[d] = fmax(h[d], __shfl_xor(h[d], off));
        }
    if ((s & (CL_G - 1)) == 0) {
        const int k = s / CL_G;
        rs.blo[k] = make_double4(l[0], l[1], l[2], 0.0);
        rs.bhi[k] = make_double4(h[0], h[1], h[2], 0.0);
        rs.fbox[2 * (size_t)k] = make_float4(__double2float_rd(l[0]), __double2float_rd(l[1]), __double2float_rd(l[2]), 1.f);
        rs.fbox[2 * (size_t)k + 1] = make_float4(__double2float_ru(h[0]), __double2float_ru(h[1]), __double2float_ru(h[2]), 0.f);
    }
}

// Rebuild, step 1 (only when the flag is up): the fresh sort becomes the pair set.
__global__ __launch_bounds__(256) void k_reuse_adopt(ClusterGeom g, ClusterBuffers b, ReuseBuffers rs) {
    if (*rs.need == 0) return;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    const size_t nslots = (size_t)CL_G * g.max_clusters;
    for (size_t s = t; s < nslots + CL_G; s += nt) { // (+ the cluster of inert slots behind the last one)
        const double4 p = b.ps[s];
        rs.ps[s] = p;
        if (s < nslots) { rs.ref[s] = p; rs.order[s] = b.order[s]; }
    }
    for (size_t k = t; k < (size_t)g.max_clusters; k += nt) {
        rs.blo[k] = b.blo[k]; rs.bhi[k] = b.bhi[k];
        rs.fbox[2 * k] = b.fbox[2 * k]; rs.fbox[2 * k + 1] = b.fbox[2 * k + 1];
    }
    for (size_t k = t; k <= (size_t)g.nbins; k += nt) rs.start[k] = b.start[k];
    for (size_t k = t; k <= (size_t)g.ncols; k += nt) { rs.clstart[k] = b.clstart[k]; rs.colncl[k] = b.colncl[k]; }
    if (t == 0) { rs.counters[0] = b.clstart[g.ncols]; rs.counters[1] += 1; }
}

// Rebuild, step 3 (step 2 is k_cl_list with cutoff + skin): the half lists.  One wave per cluster walks its cluster
// list 8 entries at a time, lane (e, s) = particle s of entry e, and tests that particle against the 8 particles of
// the cluster themselves - squared distances in packed single precision against a cutoff widened by more than the
// rounding of the converted coordinates, so that no pair double precision would keep is lost.  A particle within
// cutoff + skin of one of the first four is appended to the lower half's list, of one of the last four to the upper
// half's, in list order (ballot + lane rank), so the lists - and every sum made from them - do not depend on timing.
__global__ __launch_bounds__(256) void k_reuse_halflists(ClusterGeom g, ClusterBuffers b, ReuseBuffers rs, float a2w) {
    if (*rs.need == 0) return;
    const int lane = threadIdx.x & 63;
    const int icl = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (icl >= b.clstart[g.ncols]) return;
    const int nu = rs.n0[icl], ns = rs.n1[icl], nd = rs.n2[icl];
    if (nu < 0) return; // its cluster list overflowed: the list-free kernel takes the cluster
    v2f ix[4], iy[4], iz[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double4 p0 = b.ps[(size_t)CL_G * icl + 2 * k], p1 = b.ps[(size_t)CL_G * icl + 2 * k + 1];
        ix[k] = (v2f){(float)p0.x, (float)p1.x}; // (padding slots lie far away on either side: they pass no test)
        iy[k] = (v2f){(float)p0.y, (float)p1.y};
        iz[k] = (v2f){(float)p0.z, (float)p1.z};
    }
    const int S = rs.list_stride, MID = S / 2;
    const int *__restrict__ L = rs.list + (size_t)icl * S;
    int *__restrict__ HA = rs.hl + ((size_t)icl * 2) * rs.hcap, *__restrict__ HB = HA + rs.hcap;
    int tailA = 0, tailB = 0;
    bool overflow = false;
    const int sub = lane >> 3, sl = lane & 7;
    auto walk = [&](const int *__restrict__ base, int dir, int n, bool shifted) {
        for (int c0 = 0; c0 < n; c0 += 64) {
            const unsigned mine = c0 + lane < n ? (unsigned)base[dir * (c0 + lane)] : 0u;
            const int steps = (min(64, n - c0) + CL_G - 1) / CL_G;
            for (int k = 0; k < steps; k++) {
                const unsigned ent = __shfl(mine, CL_G * k + sub);
                const bool live = c0 + CL_G * k + sub < n;
                const unsigned slot = ((shifted ? ent & (unsigned)CL_JMASK : ent) << 3) + sl;
                double4 p = b.ps[live ? slot : (unsigned)CL_G * g.max_clusters];
                if (shifted) {
                    double sx, sy, sz;
                    shift_of_code(g, (int)(ent >> CL_CODE_SHIFT), sx, sy, sz);
                    p.x += sx; p.y += sy; p.z += sz;
                }
                const float xf = (float)p.x, yf = (float)p.y, zf = (float)p.z;
                const v2f X = (v2f){xf, xf}, Y = (v2f){yf, yf}, Z = (v2f){zf, zf};
                v2f d[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const v2f dx = X - ix[m], dy = Y - iy[m], dz = Z - iz[m];
                    d[m] = dx * dx + dy * dy + dz * dz;
                }
                const bool keepA = live && fminf(fminf(d[0].x, d[0].y), fminf(d[1].x, d[1].y)) < a2w;
                const bool keepB = live && fminf(fminf(d[2].x, d[2].y), fminf(d[3].x, d[3].y)) < a2w;
                const unsigned long long mA = __ballot(keepA), mB = __ballot(keepB);
                const int cA = __popcll(mA), cB = __popcll(mB);
                if (tailA + cA > rs.hcap || tailB + cB > rs.hcap) overflow = true;
                if (!overflow) {
                    // every entry carries a shift code (the zero shift for an unshifted neighbour)
                    const int entry = (int)(slot | (shifted ? ent & ~(unsigned)CL_JMASK : HL_PAD_CODE));
                    if (keepA) HA[tailA + lane_rank(mA)] = entry;
                    if (keepB) HB[tailB + lane_rank(mB)] = entry;
                }
                tailA += cA;
                tailB += cB;
            }
        }
        // a section ends on a whole group of 16: the rest of its last group points at the cluster of inert slots
        const int padA = (16 - (tailA & 15)) & 15, padB = (16 - (tailB & 15)) & 15;
        if (tailA + padA > rs.hcap || tailB + padB > rs.hcap) overflow = true;
        if (!overflow) {
            const int inert = (int)((unsigned)CL_G * g.max_clusters | HL_PAD_CODE);
            if (lane < padA) HA[tailA + lane] = inert;
            if (lane < padB) HB[tailB + lane] = inert;
        }
        tailA += padA;
        tailB += padB;
    };
    walk(L + 1, 1, nu - 1, false);
    const int upA = tailA, upB = tailB;
    walk(L + MID, 1, nd, false);
    const int dnA = tailA - upA, dnB = tailB - upB;
    walk(L + S - 1, -1, ns, true);
    if (lane == 0) { // in groups of 16 entries
        rs.hn[2 * (size_t)icl] = make_int4(upA >> 4, dnA >> 4, (tailA - upA - dnA) >> 4, 0);
        rs.hn[2 * (size_t)icl + 1] = make_int4(upB >> 4, dnB >> 4, (tailB - upB - dnB) >> 4, 0);
        if (overflow) rs.n0[icl] = -1; // a half list did not fit: the list-free kernel takes the cluster
    }
}

// The pair sum over the half lists: one wave per cluster, lane = (hh, jj): particle hh of the lower half and particle
// 4 + hh of the upper half against entry jj of a group of 16 of that half's list.  No test, no queue: every listed
// particle was within cutoff + skin of its half at the rebuild, so about three lanes in four hold a pair inside the
// cutoff (one in two when survivors of a box test are paired with all 8 particles of a cluster).  A step takes 64
// list entries: every lane fetches one entry and its particle (one trip through the vector cache per particle, not one
// per lane that uses it - with every lane gathering its own operand the kernel was bound by that cache, 1.0 ms), leaves
// it in the wave's staging area in LDS as two 16-byte halves (16 consecutive entries of either array cover the 64 banks
// once), and the four groups of 16 are read back as broadcasts; the particles of the next step are requested before
// the current ones are used.  Energy bookkeeping as in k_pair_clusters_filtered with every particle owned: pairs with
// clusters after this one whole, with clusters before it forces only, the cluster's own pairs and pairs through an
// image half from either side.
template <int P>
__global__ __launch_bounds__(64, 4) void k_pair_halflists(ClusterGeom g, ReuseBuffers rs, PairParams pp, double *__restrict__ acc,
                                                          double *__restrict__ epart) {
    __shared__ double2 stage[2][64]; // (x, y) and (z, q) of the 64 particles of a step
    const int lane = threadIdx.x, hh = lane >> 4, jj = lane & 15;
    const int ncl = rs.counters[0];
    const int icl = (int)blockIdx.x < ncl ? xcd_remap(blockIdx.x, ncl) : (int)blockIdx.x;
    if (blockIdx.x == 0 && lane == 0) *rs.need = 0; // the rebuild of this evaluation, if there was one, is complete
    if (icl >= ncl) {
        if (lane == 0) epart[icl] = 0.0;
        return;
    }
    if (rs.n0[icl] < 0) return; // left to the list-free kernel
    ClusterBuffers view = {};
    view.ps = rs.ps; view.order = rs.order;
    const LaneI IA = load_lane_i(g, view, pp, icl, hh), IB = load_lane_i(g, view, pp, icl, CL_G / 2 + hh);
    double u1A = 0, uhA = 0, fxA = 0, fyA = 0, fzA = 0, u1B = 0, uhB = 0, fxB = 0, fyB = 0, fzB = 0;
    {   // the cluster's own pairs
        const double4 own = rs.ps[(size_t)CL_G * icl + (jj & (CL_G - 1))];
        cl_slot<P, true>(IA.x, IA.y, IA.z, own, pp, jj < CL_G && jj != hh, uhA, fxA, fyA, fzA);
        cl_slot<P, true>(IB.x, IB.y, IB.z, own, pp, jj < CL_G && jj != CL_G / 2 + hh, uhB, fxB, fyB, fzB);
    }
    const unsigned pad = (unsigned)CL_G * g.max_clusters | HL_PAD_CODE; // the inert cluster: fails the cutoff
    const char *__restrict__ psb = (const char *)rs.ps;
    auto gather = [&](unsigned ent) { return *(const double4 *)(psb + ((size_t)(ent & (unsigned)CL_JMASK) << 5)); };
    // One half's list as one stream of groups: gu groups of clusters after this one (energy whole), gd of clusters before
    // it (forces only), gs through a periodic image (energy half).  List entries are requested two steps ahead and
    // particles one step ahead, so neither of the two dependent loads of a step is waited for.
    auto stream = [&](const int *__restrict__ list, int gu, int gd, int gs, const LaneI &I, double &u1, double &uh, double &fx,
                      double &fy, double &fz) {
        const int ng = gu + gd + gs, n = 16 * ng;
        if (ng == 0) return;
        unsigned ent0 = lane < n ? (unsigned)list[lane] : pad, ent1 = 64 + lane < n ? (unsigned)list[64 + lane] : pad;
        double4 p = gather(ent0);
        for (int e0 = 0; e0 < n; e0 += 64) {
            if (e0 + 64 > 16 * (gu + gd)) { // (uniform) a step that reaches the last section: the image of the listed particle
                double sx, sy, sz;          // next to this cluster (entries of the other sections carry the zero shift)
                shift_of_code(g, (int)(ent0 >> CL_CODE_SHIFT), sx, sy, sz);
                p.x += sx; p.y += sy; p.z += sz;
            }
            stage[0][lane] = make_double2(p.x, p.y);
            stage[1][lane] = make_double2(p.z, p.w);
            __builtin_amdgcn_wave_barrier();
            ent0 = ent1;
            p = gather(ent0);
            ent1 = e0 + 128 + lane < n ? (unsigned)list[e0 + 128 + lane] : pad;
            const int g0 = e0 >> 4, groups = min(4, ng - g0);
            for (int k = 0; k < groups; k++) {
                const double2 xy = stage[0][16 * k + jj], zq = stage[1][16 * k + jj];
                const double4 pj = make_double4(xy.x, xy.y, zq.x, zq.y);
                const int gk = g0 + k; // (uniform over the wave)
                if (gk < gu) cl_slot<P, false>(I.x, I.y, I.z, pj, pp, true, u1, fx, fy, fz);
                else if (gk < gu + gd) cl_slot_force<P>(I.x, I.y, I.z, pj, pp, fx, fy, fz);
                else cl_slot<P, false>(I.x, I.y, I.z, pj, pp, true, uh, fx, fy, fz);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    const int *__restrict__ HA = rs.hl + ((size_t)icl * 2) * rs.hcap, *__restrict__ HB = HA + rs.hcap;
    const int4 nA = rs.hn[2 * (size_t)icl], nB = rs.hn[2 * (size_t)icl + 1];
    stream(HA, nA.x, nA.y, nA.z, IA, u1A, uhA, fxA, fyA, fzA);
    stream(HB, nB.x, nB.y, nB.z, IB, u1B, uhB, fxB, fyB, fzB);
    double e = 0.0;
    auto finish = [&](const LaneI &I, double u, double fx, double fy, double fz) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            u += __shfl_xor(u, off);
            fx += __shfl_xor(fx, off);
            fy += __shfl_xor(fy, off);
            fz += __shfl_xor(fz, off);
        }
        if (jj == 0 && I.orig != CL_DUMMY) {
            const double w = pp.convention == MSM_AMD_ACC_FORCE ? -I.q : 1.0;
            acc[3 * (size_t)I.orig] = w * fx;
            acc[3 * (size_t)I.orig + 1] = w * fy;
            acc[3 * (size_t)I.orig + 2] = w * fz;
            e += 0.5 * I.q * u;
        }
    };
    finish(IA, 2.0 * u1A + uhA, fxA, fyA, fzA);
    finish(IB, 2.0 * u1B + uhB, fxB, fyB, fzB);
    e = wave_sum(e);
    if (lane == 0) epart[icl] = e;
}

void launch_reuse_pairs(hipStream_t st, const ClusterGeom &g, const ClusterBuffers &main, const ReuseBuffers &rs, const PairParams &pp,
                        const double *d_r, const double *d_q, double skin, double *acc, double *epart, hipEvent_t *ev) {
    const double a = sqrt(pp.a2), amax = std::max(g.A[0], std::max(g.A[1], g.A[2]));
    const int nslots = CL_G * g.max_clusters, nblk4 = (g.max_clusters + 3) / 4;
    hipLaunchKernelGGL(k_reuse_follow, dim3((nslots + 255) / 256), dim3(256), 0, st, g, rs, d_r, d_q, 0.25 * skin * skin);
    // rebuild (each of the three returns at once unless the flag is up)
    hipLaunchKernelGGL(k_reuse_adopt, dim3(2048), dim3(256), 0, st, g, main, rs);
    {
        ClusterGeom gw = g; // cluster lists with cutoff + skin, in the pair set's own list buffer
        const double aw = a + skin;
        gw.list_stride = rs.list_stride;
        { const double w = aw + 4e-6 * amax; gw.a2_wide = (float)(w * w * (1.0 + 1e-6)); }
        ClusterBuffers bw = main;
        bw.list = rs.list; bw.n0 = rs.n0; bw.n1 = rs.n1; bw.n2 = rs.n2;
        hipLaunchKernelGGL(k_cl_list<false>, dim3(nblk4), dim3(256), 0, st, gw, bw, aw, aw * aw, pp.n_owned, (const int *)rs.need);
        // cutoff of the single-precision test: coordinates up to 2 A (a periodic image) carry 2^-24 relative rounding each
        const double eps = 2e-6 * amax / aw + 4e-6;
        const float a2w = (float)(aw * aw * (1.0 + eps)) * 1.000001f;
        hipLaunchKernelGGL(k_reuse_halflists, dim3(nblk4), dim3(256), 0, st, g, main, rs, a2w);
    }
    if (ev) hipEventRecord(ev[0], st);
    if (pp.p == 4) hipLaunchKernelGGL((k_pair_halflists<4>), dim3(g.max_clusters), dim3(64), 0, st, g, rs, pp, acc, epart);
    else hipLaunchKernelGGL((k_pair_halflists<6>), dim3(g.max_clusters), dim3(64), 0, st, g, rs, pp, acc, epart);
    if (ev) hipEventRecord(ev[1], st);
    {   // clusters whose lists overflowed: list-free, over the frozen bins with a reach that covers the motion since the sort
        ClusterGeom go = g;
        const double ar = a + 0.5 * skin + 1e-9 * amax;
        { const double w = ar + 4e-6 * amax; go.a2_wide = (float)(w * w * (1.0 + 1e-6)); }
        ClusterBuffers bo = {};
        bo.start = rs.start; bo.clstart = rs.clstart; bo.colncl = rs.colncl; bo.order = rs.order; bo.ps = rs.ps;
        bo.blo = rs.blo; bo.bhi = rs.bhi; bo.fbox = rs.fbox; bo.n0 = rs.n0;
        if (pp.p == 4) hipLaunchKernelGGL((k_pair_clusters_overflow<4>), dim3(std::min(nblk4, 2048)), dim3(256), 0, st, go, bo, pp, ar, acc, epart, 1, 0, ar * ar);
        else hipLaunchKernelGGL((k_pair_clusters_overflow<6>), dim3(std::min(nblk4, 2048)), dim3(256), 0, st, go, bo, pp, ar, acc, epart, 1, 0, ar * ar);
    }
}

void launch_pair_clusters(hipStream_t st, const ClusterGeom &g, const ClusterBuffers &b, const PairParams &pp, double *acc,
                          double *epart, int variant, int n_total, hipEvent_t *ev, int max_wgs, hipEvent_t after_lists, int lds_pad_wpe,
                          const std::function<void()> *between, int nchunks, const std::function<void()> *before_last) {
    const double a = sqrt(pp.a2);
    // A/B: unused LDS per workgroup, waves per workgroup of the list-walking kernel (0: one), of the list build (0: four)
    const int lds_pad = lds_pad_wpe & 0xFFFFF, wpb = (lds_pad_wpe >> 20) & 0xF, list_wpb = (lds_pad_wpe >> 24) & 0xF;
    const int nblk = (g.max_clusters + 3) / 4;
    const int chunks = std::max(1, std::min(nchunks, 8));
    const int nwg = max_wgs > 0 ? std::min(nblk, max_wgs) : nblk; // workgroups of the list-walking kernel (it loops over its share)
    const int half = pp.n_owned >= n_total ? 1 : 0; // every particle owned: pair energies are taken from one side
    if (variant == 1) {
        if (after_lists) hipEventRecord(after_lists, st); // no lists: nothing to wait for
        if (between) (*between)();
        if (pp.p == 4) hipLaunchKernelGGL((k_pair_clusters_direct<4>), dim3(nblk), dim3(256), 0, st, g, b, pp, a, acc, epart, half);
        else hipLaunchKernelGGL((k_pair_clusters_direct<6>), dim3(nblk), dim3(256), 0, st, g, b, pp, a, acc, epart, half);
        return;
    }
    // the pair kernels that read the fourth list section (every instantiation but the one-sided ones of a handle without
    // halo copies) need its length written: only those one-sided kernels go with the list build that leaves it out
    const bool h = half && variant != 2; // variant 2: every pair counted half from both sides (A/B)
    if (list_wpb == 1) {
        if (h) hipLaunchKernelGGL(k_cl_list<false>, dim3(g.max_clusters), dim3(64), 0, st, g, b, a, pp.a2, pp.n_owned, (const int *)nullptr);
        else hipLaunchKernelGGL(k_cl_list<true>, dim3(g.max_clusters), dim3(64), 0, st, g, b, a, pp.a2, pp.n_owned, (const int *)nullptr);
    } else {
        if (h) hipLaunchKernelGGL(k_cl_list<false>, dim3(nblk), dim3(256), 0, st, g, b, a, pp.a2, pp.n_owned, (const int *)nullptr);
        else hipLaunchKernelGGL(k_cl_list<true>, dim3(nblk), dim3(256), 0, st, g, b, a, pp.a2, pp.n_owned, (const int *)nullptr);
    }
    if (after_lists) hipEventRecord(after_lists, st); // lets another stream start beside the pair kernel, not beside the list build
    if (between) (*between)();
#define LAUNCH_PC(P, HALF, U)                                                                                  \
    do {                                                                                                       \
        if (ev) hipEventRecord(ev[0], st);                                                                     \
        hipLaunchKernelGGL((k_pair_clusters<P, HALF, U>), dim3(nblk), dim3(256), 0, st, g, b, pp, acc, epart); \
        if (ev) hipEventRecord(ev[1], st);                                                                     \
    } while (0)
    // a slab rank's list-free kernel books the pair energies as the kernel that walks the other clusters' lists does
    const int onesided = (!half && variant == 3 && g.max_clusters < (1 << 24)) ? 1 : 0;
#define LAUNCH_PF(P, HALF)                                                                                          \
    do {                                                                                                            \
        if (ev) hipEventRecord(ev[0], st);                                                                          \
        if (max_wgs > 0 || wpb == 4) hipLaunchKernelGGL((k_pair_clusters_filtered<P, HALF, 1, 4>), dim3(nwg), dim3(256), lds_pad, st, g, b, pp, acc, epart, 0, 1); \
        else if (wpb == 2) hipLaunchKernelGGL((k_pair_clusters_filtered<P, HALF, 1, 2>), dim3((g.max_clusters + 1) / 2), dim3(128), lds_pad, st, g, b, pp, acc, epart, 0, 1); \
        else for (int c = 0; c < chunks; c++) { \
            /* chunk c holds at most a chunks-th of the slots (+1 for the rounding); the last one everything behind the */ \
            /* first chunks-1 shares of the fewest clusters n_total particles can make */ \
            const int most = c + 1 < chunks ? g.max_clusters / chunks + 1 \
                                            : g.max_clusters - (int)((long long)((n_total + CL_G - 1) / CL_G) * c / chunks); \
            if (c + 1 == chunks && chunks > 1 && before_last) (*before_last)(); \
            hipLaunchKernelGGL((k_pair_clusters_filtered<P, HALF, (HALF ? 1 : 4), 1>), dim3(most), dim3(64), lds_pad, st, g, b, pp, acc, epart, c, chunks); \
        } \
        if (ev) hipEventRecord(ev[1], st);                                                                          \
    } while (0)
    if (variant == 4 && g.max_clusters < (1 << 24)) { // particle granularity, survivors split by the half of i they are near
        // cutoff of the single-precision test: coordinates up to 2 A (a periodic image) carry 2^-24 relative rounding each
        double amax = std::max(g.A[0], std::max(g.A[1], g.A[2]));
        const double eps = 2e-6 * amax / a + 4e-6;
        const float a2w = (float)(pp.a2 * (1.0 + eps)) * 1.000001f;
#define LAUNCH_PS(P, HALF)                                                                                               \
    do {                                                                                                                 \
        if (ev) hipEventRecord(ev[0], st);                                                                               \
        hipLaunchKernelGGL((k_pair_clusters_split<P, HALF>), dim3(nblk), dim3(256), 0, st, g, b, pp, a2w, acc, epart);   \
        if (ev) hipEventRecord(ev[1], st);                                                                               \
    } while (0)
        if (pp.p == 4) { if (h) LAUNCH_PS(4, true); else LAUNCH_PS(4, false); }
        else { if (h) LAUNCH_PS(6, true); else LAUNCH_PS(6, false); }
#undef LAUNCH_PS
    } else if (variant == 3 && g.max_clusters < (1 << 24)) { // lists walked at particle granularity (32-bit byte offsets)
        if (pp.p == 4) { if (h) LAUNCH_PF(4, true); else LAUNCH_PF(4, false); }
        else { if (h) LAUNCH_PF(6, true); else LAUNCH_PF(6, false); }
    } else if (pp.p == 4) {
        if (h) LAUNCH_PC(4, true, 4);
        else LAUNCH_PC(4, false, 4);
    } else {
        if (h) LAUNCH_PC(6, true, 4);
        else LAUNCH_PC(6, false, 4);
    }
    if (pp.p == 4) {
        hipLaunchKernelGGL((k_pair_clusters_overflow<4>), dim3(std::min(nblk, 2048)), dim3(256), 0, st, g, b, pp, a, acc, epart, h ? 1 : 0, onesided, 0.0);
    } else {
        hipLaunchKernelGGL((k_pair_clusters_overflow<6>), dim3(std::min(nblk, 2048)), dim3(256), 0, st, g, b, pp, a, acc, epart, h ? 1 : 0, onesided, 0.0);
    }
#undef LAUNCH_PC
#undef LAUNCH_PF
}

} // namespace msm
