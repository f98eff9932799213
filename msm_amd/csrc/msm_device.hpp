// Small device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include "msm_kernels.hpp"

namespace msm {

// Blocks b and b+8 share an XCD (and its L2).  Give each XCD a contiguous run of
// logical ids so that neighbouring tiles / cells, which share halo data, meet in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ int wrap1(int i, int M) { // i in [-M, 2M)
    i = (i < 0) ? i + M : i;
    return (i >= M) ? i - M : i;
}
__device__ __forceinline__ int modp(int i, int M) {
    int r = i % M;
    return r < 0 ? r + M : r;
}

// i mod M for an i that is almost always within one period of [0, M): two selects, and the division only for the rest
__device__ __forceinline__ int wrapm(int i, int M) {
    i = (i < 0) ? i + M : i;
    i = (i >= M) ? i - M : i;
    if (i < 0 || i >= M) i = modp(i, M);
    return i;
}

// Column number of column (cx, cy): tiles of 8 x 8 columns, tile after tile.  Clusters are numbered in column
// order and waves take clusters in that order, so the waves in flight at any time work on a compact patch of
// columns and find their neighbours' particles in L2 instead of fetching them once per row of columns.
__device__ __forceinline__ int cl_col(const ClusterGeom &g, int cx, int cy) {
    return (((cy >> 3) * g.ntx + (cx >> 3)) << 6) + ((cy & 7) << 3) + (cx & 7);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

} // namespace msm
