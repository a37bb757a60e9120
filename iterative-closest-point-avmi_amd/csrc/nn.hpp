// nn.hpp — LDS-tiled exhaustive nearest-neighbour scan (device side).
//
// Replaces scipy.spatial.KDTree.query(k=1) at reference utilities/icp.py:179.
// Arithmetic is the one cKDTree uses for p=2: d2 = sum_axis (a-b)*(a-b) in
// float64 with separate multiply and add (the library is built with
// -ffp-contract=off), lowest target index on exact ties, so indices and
// distances agree bit for bit with the CPU oracle.
#pragma once
#include "common.hpp"

namespace icpmi {

constexpr int NN_CHUNK = 16;   // targets per running-minimum chunk

template <int DIM>
__device__ __forceinline__ double sqdist(const double (&p)[DIM], const double* __restrict__ q) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const double t = p[d] - q[d];
        s += t * t;
    }
    return s;
}

// Stage `count` target points (row-major, DIM doubles each) into LDS with
// coalesced reads — 16 B per lane for DIM == 2 — and pad the tail up to the
// next multiple of NN_CHUNK with +inf rows (their distance is +inf, never a
// minimum).  Returns the padded count.  Caller brackets with __syncthreads().
template <int DIM>
__device__ __forceinline__ int stage_targets(const double* __restrict__ g, int count, double* tile) {
    const int padded = (count + NN_CHUNK - 1) / NN_CHUNK * NN_CHUNK;
    if constexpr (DIM == 2) {
        const double2* g2 = reinterpret_cast<const double2*>(g);
        double2* t2 = reinterpret_cast<double2*>(tile);
        for (int i = threadIdx.x; i < padded; i += blockDim.x)
            t2[i] = i < count ? g2[i] : make_double2(__builtin_inf(), __builtin_inf());
    } else {
        for (int i = threadIdx.x; i < padded * DIM; i += blockDim.x)
            tile[i] = i < count * DIM ? g[i] : __builtin_inf();
    }
    return padded;
}

// Scan one staged tile for S query points held in registers.  All lanes of a
// wave read the same LDS address (broadcast), so one ds_read serves 64*S
// distance evaluations.  The inner loop keeps only a running minimum per chunk
// of NN_CHUNK targets (sub, sub, mul, mul, add, min per evaluation); the chunk
// holding the overall minimum is re-walked once at the end to recover the
// lowest index that attains it.
template <int DIM, int S>
__device__ __forceinline__ void nn_scan_tile(const double* tile, int padded, int base_j,
                                             const double (&p)[S][DIM], double (&best)[S],
                                             int (&bestj)[S]) {
    double tb[S];
    int tc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { tb[s] = best[s]; tc[s] = -1; }
    for (int c0 = 0; c0 < padded; c0 += NN_CHUNK) {
        double cm[S];
#pragma unroll
        for (int s = 0; s < S; ++s) cm[s] = __builtin_inf();
        // Two half-chunks: the 8 (DIM doubles) targets of a half are fetched into registers before
        // any of them is used, so their ds_reads are all in flight together instead of one
        // read -> wait -> 6*S flops at a time.
#pragma unroll
        for (int h = 0; h < NN_CHUNK; h += 8) {
            double q[8][DIM];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                for (int d = 0; d < DIM; ++d) q[jj][d] = tile[(c0 + h + jj) * DIM + d];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                for (int s = 0; s < S; ++s) cm[s] = fmin(cm[s], sqdist<DIM>(p[s], q[jj]));
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (cm[s] < tb[s]) { tb[s] = cm[s]; tc[s] = c0; }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (tc[s] >= 0) {
            int jf = NN_CHUNK - 1;
#pragma unroll
            for (int jj = NN_CHUNK - 1; jj >= 0; --jj)   // descending: ends on the lowest match
                if (sqdist<DIM>(p[s], tile + (tc[s] + jj) * DIM) == tb[s]) jf = jj;
            best[s] = tb[s];
            bestj[s] = base_j + tc[s] + jf;
        }
    }
}

}  // namespace icpmi
