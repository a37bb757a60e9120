// sort.hpp — workgroup-wide bitonic sort of (key, row) pairs held in LDS.
#pragma once
#include "common.hpp"

namespace icpmi {

// Sorts npad (power of two) pairs ascending by (key, row).  Rows are unique, so
// the order is total and equals a stable sort by key.  Pad with key = ~0,
// row = ~0.  Ends with a barrier.
__device__ __forceinline__ void bitonic_sort_pairs(uint64_t* keys, uint32_t* rows, int npad) {
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = threadIdx.x; p < npad / 2; p += blockDim.x) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int x = i | j;
                const uint64_t ka = keys[i], kb = keys[x];
                const uint32_t ra = rows[i], rb = rows[x];
                const bool gt = ka > kb || (ka == kb && ra > rb);
                const bool asc = (i & k) == 0;
                if (gt == asc) { keys[i] = kb; keys[x] = ka; rows[i] = rb; rows[x] = ra; }
            }
            __syncthreads();
        }
}

// The same network on ONE array of packed values (key << row_bits | row): a third (32-bit) or two thirds
// (64-bit) of the LDS traffic of the pair sort, which is what bounds it.  Pad with all ones.
template <typename T>
__device__ __forceinline__ void bitonic_sort_packed(T* v, int npad) {
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = threadIdx.x; p < npad / 2; p += blockDim.x) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int x = i | j;
                const T a = v[i], b = v[x];
                if ((a > b) == ((i & k) == 0)) { v[i] = b; v[x] = a; }
            }
            __syncthreads();
        }
}

// The same network with the elements in REGISTERS: thread t owns elements t*E .. t*E + E-1 (npad = E * blockDim.x).
// Strides below E are compare-exchanges between a thread's own registers, strides below 64 E exchange with a lane of
// the same wave (one cross-lane move per element, no index arithmetic, no barrier), only the few strides beyond that
// go through LDS.  The sorts of the voxel filter and of the prepare kernel are bound by their vector instructions,
// and this form needs about half of them (2 048 elements on 512 threads: 21 + 39 of the 66 stages never touch
// LDS).  v: packed values (key << row_bits | row), all ones = padding; lds: npad elements of scratch.
template <typename T, int E>
__device__ __forceinline__ void bitonic_sort_regs(T (&v)[E], T* lds) {
    const int npad = E * (int)blockDim.x;
    const int base = (int)threadIdx.x * E, lane = lane_id();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((e & j) == 0 && (e | j) < E) {
                        const bool asc = ((base + e) & k) == 0;
                        const T a = v[e], b = v[e | j];
                        const bool sw = (a > b) == asc;
                        v[e] = sw ? b : a;
                        v[e | j] = sw ? a : b;
                    }
            } else if (j < E * ICPMI_WAVE) {
                const int m = j / E;                                    // partner lane = lane ^ m
                const bool lower = (lane & m) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const T o = __shfl_xor(v[e], m, ICPMI_WAVE);
                    const bool asc = ((base + e) & k) == 0;
                    const bool keep_min = asc == lower;
                    const T lo = v[e] < o ? v[e] : o, hi = v[e] < o ? o : v[e];
                    v[e] = keep_min ? lo : hi;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) lds[base + e] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = base + e;
                    const T o = lds[i ^ j];
                    const bool keep_min = ((i & k) == 0) == ((i & j) == 0);
                    const T lo = v[e] < o ? v[e] : o, hi = v[e] < o ? o : v[e];
                    v[e] = keep_min ? lo : hi;
                }
                __syncthreads();
            }
        }
}

// Order-preserving map double -> uint64 (and back): a < b  <=>  enc(a) < enc(b).
__device__ __forceinline__ uint64_t f64_sortable(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unsortable(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

}  // namespace icpmi
