// common.hpp — shared device helpers for libicpmi.so (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/icpmi.h"

#define ICPMI_WAVE 64

#define ICPMI_LAUNCH_CHECK()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return ICPMI_ERR_HIP;           \
    } while (0)

namespace icpmi {

__device__ __forceinline__ int lane_id() { return threadIdx.x & (ICPMI_WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Wave-wide sum: after the call every lane holds the total (butterfly on the
// cross-lane network; 6 steps for 64 lanes).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ICPMI_WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ICPMI_WAVE);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, ICPMI_WAVE));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, ICPMI_WAVE));
    return v;
}

// Workgroup-wide sum of NV doubles per thread.  `scratch` holds NV * MAXW
// doubles of LDS (MAXW = waves per workgroup).  Result in every thread.
// Two barriers; the wave partials are added in wave order, so the result does
// not depend on timing (bitwise reproducible run to run).
template <int NV, int MAXW>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
    const int w = wave_id(), l = lane_id();
    const int nw = (blockDim.x + ICPMI_WAVE - 1) / ICPMI_WAVE;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();                      // scratch may still be read from a previous use
    if (l == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * MAXW + w] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int k = 0; k < nw; ++k) s += scratch[i * MAXW + k];
        v[i] = s;
    }
}

}  // namespace icpmi
