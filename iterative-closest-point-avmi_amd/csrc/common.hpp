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

// ── process-wide state (state.hip): options read once, side streams of the library ───────────────────
// option("NAME"): value of the ICPMI_NAME switch (environment at first use, icpmi_set_option later) or nullptr.
const char* option(const char* name);
// dyn_lds(kernel, bytes): the kernel's dynamic-LDS limit on the current device raised to at least `bytes` (state.hip: the
// attribute is set when a launch needs more than any before it, not on every launch)
hipError_t dyn_lds(const void* fn, size_t bytes);
constexpr int ICPMI_SIDE_STREAMS = 3;
struct Side {
    hipStream_t stream[ICPMI_SIDE_STREAMS] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr;
    hipEvent_t join[ICPMI_SIDE_STREAMS] = {nullptr, nullptr, nullptr};
    bool failed = false;
};
// Holds the library's side-stream lock for a whole fork / launch / join sequence; `side` is the current device's
// set (made on first use) or nullptr when side streams are off or could not be made.
struct SideLock {
    Side* side;
    SideLock();
    ~SideLock();
    SideLock(const SideLock&) = delete;
    SideLock& operator=(const SideLock&) = delete;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (ICPMI_WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ── cross-lane moves on the VALU (DPP), no LDS crossbar ─────────────────────
// dpp_ctrl: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141, row_mirror = 0x140 (rows are 16 lanes).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over each 16-lane row; afterwards every lane of a row holds its row's sum.
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<0xB1>(v);     // lane ^ 1
    v += dpp_f64<0x4E>(v);     // lane ^ 2
    v += dpp_f64<0x141>(v);    // mirror inside each half row: the other quad
    v += dpp_f64<0x140>(v);    // mirror inside the row: the other half row
    return v;
}

// Wave-wide sum, identical (wave-uniform) in every lane.  Fixed tree: quads,
// half rows, rows, then rows 0..3 left to right — reproducible run to run.
__device__ __forceinline__ double wave_sum(double v) {
    v = row_sum(v);
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
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

// Workgroup-wide sum of NV doubles per thread; the result is in every thread.
// scratch: NV4 * 16 doubles of LDS (NV4 = NV rounded up to a multiple of 4) owned
// by this call site, zero-initialised once by the caller (block_sum_init).
// ONE barrier per reduction: two consecutive uses of the same scratch must be
// separated by some other workgroup barrier (in the ICP loop the reductions
// alternate, each one's barrier protects the other's scratch).  MAXW <= 16.
// Wave partials (fixed DPP tree) are combined by a second fixed tree over the
// wave index, four values at a time (one value per 16-lane row): reproducible.
template <int NV, int MAXW>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
    static_assert(MAXW <= 16, "at most 16 waves per workgroup");
    constexpr int NV4 = (NV + 3) / 4 * 4;
    const int w = wave_id(), l = lane_id();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(v[i]);
        if (l == 0) scratch[i * 16 + w] = s;      // slots w >= #waves stay zero
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NV4 / 4; ++g) {
        const double x = row_sum(scratch[(g * 4 + (l >> 4)) * 16 + (l & 15)]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (g * 4 + q < NV) v[g * 4 + q] = readlane_f64(x, 16 * q);
    }
}

template <int NV>
constexpr int block_sum_doubles() { return (NV + 3) / 4 * 4 * 16; }

// zero a scratch area (all threads; caller adds the barrier)
__device__ __forceinline__ void block_sum_init(double* scratch, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) scratch[i] = 0.0;
}

}  // namespace icpmi
