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

// ── the register network, every stage fixed at compile time (round 4) ────────────────────────────────
// bitonic_sort_regs above walks (k, j) in run-time loops: every stage recomputes partner lanes, directions and LDS
// addresses and goes through ds_bpermute — measured (round 4): the sort is half of the voxel filter (0.70 of 1.41 ms for
// 32 768 scans) at ~18 vector instructions per element and stage.  Here the network is the FLIP form of the bitonic sort
// (level k: one stage against the mirror image inside each k-block, i <-> i ^ (k - 1), then strides k/4 .. 1 as i <-> i ^ j;
// every compare-exchange puts the minimum at the lower index: no direction bit), unrolled by templates, so that each stage
// is: its partner through the cheapest cross-lane move that reaches it — a DPP quad permutation or row (half) mirror /
// rotation, ds_swizzle inside 32 lanes, ds_bpermute across the halves of the wave (addresses computed once) —, one
// v_min, one v_max, one select on a lane mask that is a constant of the stage.  Strides inside a thread are plain
// min / max pairs; the strides that cross waves (6 of the 66 stages of 2 048 elements on 512 threads) go through LDS as
// before.  Same result as any sorting network on distinct elements (the packed values are distinct: the row is part of them).
template <int CTRL>
__device__ __forceinline__ uint32_t sort_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int PATTERN>
__device__ __forceinline__ uint32_t sort_swz(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, PATTERN); }

// value of lane (lane ^ M) for M = 1, 2, 4, 8, 16, 32; a32: byte address of lane ^ 32 for ds_bpermute
template <int M>
__device__ __forceinline__ uint32_t sort_xor_lane(uint32_t v, int a32) {
    if constexpr (M == 1) return sort_dpp<0xB1>(v);                   // quad_perm [1, 0, 3, 2]
    else if constexpr (M == 2) return sort_dpp<0x4E>(v);              // quad_perm [2, 3, 0, 1]
    else if constexpr (M == 4) return sort_swz<0x101F>(v);            // bit-mask mode: and 0x1f, xor 4
    else if constexpr (M == 8) return sort_dpp<0x128>(v);             // row_ror:8 — half a row of 16 round
    else if constexpr (M == 16) return sort_swz<0x401F>(v);           // xor 16 (inside 32 lanes)
    else return (uint32_t)__builtin_amdgcn_ds_bpermute(a32, (int)v);
}
// value of lane (lane ^ (L - 1)): the mirror image inside blocks of L = 2 .. 64 lanes; a63: byte address of lane 63 - lane
template <int L>
__device__ __forceinline__ uint32_t sort_flip_lane(uint32_t v, int a63) {
    if constexpr (L == 2) return sort_dpp<0xB1>(v);
    else if constexpr (L == 4) return sort_dpp<0x1B>(v);              // quad_perm [3, 2, 1, 0]
    else if constexpr (L == 8) return sort_dpp<0x141>(v);             // row_half_mirror
    else if constexpr (L == 16) return sort_dpp<0x140>(v);            // row_mirror
    else if constexpr (L == 32) return sort_swz<0x7C1F>(v);           // xor 31
    else return (uint32_t)__builtin_amdgcn_ds_bpermute(a63, (int)v);
}
template <int M> __device__ __forceinline__ uint64_t sort_xor_lane(uint64_t v, int a32) {
    return ((uint64_t)sort_xor_lane<M>((uint32_t)(v >> 32), a32) << 32) | sort_xor_lane<M>((uint32_t)v, a32);
}
template <int L> __device__ __forceinline__ uint64_t sort_flip_lane(uint64_t v, int a63) {
    return ((uint64_t)sort_flip_lane<L>((uint32_t)(v >> 32), a63) << 32) | sort_flip_lane<L>((uint32_t)v, a63);
}

template <typename T, int E>
struct SortNetRegs {
    T (&v)[E];
    T* lds;
    int base, lane, a32, a63;
    __device__ __forceinline__ SortNetRegs(T (&v_)[E], T* lds_) : v(v_), lds(lds_) {
        base = (int)threadIdx.x * E;
        lane = lane_id();
        a32 = (lane ^ 32) << 2;
        a63 = (63 - lane) << 2;
    }
    static __device__ __forceinline__ T mn(T a, T b) { return a < b ? a : b; }
    static __device__ __forceinline__ T mx(T a, T b) { return a < b ? b : a; }
    // one stage: FLIP (partner i ^ (S - 1), S = block size in elements) or stride S (partner i ^ S)
    template <bool FLIP, int S>
    __device__ __forceinline__ void stage() {
        if constexpr (FLIP ? S <= E : S < E) {                         // inside the thread
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int o = FLIP ? (e ^ (S - 1)) : (e ^ S);
                if (o > e) { const T a = v[e], b = v[o]; v[e] = mn(a, b); v[o] = mx(a, b); }
            }
        } else if constexpr (FLIP ? S <= E * ICPMI_WAVE : S < E * ICPMI_WAVE) {     // inside the wave
            constexpr int LB = FLIP ? S / E : S / E;                   // block (flip) or stride (xor) in lanes
            const bool lower = FLIP ? (lane & (LB / 2)) == 0 : (lane & LB) == 0;
            T o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if constexpr (FLIP) o[e] = sort_flip_lane<LB>(v[E - 1 - e], a63);   // the mirror image: the partner's elements in reverse
                else o[e] = sort_xor_lane<LB>(v[e], a32);
            }
            // the lower lane of a pair keeps the minimum (the lane mask is a constant of the stage)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if constexpr (sizeof(T) == 4) v[e] = lower ? mn(v[e], o[e]) : mx(v[e], o[e]);      // v_min, v_max, select: measured faster for 32 bits
                else v[e] = ((o[e] < v[e]) == lower) ? o[e] : v[e];                              // 64 bits: one compare, two selects
            }
        } else {                                                       // across waves: through LDS
#pragma unroll
            for (int e = 0; e < E; ++e) lds[base + e] = v[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = base + e;
                const int x = FLIP ? (i ^ (S - 1)) : (i ^ S);
                const T o = lds[x];
                if constexpr (sizeof(T) == 4) v[e] = x > i ? mn(v[e], o) : mx(v[e], o);
                else v[e] = ((o < v[e]) == (x > i)) ? o : v[e];
            }
            __syncthreads();
        }
    }
    template <int J>
    __device__ __forceinline__ void strides() {
        if constexpr (J >= 1) { stage<false, J>(); strides<J / 2>(); }
    }
    template <int K, int NPAD>
    __device__ __forceinline__ void levels() {
        if constexpr (K <= NPAD) { stage<true, K>(); strides<K / 4>(); levels<K * 2, NPAD>(); }
    }
};

// Sorts NPAD = E * THREADS packed values ascending; thread t owns elements t*E .. t*E + E-1 on entry and on return.
// lds: NPAD elements of scratch; ends with a barrier after its last LDS stage (callers that reuse `lds` at once add theirs).
template <typename T, int E, int THREADS>
__device__ __forceinline__ void bitonic_sort_regs_fixed(T (&v)[E], T* lds) {
    SortNetRegs<T, E> net(v, lds);
    net.template levels<2, E * THREADS>();
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
