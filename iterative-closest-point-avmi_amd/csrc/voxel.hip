// voxel.hip — K5: voxel_downsample (reference utilities/icp.py:117-129).
//
// Reference semantics reproduced exactly:
//   min_bound = points.min(0)                      per cloud, per call
//   key       = floor((p - min_bound) / voxel)     float64 IEEE divide, per axis
//   rows      = np.unique(keys, axis=0)            lexicographic order of the keys
//   mean      = bincount(weights=p) / bincount     sum taken in INPUT order
// A stable sort by the linearised key ((k0*E1 + k1)*E2 + k2, E = key extent per
// axis) followed by an in-order per-voxel sum gives bit-identical output.
//
// Clouds of <= 8192 points (every scan) take a single-workgroup path: keys and
// row ids live in LDS, bitonic sort on (key, row) pairs, one launch for a whole
// batch of clouds.  Larger clouds (the ~82k-point rolling submap of
// slam.py:103-108) use rocPRIM's radix sort on the same keys.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "sort.hpp"

#ifndef VOX_REGS
#define VOX_REGS 1              // 0: the round-3 path (sorted words unpacked to LDS, voxel_finish) for the usual shape too
#endif
#ifndef VOX_STOP_AFTER
#define VOX_STOP_AFTER 0        // diagnostic variants only: leave the small kernel after a phase (1 bounds, 2 keys, 3 sort)
#endif

namespace icpmi {

constexpr int VOX_THREADS = 1024;   // upper bound; the launcher picks 512 for large batches (loops use blockDim.x)
constexpr int VOX_MAXW = VOX_THREADS / ICPMI_WAVE;
constexpr int VOX_SMALL_MAX = 8192;

__device__ __forceinline__ uint64_t vox_key(const double* p, int dim, const double* mn, const double* ext, double voxel) {
    // floor((p - min) / voxel).astype(int), linearised so that integer order ==
    // lexicographic order of the per-axis keys
    uint64_t k = 0;
    for (int d = 0; d < dim; ++d) {
        const int64_t kd = (int64_t)floor((p[d] - mn[d]) / voxel);
        k = k * (uint64_t)ext[d] + (uint64_t)kd;
    }
    return k;
}

// min / max over the rows of one cloud, result in every thread
template <int DIM>
__device__ __forceinline__ void cloud_bounds(const double* pts, int n, double (&mn)[3], double (&mx)[3], double* scratch) {
    for (int d = 0; d < 3; ++d) { mn[d] = __builtin_inf(); mx[d] = -__builtin_inf(); }
    for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const double v = pts[(size_t)i * DIM + d];
            mn[d] = fmin(mn[d], v);
            mx[d] = fmax(mx[d], v);
        }
    const int w = wave_id(), l = lane_id(), nw = blockDim.x / ICPMI_WAVE;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    __syncthreads();
    if (l == 0)
#pragma unroll
        for (int d = 0; d < DIM; ++d) { scratch[d * VOX_MAXW + w] = mn[d]; scratch[(3 + d) * VOX_MAXW + w] = mx[d]; }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        double a = __builtin_inf(), b = -__builtin_inf();
        for (int k = 0; k < nw; ++k) { a = fmin(a, scratch[d * VOX_MAXW + k]); b = fmax(b, scratch[(3 + d) * VOX_MAXW + k]); }
        mn[d] = a; mx[d] = b;
    }
}

// key extents; returns false when the linear key would not fit in 63 bits
template <int DIM>
__device__ __forceinline__ bool key_extents(const double (&mn)[3], const double (&mx)[3], double voxel, double (&ext)[3]) {
    double prod = 1.0;
    bool ok = true;
    for (int d = 0; d < 3; ++d) ext[d] = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const double e = floor((mx[d] - mn[d]) / voxel) + 1.0;
        if (!(e >= 1.0) || !(e < 9.0e18)) ok = false;
        ext[d] = e;
        prod *= e;
    }
    return ok && prod < 9.0e18;
}

// Exclusive prefix sum of one int per thread over the workgroup; `total` gets
// the grand total.  scratch: VOX_MAXW ints of LDS.
__device__ __forceinline__ int block_exscan(int v, int* scratch, int& total) {
    const int l = lane_id(), w = wave_id(), nw = blockDim.x / ICPMI_WAVE;
    int inc = v;
#pragma unroll
    for (int o = 1; o < ICPMI_WAVE; o <<= 1) {
        const int t = __shfl_up(inc, o, ICPMI_WAVE);
        if (l >= o) inc += t;
    }
    __syncthreads();
    if (l == ICPMI_WAVE - 1) scratch[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < nw; ++k) { const int s = scratch[k]; if (k < w) base += s; tot += s; }
    total = tot;
    return base + inc - v;
}

// After the sort: voxel boundaries, voxel ids, in-order sums, means.
// keys/rows are sorted by (key, row); thread t owns sorted positions
// [t*ipt, (t+1)*ipt).  Works on LDS or global arrays (generic pointers).
template <int DIM>
__device__ __forceinline__ void voxel_finish(const uint64_t* keys, const uint32_t* rows, int n,
                                             const double* __restrict__ pts, double* __restrict__ out,
                                             int32_t* out_cnt, int* iscratch) {
    const int ipt = (n + blockDim.x - 1) / blockDim.x;
    const int lo = min(n, (int)threadIdx.x * ipt), hi = min(n, lo + ipt);
    int heads = 0;
    for (int r = lo; r < hi; ++r) heads += (r == 0 || keys[r] != keys[r - 1]);
    int total;
    int vid = block_exscan(heads, iscratch, total);
    for (int r = lo; r < hi; ++r) {
        if (!(r == 0 || keys[r] != keys[r - 1])) continue;
        const uint64_t k = keys[r];
        double s[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) s[d] = 0.0;
        int q = r;
        for (; q < n && keys[q] == k; ++q) {                 // rows ascending == input order
            const double* p = pts + (size_t)rows[q] * DIM;
#pragma unroll
            for (int d = 0; d < DIM; ++d) s[d] += p[d];
        }
        const double c = (double)(q - r);
#pragma unroll
        for (int d = 0; d < DIM; ++d) out[(size_t)vid * DIM + d] = s[d] / c;
        ++vid;
    }
    if (threadIdx.x == 0) *out_cnt = total;
}

// The usual shape — a 2 048-beam scan on 512 or 1 024 threads, keys and rows in one 32-bit word — from the sort to the
// means on REGISTERS (round 4; before: the sorted words were unpacked to 12 B per element of LDS, every voxel head re-read
// its neighbours' keys from there and fetched its members' points one dependent load after the other: a third of the
// kernel).  After the sort thread t holds the sorted elements t E .. t E + E - 1: it fetches THEIR points at once (E
// independent loads), finds the voxel heads among them from its own registers (the element before its first through
// LDS), and a head adds up its run out of the thread's own registers; only a run that continues past the thread's last
// element goes on through LDS and global memory.  Same sums in the same order: rows ascending inside a voxel.
template <int DIM>
__device__ __forceinline__ uint32_t vox_key32(const double* p, const double (&mn)[3], const uint32_t (&ext)[3], double voxel) {
    uint32_t k = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) k = k * ext[d] + (uint32_t)(int)floor((p[d] - mn[d]) / voxel);   // every factor below 2^31 here
    return k;
}

// Returns false (nothing written) when key and row do not fit one 32-bit word: the caller's general path takes over.
template <int DIM, int E, int THREADS>
__device__ __forceinline__ bool voxel_small_regs(const double* __restrict__ P, double* __restrict__ O, int n, double voxel,
                                                 uint32_t* sorted, double* dscratch, int* iscratch, int32_t* out_cnt) {
    constexpr uint32_t PAD = 0xffffffffu;
    constexpr int NW = THREADS / ICPMI_WAVE;
    const int tid = (int)threadIdx.x;
    // the thread's rows e THREADS + t (coalesced; any start order sorts the same) stay in registers from the bounds to the keys
    double raw[E][DIM];
    double mn[3], mx[3], extd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn[d] = __builtin_inf(); mx[d] = -__builtin_inf(); }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + tid;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            raw[e][d] = i < n ? P[(size_t)i * DIM + d] : 0.0;
            if (i < n) { mn[d] = fmin(mn[d], raw[e][d]); mx[d] = fmax(mx[d], raw[e][d]); }
        }
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    if (lane_id() == 0)
#pragma unroll
        for (int d = 0; d < DIM; ++d) { dscratch[d * VOX_MAXW + wave_id()] = mn[d]; dscratch[(3 + d) * VOX_MAXW + wave_id()] = mx[d]; }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        double a = __builtin_inf(), b = -__builtin_inf();
#pragma unroll
        for (int k = 0; k < NW; ++k) { a = fmin(a, dscratch[d * VOX_MAXW + k]); b = fmax(b, dscratch[(3 + d) * VOX_MAXW + k]); }
        mn[d] = a; mx[d] = b;
    }
    if (!key_extents<DIM>(mn, mx, voxel, extd)) { if (tid == 0) *out_cnt = -1; return true; }
    int row_bits = 6;
    while ((1 << row_bits) < E * THREADS) ++row_bits;
    double cells = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) cells *= extd[d];
    if (!(cells * (double)(E * THREADS) < 4.0e9)) return false;     // (uniform: every thread holds the same bounds)
    const uint32_t ext[3] = {(uint32_t)extd[0], (uint32_t)extd[1], (uint32_t)extd[2]};
    const uint32_t row_mask = (1u << row_bits) - 1u;
    uint32_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + tid;
        v[e] = i < n ? (vox_key32<DIM>(raw[e], mn, ext, voxel) << row_bits) | (uint32_t)i : PAD;
    }
    bitonic_sort_regs_fixed<uint32_t, E, THREADS>(v, sorted);
    __syncthreads();                                                 // the last LDS stage's reads are done
    double pt[E][DIM];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        sorted[tid * E + e] = v[e];
        const double* p = P + (size_t)(v[e] == PAD ? 0u : (v[e] & row_mask)) * DIM;
#pragma unroll
        for (int d = 0; d < DIM; ++d) pt[e][d] = p[d];
    }
    if (tid == 0) sorted[E * THREADS] = PAD;                         // one slot behind the last element
    __syncthreads();
    const uint32_t before = tid > 0 ? sorted[tid * E - 1] : PAD;
    bool head[E];
    int nh = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t prev = e > 0 ? v[e - 1] : before;
        head[e] = v[e] != PAD && ((tid == 0 && e == 0) || (prev >> row_bits) != (v[e] >> row_bits));
        nh += head[e] ? 1 : 0;
    }
    int total;
    int vid = block_exscan(nh, iscratch, total);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (!head[e]) continue;
        const uint32_t key = v[e] >> row_bits;
        double s[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) s[d] = 0.0 + pt[e][d];         // np.bincount starts from zero
        int c = 1;
        bool open = true;                                            // the run has not met another key yet
#pragma unroll
        for (int f = e + 1; f < E; ++f) {
            open = open && v[f] != PAD && (v[f] >> row_bits) == key;
            if (open) {
#pragma unroll
                for (int d = 0; d < DIM; ++d) s[d] += pt[f][d];
                ++c;
            }
        }
        if (open) {                                                  // past the thread's own elements (the slot behind the array ends it)
            for (int q = tid * E + E; ; ++q) {
                const uint32_t w = sorted[q];
                if (w == PAD || (w >> row_bits) != key) break;
                const double* p = P + (size_t)(w & row_mask) * DIM;
#pragma unroll
                for (int d = 0; d < DIM; ++d) s[d] += p[d];
                ++c;
            }
        }
        const double cd = (double)c;
#pragma unroll
        for (int d = 0; d < DIM; ++d) O[(size_t)vid * DIM + d] = s[d] / cd;
        ++vid;
    }
    if (tid == 0) *out_cnt = total;
    return true;
}

// ── small path: one workgroup per cloud, everything in LDS ─────────────────
template <int DIM>
__global__ __launch_bounds__(VOX_THREADS) void voxel_small_kernel(
    const double* __restrict__ pts, const int32_t* __restrict__ off, double voxel,
    double* __restrict__ out_pts, int32_t* __restrict__ out_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ double dscratch[6 * VOX_MAXW];
    __shared__ int iscratch[VOX_MAXW];
    const int c = blockIdx.x;
    const int n = off[c + 1] - off[c];
    if (n > VOX_SMALL_MAX) return;                            // routed to the big path by the host
    if (n <= 0) { if (threadIdx.x == 0) out_cnt[c] = 0; return; }
    const double* P = pts + (size_t)off[c] * DIM;
    double* O = out_pts + (size_t)off[c] * DIM;
    int npad = 64;
    while (npad < n) npad <<= 1;
    uint64_t* keys = reinterpret_cast<uint64_t*>(dyn);
    uint32_t* rows = reinterpret_cast<uint32_t*>(dyn + (size_t)npad * sizeof(uint64_t));

    if (VOX_REGS && VOX_STOP_AFTER == 0 && (npad == 4 * (int)blockDim.x || npad == 2 * (int)blockDim.x) && (blockDim.x == 512 || blockDim.x == 1024)) {
        uint32_t* sorted = reinterpret_cast<uint32_t*>(dyn);             // npad + 1 words
        bool done;
        if (blockDim.x == 512) {
            if (npad == 2048) done = voxel_small_regs<DIM, 4, 512>(P, O, n, voxel, sorted, dscratch, iscratch, out_cnt + c);
            else done = voxel_small_regs<DIM, 2, 512>(P, O, n, voxel, sorted, dscratch, iscratch, out_cnt + c);
        } else {
            if (npad == 4096) done = voxel_small_regs<DIM, 4, 1024>(P, O, n, voxel, sorted, dscratch, iscratch, out_cnt + c);
            else done = voxel_small_regs<DIM, 2, 1024>(P, O, n, voxel, sorted, dscratch, iscratch, out_cnt + c);
        }
        if (done) return;
        __syncthreads();                                                 // dscratch is written again below
    }
    double mn[3], mx[3], ext[3];
    cloud_bounds<DIM>(P, n, mn, mx, dscratch);
#if VOX_STOP_AFTER == 1
    if (threadIdx.x == 0) out_cnt[c] = (int)mn[0]; return;
#endif
    if (!key_extents<DIM>(mn, mx, voxel, ext)) { if (threadIdx.x == 0) out_cnt[c] = -1; return; }
    // (key, row) sorts as ONE integer key << row_bits | row when that fits: the sort is bound by LDS traffic,
    // and a 2 048-beam scan in a room needs ~18 + 11 bits
    int row_bits = 6;
    while ((1 << row_bits) < npad) ++row_bits;
    double cells = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) cells *= ext[d];                      // keys are below this product
    const double packed_range = cells * (double)npad;                   // packed values are below this
    if (packed_range < 4.0e9) {
        uint32_t* pk = rows;                                             // sorted in the row array, unpacked in place
        if (npad == 4 * (int)blockDim.x || npad == 2 * (int)blockDim.x) {
            // the usual shape (a 2 048-beam scan on 512 or 1 024 threads): the network runs on registers (sort.hpp)
            auto packed = [&](int i) {
                return i < n ? (uint32_t)(vox_key(P + (size_t)i * DIM, DIM, mn, ext, voxel) << row_bits) | (uint32_t)i : 0xffffffffu;
            };
            auto unpack = [&](int i, uint32_t v) {
                keys[i] = v == 0xffffffffu ? ~0ull : (uint64_t)(v >> row_bits);
                rows[i] = v == 0xffffffffu ? 0xffffffffu : (v & ((1u << row_bits) - 1u));
            };
            uint32_t* scratch = reinterpret_cast<uint32_t*>(keys);          // the key array is free until the unpacking
            if (npad == 4 * (int)blockDim.x) {
                uint32_t v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = packed((int)threadIdx.x * 4 + e);
#if VOX_STOP_AFTER == 2
                if (threadIdx.x == 0) out_cnt[c] = (int)(v[0] + v[1] + v[2] + v[3]); return;
#endif
                if (blockDim.x == 512) bitonic_sort_regs_fixed<uint32_t, 4, 512>(v, scratch);
                else bitonic_sort_regs_fixed<uint32_t, 4, 1024>(v, scratch);
#if VOX_STOP_AFTER == 3
                if (threadIdx.x == 0) out_cnt[c] = (int)(v[0] + v[1] + v[2] + v[3]); return;
#endif
                __syncthreads();                                            // scratch reads of the last LDS stage are done
#pragma unroll
                for (int e = 0; e < 4; ++e) unpack((int)threadIdx.x * 4 + e, v[e]);
            } else {
                uint32_t v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) v[e] = packed((int)threadIdx.x * 2 + e);
                if (blockDim.x == 512) bitonic_sort_regs_fixed<uint32_t, 2, 512>(v, scratch);
                else bitonic_sort_regs_fixed<uint32_t, 2, 1024>(v, scratch);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 2; ++e) unpack((int)threadIdx.x * 2 + e, v[e]);
            }
            __syncthreads();
        } else {
        for (int i = threadIdx.x; i < npad; i += blockDim.x)
            pk[i] = i < n ? (uint32_t)(vox_key(P + (size_t)i * DIM, DIM, mn, ext, voxel) << row_bits) | (uint32_t)i : 0xffffffffu;
        __syncthreads();
        bitonic_sort_packed<uint32_t>(pk, npad);
        for (int i = threadIdx.x; i < npad; i += blockDim.x) {
            const uint32_t v = pk[i];
            keys[i] = v == 0xffffffffu ? ~0ull : (uint64_t)(v >> row_bits);
            rows[i] = v == 0xffffffffu ? 0xffffffffu : (v & ((1u << row_bits) - 1u));
        }
        __syncthreads();
        }
    } else if (packed_range < 9.0e18) {
        for (int i = threadIdx.x; i < npad; i += blockDim.x)
            keys[i] = i < n ? (vox_key(P + (size_t)i * DIM, DIM, mn, ext, voxel) << row_bits) | (uint64_t)i : ~0ull;
        __syncthreads();
        bitonic_sort_packed<uint64_t>(keys, npad);
        for (int i = threadIdx.x; i < npad; i += blockDim.x) {
            const uint64_t v = keys[i];
            keys[i] = v == ~0ull ? ~0ull : (v >> row_bits);
            rows[i] = v == ~0ull ? 0xffffffffu : (uint32_t)(v & ((1ull << row_bits) - 1ull));
        }
        __syncthreads();
    } else {
        for (int i = threadIdx.x; i < npad; i += blockDim.x) {
            keys[i] = i < n ? vox_key(P + (size_t)i * DIM, DIM, mn, ext, voxel) : ~0ull;
            rows[i] = i < n ? (uint32_t)i : 0xffffffffu;
        }
        __syncthreads();
        bitonic_sort_pairs(keys, rows, npad);     // == stable sort by key (rows are unique)
    }
    voxel_finish<DIM>(keys, rows, n, P, O, out_cnt + c, iscratch);
}

// ── big path: every step spread over the chip ───────────────────────────────
struct VoxBounds {          // order-preserving uint64 images of the per-axis min / max (atomicMin / atomicMax)
    unsigned long long mn[3];
    unsigned long long mx[3];
};

__global__ void vox_bounds_init_kernel(VoxBounds* b) {
    if (threadIdx.x < 3) { b->mn[threadIdx.x] = ~0ull; b->mx[threadIdx.x] = 0ull; }
}

template <int DIM>
__global__ __launch_bounds__(256) void vox_bounds_kernel(const double* __restrict__ P, int n, VoxBounds* b) {
    double mn[DIM], mx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) { mn[d] = __builtin_inf(); mx[d] = -__builtin_inf(); }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const double v = P[(size_t)i * DIM + d];
            mn[d] = fmin(mn[d], v);
            mx[d] = fmax(mx[d], v);
        }
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const double a = wave_min(mn[d]), c = wave_max(mx[d]);
        if (lane_id() == 0) {
            atomicMin(&b->mn[d], (unsigned long long)f64_sortable(a));
            atomicMax(&b->mx[d], (unsigned long long)f64_sortable(c));
        }
    }
}

template <int DIM>
__device__ __forceinline__ bool vox_header(const VoxBounds* __restrict__ b, double voxel, double (&mn)[3], double (&ext)[3]) {
    double mx[3];
    for (int d = 0; d < 3; ++d) { mn[d] = 0.0; mx[d] = 0.0; }
#pragma unroll
    for (int d = 0; d < DIM; ++d) { mn[d] = f64_unsortable(b->mn[d]); mx[d] = f64_unsortable(b->mx[d]); }
    return key_extents<DIM>(mn, mx, voxel, ext);
}

template <int DIM>
__global__ void vox_keys_kernel(const double* __restrict__ P, int n, double voxel, const VoxBounds* __restrict__ b,
                                uint64_t* __restrict__ keys, uint32_t* __restrict__ rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double mn[3], ext[3];
    const bool ok = vox_header<DIM>(b, voxel, mn, ext);
    keys[i] = ok ? vox_key(P + (size_t)i * DIM, DIM, mn, ext, voxel) : 0ull;
    rows[i] = (uint32_t)i;
}

// 1 where a new voxel starts in the sorted order
__global__ void vox_heads_kernel(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ heads) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) heads[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// one thread per voxel head: in-order sum over its run of equal keys, mean, voxel id from the scan
template <int DIM>
__global__ void vox_means_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ rows,
                                 const uint32_t* __restrict__ heads, const uint32_t* __restrict__ vid, int n, double voxel,
                                 const VoxBounds* __restrict__ b, const double* __restrict__ P, double* __restrict__ O,
                                 int32_t* __restrict__ out_cnt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double mn[3], ext[3];
    const bool ok = vox_header<DIM>(b, voxel, mn, ext);
    if (r == n - 1) *out_cnt = ok ? (int32_t)(vid[r] + heads[r]) : -1;
    if (!ok || !heads[r]) return;
    const uint64_t k = keys[r];
    double s[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) s[d] = 0.0;
    int q = r;
    for (; q < n && keys[q] == k; ++q) {                     // rows ascending == input order
        const double* p = P + (size_t)rows[q] * DIM;
#pragma unroll
        for (int d = 0; d < DIM; ++d) s[d] += p[d];
    }
    const double c = (double)(q - r);
#pragma unroll
    for (int d = 0; d < DIM; ++d) O[(size_t)vid[r] * DIM + d] = s[d] / c;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t radix_temp_bytes(int n) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0, false);
    return bytes;
}

static size_t scan_temp_bytes(int n) {
    size_t bytes = 0;
    uint32_t* v = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, v, v, 0u, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)0, false);
    return bytes;
}

static size_t voxel_big_bytes(int n) {
    const size_t t1 = radix_temp_bytes(n), t2 = scan_temp_bytes(n);
    return 256 + 2 * align256((size_t)n * 8) + 4 * align256((size_t)n * 4) + align256(t1 > t2 ? t1 : t2) + 256;
}

template <int DIM>
static int voxel_big(const double* P, int n, double voxel, double* O, int32_t* out_cnt, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < voxel_big_bytes(n)) return ICPMI_ERR_WORKSPACE;
    unsigned char* base = (unsigned char*)ws;
    size_t o = 0;
    VoxBounds* h = (VoxBounds*)(base + o); o += 256;
    uint64_t* k0 = (uint64_t*)(base + o); o += align256((size_t)n * 8);
    uint64_t* k1 = (uint64_t*)(base + o); o += align256((size_t)n * 8);
    uint32_t* r0 = (uint32_t*)(base + o); o += align256((size_t)n * 4);
    uint32_t* r1 = (uint32_t*)(base + o); o += align256((size_t)n * 4);
    uint32_t* heads = (uint32_t*)(base + o); o += align256((size_t)n * 4);
    uint32_t* vid = (uint32_t*)(base + o); o += align256((size_t)n * 4);
    size_t tb = radix_temp_bytes(n), sb = scan_temp_bytes(n);
    const int blocks = (n + 255) / 256;
    vox_bounds_init_kernel<<<1, 64, 0, st>>>(h);
    // few workgroups: every wave ends with 2*DIM atomics on the same words, and those serialise
    vox_bounds_kernel<DIM><<<blocks < 64 ? blocks : 64, 256, 0, st>>>(P, n, h);
    vox_keys_kernel<DIM><<<blocks, 256, 0, st>>>(P, n, voxel, h, k0, r0);
    if (rocprim::radix_sort_pairs(base + o, tb, k0, k1, r0, r1, (size_t)n, 0, 64, st, false) != hipSuccess) return ICPMI_ERR_HIP;
    vox_heads_kernel<<<blocks, 256, 0, st>>>(k1, n, heads);
    if (rocprim::exclusive_scan(base + o, sb, heads, vid, 0u, (size_t)n, rocprim::plus<uint32_t>(), st, false) != hipSuccess) return ICPMI_ERR_HIP;
    vox_means_kernel<DIM><<<blocks, 256, 0, st>>>(k1, r1, heads, vid, n, voxel, h, P, O, out_cnt);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

}  // namespace icpmi

extern "C" size_t icpmi_voxel_workspace_bytes(int32_t max_n) {
    using namespace icpmi;
    if (max_n <= VOX_SMALL_MAX) return 256;
    return voxel_big_bytes(max_n);
}

extern "C" int icpmi_voxel_downsample_batch(const double* pts, const int32_t* off_dev, const int32_t* off_host,
                                            int32_t n_clouds, int32_t dim, double voxel_size,
                                            double* out_pts, int32_t* out_cnt, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !off_host || !out_pts || !out_cnt) return ICPMI_ERR_ARG;
    if (n_clouds < 0 || (dim != 2 && dim != 3) || !(voxel_size > 0.0)) return ICPMI_ERR_ARG;
    if (n_clouds == 0) return ICPMI_OK;
    hipStream_t st = (hipStream_t)stream;
    int max_small = 0, n_big = 0;
    for (int c = 0; c < n_clouds; ++c) {
        const int n = off_host[c + 1] - off_host[c];
        if (n < 0) return ICPMI_ERR_ARG;
        if (n <= VOX_SMALL_MAX) max_small = n > max_small ? n : max_small; else ++n_big;
    }
    if (n_big < n_clouds) {
        int npad = 64;
        while (npad < max_small) npad <<= 1;
        const size_t lds = (size_t)npad * 12;
        // The kernel is a chain of latency-bound phases (two passes over the points, 66 sort stages, a gather): with
        // many clouds, four 512-thread workgroups per CU overlap them better than two of 1 024 (2.17 -> 1.55 ms for
        // 32 768 clouds); a lone cloud finishes sooner with 1 024 threads.
        const int vox_threads = n_clouds > 256 ? 512 : VOX_THREADS;
        if (dim == 2) {
            if (dyn_lds((const void*)voxel_small_kernel<2>, lds) != hipSuccess) return ICPMI_ERR_HIP;
            voxel_small_kernel<2><<<n_clouds, vox_threads, lds, st>>>(pts, off_dev, voxel_size, out_pts, out_cnt);
        } else {
            if (dyn_lds((const void*)voxel_small_kernel<3>, lds) != hipSuccess) return ICPMI_ERR_HIP;
            voxel_small_kernel<3><<<n_clouds, vox_threads, lds, st>>>(pts, off_dev, voxel_size, out_pts, out_cnt);
        }
        ICPMI_LAUNCH_CHECK();
    }
    for (int c = 0; c < n_clouds && n_big > 0; ++c) {
        const int n = off_host[c + 1] - off_host[c];
        if (n <= VOX_SMALL_MAX) continue;
        if (!workspace) return ICPMI_ERR_WORKSPACE;
        const double* P = pts + (size_t)off_host[c] * dim;
        double* O = out_pts + (size_t)off_host[c] * dim;
        const int rc = dim == 2 ? voxel_big<2>(P, n, voxel_size, O, out_cnt + c, workspace, workspace_bytes, st)
                                : voxel_big<3>(P, n, voxel_size, O, out_cnt + c, workspace, workspace_bytes, st);
        if (rc != ICPMI_OK) return rc;
    }
    return ICPMI_OK;
}
