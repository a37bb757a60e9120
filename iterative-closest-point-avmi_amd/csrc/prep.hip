// prep.hip — prepare target clouds for the sweep search, and K4 normals on top.
//
// Per target cloud (one workgroup): pick the sort axis, sort the points along
// it in LDS, publish the sorted copy (points, original rows, axis) for the
// fused ICP kernel, and — for point_to_line — estimate_normals_2d (reference
// utilities/icp.py:51-76) by an outward sweep from every point's own sorted
// position, which is an exact k-NN search (see sweep.hpp).
#include "linalg.hpp"
#include "sort.hpp"
#include "sweep.hpp"

namespace icpmi {

constexpr int PREP_THREADS = 512;
constexpr int PREP_MAXW = PREP_THREADS / ICPMI_WAVE;
constexpr int PREP_MAX_POINTS = 4096;   // sorted copy (20 B/pt) + sort scratch (12 B/pt) stay in LDS
constexpr int PREP_BINS = 64;

// k best (d2, row, sorted position), ascending by (d2, row).  Every index is a
// compile-time constant (template recursion), so the lists stay in registers.
template <int KK>
struct TopKP {
    double d[KK];
    int j[KK];
    int p[KK];
    template <int I>
    __device__ __forceinline__ void init_from() {
        if constexpr (I < KK) { d[I] = __builtin_inf(); j[I] = 0x7fffffff; p[I] = 0; init_from<I + 1>(); }
    }
    __device__ __forceinline__ void init() { init_from<0>(); }
    template <int I>
    __device__ __forceinline__ void bubble() {
        if constexpr (I > 0) {
            // stop as soon as the new entry is in place: candidates arrive roughly by
            // increasing distance, so most insertions move one or two slots
            if (d[I] < d[I - 1] || (d[I] == d[I - 1] && j[I] < j[I - 1])) {
                const double td = d[I - 1]; const int tj = j[I - 1], tp = p[I - 1];
                d[I - 1] = d[I]; j[I - 1] = j[I]; p[I - 1] = p[I];
                d[I] = td; j[I] = tj; p[I] = tp;
                bubble<I - 1>();
            }
        }
    }
    __device__ __forceinline__ bool push(double s, int row, int pos) {
        if (s < d[KK - 1] || (s == d[KK - 1] && row < j[KK - 1])) {
            d[KK - 1] = s; j[KK - 1] = row; p[KK - 1] = pos;
            bubble<KK - 1>();
            return true;
        }
        return false;
    }
    template <int I>
    __device__ __forceinline__ double kth_from(int k, double v) const {
        if constexpr (I < KK) return kth_from<I + 1>(k, I == k ? d[I] : v);
        else return v;
    }
    __device__ __forceinline__ double kth(int k) const { return kth_from<1>(k, d[0]); }
    // sum of f(sxy[p[i]]) over i < kk, in list order
    template <int I, typename F>
    __device__ __forceinline__ void for_first(int kk, F&& f) const {
        if constexpr (I < KK) {
            if (I < kk) f(p[I]);
            for_first<I + 1>(kk, f);
        }
    }
};

template <int KK>
__device__ __forceinline__ void prep_normals(const double2* sxy, const int32_t* sorig, int M, int s_begin, int s_end, int dir, int kk,
                                             double2* __restrict__ out_sorted, double* __restrict__ out_rows) {
    const double2 c_lo = sxy[0], c_hi = sxy[M - 1];
    const double uabs = fmax(fabs(proj(dir, c_lo.x, c_lo.y)), fabs(proj(dir, c_hi.x, c_hi.y)));
    for (int s = s_begin + threadIdx.x; s < s_end; s += blockDim.x) {
        const double2 q = sxy[s];
        const double uq = proj(dir, q.x, q.y);
        TopKP<KK> top;
        top.init();
        top.push(0.0, sorig[s], s);
        // window half-width from the current kk-th best distance (inf until kk neighbours are known);
        // refreshed only when the list changes
        double thr = __builtin_inf();
        int lo = s - 1, hi = s + 1;
        while (lo >= 0 || hi < M) {
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const bool right = side == 0;
                if (right ? hi < M : lo >= 0) {
                    const int i = right ? hi : lo;
                    const double2 c = sxy[i];
                    const double du = right ? proj(dir, c.x, c.y) - uq : uq - proj(dir, c.x, c.y);
                    if (du > thr) { if (right) hi = M; else lo = -1; }
                    else {
                        const double dx = q.x - c.x, dy = q.y - c.y;
                        double d2 = 0.0;
                        d2 += dx * dx;
                        d2 += dy * dy;
                        if (top.push(d2, sorig[i], i)) thr = prune_width(dir, kk == KK ? top.d[KK - 1] : top.kth(kk - 1), uq, uabs);
                        if (right) ++hi; else --lo;
                    }
                }
            }
        }
        // np.cov over the kk neighbours, summed in ascending (distance, row) order
        double mx = 0.0, my = 0.0;
        top.template for_first<0>(kk, [&](int pos) { const double2 c = sxy[pos]; mx += c.x; my += c.y; });
        mx /= (double)kk; my /= (double)kk;
        double sxx = 0.0, sxy_ = 0.0, syy = 0.0;
        top.template for_first<0>(kk, [&](int pos) {
            const double2 c = sxy[pos];
            const double dx = c.x - mx, dy = c.y - my;
            sxx += dx * dx; sxy_ += dx * dy; syy += dy * dy;
        });
        double vx = 1.0, vy = 0.0;
        if (kk > 1) {
            const double den = (double)(kk - 1);                 // np.cov ddof = 1
            smallest_evec_2x2(sxx / den, sxy_ / den, syy / den, vx, vy);
        }
        double nn = sqrt(vx * vx + vy * vy);
        nn = nn < 1e-10 ? 1e-10 : nn;                            // icp.py:74-75
        const double2 n2 = make_double2(vx / nn, vy / nn);
        out_sorted[s] = n2;
        if (out_rows) { const int row = sorig[s]; out_rows[2 * row] = n2.x; out_rows[2 * row + 1] = n2.y; }
    }
}

// KK = capacity of the per-query neighbour list (0: no normals)
template <int KK>
__global__ __launch_bounds__(PREP_THREADS) void prep_targets_kernel(
    const double* __restrict__ pts, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const int32_t* __restrict__ cloud_ids, int k, double2* __restrict__ g_sxy, double2* __restrict__ g_snrm,
    int32_t* __restrict__ g_sorig, int32_t* __restrict__ g_dir, double* __restrict__ out_normals, int lds_points,
    int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ double dsc[8 * PREP_MAXW];
    __shared__ int hist[4 * PREP_BINS];
    // `split` workgroups share one cloud: each repeats the (cheap) axis choice and sort, then takes its
    // slice of the normal queries — small batches would otherwise leave a cloud's k-NN sweeps to one CU
    const int ci = blockIdx.x / split, part = blockIdx.x % split;
    const int c = cloud_ids ? cloud_ids[ci] : ci;
    const int M = cnt ? cnt[c] : off[c + 1] - off[c];
    if (M <= 0 || M > lds_points) { if (threadIdx.x == 0) g_dir[c] = -1; return; }
    const double* P = pts + (size_t)off[c] * 2;
    int npad = 64;
    while (npad < M) npad <<= 1;
    double2* sxy = reinterpret_cast<double2*>(dyn);                                   // lds_points * 16 B
    int32_t* sorig = reinterpret_cast<int32_t*>(dyn + (size_t)lds_points * 16);       // lds_points * 4 B
    uint64_t* keys = reinterpret_cast<uint64_t*>(dyn + (size_t)lds_points * 20);      // npad * 8 B
    uint32_t* rows = reinterpret_cast<uint32_t*>(dyn + (size_t)lds_points * 20 + (size_t)npad * 8);

    // ── range of the four projections ───────────────────────────────────────
    double mn[4], mx[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { mn[d] = __builtin_inf(); mx[d] = -__builtin_inf(); }
    for (int i = threadIdx.x; i < M; i += PREP_THREADS) {
        const double x = P[2 * i], y = P[2 * i + 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) { const double u = proj(d, x, y); mn[d] = fmin(mn[d], u); mx[d] = fmax(mx[d], u); }
    }
    const int w = wave_id(), l = lane_id();
#pragma unroll
    for (int d = 0; d < 4; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    if (l == 0)
#pragma unroll
        for (int d = 0; d < 4; ++d) { dsc[d * PREP_MAXW + w] = mn[d]; dsc[(4 + d) * PREP_MAXW + w] = mx[d]; }
    for (int i = threadIdx.x; i < 4 * PREP_BINS; i += PREP_THREADS) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        double a = __builtin_inf(), b = -__builtin_inf();
        for (int q = 0; q < PREP_MAXW; ++q) { a = fmin(a, dsc[d * PREP_MAXW + q]); b = fmax(b, dsc[(4 + d) * PREP_MAXW + q]); }
        mn[d] = a; mx[d] = b;
    }
    // ── expected window size per axis: sum of squared bin counts / bin width ─
    for (int i = threadIdx.x; i < M; i += PREP_THREADS) {
        const double x = P[2 * i], y = P[2 * i + 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const double r = mx[d] - mn[d];
            int b = r > 0.0 ? (int)((proj(d, x, y) - mn[d]) / r * PREP_BINS) : 0;
            b = b < 0 ? 0 : (b >= PREP_BINS ? PREP_BINS - 1 : b);
            atomicAdd(&hist[d * PREP_BINS + b], 1);
        }
    }
    __syncthreads();
    int dir = 0;
    double bestc = __builtin_inf();
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        double s = 0.0;
        for (int b = 0; b < PREP_BINS; ++b) { const double cn = (double)hist[d * PREP_BINS + b]; s += cn * cn; }
        const double r = mx[d] - mn[d];
        const double cost = r > 0.0 ? (d < 2 ? 1.0 : 1.4142135623730951) * s / r : __builtin_inf();
        if (cost < bestc) { bestc = cost; dir = d; }       // integer histogram: identical in every thread
    }
    // ── sort along the chosen axis ──────────────────────────────────────────
    for (int i = threadIdx.x; i < npad; i += PREP_THREADS) {
        keys[i] = i < M ? f64_sortable(proj(dir, P[2 * i], P[2 * i + 1])) : ~0ull;
        rows[i] = i < M ? (uint32_t)i : 0xffffffffu;
    }
    __syncthreads();
    bitonic_sort_pairs(keys, rows, npad);
    double2* o_sxy = g_sxy + off[c];
    int32_t* o_sorig = g_sorig + off[c];
    for (int i = threadIdx.x; i < M; i += PREP_THREADS) {
        const int row = (int)rows[i];
        const double2 p = make_double2(P[2 * row], P[2 * row + 1]);
        sxy[i] = p; sorig[i] = row;
        if (part == 0) { o_sxy[i] = p; o_sorig[i] = row; }
    }
    if (threadIdx.x == 0 && part == 0) g_dir[c] = dir;
    __syncthreads();
    if constexpr (KK > 0) {
        const int kc = min(k, M - 1);              // icp.py:61
        const int kk = kc + 1;                     // self included, icp.py:66
        double2* o_snrm = g_snrm + off[c];
        double* o_rows = out_normals ? out_normals + (size_t)off[c] * 2 : nullptr;
        const int per = (M + split - 1) / split;
        prep_normals<KK>(sxy, sorig, M, min(M, part * per), min(M, (part + 1) * per), dir, kk, o_snrm, o_rows);
    }
}

}  // namespace icpmi

// layout of a prepared-target buffer: sorted xy | sorted normals | sorted->row map | axis per cloud
extern "C" size_t icpmi_prepared_bytes(int32_t total_rows, int32_t n_clouds) {
    if (total_rows < 0 || n_clouds < 0) return 0;
    return (size_t)total_rows * (16 + 16 + 4) + (size_t)n_clouds * 4 + 256;
}

extern "C" int icpmi_prepare_targets(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                                     const int32_t* cloud_ids, int32_t n_sel, int32_t n_clouds,
                                     int32_t total_rows, int32_t max_n, int32_t normal_k,
                                     double* out_normals, void* prepared, size_t prepared_bytes, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !prepared || n_sel < 0 || n_clouds < 0 || total_rows < 0 || max_n < 0) return ICPMI_ERR_ARG;
    if (normal_k > 31) return ICPMI_ERR_UNSUPPORTED;
    if (prepared_bytes < icpmi_prepared_bytes(total_rows, n_clouds)) return ICPMI_ERR_WORKSPACE;
    if (max_n > PREP_MAX_POINTS) return ICPMI_ERR_UNSUPPORTED;
    if (n_sel == 0 || max_n == 0) return ICPMI_OK;
    unsigned char* b = (unsigned char*)prepared;
    double2* g_sxy = (double2*)b;
    double2* g_snrm = (double2*)(b + (size_t)total_rows * 16);
    int32_t* g_sorig = (int32_t*)(b + (size_t)total_rows * 32);
    int32_t* g_dir = (int32_t*)(b + (size_t)total_rows * 36);
    int npad = 64;
    while (npad < max_n) npad <<= 1;
    const size_t lds = (size_t)npad * 20 + (size_t)npad * 12;      // lds_points = npad
    int split = 512 / n_sel;                 // enough workgroups for every CU when the batch is small
    split = split < 1 ? 1 : (split > 8 ? 8 : split);
#define ICPMI_PREP_GO(KKV)                                                                                              \
    do {                                                                                                                \
        if (hipFuncSetAttribute((const void*)prep_targets_kernel<KKV>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)lds) != hipSuccess) return ICPMI_ERR_HIP;                                          \
        prep_targets_kernel<KKV><<<n_sel * (KKV > 0 ? split : 1), PREP_THREADS, lds, (hipStream_t)stream>>>(           \
            pts, off_dev, cnt_dev, cloud_ids, normal_k, g_sxy, g_snrm, g_sorig, g_dir, out_normals, npad,               \
            KKV > 0 ? split : 1);                                                                                       \
    } while (0)
    // list capacity = k + 1 exactly for the usual k (5, 10 = reference default, 12 = config.yaml), else the next size up
    if (normal_k < 0) ICPMI_PREP_GO(0);
    else if (normal_k + 1 <= 6) ICPMI_PREP_GO(6);
    else if (normal_k + 1 <= 8) ICPMI_PREP_GO(8);
    else if (normal_k + 1 <= 11) ICPMI_PREP_GO(11);
    else if (normal_k + 1 <= 13) ICPMI_PREP_GO(13);
    else if (normal_k + 1 <= 16) ICPMI_PREP_GO(16);
    else ICPMI_PREP_GO(32);
#undef ICPMI_PREP_GO
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
