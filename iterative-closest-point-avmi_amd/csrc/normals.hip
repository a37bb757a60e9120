// normals.hip — K4: estimate_normals_2d (reference utilities/icp.py:51-76).
//
// For every point: the k+1 nearest points of its own cloud (itself included,
// k clamped to n-1, icp.py:61,66), their 2x2 covariance (np.cov, ddof=1,
// icp.py:71), the eigenvector of the smaller eigenvalue (eigh column 0,
// icp.py:72-73), normalised (icp.py:74-75).  The sign is arbitrary, as it is in
// the reference.
//
// Search: one workgroup per cloud sorts the points by x in LDS (bitonic on
// order-preserving float64 keys), then every query sweeps outwards from its own
// sorted position, always taking the side that is nearer in x, and stops as
// soon as (dx)^2 exceeds its current (k+1)-th best squared distance — every
// remaining point is farther than that in x alone.  Exact (same neighbour sets
// as an exhaustive search; ties by lowest row), ~window instead of ~n distance
// evaluations per query.  Distances are float64 direct differences like the
// oracle's.
#include "linalg.hpp"
#include "sort.hpp"

namespace icpmi {

constexpr int NRM_THREADS = 512;
constexpr int NRM_LDS_MAX = 8192;     // points sortable in LDS (96 KiB)

// Sorted list of the KK best (d2, row), ascending, all in registers.
template <int KK>
struct TopK {
    double d[KK];
    int j[KK];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < KK; ++i) { d[i] = __builtin_inf(); j[i] = 0x7fffffff; }
    }
    __device__ __forceinline__ void push(double s, int row) {
        if (!(s < d[KK - 1] || (s == d[KK - 1] && row < j[KK - 1]))) return;
        d[KK - 1] = s; j[KK - 1] = row;
#pragma unroll
        for (int i = KK - 1; i > 0; --i) {
            const bool lt = d[i] < d[i - 1] || (d[i] == d[i - 1] && j[i] < j[i - 1]);
            const double td = lt ? d[i - 1] : d[i];
            const int tj = lt ? j[i - 1] : j[i];
            d[i - 1] = lt ? d[i] : d[i - 1];
            j[i - 1] = lt ? j[i] : j[i - 1];
            d[i] = td; j[i] = tj;
        }
    }
    __device__ __forceinline__ double kth(int k) const {     // d[k] with a static index chain
        double v = d[0];
#pragma unroll
        for (int i = 1; i < KK; ++i) v = (i == k) ? d[i] : v;
        return v;
    }
};

template <int KK>
__device__ __forceinline__ void normals_sweep(const uint64_t* keys, const uint32_t* rows, int M, int kk,
                                              const double* __restrict__ P, double* __restrict__ out) {
    for (int s = threadIdx.x; s < M; s += blockDim.x) {
        const int self = (int)rows[s];
        const double px = P[2 * self], py = P[2 * self + 1];
        TopK<KK> top;
        top.init();
        top.push(0.0, self);
        int lo = s - 1, hi = s + 1;
        for (;;) {
            const double dl = lo >= 0 ? px - f64_unsortable(keys[lo]) : __builtin_inf();
            const double dh = hi < M ? px - f64_unsortable(keys[hi]) : __builtin_inf();
            const double gl = dl * dl, gh = dh * dh;
            const bool left = gl <= gh;
            const double g = left ? gl : gh;
            if ((lo < 0 && hi >= M) || !(g <= top.kth(kk - 1))) break;
            const int c = left ? lo-- : hi++;
            const int row = (int)rows[c];
            const double dx = px - P[2 * row], dy = py - P[2 * row + 1];
            double d2 = 0.0;
            d2 += dx * dx;
            d2 += dy * dy;
            top.push(d2, row);
        }
        // np.cov of the kk neighbours (rows of `top` in ascending distance order)
        double mx = 0.0, my = 0.0;
#pragma unroll
        for (int i = 0; i < KK; ++i)
            if (i < kk) { mx += P[2 * top.j[i]]; my += P[2 * top.j[i] + 1]; }
        mx /= (double)kk; my /= (double)kk;
        double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
        for (int i = 0; i < KK; ++i)
            if (i < kk) {
                const double dx = P[2 * top.j[i]] - mx, dy = P[2 * top.j[i] + 1] - my;
                sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
            }
        double vx = 1.0, vy = 0.0;
        if (kk > 1) {
            const double den = (double)(kk - 1);
            smallest_evec_2x2(sxx / den, sxy / den, syy / den, vx, vy);
        }
        double nn = sqrt(vx * vx + vy * vy);
        nn = nn < 1e-10 ? 1e-10 : nn;                       // icp.py:74-75
        out[2 * self] = vx / nn;
        out[2 * self + 1] = vy / nn;
    }
}

template <typename KP, typename RP>
__device__ __forceinline__ void normals_cloud(KP keys, RP rows, int npad, int M, int k,
                                              const double* __restrict__ P, double* __restrict__ O) {
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        keys[i] = i < M ? f64_sortable(P[2 * i]) : ~0ull;
        rows[i] = i < M ? (uint32_t)i : 0xffffffffu;
    }
    __syncthreads();
    bitonic_sort_pairs(keys, rows, npad);
    const int kc = min(k, M - 1);              // icp.py:61
    const int kk = kc + 1;                     // self included, icp.py:66
    if (kk <= 8) normals_sweep<8>(keys, rows, M, kk, P, O);
    else if (kk <= 16) normals_sweep<16>(keys, rows, M, kk, P, O);
    else normals_sweep<32>(keys, rows, M, kk, P, O);
}

__global__ __launch_bounds__(NRM_THREADS) void normals_kernel(
    const double* __restrict__ pts, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const int32_t* __restrict__ cloud_ids, int k, double* __restrict__ out_normals,
    uint64_t* gkeys, uint32_t* grows, int lds_points) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int c = cloud_ids ? cloud_ids[blockIdx.x] : blockIdx.x;
    const int M = cnt ? cnt[c] : off[c + 1] - off[c];
    if (M <= 0) return;
    const double* P = pts + (size_t)off[c] * 2;
    double* O = out_normals + (size_t)off[c] * 2;
    int npad = 64;
    while (npad < M) npad <<= 1;
    // two instantiations so that each sees one address space (LDS or global)
    if (npad <= lds_points)
        normals_cloud(reinterpret_cast<uint64_t*>(dyn), reinterpret_cast<uint32_t*>(dyn + (size_t)npad * sizeof(uint64_t)), npad, M, k, P, O);
    else
        normals_cloud(gkeys + 2 * (size_t)off[c], grows + 2 * (size_t)off[c], npad, M, k, P, O);
}

}  // namespace icpmi

extern "C" size_t icpmi_normals_workspace_bytes(int32_t total_rows, int32_t max_n) {
    if (max_n <= icpmi::NRM_LDS_MAX) return 256;
    return 256 + (size_t)total_rows * 2 * 12 + 256;
}

extern "C" int icpmi_normals_2d_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                                      const int32_t* cloud_ids, int32_t n_sel, int32_t total_rows,
                                      int32_t max_n, int32_t k, double* out_normals,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !out_normals || n_sel < 0 || total_rows < 0 || max_n < 0 || k < 0) return ICPMI_ERR_ARG;
    if (k > 31) return ICPMI_ERR_UNSUPPORTED;
    if (n_sel == 0 || max_n == 0) return ICPMI_OK;
    int npad = 64;
    while (npad < max_n) npad <<= 1;
    const int lds_points = npad < NRM_LDS_MAX ? npad : NRM_LDS_MAX;
    uint64_t* gkeys = nullptr;
    uint32_t* grows = nullptr;
    if (npad > NRM_LDS_MAX) {
        if (!workspace || workspace_bytes < icpmi_normals_workspace_bytes(total_rows, max_n)) return ICPMI_ERR_WORKSPACE;
        gkeys = (uint64_t*)workspace;
        grows = (uint32_t*)((unsigned char*)workspace + (size_t)total_rows * 2 * 8);
    }
    const size_t lds = (size_t)lds_points * 12;
    if (dyn_lds((const void*)normals_kernel, lds) != hipSuccess) return ICPMI_ERR_HIP;
    normals_kernel<<<n_sel, NRM_THREADS, lds, (hipStream_t)stream>>>(pts, off_dev, cnt_dev, cloud_ids, k, out_normals, gkeys, grows, lds_points);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
