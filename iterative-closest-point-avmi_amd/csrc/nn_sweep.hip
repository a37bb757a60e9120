// nn_sweep.hip — nearest neighbour on prepared (axis-sorted) targets.
//
// Same answers as the exhaustive kernel of nn.hip (index, float64 distance,
// lowest row on ties) for KDTree.query of reference utilities/icp.py:179, by the
// exact sweep of sweep.hpp: the sorted copy of the target that
// icpmi_prepare_targets wrote is staged in LDS (20 B per point, coalesced reads),
// every lane owns one query, finds its place by binary search and walks
// outwards until the projection gap exceeds the second-best distance.  Also
// returns that second-best squared distance (what the fused ICP kernel uses to
// keep matches between iterations).
#include "sweep.hpp"

namespace icpmi {

constexpr int NNS_THREADS = 256;

__global__ __launch_bounds__(NNS_THREADS) void nn_sweep_kernel(
    const double* __restrict__ pts, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const int32_t* __restrict__ pair_src, const int32_t* __restrict__ pair_tgt,
    const double2* __restrict__ g_sxy, const int32_t* __restrict__ g_sorig, const int32_t* __restrict__ g_dir,
    int32_t* __restrict__ out_idx, double* __restrict__ out_dist, double* __restrict__ out_second2,
    int out_stride, int lds_points) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    double2* sxy = reinterpret_cast<double2*>(dyn);
    int32_t* sorig = reinterpret_cast<int32_t*>(dyn + (size_t)lds_points * 16);
    const int b = blockIdx.y;
    const int sc = pair_src[b], tc = pair_tgt[b];
    const int N = cnt ? cnt[sc] : off[sc + 1] - off[sc];
    const int M = cnt ? cnt[tc] : off[tc + 1] - off[tc];
    const int first = blockIdx.x * NNS_THREADS;
    if (first >= N) return;                                    // uniform per workgroup
    const int dir = g_dir[tc];
    const int n = first + threadIdx.x;
    int32_t* oi = out_idx + (size_t)b * out_stride;
    double* od = out_dist + (size_t)b * out_stride;
    if (M <= 0 || M > lds_points || dir < 0) {                 // nothing prepared for this target
        if (n < N) { oi[n] = -1; od[n] = __builtin_inf(); if (out_second2) out_second2[(size_t)b * out_stride + n] = __builtin_inf(); }
        return;
    }
    const double2* gx = g_sxy + off[tc];
    const int32_t* go = g_sorig + off[tc];
    for (int i = threadIdx.x; i < M; i += NNS_THREADS) { sxy[i] = gx[i]; sorig[i] = go[i]; }
    __syncthreads();
    if (n >= N) return;
    const double2 c_lo = sxy[0], c_hi = sxy[M - 1];
    const double uabs = fmax(fabs(proj(dir, c_lo.x, c_lo.y)), fabs(proj(dir, c_hi.x, c_hi.y)));
    const double* q = pts + ((size_t)off[sc] + n) * 2;
    double d2, second;
    const int pos = sweep_nn2(sxy, sorig, M, dir, uabs, q[0], q[1], -1, d2, second);
    oi[n] = sorig[pos];
    od[n] = sqrt(d2);                                          // IEEE sqrt, as KDTree returns
    if (out_second2) out_second2[(size_t)b * out_stride + n] = second;
}

}  // namespace icpmi

extern "C" int icpmi_nn_prepared_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                                       const void* prepared, const int32_t* pair_src, const int32_t* pair_tgt,
                                       int32_t n_pairs, int32_t max_src_n, int32_t max_tgt_n, int32_t total_rows,
                                       int32_t* out_idx, double* out_dist, double* out_second_sq,
                                       int32_t out_stride, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !prepared || !pair_src || !pair_tgt || !out_idx || !out_dist) return ICPMI_ERR_ARG;
    if (n_pairs < 0 || max_src_n < 0 || max_tgt_n < 0 || out_stride < max_src_n) return ICPMI_ERR_ARG;
    if (max_tgt_n > 4096) return ICPMI_ERR_UNSUPPORTED;
    if (n_pairs == 0 || max_src_n == 0) return ICPMI_OK;
    const unsigned char* b = (const unsigned char*)prepared;
    const double2* g_sxy = (const double2*)b;
    const int32_t* g_sorig = (const int32_t*)(b + (size_t)total_rows * 32);
    const int32_t* g_dir = (const int32_t*)(b + (size_t)total_rows * 40);
    int cap = 64;
    while (cap < max_tgt_n) cap <<= 1;
    const size_t lds = (size_t)cap * 20;
    if (dyn_lds((const void*)nn_sweep_kernel, lds) != hipSuccess) return ICPMI_ERR_HIP;
    for (int p0 = 0; p0 < n_pairs; p0 += 65535) {              // the pair index is the grid's y: slices of at most 65 535 pairs
        const int np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
        dim3 grid((max_src_n + NNS_THREADS - 1) / NNS_THREADS, np);
        nn_sweep_kernel<<<grid, NNS_THREADS, lds, (hipStream_t)stream>>>(
            pts, off_dev, cnt_dev, pair_src + p0, pair_tgt + p0, g_sxy, g_sorig, g_dir, out_idx + (size_t)p0 * out_stride,
            out_dist + (size_t)p0 * out_stride, out_second_sq ? out_second_sq + (size_t)p0 * out_stride : nullptr, out_stride, cap);
        ICPMI_LAUNCH_CHECK();
    }
    return ICPMI_OK;
}
