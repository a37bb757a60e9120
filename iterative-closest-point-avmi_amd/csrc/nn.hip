// nn.hip — K1: batched exhaustive nearest-neighbour kernel.
// Replaces KDTree(target).query(source) of reference utilities/icp.py:35-46,179.
#include "nn.hpp"

namespace icpmi {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE_DOUBLES = 4096;   // 32 KiB of LDS per workgroup

template <int DIM>
constexpr int nn_tile_points() { return (NN_TILE_DOUBLES / DIM) / NN_CHUNK * NN_CHUNK; }

// grid = (source tiles, pairs).  A workgroup owns NN_THREADS*S consecutive
// source rows of one pair and streams that pair's whole target cloud through
// LDS.  Source reads, target staging and result writes are coalesced.
// the body for SE <= S rows per lane (a ragged last tile of a pair uses fewer)
template <int DIM, int SE>
__device__ __forceinline__ void nn_rows(const double* __restrict__ src, const double* __restrict__ tgt, int N, int M,
                                        int first, double* tile, int32_t* __restrict__ oi, double* __restrict__ od) {
    double p[SE][DIM], best[SE];
    int bestj[SE];
#pragma unroll
    for (int s = 0; s < SE; ++s) {
        const int n = first + s * NN_THREADS + threadIdx.x;
        const int nn = n < N ? n : N - 1;                      // clamp: tail lanes repeat the last row
#pragma unroll
        for (int d = 0; d < DIM; ++d) p[s][d] = src[(size_t)nn * DIM + d];
        best[s] = __builtin_inf();
        bestj[s] = 0;
    }
    constexpr int TP = nn_tile_points<DIM>();
    for (int t0 = 0; t0 < M; t0 += TP) {
        const int c = min(TP, M - t0);
        __syncthreads();
        const int padded = stage_targets<DIM>(tgt + (size_t)t0 * DIM, c, tile);
        __syncthreads();
        nn_scan_tile<DIM, SE>(tile, padded, t0, p, best, bestj);
    }
#pragma unroll
    for (int s = 0; s < SE; ++s) {
        const int n = first + s * NN_THREADS + threadIdx.x;
        if (n < N) {
            oi[n] = M > 0 ? bestj[s] : -1;
            od[n] = sqrt(best[s]);                             // IEEE sqrt, as KDTree returns
        }
    }
}

template <int DIM, int S>
__global__ __launch_bounds__(NN_THREADS) void nn_batch_kernel(
    const double* __restrict__ pts, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const int32_t* __restrict__ pair_src, const int32_t* __restrict__ pair_tgt,
    int32_t* __restrict__ out_idx, double* __restrict__ out_dist, int out_stride) {
    __shared__ __attribute__((aligned(16))) double tile[NN_TILE_DOUBLES];
    const int b = blockIdx.y;
    const int sc = pair_src[b], tc = pair_tgt[b];
    const int N = cnt ? cnt[sc] : off[sc + 1] - off[sc];
    const int M = cnt ? cnt[tc] : off[tc + 1] - off[tc];
    const int first = blockIdx.x * (NN_THREADS * S);
    if (first >= N) return;                                    // uniform per workgroup
    const double* src = pts + (size_t)off[sc] * DIM;
    const double* tgt = pts + (size_t)off[tc] * DIM;
    int32_t* oi = out_idx + (size_t)b * out_stride;
    double* od = out_dist + (size_t)b * out_stride;
    const int se = min(S, (N - first + NN_THREADS - 1) / NN_THREADS);   // rows per lane this tile really has
    if (S >= 4 && se == 4) nn_rows<DIM, 4>(src, tgt, N, M, first, tile, oi, od);
    else if (S >= 3 && se == 3) nn_rows<DIM, 3>(src, tgt, N, M, first, tile, oi, od);
    else if (S >= 2 && se == 2) nn_rows<DIM, 2>(src, tgt, N, M, first, tile, oi, od);
    else nn_rows<DIM, 1>(src, tgt, N, M, first, tile, oi, od);
}

template <int DIM, int S>
static int launch_nn(const double* pts, const int32_t* off, const int32_t* cnt, const int32_t* ps,
                     const int32_t* pt, int n_pairs, int max_src_n, int32_t* out_idx, double* out_dist,
                     int out_stride, hipStream_t st) {
    // the pair index is the grid's y: at most 65 535 per launch, so a larger batch goes down in slices
    for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
        const int np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
        dim3 grid((max_src_n + NN_THREADS * S - 1) / (NN_THREADS * S), np);
        nn_batch_kernel<DIM, S><<<grid, NN_THREADS, 0, st>>>(pts, off, cnt, ps + p0, pt + p0, out_idx + (size_t)p0 * out_stride,
                                                             out_dist + (size_t)p0 * out_stride, out_stride);
        ICPMI_LAUNCH_CHECK();
    }
    return ICPMI_OK;
}

}  // namespace icpmi

extern "C" int icpmi_nn_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                              const int32_t* pair_src, const int32_t* pair_tgt, int32_t n_pairs,
                              int32_t max_src_n, int32_t dim, int32_t* out_idx, double* out_dist,
                              int32_t out_stride, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !pair_src || !pair_tgt || !out_idx || !out_dist) return ICPMI_ERR_ARG;
    if (n_pairs < 0 || max_src_n < 0 || out_stride < max_src_n || (dim != 2 && dim != 3)) return ICPMI_ERR_ARG;
    if (n_pairs == 0 || max_src_n == 0) return ICPMI_OK;
    hipStream_t st = (hipStream_t)stream;
    // Rows per thread: enough workgroups to cover the chip first, then register
    // blocking (one LDS broadcast read feeds S evaluations per lane).
    // The loop is FP64-VALU bound at every S (one broadcast ds_read_b128 per S evaluations is far
    // from the LDS limit already at S = 2), so tiles stay small: more, equal-sized workgroups
    // balance better over 256 CUs than fewer large ones with a ragged last tile.  S = 4 only when
    // even that leaves tens of workgroups per CU.
    const long wg4 = (long)n_pairs * ((max_src_n + NN_THREADS * 4 - 1) / (NN_THREADS * 4));
    const long wg2 = (long)n_pairs * ((max_src_n + NN_THREADS * 2 - 1) / (NN_THREADS * 2));
    const int S = wg4 >= 16384 ? 4 : (wg2 >= 512 ? 2 : 1);
#define ICPMI_NN_GO(D, SS) \
    return launch_nn<D, SS>(pts, off_dev, cnt_dev, pair_src, pair_tgt, n_pairs, max_src_n, out_idx, out_dist, out_stride, st)
    if (dim == 2) {
        if (S == 4) ICPMI_NN_GO(2, 4);
        if (S == 2) ICPMI_NN_GO(2, 2);
        ICPMI_NN_GO(2, 1);
    } else {
        if (S == 4) ICPMI_NN_GO(3, 4);
        if (S == 2) ICPMI_NN_GO(3, 2);
        ICPMI_NN_GO(3, 1);
    }
#undef ICPMI_NN_GO
}
