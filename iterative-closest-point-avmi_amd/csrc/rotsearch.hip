// rotsearch.hip — scoring pass of the correlative rotation search.
//
// Reference utilities/features.py:213-218 (`_score`, called ~270 times per
// rotation_search, features.py:221-232) and the same scoring inside
// slam.py:138-143: rotate the (centred) source by an angle, shift it, query the
// nearest target point of every row, return the mean squared distance.  Here all
// angles of a sweep are one launch: one workgroup per angle, the target staged
// in LDS, the K1 exhaustive scan of nn.hpp per row (the clouds are a few hundred
// points after the coarse voxel filter), `sqrt(d2)` squared again as
// `np.mean(dists ** 2)` does, fixed-tree workgroup sum.
#include "nn.hpp"

namespace icpmi {

constexpr int RS_THREADS = 256;
constexpr int RS_TILE_DOUBLES = 4096;                         // 32 KiB: 2048 target points per tile
constexpr int RS_TILE_POINTS = (RS_TILE_DOUBLES / 2) / NN_CHUNK * NN_CHUNK;

__global__ __launch_bounds__(RS_THREADS) void rotation_scores_kernel(
    const double* __restrict__ src_c, int n, const double* __restrict__ tgt, int m,
    const double* __restrict__ cs, double shift_x, double shift_y, double* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) double tile[RS_TILE_DOUBLES];
    __shared__ double red[block_sum_doubles<1>()];
    block_sum_init(red, block_sum_doubles<1>());
    const int a = blockIdx.x;
    const double c = cs[2 * a], s = cs[2 * a + 1];            // R = [[c, -s], [s, c]], features.py:214-215
    double acc[1] = {0.0};
    for (int first = 0; first < n; first += RS_THREADS) {     // uniform trip count
        const int i = first + threadIdx.x;
        const int ii = i < n ? i : n - 1;
        const double x = src_c[2 * ii], y = src_c[2 * ii + 1];
        double p[1][2] = {{(x * c + y * -s) + shift_x, (x * s + y * c) + shift_y}};   // src_c @ R.T + shift, features.py:216
        double best[1] = {__builtin_inf()};
        int bestj[1] = {0};
        for (int t0 = 0; t0 < m; t0 += RS_TILE_POINTS) {
            const int cnt = min(RS_TILE_POINTS, m - t0);
            __syncthreads();
            const int padded = stage_targets<2>(tgt + (size_t)t0 * 2, cnt, tile);
            __syncthreads();
            nn_scan_tile<2, 1>(tile, padded, t0, p, best, bestj);
        }
        if (i < n) {
            const double d = sqrt(best[0]);                   // KDTree distance ...
            acc[0] += d * d;                                  // ... squared, features.py:218
        }
    }
    __syncthreads();
    block_sum<1, RS_THREADS / ICPMI_WAVE>(acc, red);
    if (threadIdx.x == 0) scores[a] = acc[0] / (double)n;
}

}  // namespace icpmi

extern "C" int icpmi_rotation_scores(const double* src_c, int32_t n_src, const double* tgt, int32_t n_tgt,
                                     const double* cos_sin, int32_t n_angles, double shift_x, double shift_y,
                                     double* out_scores, void* stream) {
    using namespace icpmi;
    if (!src_c || !tgt || !cos_sin || !out_scores || n_src <= 0 || n_tgt <= 0 || n_angles < 0) return ICPMI_ERR_ARG;
    if (n_angles == 0) return ICPMI_OK;
    rotation_scores_kernel<<<n_angles, RS_THREADS, 0, (hipStream_t)stream>>>(src_c, n_src, tgt, n_tgt, cos_sin, shift_x, shift_y, out_scores);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
