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


// ── the whole search on the device (icpmi_rotation_search) ───────────────────────────────────────────
// What the drop-in `rotation_search` used to do with five host round trips (two voxel filters, their means, the
// coarse sweep, its arg-min, the fine sweep) as one chain of launches; only a 12-double record returns.
// The angle grids stay the caller's: cos / sin of every coarse angle and of every fine grid that can follow
// (one row per coarse winner) are computed with the reference's own NumPy calls and cached on the device, so
// the chosen angle — hence R and t — is the reference's bit for bit.
constexpr int RSREC_NS = 0, RSREC_NT = 1, RSREC_MUS = 2, RSREC_MUT = 4, RSREC_K = 6, RSREC_CSCORE = 7, RSREC_NF = 8,
              RSREC_J = 9, RSREC_FSCORE = 10, RSREC_DOUBLES = 12;

__global__ void rs_offsets_kernel(int32_t* off, int n_src, int n_tgt) {
    off[0] = 0; off[1] = n_src; off[2] = n_src + n_tgt;
}

// np.mean(cloud, axis=0) of a C-contiguous (n, 2) array adds the rows one after the other (the reduction runs
// along the slow axis: no pairwise summation) and divides by n: one lane per (cloud, column).
// Two waves, one per cloud: the wave copies 256 rows at a time into LDS with coalesced loads (the chain of
// dependent adds then never waits for HBM) and its lanes 0 and 1 add the two columns in row order.
constexpr int RS_MEAN_ROWS = 256;
__global__ __launch_bounds__(2 * ICPMI_WAVE) void rs_means_kernel(const double* __restrict__ vox, const int32_t* __restrict__ off,
                                                                 const int32_t* __restrict__ cnt, int centred, double shift_x,
                                                                 double shift_y, double* __restrict__ rec) {
    __shared__ double2 rows[2][RS_MEAN_ROWS];
    const int c = wave_id(), lane = lane_id(), n = cnt[c];
    const double2* p = reinterpret_cast<const double2*>(vox + (size_t)off[c] * 2);
    double s = 0.0;
    for (int i0 = 0; i0 < n; i0 += RS_MEAN_ROWS) {                       // wave-uniform trip count
        const int m = min(RS_MEAN_ROWS, n - i0);
        for (int i = lane; i < m; i += ICPMI_WAVE) rows[c][i] = p[i0 + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 2) {
            const double* col = reinterpret_cast<const double*>(rows[c]) + lane;
            int i = 0;
            for (; i + 8 <= m; i += 8) {
                const double v0 = col[2 * i], v1 = col[2 * i + 2], v2 = col[2 * i + 4], v3 = col[2 * i + 6];
                const double v4 = col[2 * i + 8], v5 = col[2 * i + 10], v6 = col[2 * i + 12], v7 = col[2 * i + 14];
                s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
            }
            for (; i < m; ++i) s += col[2 * i];
        }
        __builtin_amdgcn_wave_barrier();                                 // the rows are read before the next copy overwrites them
    }
    if (lane < 2) {
        const int d = lane;
        const double m = s / (double)n;
        if (c == 0) rec[RSREC_MUS + d] = centred ? m : 0.0;
        else rec[RSREC_MUT + d] = centred ? m : (d == 0 ? shift_x : shift_y);
        if (d == 0) rec[c == 0 ? RSREC_NS : RSREC_NT] = (double)n;
    }
}

// np.argmin: the first minimum, or the first NaN if there is one.  All threads of the workgroup get the answer.
// Every thread takes a strided share (first smallest value and first NaN of its share), the waves reduce by
// (value, index) and thread 0 settles the few wave results: no chain of dependent loads.  sh: 3 ints per wave + 1.
constexpr int RS_ARGMIN_INTS = 3 * 16 + 1;
__device__ __forceinline__ int first_argmin(const double* __restrict__ v, int n, int* sh) {
    const int big = 0x7fffffff;
    double bv = __builtin_inf();
    int bi = big, bn = big;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = v[i];
        if (x != x) bn = min(bn, i);
        else if (x < bv) { bv = x; bi = i; }                 // ascending i: the first of equal values stays
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o, ICPMI_WAVE);
        const int oi = __shfl_xor(bi, o, ICPMI_WAVE), on = __shfl_xor(bn, o, ICPMI_WAVE);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        bn = min(bn, on);
    }
    double* shv = reinterpret_cast<double*>(sh);             // value per wave (2 ints each), then indices
    const int waves = (blockDim.x + ICPMI_WAVE - 1) / ICPMI_WAVE;
    int* shi = sh + 2 * 16;
    if (lane_id() == 0) { shv[wave_id()] = bv; shi[wave_id()] = bi; atomicMin(&sh[3 * 16], bn); }
    __syncthreads();
    int best = 0;
    if (sh[3 * 16] != big) best = sh[3 * 16];                // np.argmin: the first NaN wins
    else {
        double gv = __builtin_inf();
        int gi = big;
        for (int w = 0; w < waves; ++w)
            if (shv[w] < gv || (shv[w] == gv && shi[w] < gi)) { gv = shv[w]; gi = shi[w]; }
        best = gi == big ? 0 : gi;                           // nothing below +inf: index 0, as the serial scan
    }
    __syncthreads();                                         // sh may be used again
    return best;
}
__device__ __forceinline__ void first_argmin_init(int* sh) {
    if (threadIdx.x == 0) sh[3 * 16] = 0x7fffffff;
    __syncthreads();
}

// score of one angle per workgroup, clouds and centroids read from the device: table == nullptr: angle a of
// `cs`; else angle a of row k = argmin(prev_scores) of the fine table (a >= its length: +inf)
__global__ __launch_bounds__(RS_THREADS) void rotation_scores_state_kernel(
    const double* __restrict__ vox, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const double* __restrict__ rec, const double* __restrict__ cs, const int32_t* __restrict__ table_cnt, int table_stride,
    const double* __restrict__ prev_scores, int n_prev, double* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) double tile[RS_TILE_DOUBLES];
    __shared__ double red[block_sum_doubles<1>()];
    __shared__ __attribute__((aligned(8))) int sh_k[RS_ARGMIN_INTS];
    block_sum_init(red, block_sum_doubles<1>());
    const int a = blockIdx.x;
    const int n = cnt[0], m = cnt[1];
    if (table_cnt) {
        first_argmin_init(sh_k);
        const int k = first_argmin(prev_scores, n_prev, sh_k);
        if (a >= table_cnt[k]) { if (threadIdx.x == 0) scores[a] = __builtin_inf(); return; }
        cs += (size_t)k * table_stride * 2;
    }
    if (n <= 0 || m <= 0) { if (threadIdx.x == 0) scores[a] = __builtin_nan(""); return; }
    const double* src = vox + (size_t)off[0] * 2;
    const double* tgt = vox + (size_t)off[1] * 2;
    const double mux = rec[RSREC_MUS], muy = rec[RSREC_MUS + 1], shift_x = rec[RSREC_MUT], shift_y = rec[RSREC_MUT + 1];
    const double c = cs[2 * a], s = cs[2 * a + 1];            // R = [[c, -s], [s, c]], features.py:214-215
    double acc[1] = {0.0};
    for (int first = 0; first < n; first += RS_THREADS) {     // uniform trip count
        const int i = first + threadIdx.x;
        const int ii = i < n ? i : n - 1;
        const double x = src[2 * ii] - mux, y = src[2 * ii + 1] - muy;              // src - mu_s, features.py:207
        double p[1][2] = {{(x * c + y * -s) + shift_x, (x * s + y * c) + shift_y}};   // src_c @ R.T + shift, features.py:216
        double best[1] = {__builtin_inf()};
        int bestj[1] = {0};
        for (int t0 = 0; t0 < m; t0 += RS_TILE_POINTS) {
            const int cntp = min(RS_TILE_POINTS, m - t0);
            __syncthreads();
            const int padded = stage_targets<2>(tgt + (size_t)t0 * 2, cntp, tile);
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

__global__ void rs_finish_kernel(const double* __restrict__ coarse, int n_coarse, const double* __restrict__ fine,
                                 const int32_t* __restrict__ fine_cnt, double* __restrict__ rec) {
    __shared__ __attribute__((aligned(8))) int sh[RS_ARGMIN_INTS];
    first_argmin_init(sh);
    const int k = first_argmin(coarse, n_coarse, sh);
    const int nf = fine_cnt ? fine_cnt[k] : 0;
    first_argmin_init(sh);
    const int j = nf > 0 ? first_argmin(fine, nf, sh) : 0;
    if (threadIdx.x == 0) {
        rec[RSREC_K] = (double)k; rec[RSREC_CSCORE] = coarse[k];
        rec[RSREC_NF] = (double)nf; rec[RSREC_J] = (double)j; rec[RSREC_FSCORE] = nf > 0 ? fine[j] : __builtin_nan("");
    }
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

// workspace: offsets (3 int32) | counts (2 int32) | voxel-filtered copy | coarse + fine scores | voxel scratch
static size_t rs_align(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t icpmi_rotation_search_workspace_bytes(int32_t n_src, int32_t n_tgt, int32_t n_coarse, int32_t max_fine) {
    if (n_src < 0 || n_tgt < 0 || n_coarse < 0 || max_fine < 0) return 0;
    const int mx = n_src > n_tgt ? n_src : n_tgt;
    return 256 + rs_align(((size_t)n_src + n_tgt + 1) * 16) + rs_align(((size_t)n_coarse + max_fine + 1) * 8) +
           rs_align(icpmi_voxel_workspace_bytes(mx));
}

extern "C" int icpmi_rotation_search(const double* pts, int32_t n_src, int32_t n_tgt, double voxel_size,
                                     const double* coarse_cs, int32_t n_coarse,
                                     const double* fine_cs, const int32_t* fine_cnt, int32_t max_fine,
                                     int32_t centred, double shift_x, double shift_y,
                                     double* out_record, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace icpmi;
    if (!pts || !coarse_cs || !out_record || !workspace || n_src <= 0 || n_tgt <= 0 || n_coarse <= 0 || max_fine < 0) return ICPMI_ERR_ARG;
    if (max_fine > 0 && (!fine_cs || !fine_cnt)) return ICPMI_ERR_ARG;
    if (!(voxel_size > 0.0)) return ICPMI_ERR_ARG;
    if (workspace_bytes < icpmi_rotation_search_workspace_bytes(n_src, n_tgt, n_coarse, max_fine)) return ICPMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* w = (unsigned char*)workspace;
    int32_t* off = (int32_t*)w;
    int32_t* cnt = off + 4;
    double* vox = (double*)(w + 256);
    double* sc_coarse = (double*)(w + 256 + rs_align(((size_t)n_src + n_tgt + 1) * 16));
    double* sc_fine = sc_coarse + n_coarse;
    void* vws = (unsigned char*)sc_coarse + rs_align(((size_t)n_coarse + max_fine + 1) * 8);
    const int mx = n_src > n_tgt ? n_src : n_tgt;
    rs_offsets_kernel<<<1, 1, 0, st>>>(off, n_src, n_tgt);
    const int32_t off_host[3] = {0, n_src, n_src + n_tgt};                  // read before the call returns
    int rc = icpmi_voxel_downsample_batch(pts, off, off_host, 2, 2, voxel_size, vox, cnt, vws, icpmi_voxel_workspace_bytes(mx), stream);
    if (rc != ICPMI_OK) return rc;
    rs_means_kernel<<<1, 2 * ICPMI_WAVE, 0, st>>>(vox, off, cnt, centred, shift_x, shift_y, out_record);
    rotation_scores_state_kernel<<<n_coarse, RS_THREADS, 0, st>>>(vox, off, cnt, out_record, coarse_cs, nullptr, 0, nullptr, 0, sc_coarse);
    if (max_fine > 0)
        rotation_scores_state_kernel<<<max_fine, RS_THREADS, 0, st>>>(vox, off, cnt, out_record, fine_cs, fine_cnt, max_fine, sc_coarse,
                                                                      n_coarse, sc_fine);
    rs_finish_kernel<<<1, 64, 0, st>>>(sc_coarse, n_coarse, sc_fine, max_fine > 0 ? fine_cnt : nullptr, out_record);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
