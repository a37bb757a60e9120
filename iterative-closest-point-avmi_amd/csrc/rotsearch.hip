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
#include "sweep.hpp"

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


// ── translation refinement of the submap variant, slam.py:161-181 ─────────────────────────────────────────────────
// After the sweeps: rotate the (filtered) source by the winning angle, place it at the predicted position, match every
// row to its nearest target point, keep the closest 80 % (np.percentile, linear interpolation) and take the mean of
// (matched - rotated) over them.  Two launches: the matches (K1 scan, a row per thread), then one workgroup for the
// percentile and the mean.  Every step in the reference's arithmetic: the rotated rows are NumPy's (n, 2) @ (2, 2) —
// a BLAS gemm whose element is fma(y, R[c][1], x * R[c][0]) (OpenBLAS, FMA kernels; checked on random inputs in the
// build container) — the distances the k-d tree's (direct differences, IEEE sqrt), squared again; the percentile
// numpy's _lerp on the (n - 1) * 0.8-th order statistic; the mean a row-by-row sum divided by the count.
constexpr int RSR_MAX_ROWS = 2048;            // rows the finishing workgroup holds in LDS
constexpr int RSR_THREADS = 1024;

__device__ __forceinline__ void rsr_best_cs(const double* __restrict__ rec, const double* __restrict__ coarse_cs,
                                            const double* __restrict__ fine_cs, int max_fine, double& ca, double& sa) {
    const int k = (int)rec[RSREC_K], nf = (int)rec[RSREC_NF], j = (int)rec[RSREC_J];
    const double* cs = nf > 0 ? fine_cs + ((size_t)k * max_fine + j) * 2 : coarse_cs + (size_t)k * 2;   // slam.py:157-159
    ca = cs[0]; sa = cs[1];
}

// row i: rot[i] = src[i] @ R.T, d2[i] = (distance to the nearest target)^2 as KDTree.query returns it squared, idx[i] = that target row
__global__ __launch_bounds__(RS_THREADS) void rs_refine_match_kernel(
    const double* __restrict__ vox, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt, const double* __restrict__ rec,
    const double* __restrict__ coarse_cs, const double* __restrict__ fine_cs, int max_fine, double pred_x, double pred_y,
    double2* __restrict__ rot, double* __restrict__ dsq, int32_t* __restrict__ idx) {
    __shared__ __attribute__((aligned(16))) double tile[RS_TILE_DOUBLES];
    const int n = cnt[0], m = cnt[1];
    const int first = blockIdx.x * RS_THREADS;
    if (first >= n || m <= 0) return;                              // uniform per workgroup
    double ca, sa;
    rsr_best_cs(rec, coarse_cs, fine_cs, max_fine, ca, sa);
    const double* src = vox + (size_t)off[0] * 2;
    const double* tgt = vox + (size_t)off[1] * 2;
    const int i = first + threadIdx.x, ii = i < n ? i : n - 1;
    const double x = src[2 * ii], y = src[2 * ii + 1];
    const double rx = __builtin_fma(y, -sa, x * ca), ry = __builtin_fma(y, ca, x * sa);   // src @ R_best.T, slam.py:168
    double p[1][2] = {{rx + pred_x, ry + pred_y}};                 // placed, slam.py:169
    double best[1] = {__builtin_inf()};
    int bestj[1] = {0};
    for (int t0 = 0; t0 < m; t0 += RS_TILE_POINTS) {
        const int c = min(RS_TILE_POINTS, m - t0);
        __syncthreads();
        const int padded = stage_targets<2>(tgt + (size_t)t0 * 2, c, tile);
        __syncthreads();
        nn_scan_tile<2, 1>(tile, padded, t0, p, best, bestj);
    }
    if (i < n) {
        const double d = sqrt(best[0]);
        rot[i] = make_double2(rx, ry); dsq[i] = d * d; idx[i] = bestj[0];    // slam.py:170-171
    }
}

// out[0..1] = refined t (or the predicted position), out[2] = inliers, out[3] = the 80th percentile
__global__ __launch_bounds__(RSR_THREADS) void rs_refine_finish_kernel(
    const double* __restrict__ vox, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const double2* __restrict__ rot, const double* __restrict__ dsq, const int32_t* __restrict__ idx,
    double pred_x, double pred_y, double* __restrict__ out) {
    __shared__ double v[RSR_MAX_ROWS];
    __shared__ double2 diff[RSR_MAX_ROWS];
    __shared__ double stat[2];
    __shared__ int n_in;
    const int n = cnt[0], m = cnt[1], tid = threadIdx.x;
    if (n < 5 || m < 5 || n > RSR_MAX_ROWS) {                      // slam.py:128-129 (the caller returns the prediction)
        if (tid == 0) { out[0] = pred_x; out[1] = pred_y; out[2] = 0.0; out[3] = __builtin_nan(""); }
        return;
    }
    const double* tgt = vox + (size_t)off[1] * 2;
    for (int i = tid; i < n; i += RSR_THREADS) {
        v[i] = dsq[i];
        const double2 r = rot[i];
        const int j = idx[i];
        diff[i] = make_double2(tgt[2 * j] - r.x, tgt[2 * j + 1] - r.y);           // matched - rotated_src, slam.py:178-179
    }
    if (tid == 0) n_in = 0;
    __syncthreads();
    // np.percentile(v, 80), method "linear": order statistics lo = floor((n - 1) * 0.8) and lo + 1 by rank counting
    const double vi = (double)(n - 1) * 0.8;                        // np.true_divide(80, 100) = 0.8
    const int lo = (int)floor(vi), hi = min(lo + 1, n - 1);
    for (int i = tid; i < n; i += RSR_THREADS) {
        const double x = v[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double w = v[j]; rank += (w < x || (w == x && j < i)) ? 1 : 0; }
        if (rank == lo) stat[0] = x;
        if (rank == hi) stat[1] = x;
    }
    __syncthreads();
    const double A = stat[0], B = stat[1], t = vi - (double)lo, d = B - A;
    double thresh = A + d * t;                                      // numpy _lerp
    if (t >= 0.5) thresh = B - d * (1.0 - t);
    int mine = 0;
    for (int i = tid; i < n; i += RSR_THREADS) mine += v[i] <= thresh ? 1 : 0;     // slam.py:174
    atomicAdd(&n_in, mine);
    __syncthreads();
    const int k = n_in;
    if (tid < 2) {
        // np.mean(..., axis=0) of a (k, 2) array: the rows in order, one column per lane
        double s = 0.0;
        const double* col = reinterpret_cast<const double*>(diff) + tid;
        for (int i = 0; i < n; ++i)
            if (v[i] <= thresh) s += col[2 * i];
        out[tid] = k >= 5 ? s / (double)k : (tid == 0 ? pred_x : pred_y);          // slam.py:175-181
        if (tid == 0) { out[2] = (double)k; out[3] = thresh; }
    }
}

// ── the search for a BATCH of pairs: the pre-alignment half of _run_icp_pair (slam.py:53-98 -> features.py:165-242) ──
// The reference scores every angle of the coarse sweep in full (~240 k-d tree sweeps per pair).  Only the arg-min of
// those scores matters (features.py:223), and nearly all of them lose by a wide margin: a rotated source that sticks
// out of the target by metres scores ~1 m^2, the winner ~0.01.  One workgroup per pair therefore
//   1. stages the pair on chip: the target in search order (prepare kernel: sorted along its best projection) with
//      the float32 images of sweep.hpp, the centred source;
//   2. builds a DISTANCE FIELD of the target: a W x W grid over the disc the rotated source can reach, each cell
//      holding a rigorous LOWER bound on the distance from any point of the cell to the nearest target point
//      (exhaustive float32 minimum at the cell centre, minus the half diagonal and the rounding margins);
//   3. gives every coarse angle a lower bound of its score — one field look-up per row instead of a search;
//   4. scores angles exactly in order of that bound, with the exact sorted-sweep search (same float64 distances as
//      the exhaustive scan), keeping the best exact score so far, and stops at the first angle whose BOUND exceeds it:
//      every angle not scored is provably worse than one that was, so np.argmin over the scored ones (first minimum)
//      is np.argmin over all.  The nearest-neighbour distances are the exact ones; their SUM is taken in this kernel's own
//      fixed order (a lane per row with stride 64, then a wave tree; the single-pair kernel strides by 256 and NumPy's
//      mean sums pairwise), so a score agrees with the other two to rounding (1e-13 in the tests) and the arg-min is
//      theirs unless two angles tie to within that rounding — which the goldens, the 512-pair comparisons and the
//      symmetric clouds of the tests have not produced, and which no fixture pins;
//   5. scores the fine grid around the winner (features.py:227-232) the same way and writes the record and, for the
//      ICP that follows, R_init / t_init (device memory: no host round trip between pre-alignment and ICP).
// Typically 8-20 of 240 coarse angles are scored exactly, each by searches of ~10 candidates.
constexpr int RSB_THREADS = 512;
constexpr int RSB_WAVES = RSB_THREADS / ICPMI_WAVE;
constexpr int RSB_W = 64;                        // field cells per side
constexpr int RSB_MAX_ANGLES = 1024;             // coarse angles / fine angles per grid at most
constexpr int RSB_REC_DOUBLES = 16;
constexpr int RSBREC_STATUS = 11, RSBREC_EVALS = 12, RSBREC_FEVALS = 13;
constexpr int RSB_ST_OK = 0, RSB_ST_FEW = 1, RSB_ST_CAPACITY = 2, RSB_ST_NO_FINE = 3;

struct RsbArgs {
    const double* vox;            // voxel-filtered clouds (cloud set layout)
    const int32_t* off;
    const int32_t* cnt;
    const double* means;          // [n_clouds][2]: np.mean(axis=0) of each filtered cloud
    const int32_t* pair_src;
    const int32_t* pair_tgt;
    const double2* g_sxy;         // prepared targets
    const int32_t* g_sorig;
    const float* g_skey;          // float32 bearings (bearing order)
    const int32_t* g_dir;
    const double* coarse_cs;
    int n_coarse;
    const double* fine_cs;
    const int32_t* fine_cnt;
    int max_fine;
    double* records;
    double* init;                 // [n_pairs][6] or nullptr
    int cap;                      // rows of a cloud the LDS copies hold
    int prune;                    // 0: score every angle (tests, experiments)
};

typedef float rsb_v2f __attribute__((ext_vector_type(2)));

// rounds of a row's walk before the box hierarchy takes over (sweep.hpp: sweepf_nn_far): most coarse angles the bounds
// cannot exclude put the rows decimetres to metres off, the fine grid lies about the winner
#ifndef RSB_PK
#define RSB_PK 1                // the rows' searches by the packed float32 walk / scan (sweep.hpp, round 4); 0: the round-3 searches
#endif
#ifndef RSB_COARSE_ROUNDS
#define RSB_COARSE_ROUNDS 6
#endif
#ifndef RSB_FINE_ROUNDS
#define RSB_FINE_ROUNDS SWEEP_FAR_ROUNDS
#endif

// one angle, exactly: mean over the rows of (distance to the nearest target point)^2, features.py:213-218.
// A wave per angle; `limit` (a score already reached by another angle) lets it give up as soon as the part summed so
// far exceeds it (then +inf returns: it cannot be the minimum).
__device__ __forceinline__ double rsb_score_angle(const double2* src_c, int n, const float4* sq, const double2* sxy, const SweepF& filt,
                                                  int m, int dir, double uabs, double c, double s, double shx, double shy,
                                                  const volatile double* limit, bool prune, const float4* tree, int leaves, int walk_rounds) {
    double acc = 0.0;
    const int lane = lane_id();
    for (int first = 0; first < n; first += ICPMI_WAVE) {             // wave-uniform trip count
        const int i = first + lane;
        if (i < n) {
            const double2 p = src_c[i];
            const double qx = (p.x * c + p.y * -s) + shx, qy = (p.x * s + p.y * c) + shy;   // src_c @ R.T + mu_t, features.py:216
            double d2;
#if RSB_PK
            (void)sweepf_nn_far_pk(sq, sxy, tree, leaves, filt, m, dir, uabs, qx, qy, d2, walk_rounds);
#else
            (void)sweepf_nn_far(sq, sxy, tree, leaves, filt, m, dir, uabs, qx, qy, d2, walk_rounds);
#endif
            const double d = sqrt(d2);                                  // KDTree distance ...
            acc += d * d;                                               // ... squared, features.py:218
        }
        if (prune && first + ICPMI_WAVE < n) {
            const double part = wave_sum(acc);
            if (part > *limit * (double)n * 1.000000000001) return __builtin_inf();
        }
    }
    return wave_sum(acc) / (double)n;
}

__global__ __launch_bounds__(RSB_THREADS, 4) void rotation_search_batch_kernel(RsbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ __attribute__((aligned(16))) float field[RSB_W * RSB_W];
    __shared__ float lb[RSB_MAX_ANGLES];            // float32, rounded down by the margin below
    // the field is dead once the bounds are known: the exact scores and the order live in its memory (16 KB: 8 + 2)
    static_assert(RSB_W * RSB_W * 4 >= RSB_MAX_ANGLES * 10, "scores and order alias the field");
    double* scores = reinterpret_cast<double*>(field);
    short* order = reinterpret_cast<short*>(field + 2 * RSB_MAX_ANGLES);
    __shared__ double best_score;
    __shared__ int rt_bits, rho_bits, n_evals;
    __shared__ __attribute__((aligned(8))) int sh_arg[RS_ARGMIN_INTS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int sc = a.pair_src[b], tc = a.pair_tgt[b];
    const int n = a.cnt[sc], m = a.cnt[tc];
    double* rec = a.records + (size_t)b * RSB_REC_DOUBLES;
    double* init = a.init ? a.init + (size_t)b * 6 : nullptr;
    const double musx = a.means[2 * sc], musy = a.means[2 * sc + 1], mutx = a.means[2 * tc], muty = a.means[2 * tc + 1];
    const int dir = a.g_dir[tc];
    int status = RSB_ST_OK;
    if (n < 5 || m < 5) status = RSB_ST_FEW;                           // features.py:203-204: identity, zeros, inf
    else if (n > a.cap || m > a.cap || dir < 0 || dir > SWEEP_POLAR) status = RSB_ST_CAPACITY;
    if (status != RSB_ST_OK) {                                         // uniform per workgroup, before any barrier
        if (tid == 0) {
            for (int i = 0; i < RSB_REC_DOUBLES; ++i) rec[i] = 0.0;
            rec[RSREC_NS] = (double)n; rec[RSREC_NT] = (double)m;
            rec[RSREC_MUS] = musx; rec[RSREC_MUS + 1] = musy; rec[RSREC_MUT] = mutx; rec[RSREC_MUT + 1] = muty;
            rec[RSREC_CSCORE] = __builtin_inf(); rec[RSREC_FSCORE] = __builtin_inf(); rec[RSBREC_STATUS] = (double)status;
            if (init) { init[0] = 1.0; init[1] = 0.0; init[2] = 0.0; init[3] = 1.0; init[4] = 0.0; init[5] = 0.0; }
        }
        return;
    }
    double2* sxy = reinterpret_cast<double2*>(dyn);
    double2* src_c = reinterpret_cast<double2*>(dyn + (size_t)a.cap * 16);
    float4* sq = reinterpret_cast<float4*>(dyn + (size_t)a.cap * 32) + 1;      // one padding entry at either end
    float4* tree = reinterpret_cast<float4*>(dyn + (size_t)a.cap * 48 + 32);   // box hierarchy over blocks of 16 sorted positions (sweep.hpp: far queries)

#ifdef RSB_X_TIMES          // diagnostic build: cycles per phase in the record (tools/time_prealign.py RSB_TIMES=1)
#define RSB_T(k) tph[k] = __builtin_readcyclecounter()
    unsigned long long tph[8];
#else
#define RSB_T(k)
#endif
    RSB_T(0);
    // ── 1. stage the pair ────────────────────────────────────────────────────
    if (tid == 0) { rt_bits = 0; rho_bits = 0; n_evals = 0; best_score = __builtin_inf(); }
    const double2* gx = a.g_sxy + a.off[tc];
    const int32_t* go = a.g_sorig + a.off[tc];
    const float* gk = a.g_skey + a.off[tc];
    // float32 images: relative to the point in the middle of the sort order, or — bearing order — to the frame origin the
    // bearings are taken about (as the fused ICP kernel stages them)
    SweepF filt{0.0, 0.0, 0.0, 0.0f, 0.0f};
    if (dir != SWEEP_POLAR) {
        const double2 o = gx[m >> 1];
        filt.ox = o.x; filt.oy = o.y; filt.uo = proj(dir, o.x, o.y);
    }
    __syncthreads();
    float rmax = 0.0f, rho = 0.0f;
    for (int i = tid; i < m; i += RSB_THREADS) {
        const double2 p = gx[i];
        sxy[i] = p;
        const float4 q = make_float4((float)(p.x - filt.ox), (float)(p.y - filt.oy),
                                     dir == SWEEP_POLAR ? gk[i] : (float)(proj(dir, p.x, p.y) - filt.uo), __int_as_float(go[i]));
        sq[i] = q;
        rmax = fmaxf(rmax, fmaxf(fabsf(q.x), fabsf(q.y)));
    }
    const double2* gs = reinterpret_cast<const double2*>(a.vox) + a.off[sc];
    for (int i = tid; i < n; i += RSB_THREADS) {
        const double2 p = gs[i];
        const double2 cc = make_double2(p.x - musx, p.y - musy);       // src - mu_s, features.py:207
        src_c[i] = cc;
        const float fx = (float)cc.x, fy = (float)cc.y;
        rho = fmaxf(rho, __builtin_amdgcn_sqrtf(fx * fx + fy * fy));
    }
    if (tid == 0) { sq[-1] = make_float4(0.f, 0.f, 0.f, 0.f); sq[m] = make_float4(0.f, 0.f, 0.f, 0.f); }
    atomicMax(&rt_bits, __float_as_int(rmax));                         // non-negative floats order like their bits
    atomicMax(&rho_bits, __float_as_int(rho));
    __syncthreads();
    filt.rt = __int_as_float(rt_bits) * 1.000001f;
    filt.ut = fmaxf(fabsf(sq[0].z), fabsf(sq[m - 1].z)) * 1.000001f;
    const int leaves = sweepf_tree_leaves(m);
    sweepf_build_tree(sq, m, tree, leaves, tid, RSB_THREADS);          // complete after the barrier behind the field
    const double2 c_lo = sxy[0], c_hi = sxy[m - 1];
    const double uabs = fmax(fabs(proj(dir, c_lo.x, c_lo.y)), fabs(proj(dir, c_hi.x, c_hi.y)));

    RSB_T(1);
    // ── 2. distance field: lower bounds on the distance to the target, in the frame of the float32 images ─────
    // The rotated source lies in the disc of radius rho about mu_t; the grid covers the square around it.
    const float H = __int_as_float(rho_bits) * 1.00001f + 1e-3f;       // half side
    const float g = 2.0f * H / (float)RSB_W;                           // cell side
    const float shfx = (float)(mutx - filt.ox), shfy = (float)(muty - filt.oy);   // mu_t in the image frame
    const float fox = shfx - H, foy = shfy - H;                        // the grid's corner
    const bool finite_frame = fabsf(shfx) < 1e18f && fabsf(shfy) < 1e18f && H < 1e18f && filt.rt < 1e18f;
    if (a.prune) {
        // thread: one column, 8 consecutive rows (the x difference is shared); targets are broadcast reads
        const int ix = tid & (RSB_W - 1), iy0 = (tid >> 6) * 8;
        const float cx = fox + ((float)ix + 0.5f) * g;
        rsb_v2f cy[4], mn[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            cy[p] = rsb_v2f{foy + ((float)(iy0 + 2 * p) + 0.5f) * g, foy + ((float)(iy0 + 2 * p + 1) + 0.5f) * g};
            mn[p] = rsb_v2f{__builtin_inff(), __builtin_inff()};
        }
        for (int j = 0; j < m; ++j) {
            const float4 t = sq[j];
            const float dx = cx - t.x, dx2 = dx * dx;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const rsb_v2f dy = cy[p] - t.y;
                mn[p] = __builtin_elementwise_min(mn[p], __builtin_elementwise_fma(dy, dy, rsb_v2f{dx2, dx2}));
            }
        }
        // A cell holds the distance from its CENTRE to the target, lowered by the float32 margins (coordinates and
        // arithmetic: within 1e-6 of the magnitudes involved); a query subtracts its own distance from that centre — the
        // triangle inequality with the actual offset, not the half diagonal every point of the cell would have to assume.
        const float scale = fabsf(shfx) + fabsf(shfy) + 2.0f * H + filt.rt;
        const float slack = 1e-5f * scale + 1e-6f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            field[(iy0 + 2 * p) * RSB_W + ix] = __builtin_amdgcn_sqrtf(mn[p].x) * 0.999999f - slack;
            field[(iy0 + 2 * p + 1) * RSB_W + ix] = __builtin_amdgcn_sqrtf(mn[p].y) * 0.999999f - slack;
        }
    }
    __syncthreads();

    RSB_T(2);
    // ── 3. lower bound of every coarse score: a wave per angle, one look-up per row ──────────────
    const int n_coarse = a.n_coarse;
    if (a.prune) {
        const float inv_g = 1.0f / g;
        for (int k = wave_id(); k < n_coarse; k += RSB_WAVES) {
            const float c = (float)a.coarse_cs[2 * k], s = (float)a.coarse_cs[2 * k + 1];
            float sum = 0.0f;
            bool bad = !finite_frame;
            for (int i = lane_id(); i < n; i += ICPMI_WAVE) {
                const double2 p = src_c[i];
                const float x = (float)p.x, y = (float)p.y;
                const float qx = (x * c - y * s) + shfx, qy = (x * s + y * c) + shfy;
                bad = bad || !(fabsf(qx) < 1e18f && fabsf(qy) < 1e18f);
                int jx = (int)floorf((qx - fox) * inv_g), jy = (int)floorf((qy - foy) * inv_g);
                jx = min(max(jx, 0), RSB_W - 1); jy = min(max(jy, 0), RSB_W - 1);     // (any cell gives a valid bound: the offset is the actual one)
                const float ox_ = qx - (fox + ((float)jx + 0.5f) * g), oy_ = qy - (foy + ((float)jy + 0.5f) * g);
                const float off = __builtin_amdgcn_sqrtf(fmaf(ox_, ox_, oy_ * oy_)) * 1.000001f + 1e-5f * (fabsf(qx) + fabsf(qy)) + 1e-6f;
                const float L = fmaxf(0.0f, field[jy * RSB_W + jx] - off);
                sum = fmaf(L, L, sum);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, ICPMI_WAVE);
            const bool any_bad = __any(bad);
            // float32 sum of n <= 2048 non-negative terms and the division: relative error below 2050 x 2^-24 = 1.3e-4
            if (lane_id() == 0) lb[k] = any_bad ? -__builtin_inff() : sum * 0.9997f / (float)n;
        }
    } else
        for (int k = tid; k < n_coarse; k += RSB_THREADS) lb[k] = -__builtin_inff();
    __syncthreads();
    RSB_T(3);
    // ── 4. order by bound (rank by counting), then exact scores in that order ─────────────
    for (int i = tid; i < RSB_MAX_ANGLES; i += RSB_THREADS) scores[i] = __builtin_inf();      // (the field's memory: see above)
    for (int k = tid; k < n_coarse; k += RSB_THREADS) {
        const float v = lb[k];
        int rank = 0;
        for (int j = 0; j < n_coarse; ++j) { const float w = lb[j]; rank += (w < v || (w == v && j < k)) ? 1 : 0; }
        order[rank] = (short)k;
    }
    __syncthreads();
    RSB_T(4);
    const bool prune = a.prune != 0;
    // (items are dealt to the waves round-robin — no work counter: a counter bumped by lane 0 and broadcast with
    // readfirstlane inside this loop was jump-threaded by the compiler into a per-lane loop whose other 63 lanes read item 0
    // for ever)
    for (int item = wave_id(); item < n_coarse; item += RSB_WAVES) {
        const int k = order[item];
        if ((double)lb[k] > *(volatile double*)&best_score) break;              // bounds ascend: nothing further on can win
        const double sc_k = rsb_score_angle(src_c, n, sq, sxy, filt, m, dir, uabs, a.coarse_cs[2 * k], a.coarse_cs[2 * k + 1], mutx, muty,
                                            &best_score, prune, tree, leaves, RSB_COARSE_ROUNDS);
        if (lane_id() == 0) {
            scores[k] = sc_k;
            atomicAdd(&n_evals, 1);
            if (sc_k < __builtin_inf())                                  // non-negative doubles order like their bits
                atomicMin(reinterpret_cast<unsigned long long*>(&best_score), (unsigned long long)__double_as_longlong(sc_k));
        }
    }
    __syncthreads();
    first_argmin_init(sh_arg);
    const int kbest = first_argmin(scores, n_coarse, sh_arg);
    const double cscore = scores[kbest];
    const int coarse_evals = n_evals;
    const int nf = a.max_fine > 0 ? min(a.fine_cnt[kbest], a.max_fine) : 0;
    __syncthreads();
    RSB_T(5);
    // ── 5. the fine grid around the winner, features.py:227-232 ───────────────
    for (int i = tid; i < RSB_MAX_ANGLES; i += RSB_THREADS) scores[i] = __builtin_inf();
    if (tid == 0) { best_score = __builtin_inf(); n_evals = 0; }
    __syncthreads();
    const double* fcs = a.fine_cs + (size_t)kbest * a.max_fine * 2;
    for (int j = wave_id(); j < nf; j += RSB_WAVES) {
        const double sc_j = rsb_score_angle(src_c, n, sq, sxy, filt, m, dir, uabs, fcs[2 * j], fcs[2 * j + 1], mutx, muty, &best_score, prune, tree, leaves, RSB_FINE_ROUNDS);
        if (lane_id() == 0) {
            scores[j] = sc_j;
            atomicAdd(&n_evals, 1);
            if (sc_j < __builtin_inf())
                atomicMin(reinterpret_cast<unsigned long long*>(&best_score), (unsigned long long)__double_as_longlong(sc_j));
        }
    }
    __syncthreads();
    first_argmin_init(sh_arg);
    const int jbest = nf > 0 ? first_argmin(scores, nf, sh_arg) : 0;
    RSB_T(6);
    if (tid == 0) {
        for (int i = 0; i < RSB_REC_DOUBLES; ++i) rec[i] = 0.0;
        rec[RSREC_NS] = (double)n; rec[RSREC_NT] = (double)m;
        rec[RSREC_MUS] = musx; rec[RSREC_MUS + 1] = musy; rec[RSREC_MUT] = mutx; rec[RSREC_MUT + 1] = muty;
        rec[RSREC_K] = (double)kbest; rec[RSREC_CSCORE] = cscore; rec[RSREC_NF] = (double)nf; rec[RSREC_J] = (double)jbest;
        rec[RSREC_FSCORE] = nf > 0 ? scores[jbest] : __builtin_nan("");
        rec[RSBREC_STATUS] = (double)(nf > 0 ? RSB_ST_OK : RSB_ST_NO_FINE);
        rec[RSBREC_EVALS] = (double)coarse_evals; rec[RSBREC_FEVALS] = (double)n_evals;
#ifdef RSB_X_TIMES
        for (int k = 0; k < 6; ++k) rec[k] = (double)(tph[k + 1] - tph[k]);      // stage, field, bounds, order, coarse, fine
#endif
        if (init) {
            if (nf > 0) {
                // R = [[ca, -sa], [sa, ca]], t = mu_t - R @ mu_s (features.py:235-237).  The 2 x 2 by 2 product is a BLAS
                // gemv in NumPy: y = fma(R[i][0], x0, R[i][1] * x1) (OpenBLAS, FMA kernels) — reproduced, so that t_init is
                // the number the reference hands to ICP
                const double ca = fcs[2 * jbest], sa = fcs[2 * jbest + 1];
                const double y0 = __builtin_fma(ca, musx, -sa * musy), y1 = __builtin_fma(sa, musx, ca * musy);
                init[0] = ca; init[1] = -sa; init[2] = sa; init[3] = ca; init[4] = mutx - y0; init[5] = muty - y1;
            } else { init[0] = 1.0; init[1] = 0.0; init[2] = 0.0; init[3] = 1.0; init[4] = 0.0; init[5] = 0.0; }
        }
    }
}

// np.mean(cloud, axis=0) of every filtered cloud (see rs_means_kernel): a wave per cloud
constexpr int RSB_MEAN_WAVES = 4;
__global__ __launch_bounds__(RSB_MEAN_WAVES* ICPMI_WAVE) void rsb_means_kernel(const double* __restrict__ vox, const int32_t* __restrict__ off,
                                                                                const int32_t* __restrict__ cnt, int n_clouds,
                                                                                double* __restrict__ means) {
    __shared__ double2 rows[RSB_MEAN_WAVES][RS_MEAN_ROWS];
    const int w = wave_id(), lane = lane_id();
    const int c = blockIdx.x * RSB_MEAN_WAVES + w;
    if (c >= n_clouds) return;                                          // whole waves leave: no workgroup barrier below
    const int n = cnt[c];
    const double2* p = reinterpret_cast<const double2*>(vox) + off[c];
    double s = 0.0;
    for (int i0 = 0; i0 < n; i0 += RS_MEAN_ROWS) {                       // wave-uniform trip count
        const int mm = min(RS_MEAN_ROWS, n - i0);
        for (int i = lane; i < mm; i += ICPMI_WAVE) rows[w][i] = p[i0 + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 2) {
            const double* col = reinterpret_cast<const double*>(rows[w]) + lane;
            int i = 0;
            for (; i + 8 <= mm; i += 8) {
                const double v0 = col[2 * i], v1 = col[2 * i + 2], v2 = col[2 * i + 4], v3 = col[2 * i + 6];
                const double v4 = col[2 * i + 8], v5 = col[2 * i + 10], v6 = col[2 * i + 12], v7 = col[2 * i + 14];
                s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
            }
            for (; i < mm; ++i) s += col[2 * i];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < 2) means[2 * c + lane] = s / (double)n;
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


// workspace of the batch search: filtered clouds | counts | means | prepared targets | voxel scratch
static size_t rsb_prepared_at(int32_t total_rows, int32_t n_clouds) {
    return rs_align((size_t)total_rows * 16) + rs_align((size_t)n_clouds * 4) + rs_align((size_t)n_clouds * 16);
}

extern "C" size_t icpmi_rotation_search_batch_workspace_bytes(int32_t total_rows, int32_t n_clouds, int32_t max_n) {
    if (total_rows < 0 || n_clouds < 0 || max_n < 0) return 0;
    return rsb_prepared_at(total_rows, n_clouds) + rs_align(icpmi_prepared_bytes(total_rows, n_clouds, max_n)) +
           rs_align(icpmi_voxel_workspace_bytes(max_n)) + 256;
}

extern "C" int icpmi_rotation_search_batch(const double* pts, const int32_t* off_dev, const int32_t* off_host, int32_t n_clouds,
                                           const int32_t* tgt_ids, int32_t n_tgt_ids,
                                           const int32_t* pair_src, const int32_t* pair_tgt, int32_t n_pairs,
                                           double voxel_size, const double* coarse_cs, int32_t n_coarse,
                                           const double* fine_cs, const int32_t* fine_cnt, int32_t max_fine,
                                           int32_t max_rows_hint, double* out_records, double* out_init,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !off_host || !pair_src || !pair_tgt || !coarse_cs || !out_records || !workspace) return ICPMI_ERR_ARG;
    if (n_clouds <= 0 || n_pairs < 0 || n_coarse <= 0 || max_fine < 0 || n_tgt_ids < 0 || max_rows_hint < 0) return ICPMI_ERR_ARG;
    if (max_fine > 0 && (!fine_cs || !fine_cnt)) return ICPMI_ERR_ARG;
    if (!(voxel_size > 0.0)) return ICPMI_ERR_ARG;
    if (n_coarse > RSB_MAX_ANGLES || max_fine > RSB_MAX_ANGLES) return ICPMI_ERR_UNSUPPORTED;
    if (n_pairs == 0) return ICPMI_OK;
    int max_n = 0;
    for (int c = 0; c < n_clouds; ++c) {
        const int nrow = off_host[c + 1] - off_host[c];
        if (nrow < 0) return ICPMI_ERR_ARG;
        max_n = nrow > max_n ? nrow : max_n;
    }
    const int total_rows = off_host[n_clouds];
    if (max_n > 4096) return ICPMI_ERR_UNSUPPORTED;                      // single-pair entry (icpmi_rotation_search) for larger clouds
    if (workspace_bytes < icpmi_rotation_search_batch_workspace_bytes(total_rows, n_clouds, max_n)) return ICPMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* w = (unsigned char*)workspace;
    double* vox = (double*)w;
    int32_t* cnt = (int32_t*)(w + rs_align((size_t)total_rows * 16));
    double* means = (double*)((unsigned char*)cnt + rs_align((size_t)n_clouds * 4));
    unsigned char* prepared = w + rsb_prepared_at(total_rows, n_clouds);
    const size_t prepared_bytes = rs_align(icpmi_prepared_bytes(total_rows, n_clouds, max_n));
    void* vws = prepared + prepared_bytes;
    int rc = icpmi_voxel_downsample_batch(pts, off_dev, off_host, n_clouds, 2, voxel_size, vox, cnt, vws, icpmi_voxel_workspace_bytes(max_n), stream);
    if (rc != ICPMI_OK) return rc;
    rsb_means_kernel<<<(n_clouds + RSB_MEAN_WAVES - 1) / RSB_MEAN_WAVES, RSB_MEAN_WAVES * ICPMI_WAVE, 0, st>>>(vox, off_dev, cnt, n_clouds, means);
    // search order of the targets: a projection or, for scans in their sensor frame, the bearing (the library's estimate; any
    // order is exact for any query — RS_BATCH = "projection" keeps to the projections)
    const char* oe = option("RS_BATCH");
    // a pair whose target is not among tgt_ids must report "no order" (status 2), not search whatever the workspace held
    if (hipMemsetAsync(prepared + (size_t)total_rows * 40, 0xFF, (size_t)n_clouds * sizeof(int32_t), st) != hipSuccess) return ICPMI_ERR_HIP;
    rc = icpmi_prepare_targets_ex(vox, off_dev, off_host, cnt, tgt_ids, nullptr, tgt_ids ? n_tgt_ids : n_clouds, n_clouds, total_rows, max_n, -1,
                                  nullptr, prepared, prepared_bytes, oe && oe[0] == 'p' ? 0 : 1, stream);
    if (rc != ICPMI_OK) return rc;
    RsbArgs a;
    a.vox = vox; a.off = off_dev; a.cnt = cnt; a.means = means; a.pair_src = pair_src; a.pair_tgt = pair_tgt;
    a.g_sxy = (const double2*)prepared;
    a.g_sorig = (const int32_t*)(prepared + (size_t)total_rows * 32);
    a.g_skey = (const float*)(prepared + (size_t)total_rows * 36);
    a.g_dir = (const int32_t*)(prepared + (size_t)total_rows * 40);
    a.coarse_cs = coarse_cs; a.n_coarse = n_coarse; a.fine_cs = fine_cs; a.fine_cnt = fine_cnt; a.max_fine = max_fine;
    a.records = out_records; a.init = out_init;
    // rows the on-chip copies hold: the largest raw cloud (the filter only removes rows), at most 2 048; the caller's hint
    // (an upper bound it expects for the FILTERED clouds) lowers it so that two workgroups share a CU — a pair with a
    // larger filtered cloud reports RSB_ST_CAPACITY and is left to the single-pair entry
    int cap = max_n < 2048 ? max_n : 2048;
    if (max_rows_hint > 0 && max_rows_hint < cap) cap = max_rows_hint;
    cap = (cap + 63) / 64 * 64;
    a.cap = cap;
    const char* e = option("RS_BATCH");
    a.prune = e && e[0] == 'f' ? 0 : 1;                                 // "full": every angle scored exactly
    const size_t lds = (size_t)cap * 48 + 32 + 32 * (size_t)sweepf_tree_leaves(cap);
    if (dyn_lds((const void*)rotation_search_batch_kernel, lds) != hipSuccess)
        return ICPMI_ERR_HIP;
    rotation_search_batch_kernel<<<n_pairs, RSB_THREADS, lds, st>>>(a);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}


// scratch of the refinement: rotated rows | squared distances | matched rows (n_src of each)
extern "C" size_t icpmi_rotation_refine_workspace_bytes(int32_t n_src) {
    if (n_src < 0) return 0;
    return rs_align((size_t)n_src * 16) + rs_align((size_t)n_src * 8) + rs_align((size_t)n_src * 4) + 256;
}

extern "C" int icpmi_rotation_refine(const void* search_workspace, int32_t n_src, int32_t n_tgt, const double* record,
                                     const double* coarse_cs, const double* fine_cs, int32_t max_fine,
                                     double pred_x, double pred_y, double* out4, void* scratch, size_t scratch_bytes, void* stream) {
    using namespace icpmi;
    if (!search_workspace || !record || !coarse_cs || !out4 || !scratch || n_src <= 0 || n_tgt <= 0 || max_fine < 0) return ICPMI_ERR_ARG;
    if (max_fine > 0 && !fine_cs) return ICPMI_ERR_ARG;
    if (n_src > RSR_MAX_ROWS) return ICPMI_ERR_UNSUPPORTED;             // (raw rows: the filtered count is at most that)
    if (scratch_bytes < icpmi_rotation_refine_workspace_bytes(n_src)) return ICPMI_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* w = (const unsigned char*)search_workspace;
    const int32_t* off = (const int32_t*)w;
    const int32_t* cnt = off + 4;
    const double* vox = (const double*)(w + 256);
    unsigned char* s = (unsigned char*)scratch;
    double2* rot = (double2*)s;
    double* dsq = (double*)(s + rs_align((size_t)n_src * 16));
    int32_t* idx = (int32_t*)((unsigned char*)dsq + rs_align((size_t)n_src * 8));
    rs_refine_match_kernel<<<(n_src + RS_THREADS - 1) / RS_THREADS, RS_THREADS, 0, st>>>(vox, off, cnt, record, coarse_cs, fine_cs, max_fine,
                                                                                          pred_x, pred_y, rot, dsq, idx);
    rs_refine_finish_kernel<<<1, RSR_THREADS, 0, st>>>(vox, off, cnt, rot, dsq, idx, pred_x, pred_y, out4);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
