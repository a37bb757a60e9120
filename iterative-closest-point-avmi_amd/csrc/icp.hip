// icp.hip — fused ICP: one workgroup runs one scan pair to convergence.
//
// Replaces the loop of reference utilities/icp.py:153-223 (everything after the
// two voxel_downsample calls and estimate_normals_2d): NN search (icp.py:179),
// correspondence rejection (:183-189), point-to-line normal equations + 3x3
// solve (:79-115) or point-to-point covariance + rotation (:196-207),
// accumulate/apply (:210-212), mean squared error and the convergence test
// (:215-220).  Pairs are independent, so a batch is one launch with one
// workgroup per pair and no host round trip; the iteration count differs per
// pair and each workgroup simply leaves its loop when its pair is done.
#include "linalg.hpp"
#include "nn.hpp"

namespace icpmi {

constexpr int ICP_THREADS = 512;
constexpr int ICP_MAXW = ICP_THREADS / ICPMI_WAVE;
constexpr int ICP_TILE_DOUBLES = 6144;   // 48 KiB: 3072 2-D / 2048 3-D target points resident in LDS
constexpr int ICP_SMAX = 4;              // source rows per thread per NN pass

template <int DIM>
constexpr int icp_tile_points() { return (ICP_TILE_DOUBLES / DIM) / NN_CHUNK * NN_CHUNK; }

struct IcpArgs {
    const double* pts;
    const int32_t* off;
    const int32_t* cnt;
    const double* normals;
    const int32_t* pair_src;
    const int32_t* pair_tgt;
    const double* init;
    double* results;
    double* wsP;      // [pairs][max_src_n][DIM] current (transformed) source rows
    double* wsD2;     // [pairs][max_src_n]      squared NN distance
    int32_t* wsIdx;   // [pairs][max_src_n]      NN index into the target cloud
    int max_src_n;
    double error_threshold;
    double max_corr_dist;
    int max_iterations;
    int method;
    int has_init;
};

// One NN pass for the rows first + s*ICP_THREADS + tid, s < S.
template <int DIM, int S>
__device__ __forceinline__ void icp_nn_pass(const double* __restrict__ P, int N, int first,
                                            const double* __restrict__ tgt, int M, bool resident,
                                            int resident_padded, double* tile,
                                            double* __restrict__ D2, int32_t* __restrict__ IDX) {
    double p[S][DIM], best[S];
    int bestj[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = first + s * ICP_THREADS + threadIdx.x;
        const int nn = n < N ? n : N - 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) p[s][d] = P[(size_t)nn * DIM + d];
        best[s] = __builtin_inf();
        bestj[s] = 0;
    }
    if (resident) {
        nn_scan_tile<DIM, S>(tile, resident_padded, 0, p, best, bestj);
    } else {
        constexpr int TP = icp_tile_points<DIM>();
        for (int t0 = 0; t0 < M; t0 += TP) {
            const int c = min(TP, M - t0);
            __syncthreads();
            const int padded = stage_targets<DIM>(tgt + (size_t)t0 * DIM, c, tile);
            __syncthreads();
            nn_scan_tile<DIM, S>(tile, padded, t0, p, best, bestj);
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = first + s * ICP_THREADS + threadIdx.x;
        if (n < N) { D2[n] = best[s]; IDX[n] = bestj[s]; }
    }
}

template <int DIM>
__global__ __launch_bounds__(ICP_THREADS) void icp_fused_kernel(IcpArgs a) {
    __shared__ __attribute__((aligned(16))) double tile[ICP_TILE_DOUBLES];
    __shared__ double redA[block_sum_doubles<10>()];
    __shared__ double redB[block_sum_doubles<DIM * DIM>()];
    __shared__ double redC[block_sum_doubles<1>()];
    block_sum_init(redA, block_sum_doubles<10>());
    block_sum_init(redB, block_sum_doubles<DIM * DIM>());
    block_sum_init(redC, block_sum_doubles<1>());
    __syncthreads();

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int sc = a.pair_src[b], tc = a.pair_tgt[b];
    const int N = a.cnt ? a.cnt[sc] : a.off[sc + 1] - a.off[sc];
    const int M = a.cnt ? a.cnt[tc] : a.off[tc + 1] - a.off[tc];
    const double* src = a.pts + (size_t)a.off[sc] * DIM;
    const double* tgt = a.pts + (size_t)a.off[tc] * DIM;
    const double* nrm = a.normals ? a.normals + (size_t)a.off[tc] * 2 : nullptr;
    double* P = a.wsP + (size_t)b * a.max_src_n * DIM;
    double* D2 = a.wsD2 + (size_t)b * a.max_src_n;
    int32_t* IDX = a.wsIdx + (size_t)b * a.max_src_n;
    double* res = a.results + (size_t)b * ICPMI_RES_DOUBLES;

    // running totals, identical in every thread (all inputs are uniform)
    double rt[DIM * DIM], tt[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
        tt[i] = 0.0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) rt[i * DIM + j] = (i == j) ? 1.0 : 0.0;
    }
    if (a.has_init) {                                       // icp.py:153-156
        const double* in = a.init + (size_t)b * (DIM * DIM + DIM);
#pragma unroll
        for (int i = 0; i < DIM * DIM; ++i) rt[i] = in[i];
#pragma unroll
        for (int i = 0; i < DIM; ++i) tt[i] = in[DIM * DIM + i];
    }
    double err = __builtin_inf(), prev = __builtin_inf(), delta = __builtin_inf();
    int iters = 0, status = ICPMI_ST_MAXITER;

    if (N <= 0 || M <= 0 || N > a.max_src_n) {
        status = ICPMI_ST_EMPTY;
    } else {
        for (int n = tid; n < N; n += ICP_THREADS) {
            double x[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d) x[d] = src[(size_t)n * DIM + d];
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
                double s = x[i];
                if (a.has_init) {                           // source @ R_init.T + t_init
                    s = 0.0;
#pragma unroll
                    for (int j = 0; j < DIM; ++j) s += x[j] * rt[i * DIM + j];
                    s += tt[i];
                }
                P[(size_t)n * DIM + i] = s;
            }
        }
        const bool use_p2l = (a.method == ICPMI_POINT_TO_LINE) && DIM == 2 && nrm != nullptr;
        const bool has_corr = a.max_corr_dist >= 0.0;
        const double max_corr_sq = a.max_corr_dist * a.max_corr_dist;   // icp.py:169
        const int need = max(3, N / 10);                                 // icp.py:186
        const bool resident = M <= icp_tile_points<DIM>();
        int resident_padded = 0;
        if (resident) {
            resident_padded = stage_targets<DIM>(tgt, M, tile);
            __syncthreads();
        }

        for (int it = 0; it < a.max_iterations; ++it) {
            // ── correspondences ──────────────────────────────────────────────
            for (int first = 0; first < N; first += ICP_THREADS * ICP_SMAX) {
                const int rem = N - first;
                if (rem > 3 * ICP_THREADS) icp_nn_pass<DIM, 4>(P, N, first, tgt, M, resident, resident_padded, tile, D2, IDX);
                else if (rem > 2 * ICP_THREADS) icp_nn_pass<DIM, 3>(P, N, first, tgt, M, resident, resident_padded, tile, D2, IDX);
                else if (rem > ICP_THREADS) icp_nn_pass<DIM, 2>(P, N, first, tgt, M, resident, resident_padded, tile, D2, IDX);
                else icp_nn_pass<DIM, 1>(P, N, first, tgt, M, resident, resident_padded, tile, D2, IDX);
            }
            // Every later phase touches row n from the same thread (n = tid mod
            // ICP_THREADS), so P/D2/IDX need no barrier of their own.
            double r[DIM * DIM], t[DIM];
            if (use_p2l) {
                if constexpr (DIM == 2) {
                    // ── point-to-line normal equations, icp.py:88-104 ────────
                    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    for (int n = tid; n < N; n += ICP_THREADS) {
                        const double dist = sqrt(D2[n]);
                        if (has_corr && !(dist * dist < max_corr_sq)) continue;   // icp.py:184-185
                        const int j = IDX[n];
                        const double px = P[2 * n], py = P[2 * n + 1];
                        const double qx = tgt[2 * j], qy = tgt[2 * j + 1];        // gathers go to L2, not LDS
                        const double nx = nrm[2 * j], ny = nrm[2 * j + 1];
                        const double dx = px - qx, dy = py - qy;
                        const double c = ny * px - nx * py;
                        const double bi = -(nx * dx + ny * dy);
                        acc[0] += c * c;  acc[1] += c * nx;  acc[2] += c * ny;
                        acc[3] += nx * nx; acc[4] += nx * ny; acc[5] += ny * ny;
                        acc[6] += c * bi; acc[7] += nx * bi; acc[8] += ny * bi;
                        acc[9] += 1.0;
                    }
                    block_sum<10, ICP_MAXW>(acc, redA);
                    if (has_corr && acc[9] < (double)need) { status = ICPMI_ST_FEW_INLIERS; break; }
                    double A[3][3] = {{acc[0], acc[1], acc[2]}, {acc[1], acc[3], acc[4]}, {acc[2], acc[4], acc[5]}};
                    double rhs[3] = {acc[6], acc[7], acc[8]}, x[3];
                    if (solve3(A, rhs, x)) {
                        double st, ct;
                        sincos_step(x[0], st, ct);                                  // icp.py:110-114
                        r[0] = ct; r[1] = -st; r[2] = st; r[3] = ct; t[0] = x[1]; t[1] = x[2];
                    } else {
                        r[0] = 1.0; r[1] = 0.0; r[2] = 0.0; r[3] = 1.0; t[0] = 0.0; t[1] = 0.0;
                    }
                }
            } else {
                // ── point-to-point: centroids, then centred cross-covariance ─
                double m[2 * DIM + 1];
#pragma unroll
                for (int i = 0; i < 2 * DIM + 1; ++i) m[i] = 0.0;
                for (int n = tid; n < N; n += ICP_THREADS) {
                    const double dist = sqrt(D2[n]);
                    if (has_corr && !(dist * dist < max_corr_sq)) continue;
                    const int j = IDX[n];
#pragma unroll
                    for (int d = 0; d < DIM; ++d) {
                        m[d] += P[(size_t)n * DIM + d];
                        m[DIM + d] += tgt[(size_t)j * DIM + d];
                    }
                    m[2 * DIM] += 1.0;
                }
                block_sum<2 * DIM + 1, ICP_MAXW>(m, redA);
                if (has_corr && m[2 * DIM] < (double)need) { status = ICPMI_ST_FEW_INLIERS; break; }
                double mp[DIM], mq[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d) { mp[d] = m[d] / m[2 * DIM]; mq[d] = m[DIM + d] / m[2 * DIM]; }
                double W[DIM * DIM];
#pragma unroll
                for (int i = 0; i < DIM * DIM; ++i) W[i] = 0.0;
                for (int n = tid; n < N; n += ICP_THREADS) {
                    const double dist = sqrt(D2[n]);
                    if (has_corr && !(dist * dist < max_corr_sq)) continue;
                    const int j = IDX[n];
                    double pc[DIM], qc[DIM];
#pragma unroll
                    for (int d = 0; d < DIM; ++d) {
                        pc[d] = P[(size_t)n * DIM + d] - mp[d];
                        qc[d] = tgt[(size_t)j * DIM + d] - mq[d];
                    }
#pragma unroll
                    for (int i = 0; i < DIM; ++i)
#pragma unroll
                        for (int k = 0; k < DIM; ++k) W[i * DIM + k] += pc[i] * qc[k];
                }
                block_sum<DIM * DIM, ICP_MAXW>(W, redB);
                if constexpr (DIM == 2) kabsch2(W, r); else kabsch3(W, r);
#pragma unroll
                for (int i = 0; i < DIM; ++i) {                                 // icp.py:207
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < DIM; ++k) s += r[i * DIM + k] * mp[k];
                    t[i] = mq[i] - s;
                }
            }
            // ── accumulate totals, icp.py:210-211 ────────────────────────────
            double nr[DIM * DIM], nt[DIM];
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int c = 0; c < DIM; ++c) s += r[i * DIM + c] * rt[c * DIM + k];
                    nr[i * DIM + k] = s;
                }
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < DIM; ++k) s += tt[k] * r[i * DIM + k];
                nt[i] = s + t[i];
            }
#pragma unroll
            for (int i = 0; i < DIM * DIM; ++i) rt[i] = nr[i];
#pragma unroll
            for (int i = 0; i < DIM; ++i) tt[i] = nt[i];
            // ── apply to ALL rows and take the mean squared residual, :212-215 ─
            double e[1] = {0.0};
            for (int n = tid; n < N; n += ICP_THREADS) {
                const int j = IDX[n];
                double x[DIM], y[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d) x[d] = P[(size_t)n * DIM + d];
                double se = 0.0;
#pragma unroll
                for (int i = 0; i < DIM; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < DIM; ++k) s += x[k] * r[i * DIM + k];
                    y[i] = s + t[i];
                    P[(size_t)n * DIM + i] = y[i];
                    const double dq = tgt[(size_t)j * DIM + i] - y[i];
                    se += dq * dq;
                }
                e[0] += se;
            }
            block_sum<1, ICP_MAXW>(e, redC);
            err = e[0] / (double)N;
            iters = it + 1;
            delta = fabs(prev - err);
            if (delta < a.error_threshold) { status = ICPMI_ST_CONVERGED; break; }   // icp.py:216-219
            prev = err;
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < ICPMI_RES_DOUBLES; ++i) res[i] = 0.0;
#pragma unroll
        for (int i = 0; i < DIM * DIM; ++i) res[ICPMI_RES_R + i] = rt[i];
#pragma unroll
        for (int i = 0; i < DIM; ++i) res[ICPMI_RES_T + i] = tt[i];
        res[ICPMI_RES_ERR] = err;
        res[ICPMI_RES_DELTA] = delta;
        res[ICPMI_RES_ITERS] = (double)iters;
        res[ICPMI_RES_STATUS] = (double)status;
    }
}

}  // namespace icpmi

extern "C" size_t icpmi_icp_workspace_bytes(int32_t n_pairs, int32_t max_src_n, int32_t dim) {
    if (n_pairs < 0 || max_src_n < 0 || (dim != 2 && dim != 3)) return 0;
    const size_t rows = (size_t)n_pairs * (size_t)max_src_n;
    return rows * dim * sizeof(double) + rows * sizeof(double) + rows * sizeof(int32_t) + 256;
}

namespace icpmi {
int launch_icp2(const double* pts, const int32_t* off, const int32_t* cnt, const int32_t* ps, const int32_t* pt,
                int n_pairs, int max_src_n, int max_tgt_n, int total_rows, const icpmi_icp_params* p, const double* init,
                double* results, const void* prepared, void* workspace, size_t workspace_bytes, hipStream_t st);   // icp2.hip
}

extern "C" int icpmi_icp_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                               const double* normals, const void* prepared,
                               const int32_t* pair_src, const int32_t* pair_tgt,
                               int32_t n_pairs, int32_t max_src_n, int32_t max_tgt_n, int32_t total_rows,
                               const icpmi_icp_params* p, const double* init, double* results,
                               void* workspace, size_t workspace_bytes, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !pair_src || !pair_tgt || !p || !results) return ICPMI_ERR_ARG;
    if (n_pairs < 0 || max_src_n < 0 || max_tgt_n < 0 || (p->dim != 2 && p->dim != 3)) return ICPMI_ERR_ARG;
    if (p->method != ICPMI_POINT_TO_POINT && p->method != ICPMI_POINT_TO_LINE) return ICPMI_ERR_ARG;
    if (p->has_init && !init) return ICPMI_ERR_ARG;
    if (n_pairs == 0) return ICPMI_OK;
    hipStream_t st = (hipStream_t)stream;
    // fast path: prepared (axis-sorted) targets, everything on chip
    if (prepared && p->dim == 2 && max_src_n <= 4096)
        return launch_icp2(pts, off_dev, cnt_dev, pair_src, pair_tgt, n_pairs, max_src_n, max_tgt_n, total_rows, p, init,
                           results, prepared, workspace, workspace_bytes, st);
    if (p->method == ICPMI_POINT_TO_LINE && p->dim == 2 && !normals) return ICPMI_ERR_ARG;
    if (!workspace || workspace_bytes < icpmi_icp_workspace_bytes(n_pairs, max_src_n, p->dim)) return ICPMI_ERR_WORKSPACE;
    const size_t rows = (size_t)n_pairs * (size_t)max_src_n;
    IcpArgs a;
    a.pts = pts; a.off = off_dev; a.cnt = cnt_dev; a.normals = normals;
    a.pair_src = pair_src; a.pair_tgt = pair_tgt; a.init = init; a.results = results;
    a.wsP = (double*)workspace;
    a.wsD2 = a.wsP + rows * p->dim;
    a.wsIdx = (int32_t*)(a.wsD2 + rows);
    a.max_src_n = max_src_n;
    a.error_threshold = p->error_threshold; a.max_corr_dist = p->max_corr_dist;
    a.max_iterations = p->max_iterations; a.method = p->method; a.has_init = p->has_init;
    if (p->dim == 2) icp_fused_kernel<2><<<n_pairs, ICP_THREADS, 0, st>>>(a);
    else icp_fused_kernel<3><<<n_pairs, ICP_THREADS, 0, st>>>(a);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
