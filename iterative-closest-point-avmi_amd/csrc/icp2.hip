// icp2.hip — fused 2-D ICP on prepared (axis-sorted) targets: the fast path.
//
// Same loop as icp.hip (reference utilities/icp.py:153-223), same arithmetic and
// the same correspondences, but the nearest-neighbour step is the exact sweep
// of sweep.hpp over the target copy that prep.hip sorted along its best axis,
// and everything a pair needs lives on chip: the sorted target points, their
// normals and the sorted->row map in LDS (36 B per point), the moving source
// points, their matches and squared distances in registers (up to 4 rows per
// thread).  One workgroup per pair, no traffic between workgroups, no host
// round trip.  Pairs that do not fit (more than 4 rows per thread, target
// larger than the LDS copy, 3-D) run on the exhaustive kernel of icp.hip.
#include <cstdlib>

#include "linalg.hpp"
#include "sweep.hpp"

// -DICPMI_DIAG: a diagnostic build that accumulates s_memtime cycles per phase of
// the iteration in thread 0 and stores them in the unused R slots 4..8 of the
// result record (2-D uses 0..3).  Never part of the shipped library.
#ifdef ICPMI_DIAG
#define DIAG_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define DIAG_ADD(acc, a, b) acc += (double)((b) - (a))
#else
#define DIAG_T(var)
#define DIAG_ADD(acc, a, b)
#endif

namespace icpmi {

constexpr int ICP2_SMAX = 4;

struct Icp2Args {
    const double* pts;
    const int32_t* off;
    const int32_t* cnt;
    const int32_t* pair_src;
    const int32_t* pair_tgt;
    const double* init;
    double* results;
    const double2* g_sxy;
    const double2* g_snrm;
    const int32_t* g_sorig;
    const int32_t* g_dir;
    int lds_points;           // capacity of the LDS copy (points)
    double error_threshold;
    double max_corr_dist;
    int max_iterations;
    int method;
    int has_init;
};

template <int THREADS>
__global__ __launch_bounds__(THREADS, 4) void icp2_fused_kernel(Icp2Args a) {   // 4 waves/SIMD: 2 x 512 or 1 x 1024 per CU
    constexpr int MAXW = THREADS / ICPMI_WAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ double redA[block_sum_doubles<10>()];   // normal equations / centroids
    __shared__ double redB[block_sum_doubles<4>()];    // cross-covariance
    __shared__ double redC[block_sum_doubles<1>()];    // squared error
    block_sum_init(redA, block_sum_doubles<10>());
    block_sum_init(redB, block_sum_doubles<4>());
    block_sum_init(redC, block_sum_doubles<1>());

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int sc = a.pair_src[b], tc = a.pair_tgt[b];
    const int N = a.cnt ? a.cnt[sc] : a.off[sc + 1] - a.off[sc];
    const int M = a.cnt ? a.cnt[tc] : a.off[tc + 1] - a.off[tc];
    const double* src = a.pts + (size_t)a.off[sc] * 2;
    double* res = a.results + (size_t)b * ICPMI_RES_DOUBLES;
    const int dir = a.g_dir[tc];

    double2* sxy = reinterpret_cast<double2*>(dyn);
    double2* snrm = reinterpret_cast<double2*>(dyn + (size_t)a.lds_points * 16);
    int32_t* sorig = reinterpret_cast<int32_t*>(dyn + (size_t)a.lds_points * 32);

    double rt[4] = {1.0, 0.0, 0.0, 1.0}, tt[2] = {0.0, 0.0};
    if (a.has_init) {                                       // icp.py:153-156
        const double* in = a.init + (size_t)b * 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) rt[i] = in[i];
        tt[0] = in[4]; tt[1] = in[5];
    }
    double err = __builtin_inf(), prev = __builtin_inf(), delta = __builtin_inf();
    int iters = 0, status = ICPMI_ST_MAXITER;

    if (N <= 0 || M <= 0 || dir < 0 || M > a.lds_points || N > THREADS * ICP2_SMAX) {
        status = ICPMI_ST_EMPTY;                            // the launcher only sends pairs that fit
    } else {
        const bool use_p2l = a.method == ICPMI_POINT_TO_LINE;
        // stage the prepared target: coalesced 16-B reads
        const double2* gx = a.g_sxy + a.off[tc];
        const double2* gn = a.g_snrm + a.off[tc];
        const int32_t* go = a.g_sorig + a.off[tc];
        for (int i = tid; i < M; i += THREADS) {
            sxy[i] = gx[i];
            sorig[i] = go[i];
            if (use_p2l) snrm[i] = gn[i];
        }
        // moving source rows in registers: row n = s*THREADS + tid
        double px[ICP2_SMAX], py[ICP2_SMAX], d2[ICP2_SMAX];
        int pos[ICP2_SMAX];
        const int S = (N + THREADS - 1) / THREADS;
#pragma unroll
        for (int s = 0; s < ICP2_SMAX; ++s) {
            const int n = s * THREADS + tid;
            px[s] = 0.0; py[s] = 0.0; d2[s] = 0.0; pos[s] = 0;
            if (n < N) {
                const double x = src[2 * n], y = src[2 * n + 1];
                if (a.has_init) {                           // source @ R_init.T + t_init
                    double sx = 0.0, sy = 0.0;
                    sx += x * rt[0]; sx += y * rt[1]; sx += tt[0];
                    sy += x * rt[2]; sy += y * rt[3]; sy += tt[1];
                    px[s] = sx; py[s] = sy;
                } else { px[s] = x; py[s] = y; }
            }
        }
        const bool has_corr = a.max_corr_dist >= 0.0;
        const double max_corr_sq = a.max_corr_dist * a.max_corr_dist;   // icp.py:169
        const int need = max(3, N / 10);                                 // icp.py:186
        __syncthreads();
        // largest |projection| of the target (the copy is sorted along it): rounding slack of the diagonal axes
        const double2 c_lo = sxy[0], c_hi = sxy[M - 1];
        const double uabs = fmax(fabs(proj(dir, c_lo.x, c_lo.y)), fabs(proj(dir, c_hi.x, c_hi.y)));
#pragma unroll
        for (int s = 0; s < ICP2_SMAX; ++s) pos[s] = -1;                 // no previous match yet

#ifdef ICPMI_DIAG
        double dg_nn = 0, dg_acc = 0, dg_apply = 0, dg_gather = 0, dg_red = 0, dg_solve = 0;
#endif
        for (int it = 0; it < a.max_iterations; ++it) {
            DIAG_T(c0);
            // ── correspondences: exact sweep search in LDS, icp.py:179 ───────
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s)
                if (s < S && s * THREADS + tid < N) pos[s] = sweep_nn(sxy, sorig, M, dir, uabs, px[s], py[s], pos[s], d2[s]);
#ifdef ICPMI_DIAG
            __syncthreads();          // diag only: charge the slowest wave's search to the search phase
#endif
            DIAG_T(c1);
            double r[4], t[2];
            if (use_p2l) {
                // ── point-to-line normal equations, icp.py:88-104 ────────────
                double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s) {
                    if (!(s < S && s * THREADS + tid < N)) continue;
                    if (has_corr) {                                               // icp.py:184-185
                        const double dist = sqrt(d2[s]);
                        if (!(dist * dist < max_corr_sq)) continue;
                    }
                    const double2 q = sxy[pos[s]], nm = snrm[pos[s]];
                    const double dx = px[s] - q.x, dy = py[s] - q.y;
                    const double c = nm.y * px[s] - nm.x * py[s];
                    const double bi = -(nm.x * dx + nm.y * dy);
                    acc[0] += c * c;       acc[1] += c * nm.x;    acc[2] += c * nm.y;
                    acc[3] += nm.x * nm.x; acc[4] += nm.x * nm.y; acc[5] += nm.y * nm.y;
                    acc[6] += c * bi;      acc[7] += nm.x * bi;   acc[8] += nm.y * bi;
                    acc[9] += 1.0;
                }
                DIAG_T(d0);
                block_sum<10, MAXW>(acc, redA);
                DIAG_T(d1);
                if (has_corr && acc[9] < (double)need) { status = ICPMI_ST_FEW_INLIERS; break; }
                double A[3][3] = {{acc[0], acc[1], acc[2]}, {acc[1], acc[3], acc[4]}, {acc[2], acc[4], acc[5]}};
                double rhs[3] = {acc[6], acc[7], acc[8]}, x[3];
                if (solve3(A, rhs, x)) {
                    double st, ct;
                    sincos(x[0], &st, &ct);                                        // icp.py:110-114
                    r[0] = ct; r[1] = -st; r[2] = st; r[3] = ct; t[0] = x[1]; t[1] = x[2];
                } else {
                    r[0] = 1.0; r[1] = 0.0; r[2] = 0.0; r[3] = 1.0; t[0] = 0.0; t[1] = 0.0;
                }
#ifdef ICPMI_DIAG
                DIAG_T(d2);
                DIAG_ADD(dg_gather, c1, d0); DIAG_ADD(dg_red, d0, d1); DIAG_ADD(dg_solve, d1, d2);
                if (tid == 0) { res[7] = dg_gather; res[8] = dg_red; res[10 + 1] = dg_solve; }
#endif
            } else {
                // ── point-to-point: centroids, centred cross-covariance, icp.py:197-207 ─
                double m[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s) {
                    if (!(s < S && s * THREADS + tid < N)) continue;
                    if (has_corr) {
                        const double dist = sqrt(d2[s]);
                        if (!(dist * dist < max_corr_sq)) continue;
                    }
                    const double2 q = sxy[pos[s]];
                    m[0] += px[s]; m[1] += py[s]; m[2] += q.x; m[3] += q.y; m[4] += 1.0;
                }
                block_sum<5, MAXW>(m, redA);
                if (has_corr && m[4] < (double)need) { status = ICPMI_ST_FEW_INLIERS; break; }
                const double mpx = m[0] / m[4], mpy = m[1] / m[4], mqx = m[2] / m[4], mqy = m[3] / m[4];
                double W[4] = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s) {
                    if (!(s < S && s * THREADS + tid < N)) continue;
                    if (has_corr) {
                        const double dist = sqrt(d2[s]);
                        if (!(dist * dist < max_corr_sq)) continue;
                    }
                    const double2 q = sxy[pos[s]];
                    const double pcx = px[s] - mpx, pcy = py[s] - mpy, qcx = q.x - mqx, qcy = q.y - mqy;
                    W[0] += pcx * qcx; W[1] += pcx * qcy; W[2] += pcy * qcx; W[3] += pcy * qcy;
                }
                block_sum<4, MAXW>(W, redB);
                kabsch2(W, r);
                double s0 = 0.0, s1 = 0.0;
                s0 += r[0] * mpx; s0 += r[1] * mpy;
                s1 += r[2] * mpx; s1 += r[3] * mpy;
                t[0] = mqx - s0; t[1] = mqy - s1;                                  // icp.py:207
            }
            DIAG_T(c2);
            // ── accumulate totals, icp.py:210-211 ────────────────────────────
            {
                double nr[4], nt[2];
                for (int i = 0; i < 2; ++i) {
                    for (int k = 0; k < 2; ++k) {
                        double s = 0.0;
                        for (int c = 0; c < 2; ++c) s += r[i * 2 + c] * rt[c * 2 + k];
                        nr[i * 2 + k] = s;
                    }
                    double s = 0.0;
                    for (int k = 0; k < 2; ++k) s += tt[k] * r[i * 2 + k];
                    nt[i] = s + t[i];
                }
                rt[0] = nr[0]; rt[1] = nr[1]; rt[2] = nr[2]; rt[3] = nr[3]; tt[0] = nt[0]; tt[1] = nt[1];
            }
            // ── apply to ALL rows, mean squared residual, icp.py:212-215 ─────
            double e[1] = {0.0};
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s) {
                if (!(s < S && s * THREADS + tid < N)) continue;
                const double2 q = sxy[pos[s]];
                double nx = 0.0, ny = 0.0;
                nx += px[s] * r[0]; nx += py[s] * r[1]; nx += t[0];
                ny += px[s] * r[2]; ny += py[s] * r[3]; ny += t[1];
                px[s] = nx; py[s] = ny;
                const double ex = q.x - nx, ey = q.y - ny;
                double se = 0.0;
                se += ex * ex;
                se += ey * ey;
                e[0] += se;
            }
            block_sum<1, MAXW>(e, redC);
            DIAG_T(c3);
#ifdef ICPMI_DIAG
            DIAG_ADD(dg_nn, c0, c1); DIAG_ADD(dg_acc, c1, c2); DIAG_ADD(dg_apply, c2, c3);
            if (tid == 0) { res[4] = dg_nn; res[5] = dg_acc; res[6] = dg_apply; }
#endif
            err = e[0] / (double)N;
            iters = it + 1;
            delta = fabs(prev - err);
            if (delta < a.error_threshold) { status = ICPMI_ST_CONVERGED; break; }   // icp.py:216-219
            prev = err;
        }
    }
    if (tid == 0) {
#ifndef ICPMI_DIAG
#pragma unroll
        for (int i = 0; i < ICPMI_RES_DOUBLES; ++i) res[i] = 0.0;
#endif
        res[0] = rt[0]; res[1] = rt[1]; res[2] = rt[2]; res[3] = rt[3];
        res[ICPMI_RES_T] = tt[0]; res[ICPMI_RES_T + 1] = tt[1];
        res[ICPMI_RES_ERR] = err;
        res[ICPMI_RES_DELTA] = delta;
        res[ICPMI_RES_ITERS] = (double)iters;
        res[ICPMI_RES_STATUS] = (double)status;
    }
}

// host side: called by icpmi_icp_batch (icp.hip) when a prepared buffer is given and everything fits
int launch_icp2(const double* pts, const int32_t* off, const int32_t* cnt, const int32_t* ps, const int32_t* pt,
                int n_pairs, int max_src_n, int max_tgt_n, int total_rows, const icpmi_icp_params* p, const double* init,
                double* results, const void* prepared, hipStream_t st) {
    Icp2Args a;
    const unsigned char* b = (const unsigned char*)prepared;
    a.pts = pts; a.off = off; a.cnt = cnt; a.pair_src = ps; a.pair_tgt = pt; a.init = init; a.results = results;
    a.g_sxy = (const double2*)b;
    a.g_snrm = (const double2*)(b + (size_t)total_rows * 16);
    a.g_sorig = (const int32_t*)(b + (size_t)total_rows * 32);
    a.g_dir = (const int32_t*)(b + (size_t)total_rows * 36);
    int cap = 64;
    while (cap < max_tgt_n) cap <<= 1;
    a.lds_points = cap;
    a.error_threshold = p->error_threshold; a.max_corr_dist = p->max_corr_dist;
    a.max_iterations = p->max_iterations; a.method = p->method; a.has_init = p->has_init;
    const size_t lds = (size_t)cap * 36;
    // rows per thread: 1024-thread workgroups once a source has more than 1024 rows (2 rows per thread
    // instead of 3-4 shortens the search phase, which is the critical path of a pair)
    const char* env = getenv("ICPMI_ICP2_THREADS");
    const int want = env ? atoi(env) : (max_src_n > 1024 ? 1024 : 512);
    if (want != 1024 && max_src_n <= 512 * ICP2_SMAX) {
        if (hipFuncSetAttribute((const void*)icp2_fused_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ICPMI_ERR_HIP;
        icp2_fused_kernel<512><<<n_pairs, 512, lds, st>>>(a);
    } else {
        if (hipFuncSetAttribute((const void*)icp2_fused_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ICPMI_ERR_HIP;
        icp2_fused_kernel<1024><<<n_pairs, 1024, lds, st>>>(a);
    }
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

}  // namespace icpmi
