// icp2.hip — fused 2-D ICP on prepared (axis-sorted) targets: the fast path.
//
// Same loop as icp.hip (reference utilities/icp.py:153-223), same arithmetic and
// the same correspondences, but the nearest-neighbour step is the exact sweep
// of sweep.hpp over the target copy that prep.hip sorted along its best axis,
// and everything a pair needs lives on chip: the sorted target points, their
// normals and the sorted->row map in LDS (36 B per point), the moving source
// points and their matches in registers (up to 4 rows per thread).  One
// workgroup per pair, no traffic between workgroups, no host round trip.
//
// What bounds a pair is the instruction stream of its own workgroup (the tail of
// a batch is a few pairs that run to max_iterations, each alone on a CU), so
// work that is uniform across the workgroup is done by ONE wave: wave 0 combines
// the wave partials, tests convergence, solves the 3x3 system and publishes the
// step through LDS while the other waves wait at the barrier.  The squared
// error of a step is reduced together with the normal equations of the next
// one (one reduction per iteration for point_to_line); its convergence test
// therefore arrives one search late, and the totals it returns are the ones
// held back from before that search — the reference's results exactly.
//
// Pairs that do not fit (more than 4 rows per thread, target larger than the
// LDS copy, 3-D) run on the exhaustive kernel of icp.hip.
#include <cstdio>
#include <cstdlib>

#include "linalg.hpp"
#include "sweep.hpp"

// -DICPMI_DIAG: diagnostic build; thread 0 accumulates s_memtime cycles per phase
// and stores them in the unused R slots 4..8 of its result record.  Never shipped.
#ifdef ICPMI_DIAG
#define DIAG_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define DIAG_SET(var) var = __builtin_readcyclecounter()
#define DIAG_ADD(acc, a, b) acc += (double)((b) - (a))
#else
#define DIAG_T(var)
#define DIAG_SET(var)
#define DIAG_ADD(acc, a, b)
#endif

namespace icpmi {


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
    const float* g_skey;      // float32 images of the sort key (bearing order only)
    const int32_t* g_dir;
    int lds_points;           // capacity of the LDS copy (points)
    double error_threshold;
    double max_corr_dist;
    int max_iterations;
    int method;
    int has_init;
    int n_lo, m_lo;           // second launch: only the pairs whose source has more than n_lo rows or whose target more than m_lo
    int skip_over;            // first launch: leave pairs beyond THREADS x SMAX source rows or lds_points target rows to the second
    // Two-stage run of a large batch (launch_icp2): the first launch takes every pair up to it_limit iterations and
    // parks the ones still running (moving rows, matches, totals); the second continues exactly those, all started
    // together, instead of the last of them finishing alone at the end of one long launch.
    int it_begin, it_limit;   // iterations [it_begin, min(it_limit, max_iterations)) run here
    int resume;               // 1: the pairs are list[0 .. *list_count), their state comes from st_*
    double2* st_xy;           // [pair][st_stride]: moving source rows of a parked pair
    int32_t* st_pos;          //                    and the position of each row's match in the sorted target
    int32_t* list;
    int32_t* list_count;
    int st_stride;
    int32_t* wide_list;       // pairs the first launch leaves to the wider shape (when the caller gave a workspace): the second
    int32_t* wide_count;      // launch walks this list instead of starting a workgroup per pair of the batch just to look
    // Pairs that start FAR from their target (a candidate whose pre-alignment is wrong: every row metres from the nearest
    // wall) walk most of the sorted target in every search.  A first launch that finds the mean squared error of step 0
    // (the convergence test of iteration 1 has it anyway) above far_d2 parks the pair after iteration 1 (same state as
    // above) on far_list, and icp2_far_kernel continues it with searches that give up long walks for a box hierarchy
    // (sweep.hpp).  Same matches, same arithmetic: the pair's result does not depend on which kernel finished it.
    int32_t* far_list;        // nullptr: never
    int32_t* far_count;
    double far_d2;            // +inf: never
};
constexpr int ICP2_FAR_THREADS = 1024, ICP2_FAR_SMAX = 2, ICP2_FAR_POINTS = 2048;   // the continuation's shape: most source rows, target points
constexpr int ICP2_ST_PARKED = 100;     // internal status between the two stages

// ── workgroup sums of NV values per thread ───────────────────────────────────────────────────────────
// Per wave, a TRANSPOSING reduction: v_permlane32_swap exchanges the upper half of one register with the lower
// half of another, so one add halves the partials of TWO values at once ((a_lo, b_lo) + (a_hi, b_hi)); the
// 16-lane form (v_permlane16_swap: odd rows of one register against even rows of the other) does the same for
// the two resulting registers, which leaves four different values in the four 16-lane rows of one register, and
// the four DPP steps inside the rows then finish all four together: 5 instructions per value instead of the 12
// of a DPP tree per value, and ONE total per value and wave (not one per row).  The lead wave adds the totals of
// the waves in wave order.  Every tree is fixed: bitwise reproducible run to run.
__device__ __forceinline__ double swap_add(double a, double b, bool rows16) {
    const long long ab = __double_as_longlong(a), bb = __double_as_longlong(b);
    const unsigned alo = (unsigned)ab, ahi = (unsigned)(ab >> 32), blo = (unsigned)bb, bhi = (unsigned)(bb >> 32);
    unsigned xl, xh, yl, yh;
    if (rows16) {
        const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        xl = r0[0]; yl = r0[1]; xh = r1[0]; yh = r1[1];
    } else {
        const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        xl = r0[0]; yl = r0[1]; xh = r1[0]; yh = r1[1];
    }
    return __longlong_as_double(((long long)xh << 32) | xl) + __longlong_as_double(((long long)yh << 32) | yl);
}

constexpr int RED_MAXW = 16;                                   // waves per workgroup at most
template <int NV>
constexpr int red_doubles() { return (NV + 3) / 4 * 4 * RED_MAXW; }

// totals of this wave's NV values -> scratch[value * RED_MAXW + wave]
template <int NV>
__device__ __forceinline__ void wave_totals(double* scratch, const double (&v)[NV]) {
    constexpr int NP = (NV + 3) / 4 * 4;
    const int w = wave_id(), l = lane_id();
    double h[NP / 2];
#pragma unroll
    for (int j = 0; j < NP / 2; ++j)                           // lanes 0-31: value 2j, lanes 32-63: value 2j + 1
        h[j] = swap_add(2 * j < NV ? v[2 * j] : 0.0, 2 * j + 1 < NV ? v[2 * j + 1] : 0.0, false);
#pragma unroll
    for (int k = 0; k < NP / 4; ++k) {
        // rows 0..3 of q: values 4k, 4k + 2, 4k + 1, 4k + 3 (16 partials each)
        const double q = row_sum(swap_add(h[2 * k], h[2 * k + 1], true));
        const int row = l >> 4;
        const int idx = 4 * k + ((row & 1) << 1) + (row >> 1);
        if ((l & 15) == 0 && idx < NV) scratch[idx * RED_MAXW + w] = q;
    }
}

// lead wave: v[i] = sum over the waves (in wave order) of scratch[i * RED_MAXW + wave], uniform in every lane
template <int NV>
__device__ __forceinline__ void combine_totals(const double* scratch, int n_waves, double (&v)[NV]) {
    const int l = lane_id();
    double s = 0.0;
    if (l < NV)
        for (int w = 0; w < n_waves; ++w) s += scratch[l * RED_MAXW + w];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = readlane_f64(s, i);
}

// control block published by wave 0: r (4), t (2), stop flag, mean_p (2), mean_q (2)
constexpr int CTRL_R = 0, CTRL_T = 4, CTRL_STOP = 6, CTRL_MP = 8, CTRL_MQ = 10;
// state carried from iteration to iteration by thread 0 alone (accumulated transform, errors, outcome): kept in LDS,
// not in registers — live across the search it would cost every thread of the workgroup 20 registers
constexpr int CTRL_RT = 12, CTRL_TT = 16, CTRL_ERR = 18, CTRL_PREV = 19, CTRL_DELTA = 20, CTRL_ITERS = 21, CTRL_STATUS = 22,
              CTRL_DOUBLES = 24;

#ifndef ICP2_PK
#define ICP2_PK 1               // searches of the filter instantiations by the packed float32 walk (sweep.hpp, round 4); 0: the round-2 walks
#endif
#ifndef ICP2_FAR_PK
#define ICP2_FAR_PK 0           // 1: the far continuation's searches by the packed walk and scan too (sweep.hpp) — exact (its tests pass on it) and
                                // slower there: 4.05 against 3.74 ms for the ICP half of the 3 m / 20 degree candidates.  The continuation has already
                                // packed its searching rows into few lanes, and a scan that offers every image of a block it enters issues more than
                                // one that evaluates the few images inside the threshold exactly.
#endif
#ifndef ICP2_STAGE1_ITERS
#define ICP2_STAGE1_ITERS 12    // iterations of the first stage of a large batch (launch_icp2)
#endif
#ifndef ICP2_PK_MIN
#define ICP2_PK_MIN 16          // searching lanes of a wave from which the packed walk is taken
#endif
#ifndef ICP2_PLAIN_ITERS
#define ICP2_PLAIN_ITERS 2      // iterations that search the plain nearest neighbour before budgets are kept
#endif
#ifndef ICP2_CENTRED_ITERS
#define ICP2_CENTRED_ITERS 2    // iterations whose top-two search starts at the row's own projection; later ones walk from the kept match
                                // (= ICP2_PLAIN_ITERS: none does; measured 5.22 / 5.13 ms at 3 / 2 for 16 384 pairs, 5.42 with 1 plain iteration)
#endif
#ifndef ICP2_PLAIN_CENTRED
#define ICP2_PLAIN_CENTRED ICP2_PLAIN_ITERS     // plain iterations that start at the row's own projection
#endif
// THREADS x ICP2_SMAX = most source rows a pair may have on this instantiation
// TGT_LDS: the prepared target is staged in LDS (<= 4096 points); otherwise it is read in place, through L2
// FILT (with TGT_LDS): the LDS copy carries a float32 image per point instead of the row map (48 B instead of
// 36 B per point) and the searches judge every candidate on it first (sweep.hpp, "single-precision filter")
// A value every lane of the wave holds alike, moved to scalar registers (it stays live across the whole search).
__device__ __forceinline__ double wave_uniform(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float wave_uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Finish step it-1 (icp.py:215-220: mean squared error of the step, its change, convergence) and test the inlier
// count of step it (icp.py:186).  Every lane of the lead wave evaluates the same values; the writer lane stores.
// Returns FIN_STOP when the pair is done, FIN_FAR when step 0 left a mean squared error above far_err (the pair goes to the
// far continuation after this iteration, launch_icp2), else FIN_GO.
constexpr int FIN_GO = 0, FIN_STOP = 1, FIN_FAR = 2;
__device__ __forceinline__ int finish_step(double* ctrl, int it, double err_sum, double inliers, int N, bool has_corr,
                                           int need, double error_threshold, bool writer, double far_err) {
    bool stop = false, far = false;
    if (it > 0) {
        const double err = err_sum / (double)N, delta = fabs(ctrl[CTRL_PREV] - err);
        stop = delta < error_threshold;
        far = it == 1 && err > far_err;
        if (writer) {
            ctrl[CTRL_ERR] = err; ctrl[CTRL_DELTA] = delta; ctrl[CTRL_PREV] = err; ctrl[CTRL_ITERS] = (double)it;
            if (stop) ctrl[CTRL_STATUS] = (double)ICPMI_ST_CONVERGED;
        }
    }
    if (!stop && has_corr && inliers < (double)need) {
        stop = true;
        if (writer) { ctrl[CTRL_STATUS] = (double)ICPMI_ST_FEW_INLIERS; ctrl[CTRL_ITERS] = (double)it; }
    }
    return stop ? FIN_STOP : (far ? FIN_FAR : FIN_GO);
}

// R_total = R R_total, t_total = t_total R^T + t: icp.py:210-211
__device__ __forceinline__ void accumulate_step(double* ctrl, const double (&r)[4], const double (&t)[2]) {
    const double rt0 = ctrl[CTRL_RT], rt1 = ctrl[CTRL_RT + 1], rt2 = ctrl[CTRL_RT + 2], rt3 = ctrl[CTRL_RT + 3];
    const double tt0 = ctrl[CTRL_TT], tt1 = ctrl[CTRL_TT + 1];
    ctrl[CTRL_RT] = r[0] * rt0 + r[1] * rt2; ctrl[CTRL_RT + 1] = r[0] * rt1 + r[1] * rt3;
    ctrl[CTRL_RT + 2] = r[2] * rt0 + r[3] * rt2; ctrl[CTRL_RT + 3] = r[2] * rt1 + r[3] * rt3;
    ctrl[CTRL_TT] = (tt0 * r[0] + tt1 * r[1]) + t[0]; ctrl[CTRL_TT + 1] = (tt0 * r[2] + tt1 * r[3]) + t[1];
}

// far continuation: a searching row's query on its way to the lane that runs the search, and the answer on its way back
union FarSlot {
    struct { double x, y; int seed, pad; } in;
    struct { double s1, s3; int p1, p2; } out;
};
static_assert(sizeof(FarSlot) == 24, "24 B per row");

template <int THREADS, int ICP2_SMAX, bool TGT_LDS, bool FILT, bool RESUME, bool FAR = false>
__device__ __forceinline__ void icp2_pair(const Icp2Args& a, const int b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ double redA[red_doubles<11>()];         // normal equations (10) / centroid sums (5) + carried squared error
    __shared__ double redB[red_doubles<4>()];          // cross-covariance
    __shared__ double ctrl[CTRL_DOUBLES];
    constexpr int NWAVES = THREADS / ICPMI_WAVE;

    const int tid = threadIdx.x;
    const bool lead = tid < ICPMI_WAVE;                      // wave 0 carries the uniform state
    const int sc = a.pair_src[b], tc = a.pair_tgt[b];
    const int N = a.cnt ? a.cnt[sc] : a.off[sc + 1] - a.off[sc];
    const int M = a.cnt ? a.cnt[tc] : a.off[tc + 1] - a.off[tc];
    const double* src = a.pts + (size_t)a.off[sc] * 2;
    double* res = a.results + (size_t)b * ICPMI_RES_DOUBLES;
    const int dir = a.g_dir[tc];
    // The voxel filter leaves the row counts on the device, so the launcher sizes the rows per thread for the usual
    // case and sends the (rare) larger clouds to a second launch of a wider shape: each pair is registered by
    // exactly one of the two (uniform per workgroup, before any barrier).
    if (a.skip_over ? (N > THREADS * ICP2_SMAX || (TGT_LDS && M > a.lds_points)) : (a.n_lo >= 0 && N <= a.n_lo && M <= a.m_lo)) {
        if (a.skip_over && a.wide_list && threadIdx.x == 0) a.wide_list[atomicAdd(a.wide_count, 1)] = b;
        return;
    }

    // two instantiations, each sees ONE address space behind these pointers
    double2* lds_xy = reinterpret_cast<double2*>(dyn);
    double2* lds_nrm = reinterpret_cast<double2*>(dyn + (size_t)a.lds_points * 16);
    int32_t* lds_orig = reinterpret_cast<int32_t*>(dyn + (size_t)a.lds_points * 32);      // !FILT: row map (4 B per point)
    float4* lds_sq = reinterpret_cast<float4*>(dyn + (size_t)a.lds_points * 32) + 1;     //  FILT: float32 images (16 B per point), one padding entry at either end
    // (no block boxes here: the far scan of sweep.hpp costs this kernel 11 registers — spills at six waves per SIMD — and
    // 35 % of its time on pairs that start close, to halve the time of pairs that start metres away; measured, round 3)
    const double2* sxy = TGT_LDS ? lds_xy : a.g_sxy + a.off[tc];
    const double2* snrm = TGT_LDS ? lds_nrm : a.g_snrm + a.off[tc];
    const int32_t* sorig = TGT_LDS ? lds_orig : a.g_sorig + a.off[tc];
    static_assert(TGT_LDS || !FILT, "the filter images live in LDS");
    static_assert(!FAR || (FILT && RESUME), "the far continuation: filter images, parked state");
    float4* lds_tree = reinterpret_cast<float4*>(dyn + (size_t)a.lds_points * 48 + 32);    // FAR: box hierarchy over blocks of 16 images (sweep.hpp)
    const int tree_leaves = sweepf_tree_leaves(M);
    // FAR: ICP2_SMAX slots per lane, each wave its own stretch (behind the largest tree: 2 * lds_points bytes)
    FarSlot* far_q = reinterpret_cast<FarSlot*>(dyn + (size_t)a.lds_points * 50 + 32) + (size_t)(threadIdx.x / ICPMI_WAVE) * (ICPMI_WAVE * ICP2_SMAX);
    __shared__ int rt_bits;                                      // max(|x - ox|, |y - oy|) over the target, float32 bits
    if (FILT && tid == 0) rt_bits = 0;

    if (tid == 0) {
        if (RESUME) {                                       // the totals the first stage left in the result record
            ctrl[CTRL_RT] = res[0]; ctrl[CTRL_RT + 1] = res[1]; ctrl[CTRL_RT + 2] = res[2]; ctrl[CTRL_RT + 3] = res[3];
            ctrl[CTRL_TT] = res[ICPMI_RES_T]; ctrl[CTRL_TT + 1] = res[ICPMI_RES_T + 1];
            ctrl[CTRL_ERR] = res[ICPMI_RES_ERR]; ctrl[CTRL_PREV] = res[ICPMI_RES_ERR]; ctrl[CTRL_DELTA] = res[ICPMI_RES_DELTA];
            ctrl[CTRL_ITERS] = res[ICPMI_RES_ITERS];
        } else {
            const double* in = a.init + (size_t)b * 6;      // icp.py:153-156
            ctrl[CTRL_RT] = a.has_init ? in[0] : 1.0; ctrl[CTRL_RT + 1] = a.has_init ? in[1] : 0.0;
            ctrl[CTRL_RT + 2] = a.has_init ? in[2] : 0.0; ctrl[CTRL_RT + 3] = a.has_init ? in[3] : 1.0;
            ctrl[CTRL_TT] = a.has_init ? in[4] : 0.0; ctrl[CTRL_TT + 1] = a.has_init ? in[5] : 0.0;
            ctrl[CTRL_ERR] = __builtin_inf(); ctrl[CTRL_PREV] = __builtin_inf(); ctrl[CTRL_DELTA] = __builtin_inf();
            ctrl[CTRL_ITERS] = 0.0;
        }
        ctrl[CTRL_STATUS] = (double)ICPMI_ST_MAXITER;
    }

    if (N <= 0 || M <= 0 || dir < 0 || (TGT_LDS && M > a.lds_points) || N > THREADS * ICP2_SMAX || (!FILT && dir >= SWEEP_POLAR)) {
        if (tid == 0) ctrl[CTRL_STATUS] = (double)ICPMI_ST_EMPTY;   // the launcher only sends pairs that fit
    } else {
        const bool use_p2l = a.method == ICPMI_POINT_TO_LINE;
        // stage the prepared target: coalesced 16-B reads
        const double2* gx = a.g_sxy + a.off[tc];
        const double2* gn = a.g_snrm + a.off[tc];
        const int32_t* go = a.g_sorig + a.off[tc];
        const float* gk = a.g_skey + a.off[tc];
        SweepF filt{0.0, 0.0, 0.0, 0.0f, 0.0f};
        if constexpr (TGT_LDS && !FILT)
            for (int i = tid; i < M; i += THREADS) {
                lds_xy[i] = gx[i];
                lds_orig[i] = go[i];
                if (use_p2l) lds_nrm[i] = gn[i];
            }
        if constexpr (FILT) {
            // float32 images: relative to the point in the middle of the sort order (small magnitudes), or — bearing
            // order — to the frame origin the bearings are taken about
            if (dir != SWEEP_POLAR) {
                const double2 o = gx[M >> 1];
                filt.ox = wave_uniform(o.x); filt.oy = wave_uniform(o.y); filt.uo = wave_uniform(proj(dir, o.x, o.y));
            }
            float rmax = 0.0f;
            for (int i = tid; i < M; i += THREADS) {
                const double2 p = gx[i];
                lds_xy[i] = p;
                if (use_p2l) lds_nrm[i] = gn[i];
                const float4 q = make_float4((float)(p.x - filt.ox), (float)(p.y - filt.oy),
                                             dir == SWEEP_POLAR ? gk[i] : (float)(proj(dir, p.x, p.y) - filt.uo),
                                             __int_as_float(go[i]));
                lds_sq[i] = q;
                rmax = fmaxf(rmax, fmaxf(fabsf(q.x), fabsf(q.y)));
            }
            if (tid == 0) { lds_sq[-1] = make_float4(0.f, 0.f, 0.f, 0.f); lds_sq[M] = make_float4(0.f, 0.f, 0.f, 0.f); }
            __syncthreads();                                     // rt_bits = 0 is visible, the images are complete
            atomicMax(&rt_bits, __float_as_int(rmax));           // non-negative floats order like their bits
        }
        // moving source rows in registers: row n = s*THREADS + tid
        double px[ICP2_SMAX], py[ICP2_SMAX];
        int pos[ICP2_SMAX];
        // Movement budget of each row's match.  A search returns the two nearest
        // target points and the distance d3 of the third.  While the row is
        // displaced from where it was searched (the anchor) by less than
        // (d3 - d1)/2, every other target point is still farther than the nearer
        // of those two (triangle inequality), so the match is the better of the
        // two — two distance evaluations instead of a search.  Exact: margins cover
        // rounding, equal distances fall back on the row rule or on a new search.
        // Net displacement, so a pair that oscillates in a limit cycle (the usual
        // reason for running to max_iterations) stops searching too.
        // Anchor and budget are kept in SINGLE precision (registers: four rows per
        // thread must not spill): the budget is lowered by the rounding of the
        // anchor (2^-24 of each coordinate) and rounded down, so the test stays
        // conservative — it can only send a row to a search it did not need.
        float ax[ICP2_SMAX], ay[ICP2_SMAX], budget[ICP2_SMAX];
        int pos2[ICP2_SMAX];
#pragma unroll
        for (int s = 0; s < ICP2_SMAX; ++s) { ax[s] = 0.0f; ay[s] = 0.0f; budget[s] = -1.0f; pos2[s] = -1; }
        const int S = (N + THREADS - 1) / THREADS;
#pragma unroll
        for (int s = 0; s < ICP2_SMAX; ++s) {
            const int n = s * THREADS + tid;
            px[s] = 0.0; py[s] = 0.0; pos[s] = -1;                       // -1: no previous match yet
            if (n < N && RESUME) {
                const double2 v = a.st_xy[(size_t)b * a.st_stride + n];
                px[s] = v.x; py[s] = v.y; pos[s] = a.st_pos[(size_t)b * a.st_stride + n];
            } else if (n < N) {
                const double x = src[2 * n], y = src[2 * n + 1];
                if (a.has_init) {                           // source @ R_init.T + t_init
                    const double* in = a.init + (size_t)b * 6;
                    double sx = 0.0, sy = 0.0;
                    sx += x * in[0]; sx += y * in[1]; sx += in[4];
                    sy += x * in[2]; sy += y * in[3]; sy += in[5];
                    px[s] = sx; py[s] = sy;
                } else { px[s] = x; py[s] = y; }
            }
        }
        const bool has_corr = a.max_corr_dist >= 0.0;
        const double max_corr_sq = wave_uniform(a.max_corr_dist * a.max_corr_dist);   // icp.py:169
        const int need = max(3, N / 10);                                 // icp.py:186
        __syncthreads();
        if constexpr (FAR) {
            sweepf_build_tree(lds_sq, M, lds_tree, tree_leaves, tid, THREADS);
            __syncthreads();
        }
        // largest |projection| of the target (the copy is sorted along it): rounding slack of the diagonal axes
        const double2 c_lo = sxy[0], c_hi = sxy[M - 1];
        const double uabs = wave_uniform(fmax(fabs(proj(dir, c_lo.x, c_lo.y)), fabs(proj(dir, c_hi.x, c_hi.y))));
        if constexpr (FILT) {
            filt.rt = wave_uniform(__int_as_float(rt_bits) * 1.000001f);
            filt.ut = wave_uniform(fmaxf(fabsf(lds_sq[0].z), fabsf(lds_sq[M - 1].z)) * 1.000001f);     // the images are sorted like the keys
        }

        double e_part = 0.0;          // this thread's share of the squared error of the step just applied
        if (RESUME) {
            // the squared residual of the last step of the first stage against the matches it used: the sums of the
            // apply loop below, term by term
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s) {
                if (!(s < S && s * THREADS + tid < N)) continue;
                const double2 q = sxy[pos[s]];
                const double ex = q.x - px[s], ey = q.y - py[s];
                double se = 0.0;
                se += ex * ex;
                se += ey * ey;
                e_part += se;
            }
        }
        bool stopped = false;
        // a first launch hands a pair whose FIRST step leaves a mean squared error above far_d2 to the far continuation, which
        // runs iterations 2 .. max_iterations - 1 (the test rides on the convergence test of iteration 1: no arithmetic of its own)
        const double far_err = !RESUME && a.max_iterations > 2 && N <= ICP2_FAR_THREADS * ICP2_FAR_SMAX && M <= ICP2_FAR_POINTS
                                   ? a.far_d2 : __builtin_inf();
#ifdef ICPMI_DIAG
        double dg_nn = 0, dg_red = 0, dg_lead = 0, dg_apply = 0;
#endif
        const int it_end = RESUME ? a.max_iterations : min(a.max_iterations, a.it_limit);
        for (int it = RESUME ? a.it_begin : 0; it < it_end; ++it) {
            DIAG_T(c0);
            // ── correspondences: exact sweep search in LDS, icp.py:179 ───────
            // A row whose net displacement since its last search is inside its budget keeps one of its two
            // candidates (two distances); the others search.
            bool srch[ICP2_SMAX];
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s) {
                const bool valid = s < S && s * THREADS + tid < N;
                // |dx| + |dy| >= the distance between the row and its anchor
                // (Tried in round 3: testing delta + dq < d3 with the CURRENT distance dq to the better kept candidate — about
                // twice as generous, 0.7 % instead of 2 % of the rows of a limit cycle search — costs 8 registers (a spill at
                // six waves per SIMD) and the distances of the rows that then search after all: no faster, 5.11 against 5.07 ms.)
                bool within;
                if constexpr (FAR) {
                    // The far continuation (a search costs ten times the usual) keeps d3 itself — lowered by the same
                    // margins — instead of the budget, and tests with the row's CURRENT distance dq to the better of its
                    // two candidates: every other target point was at least d3 from the anchor, so it is at least
                    // d3 - delta from the row now; dq < d3 - delta keeps the match.  About twice as generous as
                    // delta < (d3 - d1) / 2; both distances are evaluated for every row (they are cheap here).
                    within = false;
                    if (valid && budget[s] > 0.0f) {
                        const int pa = pos[s], pb = pos2[s] >= 0 ? pos2[s] : pos[s];
                        const double2 c = sxy[pa], e = sxy[pb];
                        const double dx = px[s] - c.x, dy = py[s] - c.y;
                        const double ex = px[s] - e.x, ey = py[s] - e.y;
                        double q2 = 0.0, w2 = 0.0;
                        q2 += dx * dx;
                        q2 += dy * dy;
                        w2 += ex * ex;
                        w2 += ey * ey;
                        const double g = (double)budget[s] - (fabs(px[s] - (double)ax[s]) + fabs(py[s] - (double)ay[s])) * 1.000000001;
                        within = g > 0.0 && fmin(q2, w2) * 1.000000000001 < g * g * 0.999999999999;
                    }
                } else {
                    within = valid && (fabs(px[s] - (double)ax[s]) + fabs(py[s] - (double)ay[s])) * 1.000000001 < (double)budget[s];
                }
                srch[s] = valid && !within;
                if (within) {
                    // straight-line: a missing second candidate stands in as the first (never better), and only an
                    // exact tie of the two distances takes a branch (to compare the rows)
                    const int pa = pos[s], pb = pos2[s] >= 0 ? pos2[s] : pos[s];
                    const double2 c = sxy[pa], e = sxy[pb];
                    const double dx = px[s] - c.x, dy = py[s] - c.y;
                    const double ex = px[s] - e.x, ey = py[s] - e.y;
                    double q2 = 0.0, w2 = 0.0;
                    q2 += dx * dx;
                    q2 += dy * dy;
                    w2 += ex * ex;
                    w2 += ey * ey;
                    bool second_wins = w2 < q2;
                    if (w2 == q2 && pb != pa) {
                        if constexpr (FILT) second_wins = sweepf_row(lds_sq[pb]) < sweepf_row(lds_sq[pa]);
                        else second_wins = sorig[pb] < sorig[pa];
                    }
                    pos[s] = second_wins ? pb : pa;
                    pos2[s] = pos2[s] >= 0 ? (second_wins ? pa : pb) : -1;
                }
            }
            // (a continuation starts at iteration STAGE1_ITERATIONS > ICP2_PLAIN_ITERS: its instantiation holds the top-two search only)
            if (!RESUME && it < ICP2_PLAIN_ITERS) {
                // the first steps move every row by more than any budget: plain 1-NN (smallest window), started at
                // the row's own projection — the previous match only seeds the bound (it lies a whole step away)
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s)
                    if (srch[s]) {
                        double d2s;
                        if constexpr (FILT) {
                            if (ICP2_PK && __popcll(__ballot(true)) >= ICP2_PK_MIN)
                                pos[s] = sweepf_nn_pk(lds_sq, sxy, filt, M, dir, uabs, px[s], py[s], pos[s], it < ICP2_PLAIN_CENTRED);
                            else pos[s] = sweepf_nn(lds_sq, sxy, filt, M, dir, uabs, px[s], py[s], pos[s], it < ICP2_PLAIN_CENTRED, d2s);
                        }
                        else pos[s] = sweep_nn(sxy, sorig, M, dir, uabs, px[s], py[s], pos[s], it < ICP2_PLAIN_CENTRED, d2s);
                    }
            } else {
                const bool centred = !RESUME && it < ICP2_CENTRED_ITERS;
                // The far continuation packs the searching rows of a wave: a far search costs thousands of instructions and
                // a wave pays for one per row slot in which ANY lane searches — with a quarter of the rows searching, both
                // slots, at a quarter of the lanes.  The queries of all slots go through the wave's stretch of LDS to the
                // low lanes (a slot's rows stay neighbours there) and the answers come back the same way: one round of
                // searches per 64 searching rows.  Wave-private: no barrier, LDS operations of a wave complete in order.
                int slot[ICP2_SMAX];
                if constexpr (FAR) {
                    int nq = 0;
#pragma unroll
                    for (int s = 0; s < ICP2_SMAX; ++s) {
                        const unsigned long long bal = __ballot(srch[s]);
                        slot[s] = nq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                        nq += __popcll(bal);
                        if (srch[s]) {
                            FarSlot e;
                            e.in.x = px[s]; e.in.y = py[s]; e.in.seed = pos[s]; e.in.pad = 0;
                            far_q[slot[s]] = e;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int j = tid & (ICPMI_WAVE - 1); j < nq; j += ICPMI_WAVE) {
                        const FarSlot e = far_q[j];
#ifdef ICPMI_DIAG
                        const Top2 r = sweepf_top2_far(lds_sq, sxy, lds_tree, tree_leaves, filt, M, dir, uabs, e.in.x, e.in.y, e.in.seed, &res[11]);
#elif ICP2_FAR_PK
                        const Top2 r = sweepf_top2_far_pk(lds_sq, sxy, lds_tree, tree_leaves, filt, M, dir, uabs, e.in.x, e.in.y, e.in.seed);
#else
                        const Top2 r = sweepf_top2_far(lds_sq, sxy, lds_tree, tree_leaves, filt, M, dir, uabs, e.in.x, e.in.y, e.in.seed);
#endif
                        FarSlot o;
                        o.out.s1 = r.s1; o.out.s3 = r.s3; o.out.p1 = r.p1; o.out.p2 = r.p2;
                        far_q[j] = o;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s)
                    if (srch[s]) {
                        Top2 t2;
                        // (Tried first: ONE queue for the whole workgroup, worked off by its lanes in order — 7.8 against 6.4 ms
                        // on the 3 m / 20 degree candidates: the lanes of a wave then hold rows from all over the scan and
                        // their descents diverge; two barriers more per iteration.)
                        if constexpr (FAR) {
                            const FarSlot o = far_q[slot[s]];
                            t2.p1 = o.out.p1; t2.p2 = o.out.p2; t2.s1 = o.out.s1; t2.s2 = o.out.s1; t2.s3 = o.out.s3;
                        }
                        else if constexpr (FILT) {
                            // the packed walk costs a wave the same whether one lane searches or all do; the branching walk
                            // is cheap while few do (its exact path is then rarely entered): chosen per wave by the count
                            if (ICP2_PK && __popcll(__ballot(true)) >= ICP2_PK_MIN)
                                t2 = sweepf_top2_pk(lds_sq, sxy, filt, M, dir, uabs, px[s], py[s], pos[s], centred);
                            else t2 = sweepf_top2(lds_sq, sxy, filt, M, dir, uabs, px[s], py[s], pos[s], centred);
                        }
                        else t2 = sweep_top2(sxy, sorig, M, dir, uabs, px[s], py[s], pos[s], centred);
                        pos[s] = t2.p1; pos2[s] = t2.p2;
                        float bf;
                        if constexpr (FAR) {
                            const double d1 = sqrt(t2.s1), d3 = sqrt(t2.s3);
                            // minus the rounding of the single-precision anchor; rounded down                  (see the test above)
                            bf = (float)(d3 - 1e-13 * (d3 + d1) - 1.3e-7 * (fabs(px[s]) + fabs(py[s])));
                        } else {
                            // (d3 - d1) / 2 minus the rounding of the single-precision anchor, from float32 roots rounded the safe way —
                            // d3 down, d1 up (conversion 2^-24, v_sqrt_f32 one ulp: 1.2e-7 in all, 3e-7 taken) —: two float64 roots were a
                            // tenth of a top-two search; the budget only has to be a lower bound (round 4)
                            const float r3 = __builtin_amdgcn_sqrtf((float)t2.s3) * 0.9999997f;
                            const float r1 = __builtin_amdgcn_sqrtf((float)t2.s1) * 1.0000003f + 1e-30f;
                            bf = (r3 - r1) * 0.4999999f - 1.3e-7f * (fabsf((float)px[s]) + fabsf((float)py[s]));
                        }
                        bf = bf - fabsf(bf) * 1e-6f;
                        budget[s] = t2.s3 < __builtin_inf() ? bf : __builtin_inff();
                        ax[s] = (float)px[s]; ay[s] = (float)py[s];
#ifdef ICPMI_DIAG
                        atomicAdd(&res[8], 1.0);             // diag: number of searches run by this pair
#endif
                    }
            }
#ifdef ICPMI_DIAG
            __syncthreads();          // diag only: charge the slowest wave's search to the search phase
#endif
            DIAG_T(c1);
#ifdef ICPMI_DIAG
            unsigned long long c2 = 0;
#endif
            // rows that take part in the solve: all valid rows, or those within max_corr_dist of their match
            // (icp.py:184-185: nn_dists**2 < max_corr_dist**2, the distance squared again after its root)
            bool in[ICP2_SMAX];
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s) in[s] = s < S && s * THREADS + tid < N;
            if (use_p2l) {
                // ── point-to-line normal equations, icp.py:88-104, + carried error ─
                double acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s) {
                    if (!in[s]) continue;
                    const double2 q = sxy[pos[s]], nm = snrm[pos[s]];
                    const double dx = px[s] - q.x, dy = py[s] - q.y;
                    if (has_corr) {                                   // the search's squared distance, recomputed bit for bit
                        double s2 = 0.0;
                        s2 += dx * dx;
                        s2 += dy * dy;
                        const double dist = sqrt(s2);
                        if (!(dist * dist < max_corr_sq)) continue;
                    }
                    const double c = nm.y * px[s] - nm.x * py[s];
                    const double bi = -(nm.x * dx + nm.y * dy);
                    acc[0] += c * c;       acc[1] += c * nm.x;    acc[2] += c * nm.y;
                    acc[3] += nm.x * nm.x; acc[4] += nm.x * nm.y; acc[5] += nm.y * nm.y;
                    acc[6] += c * bi;      acc[7] += nm.x * bi;   acc[8] += nm.y * bi;
                    acc[9] += 1.0;
                }
                acc[10] = e_part;
                wave_totals<11>(redA, acc);
                __syncthreads();
                DIAG_SET(c2);
                if (lead) {
                    combine_totals<11>(redA, NWAVES, acc);
                    const int fin = finish_step(ctrl, it, acc[10], acc[9], N, has_corr, need, a.error_threshold, tid == 0, far_err);
                    const bool stop = fin == FIN_STOP;
                    if (!stop) {
                        double A[3][3] = {{acc[0], acc[1], acc[2]}, {acc[1], acc[3], acc[4]}, {acc[2], acc[4], acc[5]}};
                        double rhs[3] = {acc[6], acc[7], acc[8]}, x[3];
                        double r[4], t[2];
                        if (solve3(A, rhs, x)) {
                            double st, ct;
                            sincos_step(x[0], st, ct);                             // icp.py:110-114
                            r[0] = ct; r[1] = -st; r[2] = st; r[3] = ct; t[0] = x[1]; t[1] = x[2];
                        } else {
                            r[0] = 1.0; r[1] = 0.0; r[2] = 0.0; r[3] = 1.0; t[0] = 0.0; t[1] = 0.0;
                        }
                        // accumulate totals, icp.py:210-211
                        if (tid == 0) {
                            accumulate_step(ctrl, r, t);
                            ctrl[CTRL_R] = r[0]; ctrl[CTRL_R + 1] = r[1]; ctrl[CTRL_R + 2] = r[2]; ctrl[CTRL_R + 3] = r[3];
                            ctrl[CTRL_T] = t[0]; ctrl[CTRL_T + 1] = t[1];
                        }
                    }
                    if (tid == 0) ctrl[CTRL_STOP] = (double)fin;
                }
                __syncthreads();
            } else {
                // ── point-to-point: centroids (+ carried error), icp.py:197-198 ──
                double m[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s) {
                    if (!in[s]) continue;
                    const double2 q = sxy[pos[s]];
                    if (has_corr) {                                   // the search's squared distance, recomputed bit for bit
                        const double dx = px[s] - q.x, dy = py[s] - q.y;
                        double s2 = 0.0;
                        s2 += dx * dx;
                        s2 += dy * dy;
                        const double dist = sqrt(s2);
                        in[s] = dist * dist < max_corr_sq;
                        if (!in[s]) continue;
                    }
                    m[0] += px[s]; m[1] += py[s]; m[2] += q.x; m[3] += q.y; m[4] += 1.0;
                }
                m[5] = e_part;
                wave_totals<6>(redA, m);
                __syncthreads();
                DIAG_SET(c2);
                if (lead) {
                    combine_totals<6>(redA, NWAVES, m);
                    const int fin = finish_step(ctrl, it, m[5], m[4], N, has_corr, need, a.error_threshold, tid == 0, far_err);
                    if (tid == 0) {
                        ctrl[CTRL_STOP] = (double)fin;
                        ctrl[CTRL_MP] = m[0] / m[4]; ctrl[CTRL_MP + 1] = m[1] / m[4];
                        ctrl[CTRL_MQ] = m[2] / m[4]; ctrl[CTRL_MQ + 1] = m[3] / m[4];
                    }
                }
                __syncthreads();
                if (ctrl[CTRL_STOP] == (double)FIN_STOP) { stopped = true; break; }
                const double mpx = ctrl[CTRL_MP], mpy = ctrl[CTRL_MP + 1], mqx = ctrl[CTRL_MQ], mqy = ctrl[CTRL_MQ + 1];
                // centred cross-covariance, icp.py:199-201
                double W[4] = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ICP2_SMAX; ++s) {
                    if (!in[s]) continue;
                    const double2 q = sxy[pos[s]];
                    const double pcx = px[s] - mpx, pcy = py[s] - mpy, qcx = q.x - mqx, qcy = q.y - mqy;
                    W[0] += pcx * qcx; W[1] += pcx * qcy; W[2] += pcy * qcx; W[3] += pcy * qcy;
                }
                wave_totals<4>(redB, W);
                __syncthreads();
                if (lead) {
                    combine_totals<4>(redB, NWAVES, W);
                    double r[4], t[2];
                    kabsch2(W, r);                                                 // icp.py:202-206
                    double s0 = 0.0, s1 = 0.0;
                    s0 += r[0] * mpx; s0 += r[1] * mpy;
                    s1 += r[2] * mpx; s1 += r[3] * mpy;
                    t[0] = mqx - s0; t[1] = mqy - s1;                              // icp.py:207
                    if (tid == 0) {
                        accumulate_step(ctrl, r, t);
                        ctrl[CTRL_R] = r[0]; ctrl[CTRL_R + 1] = r[1]; ctrl[CTRL_R + 2] = r[2]; ctrl[CTRL_R + 3] = r[3];
                        ctrl[CTRL_T] = t[0]; ctrl[CTRL_T + 1] = t[1];
                    }
                }
                __syncthreads();
            }
            DIAG_T(c3);
            if (ctrl[CTRL_STOP] == (double)FIN_STOP) { stopped = true; break; }
            // ── apply to ALL rows; squared residual against this search's matches, icp.py:212-215 ─
            const double r0 = ctrl[CTRL_R], r1 = ctrl[CTRL_R + 1], r2 = ctrl[CTRL_R + 2], r3 = ctrl[CTRL_R + 3];
            const double t0 = ctrl[CTRL_T], t1 = ctrl[CTRL_T + 1];
            e_part = 0.0;
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s) {
                if (!(s < S && s * THREADS + tid < N)) continue;
                const double2 q = sxy[pos[s]];
                double nx = 0.0, ny = 0.0;
                nx += px[s] * r0; nx += py[s] * r1; nx += t0;
                ny += px[s] * r2; ny += py[s] * r3; ny += t1;
                px[s] = nx; py[s] = ny;
                const double ex = q.x - nx, ey = q.y - ny;
                double se = 0.0;
                se += ex * ex;
                se += ey * ey;
                e_part += se;
            }
            if (!RESUME && it == 1 && ctrl[CTRL_STOP] == (double)FIN_FAR) break;
            DIAG_T(c4);
#ifdef ICPMI_DIAG
            DIAG_ADD(dg_nn, c0, c1); DIAG_ADD(dg_red, c1, c2); DIAG_ADD(dg_lead, c2, c3); DIAG_ADD(dg_apply, c3, c4);
            if (tid == 0) { res[4] = dg_nn; res[5] = dg_red; res[6] = dg_lead; res[7] = dg_apply; }
#endif
        }
        const bool far_parked = !RESUME && !stopped && ctrl[CTRL_STOP] == (double)FIN_FAR;      // left after iteration 1
        if (!RESUME && !stopped && (far_parked || it_end < a.max_iterations)) {
            // parked for the second stage: rows and matches as they are, totals through the result record
#pragma unroll
            for (int s = 0; s < ICP2_SMAX; ++s) {
                const int n = s * THREADS + tid;
                if (s < S && n < N) {
                    a.st_xy[(size_t)b * a.st_stride + n] = make_double2(px[s], py[s]);
                    a.st_pos[(size_t)b * a.st_stride + n] = pos[s];
                }
            }
            if (tid == 0) {
                ctrl[CTRL_STATUS] = (double)ICP2_ST_PARKED;
                if (far_parked) a.far_list[atomicAdd(a.far_count, 1)] = b;
                else a.list[atomicAdd(a.list_count, 1)] = b;
            }
        } else if (!stopped && a.max_iterations > 0) {
            // the last step's error has not been reduced yet: icp.py:215-223 for it = max_iterations - 1
            double e[1] = {e_part};
            __syncthreads();                      // redA may still be read by the lead wave of the last iteration
            wave_totals<1>(redA, e);
            __syncthreads();
            if (lead) {
                combine_totals<1>(redA, NWAVES, e);
                if (tid == 0) {
                    const double err = e[0] / (double)N, delta = fabs(ctrl[CTRL_PREV] - err);
                    ctrl[CTRL_ERR] = err; ctrl[CTRL_DELTA] = delta;
                    ctrl[CTRL_ITERS] = (double)a.max_iterations;
                    if (delta < a.error_threshold) ctrl[CTRL_STATUS] = (double)ICPMI_ST_CONVERGED;
                }
            }
        }
    }
    if (tid == 0) {
#ifndef ICPMI_DIAG
#pragma unroll
        for (int i = 0; i < ICPMI_RES_DOUBLES; ++i) res[i] = 0.0;
#endif
        res[0] = ctrl[CTRL_RT]; res[1] = ctrl[CTRL_RT + 1]; res[2] = ctrl[CTRL_RT + 2]; res[3] = ctrl[CTRL_RT + 3];
        res[ICPMI_RES_T] = ctrl[CTRL_TT]; res[ICPMI_RES_T + 1] = ctrl[CTRL_TT + 1];
        res[ICPMI_RES_ERR] = ctrl[CTRL_ERR];
        res[ICPMI_RES_DELTA] = ctrl[CTRL_DELTA];
        res[ICPMI_RES_ITERS] = ctrl[CTRL_ITERS];
        res[ICPMI_RES_STATUS] = ctrl[CTRL_STATUS];
    }
}

template <int THREADS, int ICP2_SMAX, bool TGT_LDS, bool FILT>
__global__ __launch_bounds__(THREADS, THREADS == 768 ? 6 : 4) void icp2_fused_kernel(Icp2Args a) {   // 4 waves/SIMD: 2 x 512 or 1 x 1024 per CU; 6: 2 x 768
    icp2_pair<THREADS, ICP2_SMAX, TGT_LDS, FILT, false>(a, blockIdx.x);                              // one pair per workgroup
}

// the launch for wide clouds when the first one listed them: the workgroups walk that list
template <int THREADS, int ICP2_SMAX, bool TGT_LDS, bool FILT>
__global__ __launch_bounds__(THREADS, THREADS == 768 ? 6 : 4) void icp2_wide_kernel(Icp2Args a) {
    const int count = *a.wide_count;
    for (int j = blockIdx.x; j < count; j += gridDim.x) {
        icp2_pair<THREADS, ICP2_SMAX, TGT_LDS, FILT, false>(a, __builtin_amdgcn_readfirstlane(a.wide_list[j]));
        __syncthreads();                                    // LDS is staged again for the next pair
    }
}

// second stage of a two-stage run: workgroup j continues the j-th parked pair.  No loop over the list here: with one the
// 768 x 2 shape needs 93 registers instead of 79 and spills 56 B per lane at its budget of 80 (six waves per SIMD); the
// launcher starts an eighth of the batch's pairs as workgroups (about one pair in twelve is parked; a workgroup beyond the
// list leaves at once) and ...
template <int THREADS, int ICP2_SMAX, bool TGT_LDS, bool FILT>
__global__ __launch_bounds__(THREADS, THREADS == 768 ? 6 : 4) void icp2_resume_kernel(Icp2Args a) {
    if ((int)blockIdx.x >= *a.list_count) return;
    icp2_pair<THREADS, ICP2_SMAX, TGT_LDS, FILT, true>(a, __builtin_amdgcn_readfirstlane(a.list[blockIdx.x]));
}

// ... the parked pairs beyond that (none, usually) are walked by the few workgroups of this one
template <int THREADS, int ICP2_SMAX, bool TGT_LDS, bool FILT>
__global__ __launch_bounds__(THREADS, THREADS == 768 ? 6 : 4) void icp2_resume_rest_kernel(Icp2Args a, int first) {
    const int count = *a.list_count;
    for (int j = first + blockIdx.x; j < count; j += gridDim.x) {
        // the pair index in a scalar register: everything addressed through it stays scalar
        icp2_pair<THREADS, ICP2_SMAX, TGT_LDS, FILT, true>(a, __builtin_amdgcn_readfirstlane(a.list[j]));
        __syncthreads();                                    // LDS is staged again for the next pair
    }
}

// continuation of the pairs found far from their target (Icp2Args::far_list): one shape for all of them
__global__ __launch_bounds__(ICP2_FAR_THREADS, 4) void icp2_far_kernel(Icp2Args a) {
    const int count = *a.far_count;
    for (int j = blockIdx.x; j < count; j += gridDim.x) {
        icp2_pair<ICP2_FAR_THREADS, ICP2_FAR_SMAX, true, true, true, true>(a, __builtin_amdgcn_readfirstlane(a.far_list[j]));
        __syncthreads();                                    // LDS is staged again for the next pair
    }
}

// host side: called by icpmi_icp_batch (icp.hip) when a prepared buffer is given and everything fits
int launch_icp2(const double* pts, const int32_t* off, const int32_t* cnt, const int32_t* ps, const int32_t* pt,
                int n_pairs, int max_src_n, int max_tgt_n, int total_rows, const icpmi_icp_params* p, const double* init,
                double* results, const void* prepared, void* workspace, size_t workspace_bytes, hipStream_t st) {
    Icp2Args a;
    a.it_begin = 0; a.it_limit = 0x7fffffff; a.resume = 0;
    a.st_xy = nullptr; a.st_pos = nullptr; a.list = nullptr; a.list_count = nullptr; a.st_stride = 0;
    a.wide_list = nullptr; a.wide_count = nullptr;
    a.far_list = nullptr; a.far_count = nullptr; a.far_d2 = __builtin_inf();
    const unsigned char* b = (const unsigned char*)prepared;
    a.pts = pts; a.off = off; a.cnt = cnt; a.pair_src = ps; a.pair_tgt = pt; a.init = init; a.results = results;
    a.g_sxy = (const double2*)b;
    a.g_snrm = (const double2*)(b + (size_t)total_rows * 16);
    a.g_sorig = (const int32_t*)(b + (size_t)total_rows * 32);
    a.g_skey = (const float*)(b + (size_t)total_rows * 36);
    a.g_dir = (const int32_t*)(b + (size_t)total_rows * 40);
    a.error_threshold = p->error_threshold; a.max_corr_dist = p->max_corr_dist;
    a.max_iterations = p->max_iterations; a.method = p->method; a.has_init = p->has_init;
    const bool in_lds = max_tgt_n <= 4096;
    // Workgroup shape by source size (rows per thread bounded by the instantiation).  option ICP2_SHAPE = "TxS"
    // (threads x rows per thread, one of the instantiations below) overrides the choice for experiments;
    // option ICP2_FILTER = 0 turns the single-precision filter off.
    int T = 0, SM = 0;
    if (const char* env = option("ICP2_SHAPE")) {
        if (sscanf(env, "%dx%d", &T, &SM) != 2) { T = 0; SM = 0; }
    }
    const char* fenv = option("ICP2_FILTER");
    const bool want_filter = !(fenv && fenv[0] == '0');
#define ICPMI_ICP2_GO(TT, SS)                                                                                                    \
    do {                                                                                                                         \
        if (!in_lds) ICPMI_ICP2_GO2(TT, SS, false, false);                                                                      \
        else if (filter) ICPMI_ICP2_GO2(TT, SS, true, true);                                                                    \
        else ICPMI_ICP2_GO2(TT, SS, true, false);                                                                               \
    } while (0)
#define ICPMI_ICP2_GO2(TT, SS, L, F)                                                                                             \
    do {                                                                                                                         \
        if (dyn_lds((const void*)icp2_fused_kernel<TT, SS, L, F>, lds) != hipSuccess) return ICPMI_ERR_HIP;                      \
        if (pass == 1 && a.wide_list) {                     /* the listed pairs: few workgroups walking the list */              \
            if (dyn_lds((const void*)icp2_wide_kernel<TT, SS, L, F>, lds) != hipSuccess) return ICPMI_ERR_HIP;                   \
            hipStream_t ws = st;                            /* beside the second stage when the side stream is there */          \
            if (forked && hipStreamWaitEvent(side->stream[0], side->fork, 0) == hipSuccess) ws = side->stream[0];                 \
            icp2_wide_kernel<TT, SS, L, F><<<wide_grid, TT, lds, ws>>>(a);                                                       \
            if (ws != st && (hipEventRecord(side->join[0], ws) != hipSuccess ||                                                  \
                             hipStreamWaitEvent(st, side->join[0], 0) != hipSuccess)) {                                          \
                (void)hipStreamSynchronize(ws);             /* never return with a launch the caller cannot order against */     \
                return ICPMI_ERR_HIP;                                                                                            \
            }                                                                                                                    \
        } else icp2_fused_kernel<TT, SS, L, F><<<n_pairs, TT, lds, st>>>(a);                                                      \
        if (two_stage && pass == 0 && a.wide_list && side) forked = hipEventRecord(side->fork, st) == hipSuccess;                \
        if (two_stage && pass == 0) {                       /* the parked pairs, all started together */                         \
            Icp2Args c = a;                                                                                                      \
            c.resume = 1; c.it_begin = a.it_limit; c.it_limit = 0x7fffffff;                                                      \
            if (dyn_lds((const void*)icp2_resume_kernel<TT, SS, L, F>, lds) != hipSuccess) return ICPMI_ERR_HIP;                 \
            icp2_resume_kernel<TT, SS, L, F><<<stage2_grid, TT, lds, st>>>(c);                                                   \
            if (stage2_grid < n_pairs) {                    /* more parked pairs than workgroups: the rest of the list */         \
                if (dyn_lds((const void*)icp2_resume_rest_kernel<TT, SS, L, F>, lds) != hipSuccess) return ICPMI_ERR_HIP;        \
                icp2_resume_rest_kernel<TT, SS, L, F><<<256, TT, lds, st>>>(c, stage2_grid);                                     \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
    a.n_lo = -1; a.m_lo = 0; a.skip_over = 0;
    int T2 = 0, SM2 = 0;                // second launch for the pairs the first shape cannot hold
    const bool many = n_pairs >= 1024;
    if (T == 0) {
        // A voxel-filtered 2 048-beam scan keeps ~1 400 rows: 768 threads x 2 rows, two workgroups per CU (6 waves per
        // SIMD at 80 registers; 512 x 3 at 4 waves per SIMD is 5 % slower) — one pair's serial solve and barrier waits
        // overlap the other's search.  Clouds that keep more than 1 536 rows go to a second launch (1 024 threads).
        // With few pairs (less than two per CU) 1 024 threads x 2 rows finish a pair soonest.
        if (max_src_n <= 1024) { T = 512; SM = 2; }
        else if (many) { T = 768; SM = 2; }
        else if (max_src_n <= 2048) { T = 1024; SM = 2; }
        else { T = 1024; SM = 4; }
    }
    // LDS copy of the target: 36 B per point, 48 B with the float32 images of the filter.  Two workgroups of the
    // 512-thread shapes share a CU only up to 1 536 filter points (2 x 73.7 KB): larger targets, like larger
    // sources, are left to the second launch.  The filter needs <= 2 048 points (96 KB, one workgroup per CU).
    int cap1 = ((T == 512 || T == 768) && many && in_lds && want_filter && max_tgt_n > 1536) ? 1536 : max_tgt_n;
    if (T * SM < max_src_n || cap1 < max_tgt_n) { T2 = 1024; SM2 = max_src_n <= 2048 ? 2 : 4; a.skip_over = 1; }
    // Two stages for a large batch (see Icp2Args): needs the caller's workspace for the parked state.  Below ~1 000 pairs
    // every long pair starts within the first two rounds anyway and the stages only add their own cost (512 pairs: 0.70
    // against 0.65 ms).  About one pair
    // in thirteen of a loop-closure batch runs to the iteration limit; 12 iterations settle the others.
    // option ICP2_STAGES = 1 keeps one launch (experiments, and the test that both give the same bits).
    constexpr int STAGE1_ITERATIONS = ICP2_STAGE1_ITERS;                   // measured 6.12 / 5.42 / 5.36 / 5.38 / 5.39 ms at 8 / 10 / 12 / 14 / 16
    const char* senv = option("ICP2_STAGES");
    const size_t st_rows = (size_t)n_pairs * (size_t)max_src_n;
    const size_t st_bytes = st_rows * 20 + (size_t)n_pairs * 12 + 64;
    // point-to-line only: its pairs either settle within ~10 iterations or circle to the limit; point-to-point pairs all
    // take 25-40 and would all be parked (ICP2_STAGES = 2 forces the stages for them too: tests)
    const bool have_ws = workspace && workspace_bytes >= st_bytes;
    const bool two_stage = many && have_ws && p->max_iterations >= 2 * STAGE1_ITERATIONS &&
                           !(senv && senv[0] == '1') && (p->method == ICPMI_POINT_TO_LINE || (senv && senv[0] == '2'));
    // second-stage workgroups: an eighth of the pairs, one parked pair each (about one pair in twelve is parked; the others
    // leave at once), then 256 workgroups that walk whatever the list holds beyond that
    const int stage2_grid = n_pairs / 8 > 256 ? n_pairs / 8 : 256;
    // the launch for wide clouds: a thirty-second (a launch of 4 096 workgroups that find an empty list still takes 100 us)
    const int wide_grid = n_pairs < 256 ? n_pairs : (n_pairs / 32 > 256 ? n_pairs / 32 : 256);
    // option ICP2_FAR = the mean squared error (m^2) of the first step above which a pair goes to the far continuation
    // (default 1: one metre rms — the candidates ICP converges from by itself stay below it: with 0.5 some of them are
    // sent over and the 16 384-pair batch takes 5.7 instead of 5.05 ms; 0 = never)
    double far_d2 = 1.0;
    if (const char* e = option("ICP2_FAR")) far_d2 = atof(e);
    const bool far_ok = have_ws && in_lds && want_filter && far_d2 > 0.0 && p->max_iterations > 2;
    // the far continuation's LDS: the filter layout + the box hierarchy (2 B per point at most) + a slot per row of its shape
    // (every wave owns the stretch behind its own lanes, whatever the batch's source sizes).  Asked for BEFORE the first
    // launch: without it (another ARCH than gfx950's 160 KB) no pair is parked for a kernel that could not start.
    int far_cap = 64;
    while (far_cap < max_tgt_n && far_cap < ICP2_FAR_POINTS) far_cap <<= 1;
    const size_t far_lds = (size_t)far_cap * 50 + 32 + sizeof(FarSlot) * (size_t)(ICP2_FAR_THREADS * ICP2_FAR_SMAX);
    const bool far_go = far_ok && dyn_lds((const void*)icp2_far_kernel, far_lds) == hipSuccess;
    if (have_ws && (two_stage || T2 || far_go)) {
        unsigned char* w = (unsigned char*)workspace;
        a.st_xy = (double2*)w;
        a.st_pos = (int32_t*)(w + st_rows * 16);
        a.list = (int32_t*)(w + st_rows * 20);
        a.wide_list = a.list + n_pairs;
        a.far_list = a.wide_list + n_pairs;
        a.list_count = a.far_list + n_pairs;
        a.wide_count = a.list_count + 1;
        a.far_count = a.list_count + 2;
        a.st_stride = max_src_n;
        if (!T2) { a.wide_list = nullptr; a.wide_count = nullptr; }
        if (far_go) a.far_d2 = far_d2;
        else { a.far_list = nullptr; a.far_count = nullptr; }
        if (hipMemsetAsync(a.list_count, 0, 3 * sizeof(int32_t), st) != hipSuccess) return ICPMI_ERR_HIP;
    }
    // The launch for wide clouds (a handful of pairs, ~0.1 ms at a few per cent of the chip) only needs the first stage's
    // list: it runs on a side stream beside the second stage and joins the caller's stream afterwards.
    // (state.hip: the library's side streams, one set per device; the lock is held for the fork / launch / join sequence)
    SideLock side_lock;
    Side* const side = side_lock.side;
    bool forked = false;
    for (int pass = 0; pass < 2; ++pass) {
        a.it_limit = two_stage && pass == 0 ? STAGE1_ITERATIONS : 0x7fffffff;
        if (pass == 1) {
            if (!T2) break;
            a.n_lo = T * SM; a.m_lo = cap1; a.skip_over = 0; T = T2; SM = SM2; cap1 = max_tgt_n;
        }
        int cap = 64;
        while (cap < cap1) cap <<= 1;
        if (cap1 > 1024 && cap1 <= 1536) cap = 1536;
        a.lds_points = cap;
        const bool filter = in_lds && want_filter && cap <= 2048;
        const size_t lds = in_lds ? (filter ? (size_t)cap * 48 + 32 : (size_t)cap * 36) : 0;
        if (T == 512 && SM == 2) ICPMI_ICP2_GO(512, 2);
        else if (T == 512 && SM == 3) ICPMI_ICP2_GO(512, 3);
        else if (T == 512 && SM == 4) ICPMI_ICP2_GO(512, 4);
        else if (T == 768 && SM == 2) ICPMI_ICP2_GO(768, 2);
        else if (T == 1024 && SM == 2) ICPMI_ICP2_GO(1024, 2);
        else if (T == 1024 && SM == 4) ICPMI_ICP2_GO(1024, 4);
        else return ICPMI_ERR_ARG;
    }
#undef ICPMI_ICP2_GO
#undef ICPMI_ICP2_GO2
    if (a.far_list) {                                       // after every first launch (the wide one has joined the stream)
        Icp2Args c = a;
        c.resume = 1; c.it_begin = 2; c.it_limit = 0x7fffffff; c.skip_over = 0; c.n_lo = -1; c.m_lo = 0;
        c.lds_points = far_cap;
        icp2_far_kernel<<<n_pairs < 256 ? n_pairs : 256, ICP2_FAR_THREADS, far_lds, st>>>(c);
    }
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

}  // namespace icpmi

#ifdef ICPMI_DIAG
extern "C" int icpmi_diag_read(unsigned long long* out16) {   // diagnostic build only: read and clear the phase counters of sweep.hpp
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(icpmi::icpmi_dbg), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(icpmi::icpmi_dbg), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
