// sweep.hpp — exact nearest-neighbour search on a cloud sorted along one axis.
//
// The target cloud is sorted by a projection u(p) along one of four axes
// (x, y, x+y, x-y; the prepare kernel picks the one with the fewest expected
// candidates).  A query starts at the position of its own projection (binary
// search) and walks outwards on both sides.  |u(q) - u(c)| bounds the distance
// from below (by |q-c| on the two coordinate axes, by sqrt(2)|q-c| on the
// diagonals), so a side stops at the first candidate whose projection gap
// already exceeds the best squared distance found.  Exact: the winner, its
// float64 squared distance (direct differences, no FMA) and the lowest-row
// tie rule are those of the exhaustive scan in nn.hpp — only the candidates that
// cannot win are skipped.  Typical windows are ~10 candidates instead of the
// whole cloud.
#pragma once
#include "common.hpp"

namespace icpmi {

// projection value along axis `dir`: x, y, x + y, x - y.  Written as x*a + y*b with a in {1, 0} and b in
// {0, 1, 1, -1} (every product and sum exact for finite coordinates): the axis is uniform, but selecting
// the form inside a search loop costs a chain of scalar branches per candidate; the two factors are
// loop invariant.
__device__ __forceinline__ double proj(int dir, double x, double y) {
    const double a = dir == 1 ? 0.0 : 1.0;
    const double b = dir == 0 ? 0.0 : (dir == 3 ? -1.0 : 1.0);
    return x * a + y * b;
}

// Lower bound on the squared distance implied by a projection gap, with the
// rounding of the diagonal projections accounted for (conservative: a candidate
// is only skipped when it provably cannot win).  uq, uc: projections.
__device__ __forceinline__ bool gap_exceeds(int dir, double uq, double uc, double bound_d2) {
    const double du = fabs(uq - uc);
    if (dir < 2) return du * du > bound_d2;            // same subtraction as inside the distance: exact
    const double e = 4.5e-16 * (fabs(uq) + fabs(uc));  // fl(x +- y) is within 2^-53 relative of the true sum
    const double l = du - e;
    return l > 0.0 && l * l > bound_d2 * 2.000000000000002;   // (dx +- dy)^2 <= 2 (dx^2 + dy^2)
}

// first position in [0, m) whose projection is >= uq (m if none); sxy = sorted (x, y) pairs
__device__ __forceinline__ int sweep_lower_bound(const double2* sxy, int m, int dir, double uq) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double2 c = sxy[mid];
        if (proj(dir, c.x, c.y) < uq) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Half-width of the projection window that can still hold a winner: a candidate
// with |u(q) - u(c)| > prune_width(...) has squared distance > best.  One root
// per improvement of `best` buys a two-instruction test per candidate; the root
// is a SINGLE-precision one rounded up (a float64 sqrt is ~30 instructions, and
// a bound improves several times per search): (float)best is within 2^-24 of
// best, v_sqrt_f32 within 1 ulp, so root * 1.000001f exceeds the true root for
// every normal float; +1e-18 covers best below the normal float range (root <
// 1.1e-19) and the conversion of a finite best above FLT_MAX gives +inf.  The
// diagonals add the rounding of x +- y; uabs = max |u| over the target.
// Conservative: never skips a possible winner — a wider window only costs time.
__device__ __forceinline__ double prune_width(int dir, double best, double uq, double uabs) {
    if (dir < 2) return (double)(__builtin_amdgcn_sqrtf((float)best) * 1.000001f + 1e-18f);
    return (double)(__builtin_amdgcn_sqrtf((float)(best * 2.000000000000002)) * 1.000001f + 1e-18f) + 4.5e-16 * (fabs(uq) + uabs);
}

// The same search, also returning the squared distance of the SECOND nearest
// target point (inf if the cloud has one point).  The window is the one of the
// second-best distance, so nothing that could be first or second is skipped.
// The gap between the two distances is what lets the caller keep a match for
// later iterations without searching again (see icp2.hip, "movement budget").
__device__ __forceinline__ int sweep_nn2(const double2* sxy, const int32_t* sorig, int m, int dir, double uabs,
                                         double qx, double qy, int start, double& d2_out, double& second_out) {
    const double uq = proj(dir, qx, qy);
    double best = __builtin_inf(), second = __builtin_inf(), thr = __builtin_inf();
    int bpos = 0, brow = 0x7fffffff;
    int lo, hi;
    if (start >= 0 && start < m) {
        const double2 c = sxy[start];
        const double dx = qx - c.x, dy = qy - c.y;
        double s = 0.0;
        s += dx * dx;
        s += dy * dy;
        best = s; bpos = start; brow = sorig[start];
        lo = start - 1; hi = start + 1;
    } else {
        hi = sweep_lower_bound(sxy, m, dir, uq);
        lo = hi - 1;
    }
    while (lo >= 0 || hi < m) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const bool right = side == 0;
            if (right ? hi < m : lo >= 0) {
                const int i = right ? hi : lo;
                const double2 c = sxy[i];
                const double du = right ? proj(dir, c.x, c.y) - uq : uq - proj(dir, c.x, c.y);
                if (du > thr) { if (right) hi = m; else lo = -1; }      // everything further out is farther than `second`
                else {
                    if (true) {   // a candidate before the window costs five flops; testing for it costs a mask branch
                        const double dx = qx - c.x, dy = qy - c.y;
                        double s = 0.0;
                        s += dx * dx;
                        s += dy * dy;
                        if (s <= best) {                         // new winner, or an exact tie with it
                            const int row = sorig[i];
                            if (s < best || row < brow) { second = best; best = s; bpos = i; brow = row; }
                            else second = s;                     // tie lost on the row: second == best
                            thr = prune_width(dir, second, uq, uabs);
                        } else if (s < second) {
                            second = s;
                            thr = prune_width(dir, second, uq, uabs);
                        }
                    }
                    if (right) ++hi; else --lo;
                }
            }
        }
    }
    d2_out = best;
    second_out = second;
    return bpos;
}

// The two nearest target points (positions p1, p2; p2 = -1 if the cloud has one
// point) with their squared distances s1 <= s2, and the squared distance s3 of
// the third nearest (inf if there is none).  Window of the third distance.
// Ordering rule as everywhere: (squared distance, original row) ascending.
struct Top2 {
    int p1, p2;
    double s1, s2, s3;
};

#ifdef ICPMI_SWEEP_V1
// `seed` >= 0 is a position whose distance seeds the top-two list (the previous
// match).  CENTRED = true starts the walk at the query's own projection (binary
// search) and skips the seed when the walk meets it: right after a large step the
// previous match lies many positions away from the query's place, and walking
// there from the seed would cost as much as an unseeded search.  CENTRED = false
// walks outwards from the seed itself (no binary search: cheapest when the row
// has barely moved).  Any seed and either start give the same answer.
__device__ __forceinline__ Top2 sweep_top2(const double2* sxy, const int32_t* sorig, int m, int dir, double uabs,
                                           double qx, double qy, int seed, bool CENTRED) {
    const double uq = proj(dir, qx, qy);
    Top2 t;
    t.p1 = 0; t.p2 = -1;
    t.s1 = t.s2 = t.s3 = __builtin_inf();
    int r1 = 0x7fffffff, r2 = 0x7fffffff;
    double thr = __builtin_inf();
    int lo, hi;
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) {
        const double2 c = sxy[seed];
        const double dx = qx - c.x, dy = qy - c.y;
        double s = 0.0;
        s += dx * dx;
        s += dy * dy;
        t.s1 = s; t.p1 = seed; r1 = sorig[seed];
    }
    if (seeded && !CENTRED) { lo = seed - 1; hi = seed + 1; }
    else {
        hi = sweep_lower_bound(sxy, m, dir, uq);
        lo = hi - 1;
    }
    const int skip = seeded && CENTRED ? seed : -1;
    while (lo >= 0 || hi < m) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const bool right = side == 0;
            if (right ? hi < m : lo >= 0) {
                const int i = right ? hi : lo;
                const double2 c = sxy[i];
                const double du = right ? proj(dir, c.x, c.y) - uq : uq - proj(dir, c.x, c.y);
                if (du > thr) { if (right) hi = m; else lo = -1; }      // everything further out is farther than the third
                else {
                    if (i != skip) {
                        const double dx = qx - c.x, dy = qy - c.y;
                        double s = 0.0;
                        s += dx * dx;
                        s += dy * dy;
                        if (s <= t.s2) {                             // enters the top two (or ties with the second)
                            const int row = sorig[i];
                            if (s < t.s1 || (s == t.s1 && row < r1)) {
                                t.s3 = t.s2; t.s2 = t.s1; t.p2 = t.p1; r2 = r1;
                                t.s1 = s; t.p1 = i; r1 = row;
                            } else if (s < t.s2 || row < r2) {
                                t.s3 = t.s2; t.s2 = s; t.p2 = i; r2 = row;
                            } else t.s3 = s;                         // tie with the second, lost on the row
                            thr = prune_width(dir, t.s3, uq, uabs);
                        } else if (s < t.s3) {
                            t.s3 = s;
                            thr = prune_width(dir, t.s3, uq, uabs);
                        }
                    }
                    if (right) ++hi; else --lo;
                }
            }
        }
    }
    return t;
}

// 1-NN of (qx, qy) in the sorted cloud.  Returns the sorted position; d2 is the
// squared distance; ties go to the lowest original row (sorig).  `seed` >= 0 is a
// position whose distance seeds the bound (the previous iteration's match); the
// walk starts at the query's own projection (CENTRED: binary search; meeting the
// seed again is harmless, it ties with itself) or at the seed.  Same answer for
// any seed and either start.
__device__ __forceinline__ int sweep_nn(const double2* sxy, const int32_t* sorig, int m, int dir, double uabs,
                                        double qx, double qy, int seed, bool CENTRED, double& d2_out) {
    const double uq = proj(dir, qx, qy);
    double best = __builtin_inf(), thr = __builtin_inf();
    int bpos = 0, brow = 0x7fffffff;
    int lo, hi;
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) {
        const double2 c = sxy[seed];
        const double dx = qx - c.x, dy = qy - c.y;
        double s = 0.0;
        s += dx * dx;
        s += dy * dy;
        best = s; bpos = seed; brow = sorig[seed];
        thr = prune_width(dir, best, uq, uabs);
    }
    if (seeded && !CENTRED) { lo = seed - 1; hi = seed + 1; }
    else {
        hi = sweep_lower_bound(sxy, m, dir, uq);
        lo = hi - 1;
    }
    while (lo >= 0 || hi < m) {
        if (hi < m) {
            const double2 c = sxy[hi];
            const double du = proj(dir, c.x, c.y) - uq;
            if (du > thr) hi = m;                               // everything further right is farther still
            else {
                const double dx = qx - c.x, dy = qy - c.y;
                double s = 0.0;
                s += dx * dx;
                s += dy * dy;
                if (s <= best) {
                    const int row = sorig[hi];
                    if (s < best || row < brow) {
                        if (s < best) thr = prune_width(dir, s, uq, uabs);
                        best = s; bpos = hi; brow = row;
                    }
                }
                ++hi;
            }
        }
        if (lo >= 0) {
            const double2 c = sxy[lo];
            const double du = uq - proj(dir, c.x, c.y);
            if (du > thr) lo = -1;
            else {
                const double dx = qx - c.x, dy = qy - c.y;
                double s = 0.0;
                s += dx * dx;
                s += dy * dy;
                if (s <= best) {
                    const int row = sorig[lo];
                    if (s < best || row < brow) {
                        if (s < best) thr = prune_width(dir, s, uq, uabs);
                        best = s; bpos = lo; brow = row;
                    }
                }
                --lo;
            }
        }
    }
    d2_out = best;
    return bpos;
}

#else
// ── batched, predicated walks (the searches of the fused ICP kernel) ─────────────────────────────────
// One loop round takes ONE candidate from each open side; both are loaded before either is used and the
// next pair is fetched while the current one is evaluated, so a round waits for LDS once instead of twice
// and carries no branch per candidate: the window test, the distance and the update are straight-line
// selects.  Only an exact tie of squared distances (duplicates, lattices) leaves the straight line, to
// compare original rows.  The bound `thr` is refreshed once per round (a stale bound is only wider).
// Same answers as the exhaustive scan for any seed and either start.
struct SweepAxis {
    double a, b;        // projection u = x*a + y*b, factors from {0, +-1}: exact
    double uq, slack, two;
    __device__ __forceinline__ SweepAxis(int dir, double qx, double qy, double uabs) {
        a = dir == 1 ? 0.0 : 1.0;
        b = dir == 0 ? 0.0 : (dir == 3 ? -1.0 : 1.0);
        uq = qx * a + qy * b;
        slack = dir < 2 ? 0.0 : 4.5e-16 * (fabs(uq) + uabs);          // rounding of x +- y on the diagonals
        two = dir < 2 ? 1.0 : 2.000000000000002;                      // (dx +- dy)^2 <= 2 (dx^2 + dy^2)
    }
    // prune_width() without the branch on the axis (x * 1.0 and + 0.0 are exact)
    __device__ __forceinline__ double width(double best) const {
        return (double)(__builtin_amdgcn_sqrtf((float)(best * two)) * 1.000001f + 1e-18f) + slack;
    }
    __device__ __forceinline__ double u(const double2 c) const { return c.x * a + c.y * b; }
};

__device__ __forceinline__ double2 sweep_load(const double2* sxy, int i, int m) { return sxy[min(max(i, 0), m - 1)]; }

__device__ __forceinline__ double sweep_d2(double qx, double qy, const double2 c) {
    const double dx = qx - c.x, dy = qy - c.y;
    double s = 0.0;
    s += dx * dx;
    s += dy * dy;
    return s;
}

// `seed` >= 0 is a position whose distance seeds the top-two list (the previous
// match).  CENTRED = true starts the walk at the query's own projection (binary
// search) and skips the seed when the walk meets it: right after a large step the
// previous match lies many positions away from the query's place, and walking
// there from the seed would cost as much as an unseeded search.  CENTRED = false
// walks outwards from the seed itself (no binary search: cheapest when the row
// has barely moved).  Any seed and either start give the same answer.
__device__ __forceinline__ Top2 sweep_top2(const double2* sxy, const int32_t* sorig, int m, int dir, double uabs,
                                           double qx, double qy, int seed, bool CENTRED) {
    const SweepAxis ax(dir, qx, qy, uabs);
    Top2 t;
    t.p1 = 0; t.p2 = -1;
    t.s1 = t.s2 = t.s3 = __builtin_inf();
    double thr = __builtin_inf();
    int lo, hi;
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) { t.s1 = sweep_d2(qx, qy, sxy[seed]); t.p1 = seed; }
    if (seeded && !CENTRED) { lo = seed - 1; hi = seed + 1; }
    else {
        hi = sweep_lower_bound(sxy, m, dir, ax.uq);
        lo = hi - 1;
    }
    const int skip = seeded && CENTRED ? seed : -1;
    double2 cr = sweep_load(sxy, hi, m), cl = sweep_load(sxy, lo, m);
    while (lo >= 0 || hi < m) {
        const double2 nr = sweep_load(sxy, hi + 1, m), nl = sweep_load(sxy, lo - 1, m);
        const bool inr = hi < m && !(ax.u(cr) - ax.uq > thr);          // else: everything further right is farther than the third
        const bool inl = lo >= 0 && !(ax.uq - ax.u(cl) > thr);
        const double sr = sweep_d2(qx, qy, cr), sl = sweep_d2(qx, qy, cl);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int i = side == 0 ? hi : lo;
            const double s = side == 0 ? sr : sl;
            const bool take = (side == 0 ? inr : inl) && i != skip;
            if (take && (s == t.s1 || s == t.s2)) {
                // exact tie with a kept distance: order by original row (the rule everywhere: (distance, row) ascending)
                const int row = sorig[i], r1 = sorig[t.p1], r2 = t.p2 >= 0 ? sorig[t.p2] : 0x7fffffff;
                if (s < t.s1 || (s == t.s1 && row < r1)) { t.s3 = t.s2; t.s2 = t.s1; t.p2 = t.p1; t.s1 = s; t.p1 = i; }
                else if (s < t.s2 || row < r2) { t.s3 = t.s2; t.s2 = s; t.p2 = i; }
                else t.s3 = s;                                         // tie with the second, lost on the row
            } else {
                const bool c1 = take && s < t.s1, c2 = take && s < t.s2, c3 = take && s < t.s3;
                t.s3 = c2 ? t.s2 : (c3 ? s : t.s3);
                t.s2 = c1 ? t.s1 : (c2 ? s : t.s2);
                t.p2 = c1 ? t.p1 : (c2 ? i : t.p2);
                t.s1 = c1 ? s : t.s1;
                t.p1 = c1 ? i : t.p1;
            }
        }
        thr = ax.width(t.s3);
        hi = inr ? hi + 1 : m;
        lo = inl ? lo - 1 : -1;
        cr = nr; cl = nl;
    }
    return t;
}

// 1-NN of (qx, qy) in the sorted cloud.  Returns the sorted position; d2 is the
// squared distance; ties go to the lowest original row (sorig).  `seed` >= 0 is a
// position whose distance seeds the bound (the previous iteration's match); the
// walk starts at the query's own projection (CENTRED: binary search) or at the
// seed.  Same answer for any seed and either start.
__device__ __forceinline__ int sweep_nn(const double2* sxy, const int32_t* sorig, int m, int dir, double uabs,
                                        double qx, double qy, int seed, bool CENTRED, double& d2_out) {
    const SweepAxis ax(dir, qx, qy, uabs);
    double best = __builtin_inf(), thr = __builtin_inf();
    int bpos = 0;
    int lo, hi;
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) {
        best = sweep_d2(qx, qy, sxy[seed]); bpos = seed;
        thr = ax.width(best);
    }
    if (seeded && !CENTRED) { lo = seed - 1; hi = seed + 1; }
    else {
        hi = sweep_lower_bound(sxy, m, dir, ax.uq);
        lo = hi - 1;
    }
    double2 cr = sweep_load(sxy, hi, m), cl = sweep_load(sxy, lo, m);
    while (lo >= 0 || hi < m) {
        const double2 nr = sweep_load(sxy, hi + 1, m), nl = sweep_load(sxy, lo - 1, m);
        const bool inr = hi < m && !(ax.u(cr) - ax.uq > thr);          // else: everything further right is farther still
        const bool inl = lo >= 0 && !(ax.uq - ax.u(cl) > thr);
        const double sr = sweep_d2(qx, qy, cr), sl = sweep_d2(qx, qy, cl);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int i = side == 0 ? hi : lo;
            const double s = side == 0 ? sr : sl;
            const bool take = side == 0 ? inr : inl;
            if (take && s == best && i != bpos) {                      // exact tie: the lowest original row wins
                if (sorig[i] < sorig[bpos]) bpos = i;
            }
            const bool lt = take && s < best;
            best = lt ? s : best;
            bpos = lt ? i : bpos;
        }
        thr = ax.width(best);
        hi = inr ? hi + 1 : m;
        lo = inl ? lo - 1 : -1;
        cr = nr; cl = nl;
    }
    d2_out = best;
    return bpos;
}


// ── single-precision filter in front of the exact walk ───────────────────────────────────────────────
// The walks above spend ten float64 instructions on every candidate of the window (projection, gap,
// distance), and float64 issues at a fraction of the float32 rate.  Nearly all candidates lose.  So the
// target also carries a float32 image per sorted position — (x - ox, y - oy, u - uo, original row), relative
// to a point o of the cloud so that the magnitudes stay small — and a candidate is first judged on that
// image with rigorous margins; only one that MIGHT enter the result is then evaluated in float64 exactly as
// before.  Results are therefore bit-identical; the filter can only cost a spurious exact evaluation.
//
// Margins (E, mu below; 2^-23 = 1.19e-7).  Every float32 coordinate is the rounding of a float64 difference,
// so it is within 2^-24 relative of the true recentred coordinate, and the float32 difference q - c within
// e <= 2^-23 (|q'| + |c'|) of the true one (both roundings + the subtraction's).  With d the true distance:
// |s32 - d^2| <= 2 sqrt(2) e d + 2 e^2 + 2^-22 s32, hence d <= B  =>  s32 <= (B + 1.5 e)^2 (1 + 2^-21): a
// candidate is skipped only when s32 exceeds that, with B a float32 upper bound of the root of the bound.
// The window test uses u32 = fl32(fl64(u - uo)), a monotone function of the sort key, so "this candidate's gap
// exceeds W" still implies it for every candidate further out; W = kappa B (1 + 1e-6) + mu covers the
// roundings of both projections.  NaN / inf (coordinates beyond float32) compare false and fall through to
// the exact path.
// The sort order of a prepared target ("dir"): 0..3 = the projections x, y, x + y, x - y; SWEEP_POLAR = the bearing
// atan2(y, x) about the frame origin.  A lidar scan in its sensor frame has at most one return per bearing, so
// the points within distance B of a query q all lie in the wedge |bearing - bearing(q)| <= asin(B / |q|) — a few
// points — whereas a slab |u - u(q)| <= B of a projection holds every wall that crosses it (measured on the bench
// scans: 15 instead of 34 candidates for the first search, and a 2.5x smaller longest window).  The bound:
// a point c at angle dth from q (seen from the origin) is at least |q| sin|dth| away for |dth| <= 90 deg and at
// least |q| beyond, so dist(q, c) <= B < |q| implies |dth| <= asin(B / |q|).  The walk is the same linear one
// on the bearing, with the half-width asin(B / |q|) instead of kappa * B, and one continuation across the seam at
// +-pi.  A query closer to the origin than ~1.4 B has no wedge and walks the whole cloud (still exact).
constexpr int SWEEP_POLAR = 4;
__device__ __forceinline__ double polar_key(double x, double y) { return atan2(y, x); }

struct SweepF {
    double ox, oy, uo;        // origin of the float32 images (polar: the frame origin) and its sort key
    float ut, rt;             // max |key - uo| and max(|x - ox|, |y - oy|) over the target (upper bounds)
};
struct SweepFQuery {
    float x, y, u;            // the query in the single-precision frame, and its sort key
    float e15, mu;            // 1.5 e; key slack
    float kw;                 // window half-width per unit of B: kappa (projections) or (1 + 1e-5) / |q| (polar)
    bool polar;
    __device__ __forceinline__ SweepFQuery(const SweepF& f, int dir, double uabs, double qx, double qy) {
        polar = dir == SWEEP_POLAR;
        x = (float)(qx - f.ox); y = (float)(qy - f.oy);
        e15 = 1.8e-7f * (fabsf(x) + fabsf(y) + 2.0f * f.rt) + 1e-30f;
        if (polar) {
            u = atan2f(y, x);
            // images: the float32 bearing quantised to 2 pi / 2^21 (prep.hip: within 5.3e-6 of the true bearing); here: atan2f
            // of the rounded coordinates (1e-7 + its own few ulp) and the float32 subtraction of the two: < 7e-6 in all
            mu = 1e-5f;
            kw = 1.00001f / __builtin_amdgcn_sqrtf(x * x + y * y);       // >= 1 / |q| (inf at the origin: no wedge)
        } else {
            const SweepAxis ax(dir, qx, qy, uabs);
            u = (float)(ax.uq - f.uo);
            mu = 2.4e-7f * (fabsf(u) + f.ut) + (float)ax.slack * 1.000001f + 1e-30f;
            kw = dir < 2 ? 1.0f : 1.4142137f;
        }
    }
    // float32 bounds for a float64 squared-distance bound `best`: window half-width W and filter threshold T
    __device__ __forceinline__ void bounds(double best, float& W, float& T) const {
        const float B = __builtin_amdgcn_sqrtf((float)best) * 1.000001f + 1e-18f;
        const float t = B * kw;
        if (polar) W = t <= 0.7f ? (t + 0.3f * t * t * t) * 1.000001f + mu : __builtin_inff();   // asin t <= t + 0.3 t^3 on [0, 0.7]
        else W = t * 1.000001f + mu;
        const float b = B + e15;
        T = b * b * 1.000002f;
    }
};

typedef float sweep_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int sweepf_row(const float4 c) { return __float_as_int(c.w); }

// first position whose float32 key image is >= u (m if none): same place as sweep_lower_bound up to
// float32 ties, and any start is a correct start (both directions are walked until the gap rules the rest out)
__device__ __forceinline__ int sweepf_lower_bound(const float4* sq, int m, float u) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sq[mid].z < u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Index state of a filtered walk.  The right side takes hi, hi + 1, ... while hi < hi_end, the left side lo,
// lo - 1, ... while lo > lo_end; a side closes for good at the first candidate whose key gap exceeds the window
// (and stays ON that candidate).  Polar order: when exactly one side has run into the end of the array while the
// other closed on a gap, the side at the end continues once across the seam (key offset 2 pi), up to where the
// other side stopped — so every candidate is visited at most once.
struct SweepFWalk {
    int lo, hi, lo_end, hi_end;
    bool openr, openl;
    sweep_v2f uu;             // (-u + right offset, u + left offset): gap = (key_right, -key_left) + uu
    __device__ __forceinline__ SweepFWalk(int lo0, int hi0, int m, float u) {
        lo = lo0; hi = hi0; lo_end = -1; hi_end = m; openr = true; openl = true;
        uu = sweep_v2f{-u, u};
    }
    __device__ __forceinline__ bool more() const { return (openr && hi < hi_end) || (openl && lo > lo_end); }
    // after a pass: set up the continuation across the seam; false = done
    __device__ __forceinline__ bool wrap(int m) {
        const bool endr = openr && hi >= hi_end, endl = openl && lo <= lo_end;
        if (endr == endl) return false;                                // both closed on gaps, or the whole array was walked
        if (endr) { hi_end = lo + 1; hi = 0; uu.x += 6.2831855f; openl = false; lo_end = lo; }
        else { lo_end = hi - 1; lo = m - 1; uu.y += 6.2831855f; openr = false; hi_end = hi; }
        return hi < hi_end || lo > lo_end;
    }
};

// One round of a filtered walk: the candidates at hi (right) and lo (left), judged together on their float32
// images with packed arithmetic (one v_pk_* instruction serves both sides).  sq[-1] and sq[m] exist (padding),
// so a closed side may still be loaded.  Sets inr / inl (side still inside the window) and pr / pl (candidate may
// enter the result: evaluate it exactly).
struct SweepFRound {
    bool inr, inl, pr, pl;
    float4 cr, cl;
    __device__ __forceinline__ SweepFRound(const float4* sq, const SweepFWalk& w, const SweepFQuery& fq, float W, float T) {
        cr = sq[w.hi]; cl = sq[w.lo];
        const sweep_v2f cz = {cr.z, -cl.z};
        const sweep_v2f g = cz + w.uu;                                 // (key_right - key_q, key_q - key_left)
        inr = w.openr && w.hi < w.hi_end && !(g.x > W);                // else: everything further out is farther still
        inl = w.openl && w.lo > w.lo_end && !(g.y > W);
        const sweep_v2f cx = {cr.x, cl.x}, cy = {cr.y, cl.y};
        const sweep_v2f dx = fq.x - cx, dy = fq.y - cy;
        const sweep_v2f s2 = __builtin_elementwise_fma(dx, dx, dy * dy);
        pr = inr && !(s2.x > T);
        pl = inl && !(s2.y > T);
    }
    __device__ __forceinline__ void advance(SweepFWalk& w) const {
        w.openr = w.openr && (inr || w.hi >= w.hi_end);                // closed by the gap: stays closed, hi stays on that candidate
        w.openl = w.openl && (inl || w.lo <= w.lo_end);
        w.hi += inr ? 1 : 0;
        w.lo -= inl ? 1 : 0;
    }
};

// sweep_nn with the filter: sq = float32 images (padded by one entry at either end), sxy = exact points (read only
// for candidates that pass)
__device__ __forceinline__ int sweepf_nn(const float4* sq, const double2* sxy, const SweepF& f, int m, int dir, double uabs,
                                         double qx, double qy, int seed, bool CENTRED, double& d2_out) {
    const SweepFQuery fq(f, dir, uabs, qx, qy);
    double best = __builtin_inf();
    float W = __builtin_inff(), T = __builtin_inff();
    int bpos = 0;
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) {
        best = sweep_d2(qx, qy, sxy[seed]); bpos = seed;
        fq.bounds(best, W, T);
    }
    // a seed on the other side of the seam at +-pi is no place to start from (the walk would cross the whole array)
    const bool from_seed = seeded && !CENTRED && !(fq.polar && fabsf(sq[seed].z - fq.u) > 3.0f);
    const int h0 = from_seed ? seed + 1 : sweepf_lower_bound(sq, m, fq.u);
    SweepFWalk w(from_seed ? seed - 1 : h0 - 1, h0, m, fq.u);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        while (w.more()) {
            const SweepFRound r(sq, w, fq, W, T);
            if (r.pr || r.pl) {                                        // might win (or tie): the exact test
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int i = side == 0 ? w.hi : w.lo;
                    if (side == 0 ? r.pr : r.pl) {
                        const double s = sweep_d2(qx, qy, sxy[i]);
                        if (s == best && i != bpos) {                  // exact tie: the lowest original row wins
                            if (sweepf_row(side == 0 ? r.cr : r.cl) < sweepf_row(sq[bpos])) bpos = i;
                        }
                        if (s < best) { best = s; bpos = i; }
                    }
                }
                fq.bounds(best, W, T);
            }
            r.advance(w);
        }
        if (!fq.polar || !w.wrap(m)) break;
    }
    d2_out = best;
    return bpos;
}

// sweep_top2 with the filter (bounds from the third distance)
__device__ __forceinline__ Top2 sweepf_top2(const float4* sq, const double2* sxy, const SweepF& f, int m, int dir, double uabs,
                                            double qx, double qy, int seed, bool CENTRED) {
    const SweepFQuery fq(f, dir, uabs, qx, qy);
    Top2 t;
    t.p1 = 0; t.p2 = -1;
    t.s1 = t.s2 = t.s3 = __builtin_inf();
    float W = __builtin_inff(), T = __builtin_inff();
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) { t.s1 = sweep_d2(qx, qy, sxy[seed]); t.p1 = seed; }
    // a seed on the other side of the seam at +-pi is no place to start from (the walk would cross the whole array)
    const bool from_seed = seeded && !CENTRED && !(fq.polar && fabsf(sq[seed].z - fq.u) > 3.0f);
    const int h0 = from_seed ? seed + 1 : sweepf_lower_bound(sq, m, fq.u);
    SweepFWalk w(from_seed ? seed - 1 : h0 - 1, h0, m, fq.u);
    const int skip = seeded ? seed : -1;                               // the seed is in the list already
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        while (w.more()) {
            const SweepFRound r(sq, w, fq, W, T);
            const bool pr = r.pr && w.hi != skip, pl = r.pl && w.lo != skip;
            if (pr || pl) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int i = side == 0 ? w.hi : w.lo;
                    if (side == 0 ? pr : pl) {
                        const double s = sweep_d2(qx, qy, sxy[i]);
                        if (s == t.s1 || s == t.s2) {
                            // exact tie with a kept distance: order by original row (the rule everywhere: (distance, row) ascending)
                            const int row = sweepf_row(side == 0 ? r.cr : r.cl), r1 = sweepf_row(sq[t.p1]);
                            const int r2 = t.p2 >= 0 ? sweepf_row(sq[t.p2]) : 0x7fffffff;
                            if (s < t.s1 || (s == t.s1 && row < r1)) { t.s3 = t.s2; t.s2 = t.s1; t.p2 = t.p1; t.s1 = s; t.p1 = i; }
                            else if (s < t.s2 || row < r2) { t.s3 = t.s2; t.s2 = s; t.p2 = i; }
                            else t.s3 = s;                             // tie with the second, lost on the row
                        } else {
                            const bool c1 = s < t.s1, c2 = s < t.s2, c3 = s < t.s3;
                            t.s3 = c2 ? t.s2 : (c3 ? s : t.s3);
                            t.s2 = c1 ? t.s1 : (c2 ? s : t.s2);
                            t.p2 = c1 ? t.p1 : (c2 ? i : t.p2);
                            t.s1 = c1 ? s : t.s1;
                            t.p1 = c1 ? i : t.p1;
                        }
                    }
                }
                fq.bounds(t.s3, W, T);
            }
            r.advance(w);
        }
        if (!fq.polar || !w.wrap(m)) break;
    }
    return t;
}

// ── packed walks: the whole walk in float32, exact arithmetic only for the listed survivors (round 4) ─
// In the walks above a round is cheap while both candidates are rejected (24 vector instructions) — but the 64 lanes of
// a wave share one instruction stream, and in nearly every round SOME lane has a candidate that passes the filter: the
// wave then runs the exact float64 evaluation, the tie test and the new bounds (60-70 instructions) for that round.
// Measured (round 4): the first top-two search of a batch — proof windows of ~9 candidates for the slowest of 64 lanes —
// cost more than the unseeded first search.  So here the walk never leaves float32 and has no branch but its loop:
// every candidate's filter value s32 = fma(dx, dx, dy * dy) (>= 0) is PACKED with its sorted position into one word,
// (bits(s32) & ~0x7FF) | position (targets of at most 2 048 points; non-negative floats order like their bits), and the
// K smallest words are kept in registers by integer v_med3_u32 / v_min_u32 — one instruction per list entry and
// candidate, value and index together.  The window follows from the K-th word: the candidate behind it lies within
//     B = sqrt(s32) (1 + 2^-22)^(1/2) + 3.42 e   of the query      (from |s32 - d^2| <= 2 sqrt2 e d + 2 e^2 + 2^-22 s32,
// the bound of the filter section solved for d; e <= e15 / 1.5; the 11 index bits are set before the root is taken, so
// the truncation only widens it), hence the K-th nearest true distance is at most B, and a side closes where the key
// gap alone exceeds W(B) exactly as before.  After the walk, every candidate that can be among the K nearest has a filter
// value of at most T(B) (the filter's guarantee d <= B => s32 <= T(B)); if the (K+1)-th word of the list lies above T(B)
// the list holds them all, and only they are evaluated in float64 — (squared distance, original row) ascending, the
// rule everywhere.  Otherwise (more candidates within the filter's resolution than the list holds: duplicates, lattices,
// about one row in a thousand of an ordinary scan) or when the images are not finite-small, the lane falls back on the
// exact walk above.  Same answers as the exhaustive scan by construction; the tests that hold the fast path to the
// exhaustive kernel and the oracle bit for bit are the gate.
// (measured: two or three rounds per update of the window change nothing, 4.61 / 4.67 against 4.65 ms for 16 384 pairs)
constexpr unsigned SWEEP_PK_IDX = 0x7FFu;                              // positions below 2 048
constexpr unsigned SWEEP_PK_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned sweep_med3_u32(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned sweep_min_u32(unsigned a, unsigned b) { return a < b ? a : b; }

template <int K>
struct SweepPkList {
    unsigned m[K];                                                     // ascending
    __device__ __forceinline__ SweepPkList() {
#pragma unroll
        for (int k = 0; k < K; ++k) m[k] = SWEEP_PK_NONE;
    }
    __device__ __forceinline__ void offer(unsigned sp) {
#pragma unroll
        for (int k = K - 1; k > 0; --k) m[k] = sweep_med3_u32(m[k - 1], m[k], sp);
        m[0] = sweep_min_u32(m[0], sp);
    }
};
__device__ __forceinline__ unsigned sweep_pk(float s32, int i, bool in) {
    return in ? ((__float_as_uint(s32) & ~SWEEP_PK_IDX) | (unsigned)i) : SWEEP_PK_NONE;
}
__device__ __forceinline__ float sweep_pk_floor(unsigned m) { return __uint_as_float(m & ~SWEEP_PK_IDX); }   // <= the filter value behind m

struct SweepPkQuery : SweepFQuery {
    float c25;                // 3.42 e + the float32 roundings of B
    bool bad;                 // images not finite-small: the float32 values may overflow — the exact walk decides
    __device__ __forceinline__ SweepPkQuery(const SweepF& f, int dir, double uabs, double qx, double qy) : SweepFQuery(f, dir, uabs, qx, qy) {
        c25 = 2.5f * e15 + 1e-18f;
        bad = !(fabsf(x) + fabsf(y) < 1e18f) || !(f.rt < 1e18f);
    }
    // float32 upper bound of the true distance of the candidate behind the packed word m (NaN while the list is not full:
    // compares false everywhere, so nothing closes).  The fused multiply-adds are this bound's own (no reference arithmetic).
    __device__ __forceinline__ float dist_bound(unsigned m) const {
        return __builtin_fmaf(__builtin_amdgcn_sqrtf(__uint_as_float(m | SWEEP_PK_IDX)), 1.000002f, c25);
    }
    template <bool POLAR>
    __device__ __forceinline__ float width(float B) const {
        const float t = B * kw;
        if (POLAR) {                                                   // asin t <= t + 0.3 t^3 on [0, 0.7]
            const float w = __builtin_fmaf(t, __builtin_fmaf(0.3000004f, t * t, 1.000001f), mu);
            return t <= 0.7f ? w : __builtin_inff();
        }
        return __builtin_fmaf(t, 1.000001f, mu);
    }
    __device__ __forceinline__ float threshold(float B) const {
        const float b = B + e15;
        return b * b * 1.000002f;
    }
};

// The walk: every candidate of the window into the list.  lo / hi: the next candidate of either side.  A side takes its
// candidate while the key gap is within W; W never grows (list entries only fall) and a side that stops stays on its
// candidate, whose gap does not change — so a stopped side needs no flag, the same test keeps it stopped.  Bearing order
// (POLAR): the sorted array is a circle — indices run on past either end (hi up to lo + m, the key shifted by 2 pi) until
// every point has been visited once; the right side goes first when one point is left.
// NB: the list entry that bounds the window (0: nearest neighbour, 2: the third distance); the entries behind it only tell
// whether the list is complete.
// max_rounds > 0: the walk gives up after that many rounds (a query far from the cloud: the caller finishes on the box
// hierarchy); returns true when it ended by itself — both sides stopped: the window is covered.
template <int K, int NB, bool POLAR>
__device__ __forceinline__ bool sweep_pk_walk(const float4* sq, const SweepPkQuery& fq, int lo, int hi, int m, int skip, SweepPkList<K>& L,
                                              int max_rounds = 0) {
    float W = fq.width<POLAR>(fq.dist_bound(L.m[NB]));
    const float ur = -fq.u, ul = fq.u, ur2 = 6.2831855f - fq.u, ul2 = fq.u + 6.2831855f;
    int rounds = 0;
    for (;;) {
        int hp = hi, lp = lo;
        float o_r = ur, o_l = ul;
        bool okr, okl;
        if (POLAR) {
            const bool wr = hi >= m, wl = lo < 0;
            hp = wr ? hi - m : hi;
            lp = wl ? lo + m : lo;
            o_r = wr ? ur2 : ur;
            o_l = wl ? ul2 : ul;
            okr = hi - lo <= m;                                        // a point not visited yet is left
        } else {
            okr = hi < m;
            okl = lo >= 0;
        }
        const float4 cr = sq[hp], cl = sq[lp];                         // sq[-1] and sq[m] exist (padding)
        const bool inr = okr && !(cr.z + o_r > W);                     // key_right - key_q; else: everything further out is farther still
        if (POLAR) okl = hi + (inr ? 1 : 0) - lo <= m;
        const bool inl = okl && !(o_l - cl.z > W);                     // key_q - key_left
        const sweep_v2f cx = {cr.x, cl.x}, cy = {cr.y, cl.y};
        const sweep_v2f dx = fq.x - cx, dy = fq.y - cy;
        const sweep_v2f s2 = __builtin_elementwise_fma(dx, dx, dy * dy);
        L.offer(sweep_pk(s2.x, hp, inr && hp != skip));
        L.offer(sweep_pk(s2.y, lp, inl && lp != skip));
        if (!(inr || inl)) return true;
        if (max_rounds > 0 && ++rounds >= max_rounds) return false;
        W = fq.width<POLAR>(fq.dist_bound(L.m[NB]));
        hi += inr ? 1 : 0;
        lo -= inl ? 1 : 0;
    }
}

template <int K, int NB>
__device__ __forceinline__ void sweep_pk_search(const float4* sq, const SweepPkQuery& fq, int m, int seed, bool CENTRED, SweepPkList<K>& L) {
    const bool seeded = seed >= 0 && seed < m;
    int skip = -1;
    // a seed on the other side of the seam at +-pi is no place to start from (the walk would cross the whole array)
    const bool from_seed = seeded && !CENTRED && !(fq.polar && fabsf(sq[seed].z - fq.u) > 3.0f);
    if (seeded && !from_seed) {                                        // the previous match bounds the window from the first round on
        const float4 c = sq[seed];
        const float ex = fq.x - c.x, ey = fq.y - c.y;
        L.offer(sweep_pk(__builtin_fmaf(ex, ex, ey * ey), seed, true));
        skip = seed;
    }
    const int h0 = from_seed ? seed + 1 : sweepf_lower_bound(sq, m, fq.u);    // from the seed: it is the first candidate of the left side
    if (fq.polar) sweep_pk_walk<K, NB, true>(sq, fq, h0 - 1, h0, m, skip, L);
    else sweep_pk_walk<K, NB, false>(sq, fq, h0 - 1, h0, m, skip, L);
}

// sweepf_nn by the packed walk (seed, CENTRED as there; the squared distance is not returned: no caller of the plain
// iterations uses it)
__device__ __forceinline__ int sweepf_nn_pk(const float4* sq, const double2* sxy, const SweepF& f, int m, int dir, double uabs,
                                            double qx, double qy, int seed, bool CENTRED) {
    const SweepPkQuery fq(f, dir, uabs, qx, qy);
    SweepPkList<3> L;
    sweep_pk_search<3, 0>(sq, fq, m, seed, CENTRED, L);
    const float T = fq.threshold(fq.dist_bound(L.m[0]));
    int bpos = (int)(L.m[0] & SWEEP_PK_IDX);
    const bool a2 = L.m[1] != SWEEP_PK_NONE && !(sweep_pk_floor(L.m[1]) > T);
    const bool a3 = L.m[2] != SWEEP_PK_NONE && !(sweep_pk_floor(L.m[2]) > T);
    if (a3 || fq.bad || L.m[0] == SWEEP_PK_NONE) {
        double d2;
        bpos = sweepf_nn(sq, sxy, f, m, dir, uabs, qx, qy, seed, CENTRED, d2);
    } else if (a2) {                                                   // two candidates within the filter's resolution: exact, rows on a tie
        const int i2 = (int)(L.m[1] & SWEEP_PK_IDX);
        const double s1 = sweep_d2(qx, qy, sxy[bpos]), s2 = sweep_d2(qx, qy, sxy[i2]);
        if (s2 < s1 || (s2 == s1 && sweepf_row(sq[i2]) < sweepf_row(sq[bpos]))) bpos = i2;
    }
    return bpos;
}

// the three listed candidates (packed words, NONE = absent) in float64, ordered by (squared distance, original row)
__device__ __forceinline__ Top2 sweep_pk_top2_exact(const float4* sq, const double2* sxy, double qx, double qy, unsigned w0, unsigned w1, unsigned w2) {
    const unsigned w[3] = {w0, w1, w2};
    double s[3];
    int p[3], r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool have = w[k] != SWEEP_PK_NONE;
        p[k] = have ? (int)(w[k] & SWEEP_PK_IDX) : -1;
        const int j = have ? p[k] : 0;
        s[k] = have ? sweep_d2(qx, qy, sxy[j]) : __builtin_inf();
        r[k] = have ? sweepf_row(sq[j]) : 0x7fffffff;
    }
#define ICPMI_CS(a, b)                                                                              \
    do {                                                                                            \
        const bool sw = s[b] < s[a] || (s[b] == s[a] && r[b] < r[a]);                              \
        const double sa = sw ? s[b] : s[a], sb = sw ? s[a] : s[b];                                  \
        const int pa = sw ? p[b] : p[a], pb = sw ? p[a] : p[b], ra = sw ? r[b] : r[a], rb = sw ? r[a] : r[b]; \
        s[a] = sa; s[b] = sb; p[a] = pa; p[b] = pb; r[a] = ra; r[b] = rb;                           \
    } while (0)
    ICPMI_CS(0, 1); ICPMI_CS(1, 2); ICPMI_CS(0, 1);
#undef ICPMI_CS
    Top2 t;
    t.p1 = p[0]; t.p2 = p[1]; t.s1 = s[0]; t.s2 = s[1]; t.s3 = s[2];
    return t;
}

// sweepf_top2 by the packed walk
__device__ __forceinline__ Top2 sweepf_top2_pk(const float4* sq, const double2* sxy, const SweepF& f, int m, int dir, double uabs,
                                               double qx, double qy, int seed, bool CENTRED) {
    const SweepPkQuery fq(f, dir, uabs, qx, qy);
    SweepPkList<4> L;
    sweep_pk_search<4, 2>(sq, fq, m, seed, CENTRED, L);
    const float T = fq.threshold(fq.dist_bound(L.m[2]));
    const bool a4 = L.m[3] != SWEEP_PK_NONE && !(sweep_pk_floor(L.m[3]) > T);
    if (a4 || fq.bad || L.m[0] == SWEEP_PK_NONE) return sweepf_top2(sq, sxy, f, m, dir, uabs, qx, qy, seed, CENTRED);
    return sweep_pk_top2_exact(sq, sxy, qx, qy, L.m[0], L.m[1], L.m[2]);
}

// ── far queries: a box hierarchy over the sort order ─────────────────────────────────────────────────
// A walk visits every point whose key lies within the bound of the query's key: a query that is metres from every
// target point (a pair started from a wrong pre-alignment, a rotation far from the right one) walks most of the
// cloud — ~1 400 candidates instead of ~10 — and a wave waits for its longest lane.  So a walk that has taken a few
// rounds (the caller's choice; what they meet is the descent's first bound) gives up and the search is finished on a
// hierarchy of bounding boxes of the float32 images:
// the leaves are BLOCKS of SWEEP_BLOCK consecutive sorted positions (in bearing order a contiguous piece of wall), every
// inner node is the box of its two children — a complete binary tree in heap order over `leaves` (a power of two) blocks,
// tree[1] the root, tree[leaves + b] the box of block b (min x, min y, max x, max y; an empty block is an empty box,
// infinitely far from everything).  A query descends to the nearer child first, so the first block it looks at is (nearly)
// the nearest and what it holds bounds the rest: ~20 box tests and two or three blocks of points per query, whatever its
// distance.  (Flat versions, measured in round 3: testing all ~80 block boxes costs ~2 000 instructions per query, more
// than everything else together; scanning the blocks that pass in index order offers a third of their points to the
// exact test before the bound is tight.)
// Exact: the box distance is computed with the very operations of the point filter (monotone roundings), so it never
// exceeds the filter value of an image inside the box; the points go through the same float32 filter and exact test as
// in the walk.
constexpr int SWEEP_BLOCK = 16;
#ifndef SWEEP_FAR_ROUNDS
#define SWEEP_FAR_ROUNDS 24
#endif
#ifdef ICPMI_DIAG
static __device__ unsigned long long icpmi_dbg[16];      // diagnostic build: cycles of the phases of sweepf_top2_far (per lane-call, summed)
#define DBG_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define DBG_ADD(k, a, b) atomicAdd(&icpmi_dbg[k], (b) - (a))
#else
#define DBG_T(v)
#define DBG_ADD(k, a, b)
#endif

// leaves of the tree over m points (a power of two, at least 2); the tree takes 2 * leaves entries of 16 B
__host__ __device__ __forceinline__ int sweepf_tree_leaves(int m) {
    int l = 2;
    while (l * SWEEP_BLOCK < m) l <<= 1;
    return l;
}

// build the tree of m staged images (all threads of the workgroup; the images must be complete: barrier before; holds a
// barrier itself, the tree is complete after the caller's next one)
__device__ __forceinline__ void sweepf_build_tree(const float4* sq, int m, float4* tree, int leaves, int tid, int nthreads) {
    for (int b = tid; b < leaves; b += nthreads) {
        const int i1 = min(m, (b + 1) * SWEEP_BLOCK);
        float x0 = __builtin_inff(), y0 = __builtin_inff(), x1 = -__builtin_inff(), y1 = -__builtin_inff();
        for (int i = b * SWEEP_BLOCK; i < i1; ++i) {
            const float4 c = sq[i];
            x0 = fminf(x0, c.x); y0 = fminf(y0, c.y); x1 = fmaxf(x1, c.x); y1 = fmaxf(y1, c.y);
        }
        tree[leaves + b] = make_float4(x0, y0, x1, y1);
    }
    __syncthreads();
    for (int k = 1 + tid; k < leaves; k += nthreads) {                 // inner node k: the union of the leaves below it
        int lo = k, hi = k;
        while (lo < leaves) { lo = 2 * lo; hi = 2 * hi + 1; }
        float x0 = __builtin_inff(), y0 = __builtin_inff(), x1 = -__builtin_inff(), y1 = -__builtin_inff();
        for (int j = lo; j <= hi; ++j) {
            const float4 c = tree[j];
            x0 = fminf(x0, c.x); y0 = fminf(y0, c.y); x1 = fmaxf(x1, c.z); y1 = fmaxf(y1, c.w);
        }
        tree[k] = make_float4(x0, y0, x1, y1);
    }
}

// float32 lower bound of the filter value (SweepFRound: fma(dx, dx, dy * dy)) of any image inside the box
__device__ __forceinline__ float sweepf_box_s2(const float4 bb, float qx, float qy) {
    const float dx = fmaxf(fmaxf(bb.x - qx, qx - bb.z), 0.0f), dy = fmaxf(fmaxf(bb.y - qy, qy - bb.w), 0.0f);
    return __builtin_fmaf(dx, dx, dy * dy);
}

// The far scan of one query: depth-first over the tree, nearer child first, a subtree left out when its box is beyond
// the filter threshold T.  No stack: `trail` holds one bit per level on the path from the root (is the sibling still to
// be looked at?), the path itself is the node's index.  offer(i) is called for the points within T and lowers it.
template <class Offer>
__device__ __forceinline__ void sweepf_far_scan(const float4* sq, const float4* tree, int leaves, int m, const SweepFQuery& fq,
                                                const float& T, Offer offer) {
    int node = 1;
    unsigned trail = 0;
    bool tested = true;                                                // the root is entered untested
    for (;;) {
        bool down = tested || !(sweepf_box_s2(tree[node], fq.x, fq.y) > T);          // NaN compares false: entered
        if (down && node < leaves) {
            const float la = sweepf_box_s2(tree[2 * node], fq.x, fq.y), lb = sweepf_box_s2(tree[2 * node + 1], fq.x, fq.y);
            const bool b_first = lb < la;
            const float l_near = b_first ? lb : la, l_far = b_first ? la : lb;
            if (!(l_near > T)) {
                trail = (trail << 1) | (l_far > T ? 0u : 1u);
                node = 2 * node + (b_first ? 1 : 0);
                tested = true;
                continue;
            }
            down = false;                                              // both children are beyond the threshold
        }
        if (down) {
            // a block: its points against the filter, eight image loads in flight (the image copy is padded to whole
            // blocks; entries past m are never offered)
            const int b = node - leaves;
#pragma unroll
            for (int e = 0; e < SWEEP_BLOCK / 8; ++e) {
                const int i0 = b * SWEEP_BLOCK + e * 8;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float2 c = *reinterpret_cast<const float2*>(sq + i0 + k);
                    const float ex = fq.x - c.x, ey = fq.y - c.y;
                    v[k] = __builtin_fmaf(ex, ex, ey * ey);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (!(v[k] > T) && i0 + k < m) offer(i0 + k);
            }
        }
        // back up to the nearest level whose sibling is still to be looked at
        if (trail == 0) break;
        const int up = __builtin_ctz(trail);
        node = (node >> up) ^ 1;
        trail = (trail >> up) ^ 1u;
        tested = false;
    }
}

// finish a 1-NN search over all blocks (best / bpos: what the walk has found so far)
__device__ __forceinline__ void sweepf_far_nn(const float4* sq, const double2* sxy, const float4* tree, int leaves, const SweepFQuery& fq, int m,
                                              double qx, double qy, double& best, int& bpos, float& T) {
    float W;
    sweepf_far_scan(sq, tree, leaves, m, fq, T, [&](int i) {
        const double s = sweep_d2(qx, qy, sxy[i]);
        if (s == best && i != bpos) {                                  // exact tie: the lowest original row wins
            if (sweepf_row(sq[i]) < sweepf_row(sq[bpos])) bpos = i;
        }
        if (s < best) { best = s; bpos = i; fq.bounds(best, W, T); }
    });
}

// sweepf_nn for callers whose queries may lie far from the cloud (the rotation search: most angles put the source metres
// off the target): the same walk, abandoned after SWEEP_FAR_ROUNDS rounds for the scan over block boxes.  A separate
// function on purpose — the same logic inside sweepf_nn / sweepf_top2 cost the fused ICP kernel 11 registers (spills at
// six waves per SIMD) and a third of its speed on pairs that start close (measured, round 3).
// max_rounds: rounds of the walk before the tree takes over.  The first few are never wasted — what they meet bounds the
// descent (started without any bound it costs twice as much) —, more pay only where most queries end within them.
__device__ __forceinline__ int sweepf_nn_far(const float4* sq, const double2* sxy, const float4* tree, int leaves, const SweepF& f, int m, int dir,
                                             double uabs, double qx, double qy, double& d2_out, int max_rounds = SWEEP_FAR_ROUNDS) {
    const SweepFQuery fq(f, dir, uabs, qx, qy);
    double best = __builtin_inf();
    float W = __builtin_inff(), T = __builtin_inff();
    int bpos = 0;
    const int h0 = sweepf_lower_bound(sq, m, fq.u);
    SweepFWalk w(h0 - 1, h0, m, fq.u);
    int rounds = 0;
    bool far = false;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        while (w.more()) {
            if (++rounds > max_rounds) { far = true; break; }
            const SweepFRound r(sq, w, fq, W, T);
            if (r.pr || r.pl) {                                        // might win (or tie): the exact test
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int i = side == 0 ? w.hi : w.lo;
                    if (side == 0 ? r.pr : r.pl) {
                        const double s = sweep_d2(qx, qy, sxy[i]);
                        if (s == best && i != bpos) {                  // exact tie: the lowest original row wins
                            if (sweepf_row(side == 0 ? r.cr : r.cl) < sweepf_row(sq[bpos])) bpos = i;
                        }
                        if (s < best) { best = s; bpos = i; }
                    }
                }
                fq.bounds(best, W, T);
            }
            r.advance(w);
        }
        if (far || !fq.polar || !w.wrap(m)) break;
    }
    if (far) sweepf_far_nn(sq, sxy, tree, leaves, fq, m, qx, qy, best, bpos, T);
    d2_out = best;
    return bpos;
}

// ── far queries by the packed walk and a packed scan of the box hierarchy (round 4) ──────────────────
// As in the packed walks: nothing but float32 until the list is complete.  The scan offers every image of a block it
// enters (a select, no branch) and prunes subtrees against T(B) of the list's bounding entry.
// T0: a threshold known beforehand (from the candidates a walk has met; NaN: none) — the scan's own list may start empty
template <int K, int NB>
__device__ __forceinline__ void sweep_pk_far_scan(const float4* sq, const float4* tree, int leaves, int m, const SweepPkQuery& fq, SweepPkList<K>& L,
                                                  float T0 = __builtin_nanf("")) {
    auto tighter = [](float a, float b) { return (a != a || b < a) ? (b != b ? a : b) : a; };    // the smaller of two thresholds, NaN = none
    float T = tighter(T0, fq.threshold(fq.dist_bound(L.m[NB])));       // NaN while nothing bounds: nothing is pruned
    int node = 1;
    unsigned trail = 0;
    bool tested = true;                                                // the root is entered untested
    for (;;) {
        bool down = tested || !(sweepf_box_s2(tree[node], fq.x, fq.y) > T);
        if (down && node < leaves) {
            const float la = sweepf_box_s2(tree[2 * node], fq.x, fq.y), lb = sweepf_box_s2(tree[2 * node + 1], fq.x, fq.y);
            const bool b_first = lb < la;
            const float l_near = b_first ? lb : la, l_far = b_first ? la : lb;
            if (!(l_near > T)) {
                trail = (trail << 1) | (l_far > T ? 0u : 1u);
                node = 2 * node + (b_first ? 1 : 0);
                tested = true;
                continue;
            }
            down = false;                                              // both children are beyond the threshold
        }
        if (down) {
            const int b = node - leaves;
#pragma unroll
            for (int e = 0; e < SWEEP_BLOCK / 8; ++e) {
                const int i0 = b * SWEEP_BLOCK + e * 8;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float2 c = *reinterpret_cast<const float2*>(sq + i0 + k);
                    const float ex = fq.x - c.x, ey = fq.y - c.y;
                    v[k] = __builtin_fmaf(ex, ex, ey * ey);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) L.offer(sweep_pk(v[k], i0 + k, i0 + k < m));
            }
            T = tighter(T, fq.threshold(fq.dist_bound(L.m[NB])));
        }
        if (trail == 0) break;
        const int up = __builtin_ctz(trail);
        node = (node >> up) ^ 1;
        trail = (trail >> up) ^ 1u;
        tested = false;
    }
}

// sweepf_nn_far by the packed walk and scan: position of the nearest point and its exact squared distance
__device__ __forceinline__ int sweepf_nn_far_pk(const float4* sq, const double2* sxy, const float4* tree, int leaves, const SweepF& f, int m, int dir,
                                                double uabs, double qx, double qy, double& d2_out, int max_rounds = SWEEP_FAR_ROUNDS) {
    const SweepPkQuery fq(f, dir, uabs, qx, qy);
    SweepPkList<3> L;
    const int h0 = sweepf_lower_bound(sq, m, fq.u);
    const bool done = fq.polar ? sweep_pk_walk<3, 0, true>(sq, fq, h0 - 1, h0, m, -1, L, max_rounds)
                               : sweep_pk_walk<3, 0, false>(sq, fq, h0 - 1, h0, m, -1, L, max_rounds);
    if (!done) {                                                       // the scan's own list (see sweepf_top2_far_pk): no word twice
        SweepPkList<3> S;
        sweep_pk_far_scan<3, 0>(sq, tree, leaves, m, fq, S, fq.threshold(fq.dist_bound(L.m[0])));
        L = S;
    }
    const float T = fq.threshold(fq.dist_bound(L.m[0]));
    const bool a2 = L.m[1] != SWEEP_PK_NONE && !(sweep_pk_floor(L.m[1]) > T);
    const bool a3 = L.m[2] != SWEEP_PK_NONE && !(sweep_pk_floor(L.m[2]) > T);
    if (a3 || fq.bad || L.m[0] == SWEEP_PK_NONE) return sweepf_nn_far(sq, sxy, tree, leaves, f, m, dir, uabs, qx, qy, d2_out, max_rounds);
    int bpos = (int)(L.m[0] & SWEEP_PK_IDX);
    double best = sweep_d2(qx, qy, sxy[bpos]);
    if (a2) {                                                          // two candidates within the filter's resolution: exact, rows on a tie
        const int i2 = (int)(L.m[1] & SWEEP_PK_IDX);
        const double s2 = sweep_d2(qx, qy, sxy[i2]);
        if (s2 < best || (s2 == best && sweepf_row(sq[i2]) < sweepf_row(sq[bpos]))) { bpos = i2; best = s2; }
    }
    d2_out = best;
    return bpos;
}

// sweepf_top2 for the same kind of query (the continuation kernel of pairs the fused ICP finds metres off their target,
// icp2.hip): the walk, abandoned after SWEEP_FAR_ROUNDS rounds for the scan over block boxes with the third distance as
// the bound.  The scan passes over the positions the walk has visited once more: the two kept positions are skipped by
// index, any other revisited point is at least as far as the kept third distance and changes nothing.
__device__ __forceinline__ void top2_offer(Top2& t, const float4* sq, double s, int i, int row) {
    if (s == t.s1 || s == t.s2) {
        // exact tie with a kept distance: order by original row (the rule everywhere: (distance, row) ascending)
        const int r1 = sweepf_row(sq[t.p1]);
        const int r2 = t.p2 >= 0 ? sweepf_row(sq[t.p2]) : 0x7fffffff;
        if (s < t.s1 || (s == t.s1 && row < r1)) { t.s3 = t.s2; t.s2 = t.s1; t.p2 = t.p1; t.s1 = s; t.p1 = i; }
        else if (s < t.s2 || row < r2) { t.s3 = t.s2; t.s2 = s; t.p2 = i; }
        else t.s3 = s;                                             // tie with the second, lost on the row
    } else {
        const bool c1 = s < t.s1, c2 = s < t.s2, c3 = s < t.s3;
        t.s3 = c2 ? t.s2 : (c3 ? s : t.s3);
        t.s2 = c1 ? t.s1 : (c2 ? s : t.s2);
        t.p2 = c1 ? t.p1 : (c2 ? i : t.p2);
        t.s1 = c1 ? s : t.s1;
        t.p1 = c1 ? i : t.p1;
    }
}

// (measured on the 3 m / 20 degree candidates, ICP half: 24 rounds 6.35 ms, 12 5.73, 6 5.46, 3 5.42 — and 3.84 once rows far
// from their previous match stopped skipping the walk altogether: two or three rounds about the seed are what gives the
// descent its bound; 1 / 2 / 3 / 4 rounds 3.84 / 3.86 / 3.84 / 3.95)
#ifndef SWEEP_FAR_ROUNDS_TOP2
#define SWEEP_FAR_ROUNDS_TOP2 3
#endif
__device__ __forceinline__ Top2 sweepf_top2_far(const float4* sq, const double2* sxy, const float4* tree, int leaves, const SweepF& f, int m, int dir,
                                                double uabs, double qx, double qy, int seed, double* diag = nullptr) {
    DBG_T(d0);
    const SweepFQuery fq(f, dir, uabs, qx, qy);
    Top2 t;
    t.p1 = 0; t.p2 = -1;
    t.s1 = t.s2 = t.s3 = __builtin_inf();
    float W = __builtin_inff(), T = __builtin_inff();
    const bool seeded = seed >= 0 && seed < m;
    if (seeded) { t.s1 = sweep_d2(qx, qy, sxy[seed]); t.p1 = seed; }
    const bool from_seed = seeded && !(fq.polar && fabsf(sq[seed].z - fq.u) > 3.0f);
    const int h0 = from_seed ? seed + 1 : sweepf_lower_bound(sq, m, fq.u);
    SweepFWalk w(from_seed ? seed - 1 : h0 - 1, h0, m, fq.u);
    const int skip = seeded ? seed : -1;                               // the seed is in the list already
    int rounds = 0;
    bool far = false;
    DBG_T(d1);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        while (w.more()) {
            if (++rounds > SWEEP_FAR_ROUNDS_TOP2) { far = true; break; }
            const SweepFRound r(sq, w, fq, W, T);
            const bool pr = r.pr && w.hi != skip, pl = r.pl && w.lo != skip;
            if (pr || pl) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int i = side == 0 ? w.hi : w.lo;
                    if (side == 0 ? pr : pl) top2_offer(t, sq, sweep_d2(qx, qy, sxy[i]), i, sweepf_row(side == 0 ? r.cr : r.cl));
                }
                fq.bounds(t.s3, W, T);
            }
            r.advance(w);
        }
        if (far || !fq.polar || !w.wrap(m)) break;
    }
    DBG_T(d2);
    DBG_ADD(0, d0, d1); DBG_ADD(1, d1, d2); DBG_ADD(5, 0ull, 1ull);
    if (far) {
#ifdef ICPMI_DIAG
        if (diag) atomicAdd(diag, 4294967296.0);
#endif
        sweepf_far_scan(sq, tree, leaves, m, fq, T, [&](int i) {
            if (i == t.p2 || (i == t.p1 && t.s1 < __builtin_inf())) return;
            top2_offer(t, sq, sweep_d2(qx, qy, sxy[i]), i, sweepf_row(sq[i]));
            fq.bounds(t.s3, W, T);
        });
        DBG_T(d4);
        DBG_ADD(3, d2, d4); DBG_ADD(6, 0ull, 1ull);
    }
    return t;
}

// sweepf_top2_far by the packed walk and scan (the far continuation's searches).  A word offered twice would stand for two
// of the three nearest and bound the window by the SECOND distance: so the scan keeps a list of its own — everything
// the walk has met within its bound is met again, the walk only lends its threshold.
__device__ __forceinline__ Top2 sweepf_top2_far_pk(const float4* sq, const double2* sxy, const float4* tree, int leaves, const SweepF& f, int m, int dir,
                                                   double uabs, double qx, double qy, int seed) {
    const SweepPkQuery fq(f, dir, uabs, qx, qy);
    SweepPkList<4> L;
    const bool seeded = seed >= 0 && seed < m;
    const bool from_seed = seeded && !(fq.polar && fabsf(sq[seed].z - fq.u) > 3.0f);
    const int h0 = from_seed ? seed + 1 : sweepf_lower_bound(sq, m, fq.u);
    const bool done = fq.polar ? sweep_pk_walk<4, 2, true>(sq, fq, h0 - 1, h0, m, -1, L, SWEEP_FAR_ROUNDS_TOP2)
                               : sweep_pk_walk<4, 2, false>(sq, fq, h0 - 1, h0, m, -1, L, SWEEP_FAR_ROUNDS_TOP2);
    if (!done) {
        SweepPkList<4> S;
        sweep_pk_far_scan<4, 2>(sq, tree, leaves, m, fq, S, fq.threshold(fq.dist_bound(L.m[2])));
        L = S;
    }
    const float T = fq.threshold(fq.dist_bound(L.m[2]));
    const bool a4 = L.m[3] != SWEEP_PK_NONE && !(sweep_pk_floor(L.m[3]) > T);
    if (a4 || fq.bad || L.m[0] == SWEEP_PK_NONE) return sweepf_top2_far(sq, sxy, tree, leaves, f, m, dir, uabs, qx, qy, seed);
    return sweep_pk_top2_exact(sq, sxy, qx, qy, L.m[0], L.m[1], L.m[2]);
}

#endif


}  // namespace icpmi
