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

}  // namespace icpmi
