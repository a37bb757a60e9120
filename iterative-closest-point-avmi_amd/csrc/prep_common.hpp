// prep_common.hpp — pieces shared by the LDS (prep.hip) and global-memory (prep_big.hip) target preparation.
#pragma once
#include "linalg.hpp"
#include "sort.hpp"
#include "sweep.hpp"

namespace icpmi {

constexpr int PREP_BINS = 64;
#ifndef PREP_ABLATE
#define PREP_ABLATE 0           // diagnostic variants only (wrong results): 1 no walk, 2 no normal, 3 neither network nor walk nor normal
#endif

// Search axis of a cloud: ranges of the four projections x, y, x+y, x-y, a 64-bin histogram per axis, and
// the axis with the smallest expected search window (sum of squared bin counts / bin width; the diagonals
// pay sqrt(2) because their projection gap bounds the distance only up to that factor).  All threads of the
// workgroup call it and get the same answer.  dsc: 8 * THREADS/64 doubles, hist: 4 * PREP_BINS ints of LDS.
// bounds (optional): min x, max x, min y, max y of the cloud.
// polar > 0 adds a fifth candidate, the bearing about the frame origin (SWEEP_POLAR, sweep.hpp): a query at range r
// sees a window of half-width B / r in the bearing, so a bin's points cost (points of the bin) x (sum of 1 / r over
// the bin) / bin width — in the same unit as the projections' sum of squares / range.  1 / r is summed in fixed
// point (integer atomics: the choice is reproducible) and capped at 1 mm, so a cloud with points at the origin
// (no wedge there) never picks it.  polar == 2 forces it (tests).  hist: 6 * PREP_BINS ints.
template <int THREADS>
__device__ __forceinline__ int choose_axis(const double* __restrict__ P, int M, double* dsc, int* hist, double* bounds = nullptr,
                                           int polar = 0, int step = 1) {
    // step > 1: every step-th point only — the choice is an estimate of search cost, any answer is correct, and the
    // sample is the same every run; callers that need exact `bounds` (the grid) pass 1
    constexpr int MAXW = THREADS / ICPMI_WAVE;
    double mn[4], mx[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { mn[d] = __builtin_inf(); mx[d] = -__builtin_inf(); }
    for (int i = threadIdx.x * step; i < M; i += THREADS * step) {
        const double x = P[2 * i], y = P[2 * i + 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) { const double u = proj(d, x, y); mn[d] = fmin(mn[d], u); mx[d] = fmax(mx[d], u); }
    }
    const int w = wave_id(), l = lane_id();
#pragma unroll
    for (int d = 0; d < 4; ++d) { mn[d] = wave_min(mn[d]); mx[d] = wave_max(mx[d]); }
    if (l == 0)
#pragma unroll
        for (int d = 0; d < 4; ++d) { dsc[d * MAXW + w] = mn[d]; dsc[(4 + d) * MAXW + w] = mx[d]; }
    for (int i = threadIdx.x; i < (polar ? 6 : 4) * PREP_BINS; i += THREADS) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        double a = __builtin_inf(), b = -__builtin_inf();
        for (int q = 0; q < MAXW; ++q) { a = fmin(a, dsc[d * MAXW + q]); b = fmax(b, dsc[(4 + d) * MAXW + q]); }
        mn[d] = a; mx[d] = b;
    }
    double scale[4];                                       // bins per unit of the projection: one division per axis, not per point
#pragma unroll
    for (int d = 0; d < 4; ++d) { const double r = mx[d] - mn[d]; scale[d] = r > 0.0 ? (double)PREP_BINS / r : 0.0; }
    for (int i = threadIdx.x * step; i < M; i += THREADS * step) {
        const double x = P[2 * i], y = P[2 * i + 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            int b = (int)((proj(d, x, y) - mn[d]) * scale[d]);
            b = b < 0 ? 0 : (b >= PREP_BINS ? PREP_BINS - 1 : b);
            atomicAdd(&hist[d * PREP_BINS + b], 1);
        }
        if (polar) {
            // an estimate only: float32 is plenty (and every thread computes the same numbers: reproducible)
            const float xf = (float)x, yf = (float)y;
            int b = (int)((atan2f(yf, xf) + 3.14159265f) * (PREP_BINS / 6.2831853f));
            b = b < 0 ? 0 : (b >= PREP_BINS ? PREP_BINS - 1 : b);
            const float r = __builtin_amdgcn_sqrtf(xf * xf + yf * yf);
            const float w = r > 1e-3f ? 1.0f / r : 1e3f;
            atomicAdd(&hist[4 * PREP_BINS + b], 1);
            atomicAdd(reinterpret_cast<unsigned int*>(&hist[5 * PREP_BINS + b]), (unsigned int)(w * 1024.0f));   // <= 4096 points x 1e3 x 1024 < 2^32
        }
    }
    __syncthreads();
    if (bounds) { bounds[0] = mn[0]; bounds[1] = mx[0]; bounds[2] = mn[1]; bounds[3] = mx[1]; }
    // the costs: a lane per bin (PREP_BINS = a wave), fixed-tree wave sums of integer-valued terms — every wave gets
    // the same numbers, so every thread the same answer without another barrier
    static_assert(PREP_BINS == ICPMI_WAVE, "one lane per bin");
    int dir = 0;
    double bestc = __builtin_inf();
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const double cn = (double)hist[d * PREP_BINS + l];
        const double s = wave_sum(cn * cn);
        const double r = mx[d] - mn[d];
        const double cost = r > 0.0 ? (d < 2 ? 1.0 : 1.4142135623730951) * s / r : __builtin_inf();
        if (cost < bestc) { bestc = cost; dir = d; }
    }
    if (polar) {
        const double s = wave_sum((double)hist[4 * PREP_BINS + l] * ((double)(unsigned int)hist[5 * PREP_BINS + l] * (1.0 / 1024.0)));
        if (s * (1.0 / 6.283185307179586) < bestc || polar == 2) dir = SWEEP_POLAR;
    }
    return dir;
}

// Batcher's odd-even merge sort as a list of compare-exchange pairs (a < b), for N inputs: the network for the next
// power of two with every pair that touches an index >= N dropped (those inputs would be +inf and never move).
template <int N>
struct SortNet {
    int n;
    int a[N * 8], b[N * 8];
};
template <int N>
constexpr SortNet<N> make_sortnet() {
    SortNet<N> r{};
    int P = 1;
    while (P < N) P <<= 1;
    for (int p = 1; p < P; p <<= 1)
        for (int k = p; k >= 1; k >>= 1)
            for (int j = k % p; j <= P - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < P - j - k - 1 ? k - 1 : P - j - k - 1); ++i)
                    if ((i + j) / (p * 2) == (i + j + k) / (p * 2) && i + j + k < N) {
                        r.a[r.n] = i + j; r.b[r.n] = i + j + k; ++r.n;
                    }
    return r;
}

// k best (d2, sorted position), ascending by (d2, original row).  Every index is a compile-time constant
// (template recursion), so the lists stay in registers.  The rows are not kept: they only order candidates
// at exactly equal distances, where they are read from `sorig` (LDS / L2) on the spot.
template <int KK>
struct TopKP {
    double d[KK];
    int p[KK];
    template <int I>
    __device__ __forceinline__ void init_from() {
        if constexpr (I < KK) { d[I] = __builtin_inf(); p[I] = 0; init_from<I + 1>(); }
    }
    __device__ __forceinline__ void init() { init_from<0>(); }
    // The list filled with the KK consecutive sorted positions b0 .. b0+KK-1 and put in order by a fixed sorting
    // network: every lane does the same KK distances and the same compare-exchanges (selects, no branch but on an
    // exact tie of distances), instead of KK insertions that each lane would bubble to a different depth.
    template <int I>
    __device__ __forceinline__ void load_from(const double2* sxy, int b0, const double2 q) {
        if constexpr (I < KK) {
            const double2 c = sxy[b0 + I];
            const double dx = q.x - c.x, dy = q.y - c.y;
            double s = 0.0;
            s += dx * dx;
            s += dy * dy;
            d[I] = s; p[I] = b0 + I;
            load_from<I + 1>(sxy, b0, q);
        }
    }
    static constexpr SortNet<KK> NET = make_sortnet<KK>();
    template <int C>
    __device__ __forceinline__ void net_from(const int32_t* sorig) {
        if constexpr (C < NET.n) {
            constexpr int A = NET.a[C], B = NET.b[C];
            bool sw = d[B] < d[A];
            if (d[B] == d[A]) sw = sorig[p[B]] < sorig[p[A]];        // exact tie: original rows decide
            const double da = d[A], db = d[B];
            const int pa = p[A], pb = p[B];
            d[A] = sw ? db : da; d[B] = sw ? da : db;
            p[A] = sw ? pb : pa; p[B] = sw ? pa : pb;
            net_from<C + 1>(sorig);
        }
    }
    __device__ __forceinline__ void init_block(const double2* sxy, const int32_t* sorig, int b0, const double2 q) {
        load_from<0>(sxy, b0, q);
        net_from<0>(sorig);
    }
    template <int I>
    __device__ __forceinline__ void bubble(const int32_t* sorig) {
        if constexpr (I > 0) {
            // stop as soon as the new entry is in place: candidates arrive roughly by
            // increasing distance, so most insertions move one or two slots
            if (d[I] < d[I - 1] || (d[I] == d[I - 1] && sorig[p[I]] < sorig[p[I - 1]])) {
                const double td = d[I - 1]; const int tp = p[I - 1];
                d[I - 1] = d[I]; p[I - 1] = p[I];
                d[I] = td; p[I] = tp;
                bubble<I - 1>(sorig);
            }
        }
    }
    __device__ __forceinline__ bool push(double s, int pos, const int32_t* sorig) {
        if (s < d[KK - 1] || (s == d[KK - 1] && sorig[pos] < sorig[p[KK - 1]])) {
            d[KK - 1] = s; p[KK - 1] = pos;
            bubble<KK - 1>(sorig);
            return true;
        }
        return false;
    }
    template <int I>
    __device__ __forceinline__ double kth_from(int k, double v) const {
        if constexpr (I < KK) return kth_from<I + 1>(k, I == k ? d[I] : v);
        else return v;
    }
    __device__ __forceinline__ double kth(int k) const { return kth_from<1>(k, d[0]); }
    // sum of f(sxy[p[i]]) over i < kk, in list order
    template <int I, typename F>
    __device__ __forceinline__ void for_first(int kk, F&& f) const {
        if constexpr (I < KK) {
            if (I < kk) f(p[I]);
            for_first<I + 1>(kk, f);
        }
    }
};

// np.cov over the kk neighbours of sorted position s (summed in ascending (distance, row) order), eigenvector of the
// smaller eigenvalue, unit length (icp.py:66-76); written to the sorted copy and, optionally, to the row layout.
template <int KK>
__device__ __forceinline__ void emit_normal(const TopKP<KK>& top, int kk, const double2* sxy, const int32_t* sorig, int s,
                                            double2* __restrict__ out_sorted, double* __restrict__ out_rows) {
    double mx = 0.0, my = 0.0;
    top.template for_first<0>(kk, [&](int pos) { const double2 c = sxy[pos]; mx += c.x; my += c.y; });
    mx /= (double)kk; my /= (double)kk;
    double sxx = 0.0, sxy_ = 0.0, syy = 0.0;
    top.template for_first<0>(kk, [&](int pos) {
        const double2 c = sxy[pos];
        const double dx = c.x - mx, dy = c.y - my;
        sxx += dx * dx; sxy_ += dx * dy; syy += dy * dy;
    });
    double vx = 1.0, vy = 0.0;
    if (kk > 1) {
        const double den = (double)(kk - 1);                 // np.cov ddof = 1
        smallest_evec_2x2(sxx / den, sxy_ / den, syy / den, vx, vy);
    }
    double nn = sqrt(vx * vx + vy * vy);
    nn = nn < 1e-10 ? 1e-10 : nn;                            // icp.py:74-75
    const double2 n2 = make_double2(vx / nn, vy / nn);
    out_sorted[s] = n2;
    if (out_rows) { const int row = sorig[s]; out_rows[2 * row] = n2.x; out_rows[2 * row + 1] = n2.y; }
}

template <int KK>
__device__ __forceinline__ void prep_normals(const double2* sxy, const int32_t* sorig, int M, int s_begin, int s_end, int dir, int kk,
                                             double2* __restrict__ out_sorted, double* __restrict__ out_rows) {
    const double2 c_lo = sxy[0], c_hi = sxy[M - 1];
    const double uabs = fmax(fabs(proj(dir, c_lo.x, c_lo.y)), fabs(proj(dir, c_hi.x, c_hi.y)));
    const double pa = dir == 1 ? 0.0 : 1.0, pb = dir == 0 ? 0.0 : (dir == 3 ? -1.0 : 1.0);    // u = x*pa + y*pb (proj)
    for (int s = s_begin + threadIdx.x; s < s_end; s += blockDim.x) {
        const double2 q = sxy[s];
        const double uq = q.x * pa + q.y * pb;
        TopKP<KK> top;
        // the sweep on one side ends when the gap along the axis alone exceeds the current kk-th best
        // distance (inf until kk neighbours are known), compared in squares: no sqrt on the path
        // (gap_exceeds, sweep.hpp: exact on x / y; the diagonals allow for the rounding of x +- y)
        double bound = __builtin_inf();
        const double eps = dir < 2 ? 0.0 : 4.5e-16 * (fabs(uq) + uabs);
        const double widen = dir < 2 ? 1.0 : 2.000000000000002;
        int lo, hi;
        if (M >= KK) {
            // start from the KK points around the query's own sorted position (it is one of them): on a wall that
            // runs alone through the sweep window these ARE its neighbours and everything after is rejected
            const int b0 = min(max(s - KK / 2, 0), M - KK);
            top.init_block(sxy, sorig, b0, q);
            bound = (kk == KK ? top.d[KK - 1] : top.kth(kk - 1)) * widen;
            lo = b0 - 1; hi = b0 + KK;
        } else {
            top.init();
            top.push(0.0, s, sorig);
            lo = s - 1; hi = s + 1;
        }
        // one candidate from each open side per round, both loaded before use, the next pair fetched meanwhile
        double2 cr = sweep_load(sxy, hi, M), cl = sweep_load(sxy, lo, M);
        while (lo >= 0 || hi < M) {
            const double2 nr = sweep_load(sxy, hi + 1, M), nl = sweep_load(sxy, lo - 1, M);
            const double gr = (cr.x * pa + cr.y * pb) - uq - eps, gl = uq - (cl.x * pa + cl.y * pb) - eps;
            const bool inr = hi < M && !(gr > 0.0 && gr * gr > bound);
            const bool inl = lo >= 0 && !(gl > 0.0 && gl * gl > bound);
            const double sr = sweep_d2(q.x, q.y, cr), sl = sweep_d2(q.x, q.y, cl);
            bool changed = false;
            if (inr) changed = top.push(sr, hi, sorig);
            if (inl) changed = top.push(sl, lo, sorig) || changed;
            if (changed) bound = (kk == KK ? top.d[KK - 1] : top.kth(kk - 1)) * widen;
            hi = inr ? hi + 1 : M;
            lo = inl ? lo - 1 : -1;
            cr = nr; cl = nl;
        }
        emit_normal<KK>(top, kk, sxy, sorig, s, out_sorted, out_rows);
    }
}

// The same k-NN search on a cloud sorted by bearing (SWEEP_POLAR): the window is the wedge |dth| <= asin(B / |q|) with
// B the kk-th best distance so far, walked from the query's own position in both directions and once across the
// seam at +-pi (sweep.hpp has the bound and its margins).  sth = float32 images of the sort keys.  Same lists, same
// order, same normals as any other exact search.
template <int KK>
__device__ __forceinline__ void prep_normals_polar(const double2* sxy, const int32_t* sorig, const float* sth, int M, int s_begin, int s_end,
                                                   int kk, double2* __restrict__ out_sorted, double* __restrict__ out_rows) {
    for (int s = s_begin + threadIdx.x; s < s_end; s += blockDim.x) {
        const double2 q = sxy[s];
        const float thq = sth[s];
        const float kw = 1.00001f / __builtin_amdgcn_sqrtf((float)(q.x * q.x + q.y * q.y));     // >= 1 / |q| (inf at the origin: no wedge)
        TopKP<KK> top;
        float W = __builtin_inff();
        auto window = [&](double kth) {
            const float t = (__builtin_amdgcn_sqrtf((float)kth) * 1.000001f + 1e-18f) * kw;
            W = t <= 0.7f ? (t + 0.3f * t * t * t) * 1.000001f + 1.6e-5f : __builtin_inff();    // asin t <= t + 0.3 t^3 on [0, 0.7]; two images, each within 5.3e-6 of its bearing (prep.hip)
        };
        int lo, hi, remaining;                                       // remaining: positions not yet looked at
        if (M >= KK) {
            const int b0 = min(max(s - KK / 2, 0), M - KK);
            if (PREP_ABLATE == 3) { top.init(); top.template load_from<0>(sxy, b0, q); }     // distances only: no network
            else top.init_block(sxy, sorig, b0, q);
            window(kk == KK ? top.d[KK - 1] : top.kth(kk - 1));
            lo = b0 - 1; hi = b0 + KK; remaining = PREP_ABLATE == 3 ? 0 : M - KK;
        } else {
            top.init();
            top.push(0.0, s, sorig);
            lo = s - 1; hi = s + 1; remaining = M - 1;
        }
        // The array is a circle: a side that runs off its end continues at the other one with its bearings shifted by
        // 2 pi (a query next to the seam at +-pi has half of its neighbours there, and until they are seen the bound —
        // hence the window — may be as wide as the whole cloud).  `remaining` stops the two sides where they meet.
        float offr = 0.0f, offl = 0.0f;
        if (hi >= M) { hi -= M; offr = 6.2831855f; }
        if (lo < 0) { lo += M; offl = 6.2831855f; }
        bool openr = PREP_ABLATE != 1, openl = PREP_ABLATE != 1;          // (PREP_ABLATE: diagnostic variants only — 1 no walk, 2 no normal)
        // one candidate from each open side per round; bearings and points of both are loaded before either is used and
        // the next pair is fetched meanwhile
        float tr = sth[hi], tl = sth[lo];
        double2 cr = sxy[hi], cl = sxy[lo];
        while ((openr || openl) && remaining > 0) {
            const int nh = hi + 1 == M ? 0 : hi + 1, nl = lo == 0 ? M - 1 : lo - 1;
            const float ntr = sth[nh], ntl = sth[nl];
            const double2 ncr = sxy[nh], ncl = sxy[nl];
            const bool inr = openr && !((tr + offr) - thq > W);             // else: everything further round is farther than the kk-th
            bool inl = openl && !(thq - (tl - offl) > W);
            if (inr && inl && remaining == 1) inl = false;                   // both sides at the last position
            const double sr = sweep_d2(q.x, q.y, cr), sl = sweep_d2(q.x, q.y, cl);
            bool changed = false;
            if (inr) changed = top.push(sr, hi, sorig);
            if (inl) changed = top.push(sl, lo, sorig) || changed;
            if (changed) window(kk == KK ? top.d[KK - 1] : top.kth(kk - 1));
            openr = inr; openl = inl;
            remaining -= (inr ? 1 : 0) + (inl ? 1 : 0);
            if (inr) { if (nh == 0) offr = 6.2831855f; hi = nh; tr = ntr; cr = ncr; }
            if (inl) { if (lo == 0) offl = 6.2831855f; lo = nl; tl = ntl; cl = ncl; }
        }
        if (PREP_ABLATE == 2) { out_sorted[s] = make_double2(top.d[KK - 1], (double)top.p[0]); continue; }
        if (PREP_ABLATE == 3) { out_sorted[s] = make_double2((double)hi, (double)lo); continue; }
        emit_normal<KK>(top, kk, sxy, sorig, s, out_sorted, out_rows);
    }
}

// Grid over the cloud for the k-NN search of the normals (LDS, built in the scratch the sort has released):
// cell_end[c] = end of cell c in cell_pts (its start is cell_end[c-1]), cell_pts = sorted positions by cell.
struct PrepGrid {
    double min_x, min_y, h;
    int nx, ny;
    const uint32_t* cell_end;
    const uint16_t* cell_pts;
};

__device__ __forceinline__ int grid_cell_1d(double v, double mn, double h, int n) {
    const int c = (int)floor((v - mn) / h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// Build the grid (all threads of the workgroup).  bounds: min x, max x, min y, max y.  kk scales the cell so
// that the kk-th neighbour of a point on a wall lies about one cell away: points at spacing ~ perimeter / M.
__device__ __forceinline__ PrepGrid prep_grid_build(const double2* sxy, int M, const double* bounds, int kk, uint32_t* cell_end,
                                                    uint16_t* cell_pts, int cells_cap, int* wave_tot) {
    PrepGrid g;
    g.min_x = bounds[0]; g.min_y = bounds[2];
    const double w = bounds[1] - bounds[0], hg = bounds[3] - bounds[2];
    double h = 0.55 * (double)kk * 2.0 * (w + hg) / (double)M;
    if (!(h > 0.0) || !(h < 1e300)) h = 1.0;
    int nx, ny;
    for (;;) {
        const double fx = floor(w / h) + 1.0, fy = floor(hg / h) + 1.0;
        if (fx * fy <= (double)cells_cap) { nx = (int)fx; ny = (int)fy; break; }
        h *= 1.3;
    }
    g.h = h; g.nx = nx; g.ny = ny;
    g.cell_end = cell_end; g.cell_pts = cell_pts;
    const int nc = nx * ny;
    for (int c = threadIdx.x; c < nc; c += blockDim.x) cell_end[c] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const double2 p = sxy[i];
        atomicAdd(&cell_end[grid_cell_1d(p.y, g.min_y, h, ny) * nx + grid_cell_1d(p.x, g.min_x, h, nx)], 1u);
    }
    __syncthreads();
    // exclusive scan of the counts, in place: consecutive chunks per thread, wave scan of the chunk sums,
    // wave totals through `wave_tot` (one int per wave of LDS)
    {
        const int per = (nc + (int)blockDim.x - 1) / (int)blockDim.x;
        const int lo = min(nc, (int)threadIdx.x * per), hi = min(nc, lo + per);
        uint32_t sum = 0;
        for (int c = lo; c < hi; ++c) sum += cell_end[c];
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < ICPMI_WAVE; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, ICPMI_WAVE);
            if (lane_id() >= o) inc += t;
        }
        if (lane_id() == ICPMI_WAVE - 1) wave_tot[wave_id()] = (int)inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int w = 0; w < wave_id(); ++w) run += (uint32_t)wave_tot[w];
        for (int c = lo; c < hi; ++c) { const uint32_t n = cell_end[c]; cell_end[c] = run; run += n; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += blockDim.x) {               // fill: cursor = start, ends as the cell's end
        const double2 p = sxy[i];
        const uint32_t slot = atomicAdd(&cell_end[grid_cell_1d(p.y, g.min_y, h, ny) * nx + grid_cell_1d(p.x, g.min_x, h, nx)], 1u);
        cell_pts[slot] = (uint16_t)i;
    }
    __syncthreads();
    return g;
}

// estimate_normals_2d over the grid: rings of cells around the query's cell until the kk-th best distance is
// below the distance to every cell not yet visited (k rings done -> everything else is farther than k*h).
// Exact: the same (distance, row) order decides, only the candidates that cannot enter the list are skipped.
template <int KK>
__device__ __forceinline__ void prep_normals_grid(const double2* sxy, const int32_t* sorig, int M, int s_begin, int s_end, int kk,
                                                  const PrepGrid g, double2* __restrict__ out_sorted, double* __restrict__ out_rows) {
    for (int s = s_begin + threadIdx.x; s < s_end; s += blockDim.x) {
        const double2 q = sxy[s];
        TopKP<KK> top;
        top.init();
        const int cx = grid_cell_1d(q.x, g.min_x, g.h, g.nx), cy = grid_cell_1d(q.y, g.min_y, g.h, g.ny);
        const int kmax = max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy));
        for (int k = 0; k <= kmax; ++k) {
            // the 8k cells of ring k through ONE loop (one copy of the list insertion in the code): first the two
            // full rows cy -+ k, then the two columns cx -+ k between them
            const int n_row = k == 0 ? 1 : 2 * (2 * k + 1), n_ring = k == 0 ? 1 : 8 * k;
            for (int idx = 0; idx < n_ring; ++idx) {
                int xx, yy;
                if (idx < n_row) { xx = cx - k + (idx >> 1); yy = (idx & 1) ? cy + k : cy - k; }
                else { const int j = idx - n_row; yy = cy - k + 1 + (j >> 1); xx = (j & 1) ? cx + k : cx - k; }
                if (xx < 0 || xx >= g.nx || yy < 0 || yy >= g.ny) continue;
                const int c = yy * g.nx + xx;
                const uint32_t e = g.cell_end[c];
                for (uint32_t t = c ? g.cell_end[c - 1] : 0u; t < e; ++t) {
                    const int i = g.cell_pts[t];
                    const double2 p = sxy[i];
                    const double dx = q.x - p.x, dy = q.y - p.y;
                    double d2 = 0.0;
                    d2 += dx * dx;
                    d2 += dy * dy;
                    top.push(d2, i, sorig);
                }
            }
            const double kth = kk == KK ? top.d[KK - 1] : top.kth(kk - 1);
            const double reach = (double)k * g.h;
            if (kth < reach * reach * 0.999999999) break;              // nothing farther out can enter (or tie)
        }
        emit_normal<KK>(top, kk, sxy, sorig, s, out_sorted, out_rows);
    }
}

}  // namespace icpmi
