// raycast.hip — K6/K7: OccupancyGrid2D.update_scan (reference utilities/mapping.py:103-141).
//
// Reference order per scan: every in-bounds hit cell += l_hit (np.add.at, so
// duplicates accumulate), then every in-bounds cell of every Bresenham ray
// += l_miss, then the whole grid is clipped once.  Each += is
// float32(float64(cell) + l).  All hit adds are equal and all miss adds are
// equal, so a cell's new value is a function of (old value, H, M) only:
//   K6 ray_count:    integer atomics count H (high 16 bits) and M (low 16 bits)
//                    per cell — order independent, hence exact and reproducible;
//   K7 ray_finalize: walks the bounding box of the scan, and for every counted
//                    cell replays H + M rounded adds and the clip, then zeroes
//                    the counter.
// Float atomics cannot do this: they are order dependent and f32+f32 differs
// from f32(f64+f64).
//
// Bresenham cells (mapping.py:68-89: start included, end excluded,
// n = max(|dx|,|dy|) cells): cell k lies at major = m0 + sm*k and
// minor = n0 + sn*((2*k*dmin + dmaj - 1) / (2*dmaj)) — integer division — with
// the x axis as major when dx >= dy.  A wave covers 64 neighbouring beams at
// the same 16 steps; inside a chunk the quotient/remainder pair is advanced
// incrementally, and equal cells in adjacent lanes are merged into one atomic
// (all 2 048 beams start in the origin cell).
#include "common.hpp"

namespace icpmi {

constexpr int RC_THREADS = 256;
#ifndef ICPMI_RC_STEPS
#define ICPMI_RC_STEPS 16
#define ICPMI_RC_SLOTS 16
#define ICPMI_RC_FIN_BLOCKS 1024
#endif
constexpr int RC_STEPS = ICPMI_RC_STEPS;   // Bresenham steps per chunk
constexpr int RC_SLOTS = ICPMI_RC_SLOTS;   // chunk slots per 64-beam group (grid-stride over longer rays)
constexpr int RC_COORD_MAX = 1 << 29; // cell coordinates are clamped to +-2^29
constexpr int RC_FIN_BLOCKS = ICPMI_RC_FIN_BLOCKS;

struct GridDesc {
    int nx, ny;
    double min_x, min_y, res;
    int wx0, wx1, wy0, wy1;   // cells [wx0, wx1) x [wy0, wy1) this call may write: the grid (or one rank's band of rows in a
                              // sharded replay) cut to the box the caller says every ray stays in
    int bx0, by0, bw;         // counter region of one scan: cell (x, y) counts at (y - by0) * bw + (x - bx0); covers the window
    int ry0, ry1;             // the band of rows itself (a whole-band clip walks it; rays are only cut to it when it is not the grid)
};
__device__ __forceinline__ size_t counter_index(const GridDesc& g, int x, int y) { return (size_t)(y - g.by0) * g.bw + (x - g.bx0); }

// bounding-box slot: every field grows by atomicMax and 0 means "empty"
struct BBox {
    uint32_t inv_x0;   // nx - x0
    uint32_t inv_y0;   // ny - y0
    uint32_t x1p;      // x1 + 1
    uint32_t y1p;      // y1 + 1
};

__device__ __forceinline__ bool world_to_cell(double w, double mn, double res, int& out) {
    const double f = floor((w - mn) / res);          // mapping.py:58-59,96-97
    if (!(fabs(f) < 1e300)) return false;            // NaN / inf: the reference raises on int()
    out = f > (double)RC_COORD_MAX ? RC_COORD_MAX : (f < -(double)RC_COORD_MAX ? -RC_COORD_MAX : (int)f);
    return true;
}

// The same index with ONE multiplication: floor(fl((w - mn) / res)) can differ from floor((w - mn) * (1 / res)) only when
// the product lies within a few ulp of an integer — then (and for anything out of range) the division is done after all.
__device__ __forceinline__ bool world_to_cell_fast(double w, double mn, double res, double rinv, int& out) {
    const double d = (w - mn) * rinv, r = rint(d);
    if (!(fabs(d) < 4.0e8) || fabs(d - r) <= 1e-12 * fabs(d) + 1e-300) return world_to_cell(w, mn, res, out);
    out = (int)floor(d);
    return true;
}

__device__ __forceinline__ bool world_to_cell_q(double w, double mn, double res, int& out) {   // same index, the division only near an integer
    return world_to_cell_fast(w, mn, res, 1.0 / res, out);
}

// Walker over the cells of one ray.
struct Ray {
    int m0, n0, sm, sn;       // start on major / minor axis, step signs
    int dmaj, dmin, n;        // |delta| on major / minor axis, number of cells
    bool xmajor;
    int q;                    // minor offset at the current step
    long long rem;            // (2*k*dmin + dmaj - 1) mod (2*dmaj)

    __device__ __forceinline__ void init(int x0, int y0, int x1, int y1) {
        const int dx = abs(x1 - x0), dy = abs(y1 - y0);
        const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
        xmajor = dx >= dy;
        m0 = xmajor ? x0 : y0; n0 = xmajor ? y0 : x0;
        sm = xmajor ? sx : sy; sn = xmajor ? sy : sx;
        dmaj = xmajor ? dx : dy; dmin = xmajor ? dy : dx;
        n = dmaj;
        q = 0; rem = 0;
    }
    // position the walker on step k (k < n, so dmaj >= 1)
    __device__ __forceinline__ void seek(int k) {
        const long long num = 2ll * k * dmin + dmaj - 1, den = 2ll * dmaj;
        long long qq = (long long)floor((double)num / (double)den);   // quotient < 2^30: off by at most one
        long long r = num - qq * den;
        while (r < 0) { --qq; r += den; }
        while (r >= den) { ++qq; r -= den; }
        q = (int)qq; rem = r;
    }
    __device__ __forceinline__ void step() {
        rem += 2ll * dmin;
        if (rem >= 2ll * dmaj) { rem -= 2ll * dmaj; ++q; }
    }
    __device__ __forceinline__ void cell(int k, int& x, int& y) const {
        const int mj = m0 + sm * k, mn = n0 + sn * q;
        x = xmajor ? mj : mn; y = xmajor ? mn : mj;
    }
    // steps whose MAJOR coordinate lies inside [lo, hi): [klo, khi)
    __device__ __forceinline__ void clip_major(int lo, int hi, int& klo, int& khi) const {
        if (sm > 0) { klo = max(0, lo - m0); khi = min(n, hi - m0); }
        else { klo = max(0, m0 - hi + 1); khi = min(n, m0 - lo + 1); }
        if (khi < klo) khi = klo;
    }
    // first step whose minor offset is >= qq (n if none): q(k) >= qq  <=>  2*k*dmin + dmaj - 1 >= 2*dmaj*qq
    __device__ __forceinline__ int first_step_with_offset(long long qq) const {
        if (qq <= 0) return 0;
        if (qq > dmin) return n;
        const long long num = 2ll * dmaj * qq - dmaj + 1, den = 2ll * dmin;       // num > 0, den > 0 (qq <= dmin)
        // ceil(num / den) without a 64-bit integer division (a few hundred instructions on this machine, and every beam
        // that crosses a tile pays two): the float64 quotient is within a few units of it (num < 2^61), exact products settle it
        long long k = (long long)((double)num / (double)den);
        while (k * den < num) ++k;
        while (k > 0 && (k - 1) * den >= num) --k;
        return k > n ? n : (int)k;
    }
    // narrow [klo, khi) to the steps whose MINOR coordinate lies inside [lo, hi) (the offset grows with k)
    __device__ __forceinline__ void clip_minor(int lo, int hi, int& klo, int& khi) const {
        const long long qa = sn > 0 ? (long long)lo - n0 : (long long)n0 - hi + 1;   // offsets qa .. qb are inside
        const long long qb = sn > 0 ? (long long)hi - 1 - n0 : (long long)n0 - lo;
        if (qb < qa) { khi = klo; return; }
        klo = max(klo, first_step_with_offset(qa));
        khi = min(khi, first_step_with_offset(qb + 1));
        if (khi < klo) khi = klo;
    }
};

// mode bits
constexpr int RC_DO_HITS = 1, RC_DO_MISS = 2, RC_PACKED = 4;

// K6 body; `block` = index of this workgroup among the counting workgroups of the launch
__device__ __forceinline__ void ray_count_body(
    const GridDesc& g, const double* __restrict__ origin, const double* __restrict__ hits, int nb,
    uint32_t* __restrict__ counts, BBox* bbox, int mode, int block) {
    const int lane = lane_id();
    const int gw = (block * RC_THREADS + threadIdx.x) >> 6;             // global wave id
    const int group = gw / RC_SLOTS, slot = gw % RC_SLOTS;
    const int beam = group * ICPMI_WAVE + lane;
    int ox = 0, oy = 0, hx = 0, hy = 0;
    const bool okx = world_to_cell(origin[0], g.min_x, g.res, ox);
    const bool oky = world_to_cell(origin[1], g.min_y, g.res, oy);
    bool valid = okx && oky && beam < nb;
    if (valid) {
        const bool a = world_to_cell_q(hits[2 * (size_t)beam], g.min_x, g.res, hx);
        const bool b = world_to_cell_q(hits[2 * (size_t)beam + 1], g.min_y, g.res, hy);
        valid = a && b;
    }

    if (slot == 0) {
        // occupied cell, mapping.py:124-129
        const bool hit_in = valid && hx >= g.wx0 && hx < g.wx1 && hy >= g.wy0 && hy < g.wy1;
        if ((mode & RC_DO_HITS) && hit_in)
            atomicAdd(&counts[counter_index(g, hx, hy)], (mode & RC_PACKED) ? 0x10000u : 1u);
        // bounding box of everything this beam can touch: Bresenham stays inside
        // the rectangle spanned by its end points
        int bx0 = max(g.wx0, min(ox, hx)), bx1 = min(g.wx1 - 1, max(ox, hx));
        int by0 = max(g.wy0, min(oy, hy)), by1 = min(g.wy1 - 1, max(oy, hy));
        const bool any = valid && bx0 <= bx1 && by0 <= by1;
        uint32_t a = any ? (uint32_t)(g.nx - bx0) : 0u, b = any ? (uint32_t)(g.ny - by0) : 0u;
        uint32_t c = any ? (uint32_t)(bx1 + 1) : 0u, d = any ? (uint32_t)(by1 + 1) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = max(a, (uint32_t)__shfl_xor((int)a, o, ICPMI_WAVE));
            b = max(b, (uint32_t)__shfl_xor((int)b, o, ICPMI_WAVE));
            c = max(c, (uint32_t)__shfl_xor((int)c, o, ICPMI_WAVE));
            d = max(d, (uint32_t)__shfl_xor((int)d, o, ICPMI_WAVE));
        }
        if (lane == 0 && a) {
            atomicMax(&bbox->inv_x0, a); atomicMax(&bbox->inv_y0, b);
            atomicMax(&bbox->x1p, c); atomicMax(&bbox->y1p, d);
        }
    }
    if (!(mode & RC_DO_MISS)) return;

    // free cells along the ray, mapping.py:135-139
    Ray ray;
    ray.init(ox, oy, hx, hy);
    int klo = 0, khi = 0;
    const int minor_lo = ray.xmajor ? g.wy0 : g.wx0, minor_hi = ray.xmajor ? g.wy1 : g.wx1;
    if (valid) {
        if (ray.xmajor) ray.clip_major(g.wx0, g.wx1, klo, khi); else ray.clip_major(g.wy0, g.wy1, klo, khi);
        if (g.ry0 > 0 || g.ry1 < g.ny) ray.clip_minor(minor_lo, minor_hi, klo, khi);   // a band: skip the steps outside it
    }
    const int len = khi - klo;
    int lmax = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o, ICPMI_WAVE));
    for (int c0 = slot * RC_STEPS; c0 < lmax; c0 += RC_SLOTS * RC_STEPS) {      // wave-uniform trip count
        const bool live = c0 < len;
        if (live) ray.seek(klo + c0);
#pragma unroll 4
        for (int i = 0; i < RC_STEPS; ++i) {
            const int k = klo + c0 + i;
            long long cellid = -1;
            if (live && k < khi) {
                const int mn = ray.n0 + ray.sn * ray.q;
                if (mn >= minor_lo && mn < minor_hi) {
                    int x, y;
                    ray.cell(k, x, y);
                    cellid = (long long)counter_index(g, x, y);
                }
                ray.step();
            }
            // merge runs of equal cells in adjacent lanes into one atomic
            const long long prev = __shfl_up(cellid, 1, ICPMI_WAVE);
            const bool leader = lane == 0 || cellid != prev;
            const unsigned long long lead = __ballot(leader);
            if (leader && cellid >= 0) {
                const unsigned long long above = lane == 63 ? 0ull : (lead >> (lane + 1));
                const int run = above ? __ffsll((long long)above) : 64 - lane;
                atomicAdd(&counts[cellid], (uint32_t)run);
            }
        }
    }
}

// float32(float64(v) + l) repeated H then M times, then the per-scan clip.
__device__ __forceinline__ float apply_counts(float v0, uint32_t H, uint32_t M, double l_hit, double l_miss,
                                              float lo32, float hi32, bool clip) {
    if (clip) {
        // Saturated far beyond a clamp?  Every rounded add is within half an ulp of
        // the exact one, so |v_final - exact| <= (H+M) * ulp(vmax)/2 < bound.
        const double exact = (double)v0 + (double)H * l_hit + (double)M * l_miss;
        const double vmax = fabs((double)v0) + (double)H * fabs(l_hit) + (double)M * fabs(l_miss);
        const double bound = ((double)H + (double)M) * vmax * 1.2e-7 + 1e-30;
        if (exact + bound < (double)lo32) return lo32;
        if (exact - bound > (double)hi32) return hi32;
    }
    float v = v0;
    for (uint32_t i = 0; i < H; ++i) {
        const float nv = (float)((double)v + l_hit);
        if (nv == v) break;                       // fixed point: further equal adds change nothing
        v = nv;
    }
    for (uint32_t i = 0; i < M; ++i) {
        const float nv = (float)((double)v + l_miss);
        if (nv == v) break;
        v = nv;
    }
    if (clip) {                                    // np.clip on float32, mapping.py:141
        if (v < lo32) v = lo32;
        if (v > hi32) v = hi32;
    }
    return v;
}

// A group of consecutive scans counted in ONE launch, each into its own counter grid (cells are independent and
// the finalise pass replays the grids in scan order, so the result is the sequential one bit for bit); the
// launches of a replay drop from one per scan to one per group.
#ifndef ICPMI_RC_GROUP
#define ICPMI_RC_GROUP 32
#endif
constexpr int RC_GROUP_MAX = ICPMI_RC_GROUP;
struct ScanGroup {
    int n;                               // scans in the group
    int first_block[RC_GROUP_MAX + 1];   // counting workgroups of scan s: [first_block[s], first_block[s + 1])
    int nb[RC_GROUP_MAX];                // beams
    int origin_row[RC_GROUP_MAX];        // row of the scan in `origins`
    long long hit_row[RC_GROUP_MAX];     // first row of the scan in `hits`
};

struct FinArgs {
    float* log_odds;
    uint32_t* counts;     // first counter grid of the group
    size_t grid_stride;   // cells between the counter grids of consecutive scans of the group
    int n_grids;          // scans in the group (1 for the single-scan paths)
    const BBox* bbox;
    BBox* other;          // bounding-box slot to zero for a later group (may be null)
    double l_hit, l_miss;
    float lo32, hi32;
    int count_kind;       // 0 = packed (H<<16 | M), 1 = counts are hits, 2 = counts are misses
    int clip, full_clip;
    int use_window;       // 1: walk the caller's window (a single scan: its box IS the window) instead of the counted box
};

// K7 body; `block` of `nblocks` finalising workgroups
__device__ __forceinline__ void ray_finalize_body(const GridDesc& g, const FinArgs& f, int block, int nblocks) {
    float* __restrict__ log_odds = f.log_odds;
    uint32_t* __restrict__ counts = f.counts;
    const BBox* bbox = f.bbox;
    BBox* other = f.other;
    const double l_hit = f.l_hit, l_miss = f.l_miss;
    const float lo32 = f.lo32, hi32 = f.hi32;
    const int count_kind = f.count_kind, clip = f.clip, full_clip = f.full_clip;
    int x0, y0, x1, y1;
    const BBox bb = *bbox;
    if (full_clip) { x0 = 0; y0 = g.ry0; x1 = g.nx - 1; y1 = g.ry1 - 1; }        // the whole band, also outside the counted window
    else if (f.use_window) { x0 = g.wx0; y0 = g.wy0; x1 = g.wx1 - 1; y1 = g.wy1 - 1; }
    else {
        if (bb.inv_x0 == 0) { x0 = 0; y0 = 0; x1 = -1; y1 = -1; }
        else { x0 = g.nx - (int)bb.inv_x0; y0 = g.ny - (int)bb.inv_y0; x1 = (int)bb.x1p - 1; y1 = (int)bb.y1p - 1; }   // inside the band by construction
    }
    for (int y = y0 + block; y <= y1; y += nblocks)
        for (int x = x0 + threadIdx.x; x <= x1; x += RC_THREADS) {
            const size_t c = (size_t)y * g.nx + x;
            const bool counted = x >= g.wx0 && x < g.wx1 && y >= g.wy0 && y < g.wy1;      // only a whole-band clip walks beyond the window
            const size_t cc = counted ? counter_index(g, x, y) : 0;
            uint32_t cn[RC_GROUP_MAX];
            uint32_t any = 0;
#pragma unroll
            for (int s = 0; s < RC_GROUP_MAX; ++s) {                 // independent loads first, then the ordered replay
                cn[s] = counted && s < f.n_grids ? counts[(size_t)s * f.grid_stride + cc] : 0u;
                any |= cn[s];
            }
            if (any) {
                float v = log_odds[c];
#pragma unroll
                for (int s = 0; s < RC_GROUP_MAX; ++s)
                    if (cn[s]) {
                        counts[(size_t)s * f.grid_stride + cc] = 0;
                        const uint32_t H = count_kind == 0 ? cn[s] >> 16 : (count_kind == 1 ? cn[s] : 0u);
                        const uint32_t M = count_kind == 0 ? cn[s] & 0xffffu : (count_kind == 2 ? cn[s] : 0u);
                        v = apply_counts(v, H, M, l_hit, l_miss, lo32, hi32, clip != 0);
                    } else if (s == 0 && full_clip && clip) {        // untouched by the first scan, but its clip is whole-grid
                        v = v < lo32 ? lo32 : v;
                        v = v > hi32 ? hi32 : v;
                    }
                log_odds[c] = v;
            } else if (full_clip && clip) {
                float v = log_odds[c];
                if (v < lo32) v = lo32;
                if (v > hi32) v = hi32;
                log_odds[c] = v;
            }
        }
    if (other && block == 0 && threadIdx.x == 0) { other->inv_x0 = 0; other->inv_y0 = 0; other->x1p = 0; other->y1p = 0; }
}

__global__ __launch_bounds__(RC_THREADS) void ray_count_kernel(
    GridDesc g, const double* __restrict__ origin, const double* __restrict__ hits, int nb,
    uint32_t* __restrict__ counts, BBox* bbox, int mode) {
    ray_count_body(g, origin, hits, nb, counts, bbox, mode, blockIdx.x);
}

__global__ __launch_bounds__(RC_THREADS) void ray_finalize_kernel(GridDesc g, FinArgs f) {
    ray_finalize_body(g, f, blockIdx.x, gridDim.x);
}

// counting workgroup `block` of a group launch: find its scan, count into that scan's grid
__device__ __forceinline__ void ray_count_group_body(const GridDesc& g, const double* __restrict__ origins,
                                                     const double* __restrict__ hits, const ScanGroup& grp,
                                                     uint32_t* __restrict__ counts, size_t grid_stride, BBox* bbox, int block) {
    int s = 0;
#pragma unroll
    for (int t = 1; t < RC_GROUP_MAX; ++t) s += (t < grp.n && block >= grp.first_block[t]) ? 1 : 0;
    ray_count_body(g, origins + 2 * (size_t)grp.origin_row[s], hits + 2 * (size_t)grp.hit_row[s], grp.nb[s],
                   counts + (size_t)s * grid_stride, bbox, RC_DO_HITS | RC_DO_MISS | RC_PACKED, block - grp.first_block[s]);
}

__global__ __launch_bounds__(RC_THREADS) void ray_count_group_kernel(
    GridDesc g, const double* __restrict__ origins, const double* __restrict__ hits, ScanGroup grp,
    uint32_t* __restrict__ counts, size_t grid_stride, BBox* bbox) {
    ray_count_group_body(g, origins, hits, grp, counts, grid_stride, bbox, blockIdx.x);
}

// One launch of a replay: the first n_count workgroups count group k into one set of counter grids while the
// others finalise group k-1 from the other set — the two touch disjoint memory, so a replay of S scans in
// groups of G is S/G + 1 dependent launches instead of 2S.
__global__ __launch_bounds__(RC_THREADS) void ray_step_kernel(
    GridDesc g, const double* __restrict__ origins, const double* __restrict__ hits, ScanGroup grp,
    uint32_t* __restrict__ counts, size_t grid_stride, BBox* bbox, int n_count, FinArgs f) {
    if ((int)blockIdx.x < n_count) ray_count_group_body(g, origins, hits, grp, counts, grid_stride, bbox, blockIdx.x);
    else ray_finalize_body(g, f, blockIdx.x - n_count, gridDim.x - n_count);
}

// ── K6 by tiles: counters of a 64 x 64 tile in LDS ────────────────────────────────────────────────────────────────
// The counting pass above is bound by the rate of scattered atomic line requests at L2 (~27 G/s on an MI355X
// whatever the scope or the size of the region, tools/ubench/atomic_scope.hip; ~413 G/s when the 64 lanes of an
// instruction fall on consecutive words), about one request per distinct (64-beam wave, cell).  Here a workgroup
// owns one tile of one scan's counter grid and a quarter of the scan's beams: it cuts every beam to the tile
// (exact step range from the closed form, after a bounding-box and a side-of-line rejection), walks the part
// inside into LDS counters (LDS atomics: conflicts cost cycles, not requests), and adds the tile's non-zero
// counters to the scan's grid row by row — consecutive words, one or two line requests per 64 cells.  Integer
// counts, so the grids (and everything after them) are the same bit for bit.
//   ray_scan_boxes_kernel  one workgroup per scan of the group: cell box of what its beams can touch (cut to the
//                          window), so that the workgroups of the other tiles leave at once;
//   ray_tile_body          one workgroup per (scan, tile, beam chunk).
constexpr int RT_TILE = 64, RT_THREADS = 256;
constexpr int RT_FAR = 4, RT_CHUNKS = 16;           // workgroups per (scan, tile): 4, and 16 for the 3 x 3 tiles around the origin
__host__ __device__ constexpr int rt_blocks_per_scan(int n_tiles) { return n_tiles * RT_FAR + 9 * (RT_CHUNKS - RT_FAR); }
constexpr int RT_STEPS = 16, RT_BLOCKS = RT_TILE / RT_STEPS;   // a thread walks blocks of 16 steps

struct ScanBox { int x0, y0, x1, y1; };           // inclusive; x1 < x0: nothing to touch

// beams of a scan are cut into RT_CHUNKS parts of rt_part(nb) consecutive beams (whole waves)
__host__ __device__ constexpr int rt_part(int nb) { return ((nb + RT_CHUNKS - 1) / RT_CHUNKS + ICPMI_WAVE - 1) & ~(ICPMI_WAVE - 1); }

constexpr int RT_BOXES = 1 + RT_CHUNKS;           // per scan: the box of the scan, then one per part

// boxes[s * RT_BOXES]: what the beams of scan s can touch, cut to the window; boxes[s * RT_BOXES + 1 + p]: the same
// for part p alone — a lidar reports its beams in angular order, so a part is a wedge and most tiles lie outside
// its box (any order is correct; it only decides how many beams a tile has to look at).  One workgroup per scan
// (one per part measured slower: 16 times the workgroups for the dispatcher).
__device__ __forceinline__ void ray_scan_boxes_body(const GridDesc& g, const double* __restrict__ origins,
                                                    const double* __restrict__ hits, const ScanGroup& grp,
                                                    ScanBox* __restrict__ boxes, int s) {
    __shared__ uint32_t bb[RT_BOXES][4];
    const int tid = threadIdx.x;
    const int nb = grp.nb[s], part = rt_part(nb);
    const double* h = hits + 2 * (size_t)grp.hit_row[s];
    const double* origin = origins + 2 * (size_t)grp.origin_row[s];
    if (tid < RT_BOXES * 4) (&bb[0][0])[tid] = 0u;
    int ox = 0, oy = 0;
    const bool ok = world_to_cell(origin[0], g.min_x, g.res, ox) && world_to_cell(origin[1], g.min_y, g.res, oy);
    __syncthreads();
    for (int base = 0; ok && base < nb; base += RT_THREADS) {           // a wave's 64 beams belong to one part
        const int i = base + tid;
        uint32_t a = 0, b = 0, c = 0, d = 0;
        int hx = 0, hy = 0;
        if (i < nb && world_to_cell_q(h[2 * (size_t)i], g.min_x, g.res, hx) && world_to_cell_q(h[2 * (size_t)i + 1], g.min_y, g.res, hy)) {
            const int bx0 = max(g.wx0, min(ox, hx)), bx1 = min(g.wx1 - 1, max(ox, hx));
            const int by0 = max(g.wy0, min(oy, hy)), by1 = min(g.wy1 - 1, max(oy, hy));
            if (bx0 <= bx1 && by0 <= by1) { a = (uint32_t)(g.nx - bx0); b = (uint32_t)(g.ny - by0); c = (uint32_t)(bx1 + 1); d = (uint32_t)(by1 + 1); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = max(a, (uint32_t)__shfl_xor((int)a, o, ICPMI_WAVE));
            b = max(b, (uint32_t)__shfl_xor((int)b, o, ICPMI_WAVE));
            c = max(c, (uint32_t)__shfl_xor((int)c, o, ICPMI_WAVE));
            d = max(d, (uint32_t)__shfl_xor((int)d, o, ICPMI_WAVE));
        }
        if (lane_id() == 0 && a) {
            const int p = 1 + i / part;
            atomicMax(&bb[0][0], a); atomicMax(&bb[0][1], b); atomicMax(&bb[0][2], c); atomicMax(&bb[0][3], d);
            atomicMax(&bb[p][0], a); atomicMax(&bb[p][1], b); atomicMax(&bb[p][2], c); atomicMax(&bb[p][3], d);
        }
    }
    __syncthreads();
    if (tid < RT_BOXES) {
        ScanBox r{0, 0, -1, -1};
        if (bb[tid][0]) r = ScanBox{g.nx - (int)bb[tid][0], g.ny - (int)bb[tid][1], (int)bb[tid][2] - 1, (int)bb[tid][3] - 1};
        boxes[(size_t)s * RT_BOXES + tid] = r;
    }
}

__global__ __launch_bounds__(RT_THREADS) void ray_scan_boxes_kernel(GridDesc g, const double* __restrict__ origins,
                                                                   const double* __restrict__ hits, ScanGroup grp,
                                                                   ScanBox* __restrict__ boxes) {
    ray_scan_boxes_body(g, origins, hits, grp, boxes, blockIdx.x);
}

struct TileArgs {
    const ScanBox* boxes;      // boxes of the scans of the group
    int tiles_x, tiles_y;
};

__device__ __forceinline__ void ray_tile_body(const GridDesc& g, const double* __restrict__ origins,
                                              const double* __restrict__ hits, const ScanGroup& grp,
                                              uint32_t* __restrict__ counts, size_t grid_stride, BBox* bbox,
                                              const TileArgs& ta, int block) {
    __shared__ __attribute__((aligned(16))) uint32_t cnt[RT_TILE * RT_TILE];
    __shared__ int4 seg[RT_THREADS];                    // beams of this pass that cross the tile: hit cell, step range
    __shared__ uint16_t items[RT_THREADS * RT_BLOCKS];  // (beam << 2 | block of RT_STEPS steps)
    __shared__ int n_items;
    const int n_tiles = ta.tiles_x * ta.tiles_y, per_scan = rt_blocks_per_scan(n_tiles);
    const int s = block / per_scan, rem = block - s * per_scan;
    const int tid = threadIdx.x;
    const double* origin = origins + 2 * (size_t)grp.origin_row[s];
    int ox = 0, oy = 0;
    const bool ok = world_to_cell(origin[0], g.min_x, g.res, ox) && world_to_cell(origin[1], g.min_y, g.res, oy);
    // Every tile has RT_FAR workgroups, each with a quarter of the beams.  The 3 x 3 tiles around the origin see every
    // beam at full length: their beams are cut into RT_CHUNKS parts, and the extra workgroups come after the regular ones.
    const int otx = (ox - g.wx0) >> 6, oty = (oy - g.wy0) >> 6;            // arithmetic shift: floor, also left of the window
    int tcx, tcy, chunk;
    if (rem < n_tiles * RT_FAR) {
        const int tile = rem / RT_FAR;
        chunk = rem - tile * RT_FAR;
        tcx = tile % ta.tiles_x; tcy = tile / ta.tiles_x;
    } else {
        const int e = rem - n_tiles * RT_FAR, nt = e / (RT_CHUNKS - RT_FAR);
        chunk = RT_FAR + e - nt * (RT_CHUNKS - RT_FAR);
        tcx = otx + nt % 3 - 1; tcy = oty + nt / 3 - 1;
        if (tcx < 0 || tcx >= ta.tiles_x || tcy < 0 || tcy >= ta.tiles_y) return;
    }
    const bool near = abs(tcx - otx) <= 1 && abs(tcy - oty) <= 1;
    const bool first = rem == 0;
    const int x0 = g.wx0 + tcx * RT_TILE, y0 = g.wy0 + tcy * RT_TILE;
    const int x1 = min(x0 + RT_TILE, g.wx1), y1 = min(y0 + RT_TILE, g.wy1);
    if (ta.boxes) {
        const ScanBox sb = ta.boxes[(size_t)s * RT_BOXES];
        if (first && tid == 0 && sb.x1 >= sb.x0) {                       // the group's box for the finalise pass
            atomicMax(&bbox->inv_x0, (uint32_t)(g.nx - sb.x0)); atomicMax(&bbox->inv_y0, (uint32_t)(g.ny - sb.y0));
            atomicMax(&bbox->x1p, (uint32_t)(sb.x1 + 1)); atomicMax(&bbox->y1p, (uint32_t)(sb.y1 + 1));
        }
        if (sb.x1 < x0 || sb.x0 >= x1 || sb.y1 < y0 || sb.y0 >= y1) return;              // uniform: the scan does not reach this tile
    } else if (first && tid == 0) {                                     // a single scan: the caller's box is the scan's
        atomicMax(&bbox->inv_x0, (uint32_t)(g.nx - g.wx0)); atomicMax(&bbox->inv_y0, (uint32_t)(g.ny - g.wy0));
        atomicMax(&bbox->x1p, (uint32_t)g.wx1); atomicMax(&bbox->y1p, (uint32_t)g.wy1);
    }
    const double* h = hits + 2 * (size_t)grp.hit_row[s];
    // a near workgroup takes one part of the beams, a far one four
    const int nb = grp.nb[s], part = rt_part(nb);
    const int p0 = near ? chunk : chunk * (RT_CHUNKS / RT_FAR), p1 = near ? chunk + 1 : (chunk + 1) * (RT_CHUNKS / RT_FAR);
    if ((!near && chunk >= RT_FAR) || p0 * part >= nb || !ok) return;
    uint4* cnt4 = reinterpret_cast<uint4*>(cnt);
    bool dirty = false;                                                  // uniform: the counters are zeroed by the first pass that needs them
    for (int pt = p0; pt < p1 && pt * part < nb; ++pt) {
      if (ta.boxes) {                                                    // the part's wedge misses the tile: skip its beams unread
          const ScanBox pb = ta.boxes[(size_t)s * RT_BOXES + 1 + pt];
          if (pb.x1 < x0 || pb.x0 >= x1 || pb.y1 < y0 || pb.y0 >= y1) continue;
      }
      const int b1 = min(nb, (pt + 1) * part);
      for (int base = pt * part; base < b1; base += RT_THREADS) {        // uniform trip counts
        if (tid == 0) n_items = 0;
        const int i = base + tid;
        int hx = 0, hy = 0, klo = 0, khi = 0;
        bool hit_in = false;
        if (i < b1 && world_to_cell_q(h[2 * (size_t)i], g.min_x, g.res, hx) && world_to_cell_q(h[2 * (size_t)i + 1], g.min_y, g.res, hy)) {
            hit_in = hx >= x0 && hx < x1 && hy >= y0 && hy < y1;
            // Bresenham stays inside the rectangle of its end points ...
            bool cross = !(max(ox, hx) < x0 || min(ox, hx) >= x1 || max(oy, hy) < y0 || min(oy, hy) >= y1);
            if (cross) {
                // ... and within one cell of the straight line: a tile (grown by a cell) whose corners all lie
                // strictly on one side of the line cannot be touched
                const long long dx = hx - ox, dy = hy - oy;
                const long long cxa = x0 - 1 - ox, cxb = x1 - ox, cya = y0 - 1 - oy, cyb = y1 - oy;
                const long long c0 = dx * cya - dy * cxa, c1 = dx * cya - dy * cxb, c2 = dx * cyb - dy * cxa, c3 = dx * cyb - dy * cxb;
                cross = !((c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0) || (c0 < 0 && c1 < 0 && c2 < 0 && c3 < 0));
            }
            if (cross) {
                Ray ray;
                ray.init(ox, oy, hx, hy);
                if (ray.xmajor) { ray.clip_major(x0, x1, klo, khi); ray.clip_minor(y0, y1, klo, khi); }
                else { ray.clip_major(y0, y1, klo, khi); ray.clip_minor(x0, x1, klo, khi); }
            }
        }
        if (!__syncthreads_or(hit_in || khi > klo)) continue;            // nothing of this pass touches the tile
        if (!dirty) {
            for (int c = tid; c < RT_TILE * RT_TILE / 4; c += RT_THREADS) cnt4[c] = make_uint4(0u, 0u, 0u, 0u);
            dirty = true;
            __syncthreads();
        }
        if (hit_in) atomicAdd(&cnt[(hy - y0) * RT_TILE + (hx - x0)], 0x10000u);     // mapping.py:124-129
        if (khi > klo) {                                                 // at most RT_TILE steps: RT_BLOCKS blocks
            const int nblk = (khi - klo + RT_STEPS - 1) / RT_STEPS;
            const int slot = atomicAdd(&n_items, nblk);
            seg[tid] = make_int4(hx, hy, klo, khi);
            for (int b = 0; b < nblk; ++b) items[slot + b] = (uint16_t)(tid << 2 | b);
        }
        __syncthreads();
        const int total = n_items;
        for (int it = tid; it < total; it += RT_THREADS) {               // free cells, mapping.py:135-139
            const int e = items[it];
            const int4 sg = seg[e >> 2];
            Ray ray;
            ray.init(ox, oy, sg.x, sg.y);
            const int k0 = sg.z + (e & 3) * RT_STEPS, k1 = min(sg.w, k0 + RT_STEPS);
            ray.seek(k0);
            for (int k = k0; k < k1; ++k) {
                int x, y;
                ray.cell(k, x, y);
                atomicAdd(&cnt[(y - y0) * RT_TILE + (x - x0)], 1u);
                ray.step();
            }
        }
        __syncthreads();
      }
    }
    if (!dirty) return;
    uint32_t* plane = counts + (size_t)s * grid_stride;
    for (int c = tid; c < RT_TILE * RT_TILE / 4; c += RT_THREADS) {      // four cells of a row per lane: a wave covers four rows
        const uint4 v = cnt4[c];
        if (!(v.x | v.y | v.z | v.w)) continue;
        uint32_t* row = &plane[counter_index(g, x0 + ((c * 4) & (RT_TILE - 1)), y0 + ((c * 4) >> 6))];   // only cells inside the tile's cut were counted
        if (v.x) atomicAdd(row, v.x);
        if (v.y) atomicAdd(row + 1, v.y);
        if (v.z) atomicAdd(row + 2, v.z);
        if (v.w) atomicAdd(row + 3, v.w);
    }
}

// One launch of a replay on the tile path: the first n_count workgroups count group k, the next n_fin finalise group
// k-1 from the other set of counter grids (disjoint memory), the last ones find the scan boxes of group k+1.
__global__ __launch_bounds__(RT_THREADS) void ray_tile_step_kernel(GridDesc g, const double* __restrict__ origins,
                                                                  const double* __restrict__ hits, ScanGroup grp,
                                                                  uint32_t* __restrict__ counts, size_t grid_stride, BBox* bbox, TileArgs ta,
                                                                  int n_items, int n_count, int n_fin, FinArgs f, ScanGroup nxt,
                                                                  ScanBox* nxt_boxes) {
    // the boxes of the next group first: few workgroups with a long chain of dependent loads, they must not start last
    if ((int)blockIdx.x < nxt.n) { ray_scan_boxes_body(g, origins, hits, nxt, nxt_boxes, blockIdx.x); return; }
    const int b = blockIdx.x - nxt.n;
    if (b < n_count) {
        // n_count workgroups share the n_items (scan, tile, chunk) items: launching one workgroup per item is bound by
        // the dispatch rate (~190 workgroups/us with 23 KB of LDS each), not by the work
        for (int w = b; w < n_items; w += n_count) {
            ray_tile_body(g, origins, hits, grp, counts, grid_stride, bbox, ta, w);
            __syncthreads();                    // the counters are zeroed again by the next item
        }
    } else ray_finalize_body(g, f, b - n_count, n_fin);
}


// ── the live update in ONE launch: a workgroup OWNS a piece of the window ─────────────────────────────────────────
// One update_scan of the running SLAM loop (slam.py:552-557) is ~250 000 cell visits: microseconds of work, and the two
// passes above are two dependent launches with a round trip of every counter through L2 atomics in between (9 + 5 us
// of kernels, 18.6 us per call).  Here a workgroup owns a rectangle of cells — a 64 x 64 tile, or a 16 x 16 piece of the
// 3 x 3 tiles around the origin, which every beam crosses — walks every beam's part inside it into LDS counters (a wave
// = 64 consecutive beams at the same step; a lidar reports beams in angular order, so near the origin the lanes of a
// wave sit in the same few cells and their runs merge into one LDS atomic each) and then applies the H / M replay and
// the clip to its own cells of the grid directly: no counter grid, no second launch, nothing to zero afterwards.  Same
// integer counts per cell as the other passes, same replay: the same grid bit for bit.
constexpr int RO_THREADS = 1024;                                 // 16 waves: a 2 048-beam scan is two passes (what a piece at the
                                                                 // origin waits for is its passes over ALL the beams)
constexpr int RO_STEPS = 16;                                     // steps per walk item
constexpr int RO_FAR = 64;                                       // pieces away from the origin: whole tiles (32 x 32 pieces quarter their
                                                                 // walks but start four times the workgroups: 15.9 against 13.4 us)
constexpr int RO_SUB0 = 8, RO_SUB1 = 16;                         // the origin's tile in 8 x 8 pieces, the 8 tiles around it in 16 x 16
constexpr int RO_N0 = (RT_TILE / RO_SUB0) * (RT_TILE / RO_SUB0), RO_N1 = (RT_TILE / RO_SUB1) * (RT_TILE / RO_SUB1);
constexpr int RO_NEAR = RO_N0 + 8 * RO_N1;                       // 64 + 128 workgroups for the 3 x 3 tiles around the origin

__global__ __launch_bounds__(RO_THREADS) void ray_owner_kernel(GridDesc g, const double* __restrict__ origin,
                                                              const double* __restrict__ hits, int nb, FinArgs f,
                                                              int tiles_x, int tiles_y, unsigned long long* dbg) {
#ifdef RO_X_TIMES        // diagnostic build: cycles per phase of every workgroup into the workspace (tools/time_livescan.py)
    const unsigned long long tt0 = __builtin_readcyclecounter();
    unsigned long long t_sel = 0, t_walk = 0;
    if (threadIdx.x == 0) { dbg[4 * blockIdx.x] = 0; dbg[4 * blockIdx.x + 1] = 0; dbg[4 * blockIdx.x + 2] = 0; dbg[4 * blockIdx.x + 3] = 1; }
#endif
    __shared__ uint32_t cnt[RT_TILE * RT_TILE];
    __shared__ int4 seg[2 * RO_THREADS];                             // this chunk's beams: hit cell, step range inside the rectangle
    __shared__ uint16_t items[2 * RO_THREADS / ICPMI_WAVE * (RT_TILE / RO_STEPS)];   // (wave of beams << 8 | block of RO_STEPS steps)
    __shared__ int n_items[2];                                       // by pass parity: the other one is reset while this one is read
    const int tid = threadIdx.x, lane = lane_id();
    // What this thread will need from HBM is asked for before anything else — its beams of the first chunk here, the cells
    // it finalises as soon as the origin (itself a round trip) has told the workgroup which cells it owns: each is a ~1 us
    // round trip, and the phases below are short chains of dependent steps with nothing to hide one behind.
    constexpr int PER_CHUNK = 2;                                     // a chunk = 2 x 1 024 beams: a 2 048-beam scan is one chunk
    double hwx[PER_CHUNK], hwy[PER_CHUNK];
#pragma unroll
    for (int j = 0; j < PER_CHUNK; ++j) {
        const int i = j * RO_THREADS + tid;
        hwx[j] = i < nb ? hits[2 * (size_t)i] : 0.0; hwy[j] = i < nb ? hits[2 * (size_t)i + 1] : 0.0;
    }
    int ox = 0, oy = 0;
    if (!(world_to_cell(origin[0], g.min_x, g.res, ox) && world_to_cell(origin[1], g.min_y, g.res, oy))) return;   // mapping.py: int() raises
    const int otx = (ox - g.wx0) >> 6, oty = (oy - g.wy0) >> 6;      // arithmetic shift: floor, also left of the window
    // the rectangle this workgroup owns (uniform)
    int x0, y0, x1, y1;
    const int b = blockIdx.x;
    if (b < RO_N0) {                                                 // the origin's own tile: 8 x 8 pieces of 8 x 8 cells
        constexpr int PER = RT_TILE / RO_SUB0;
        if (otx < 0 || otx >= tiles_x || oty < 0 || oty >= tiles_y) return;
        x0 = g.wx0 + otx * RT_TILE + (b % PER) * RO_SUB0; y0 = g.wy0 + oty * RT_TILE + (b / PER) * RO_SUB0;
        x1 = x0 + RO_SUB0; y1 = y0 + RO_SUB0;
    } else if (b < RO_NEAR) {                                        // the eight tiles around it: 4 x 4 pieces of 16 x 16 cells
        constexpr int PER = RT_TILE / RO_SUB1;
        const int e = b - RO_N0, t8 = e / RO_N1, sub = e - t8 * RO_N1, t9 = t8 < 4 ? t8 : t8 + 1;   // 0..8 without the centre (4)
        const int tcx = otx + t9 % 3 - 1, tcy = oty + t9 / 3 - 1;
        if (tcx < 0 || tcx >= tiles_x || tcy < 0 || tcy >= tiles_y) return;
        x0 = g.wx0 + tcx * RT_TILE + (sub % PER) * RO_SUB1; y0 = g.wy0 + tcy * RT_TILE + (sub / PER) * RO_SUB1;
        x1 = x0 + RO_SUB1; y1 = y0 + RO_SUB1;
    } else {                                                         // everywhere else (tiles_x, tiles_y count 64 x 64 tiles)
        constexpr int PER = RT_TILE / RO_FAR;
        const int t = b - RO_NEAR, fx = t % (tiles_x * PER), fy = t / (tiles_x * PER);
        const int tcx = fx / PER, tcy = fy / PER;
        if (tcy >= tiles_y || (abs(tcx - otx) <= 1 && abs(tcy - oty) <= 1)) return;   // the pieces above cover those
        x0 = g.wx0 + fx * RO_FAR; y0 = g.wy0 + fy * RO_FAR;
        x1 = x0 + RO_FAR; y1 = y0 + RO_FAR;
    }
    x1 = min(x1, g.wx1); y1 = min(y1, g.wy1);
    if (x0 >= x1 || y0 >= y1) return;
    const int w = x1 - x0, h = y1 - y0;
    // the cells this thread finalises (this workgroup is their only writer)
    constexpr int FIN = RT_TILE * RT_TILE / RO_THREADS;              // cells per thread at most
    float old[FIN];
#pragma unroll
    for (int j = 0; j < FIN; ++j) {
        const int c = j * RO_THREADS + tid;
        old[j] = c < w * h ? f.log_odds[(size_t)(y0 + c / w) * g.nx + (x0 + c % w)] : 0.0f;
    }
    for (int c = tid; c < w * h; c += RO_THREADS) cnt[c] = 0u;
    if (tid == 0) { n_items[0] = 0; n_items[1] = 0; }
    __syncthreads();
    // the rectangle in world coordinates, grown by two cells: a beam whose end points both lie on one side of it, or whose
    // line passes all four corners on one side, cannot touch it (Bresenham stays within a cell of the line) — decided
    // without the divisions of a cell index; nearly every beam of a tile away from the origin leaves here
    const double wxa = g.min_x + ((double)x0 - 2.0) * g.res, wxb = g.min_x + ((double)x1 + 2.0) * g.res;
    const double wya = g.min_y + ((double)y0 - 2.0) * g.res, wyb = g.min_y + ((double)y1 + 2.0) * g.res;
    const double owx = origin[0], owy = origin[1];
    const double rinv = 1.0 / g.res;
    for (int base = 0; base < nb; base += PER_CHUNK * RO_THREADS) {  // uniform trip count
#ifdef RO_X_TIMES
        const unsigned long long ta = __builtin_readcyclecounter();
#endif
        const int par = (base / (PER_CHUNK * RO_THREADS)) & 1;
#pragma unroll
        for (int j = 0; j < PER_CHUNK; ++j) {                        // a wave's 64 beams are neighbours
            const int i = base + j * RO_THREADS + tid;
            int hx = 0, hy = 0, klo = 0, khi = 0;
            bool maybe = i < nb;
            if (maybe) {
                // (comparisons with NaN are false: such a beam goes on to the cell index, which refuses it)
                maybe = !((owx < wxa && hwx[j] < wxa) || (owx > wxb && hwx[j] > wxb) || (owy < wya && hwy[j] < wya) || (owy > wyb && hwy[j] > wyb));
                const double dxw = hwx[j] - owx, dyw = hwy[j] - owy;
                if (maybe && fabs(dxw) + fabs(dyw) > 4.0 * g.res) {
                    const double c0 = dxw * (wya - owy) - dyw * (wxa - owx), c1 = dxw * (wya - owy) - dyw * (wxb - owx);
                    const double c2 = dxw * (wyb - owy) - dyw * (wxa - owx), c3 = dxw * (wyb - owy) - dyw * (wxb - owx);
                    maybe = !((c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0) || (c0 < 0 && c1 < 0 && c2 < 0 && c3 < 0));
                }
            }
            if (maybe && world_to_cell_fast(hwx[j], g.min_x, g.res, rinv, hx) && world_to_cell_fast(hwy[j], g.min_y, g.res, rinv, hy)) {
                if (hx >= x0 && hx < x1 && hy >= y0 && hy < y1) atomicAdd(&cnt[(hy - y0) * w + (hx - x0)], 0x10000u);   // mapping.py:124-129
                // Bresenham stays inside the rectangle of its end points ...
                bool cross = !(max(ox, hx) < x0 || min(ox, hx) >= x1 || max(oy, hy) < y0 || min(oy, hy) >= y1);
                if (cross) {
                    // ... and within one cell of the straight line: a rectangle (grown by a cell) whose corners all lie
                    // strictly on one side of the line cannot be touched
                    const long long dx = hx - ox, dy = hy - oy;
                    const long long cxa = x0 - 1 - ox, cxb = x1 - ox, cya = y0 - 1 - oy, cyb = y1 - oy;
                    const long long c0 = dx * cya - dy * cxa, c1 = dx * cya - dy * cxb, c2 = dx * cyb - dy * cxa, c3 = dx * cyb - dy * cxb;
                    cross = !((c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0) || (c0 < 0 && c1 < 0 && c2 < 0 && c3 < 0));
                }
                if (cross) {
                    Ray ray;
                    ray.init(ox, oy, hx, hy);
                    if (ray.xmajor) { ray.clip_major(x0, x1, klo, khi); ray.clip_minor(y0, y1, klo, khi); }
                    else { ray.clip_major(y0, y1, klo, khi); ray.clip_minor(x0, x1, klo, khi); }
                }
            }
            // The walk is shared out as (64 neighbouring beams, block of RO_STEPS steps) items: a far tile is crossed by two
            // or three waves' worth of beams for up to 64 steps each, and a step is a chain of dependent LDS operations
            // that one wave alone would queue up 64 deep; 16 waves take a block each.
            // blocks of RO_STEPS steps the longest of the wave's 64 segments needs (a segment inside a 64 x 64 rectangle has at
            // most 64 steps): four ballots, no trip through LDS
            const int len = khi - klo;
            int nblk = 0;
#pragma unroll
            for (int bk = 0; bk < RT_TILE / RO_STEPS; ++bk) nblk += __ballot(len > bk * RO_STEPS) != 0ull ? 1 : 0;
            seg[j * RO_THREADS + tid] = make_int4(hx, hy, klo, khi);
            if (lane == 0 && nblk > 0) {
                const int slot = atomicAdd(&n_items[par], nblk);
                for (int bk = 0; bk < nblk; ++bk) items[slot + bk] = (uint16_t)((j * (RO_THREADS / ICPMI_WAVE) + wave_id()) << 8 | bk);
            }
            // the beams of the next chunk (a scan of more than 2 048 beams)
            const int in = base + (PER_CHUNK + j) * RO_THREADS + tid;
            if (in < nb) { hwx[j] = hits[2 * (size_t)in]; hwy[j] = hits[2 * (size_t)in + 1]; }
        }
        __syncthreads();
#ifdef RO_X_TIMES
        const unsigned long long tb = __builtin_readcyclecounter();
        t_sel += tb - ta;
#endif
        const int total = n_items[par];
        if (tid == 0) n_items[par ^ 1] = 0;                          // nobody touches it between this barrier and the next
#ifndef RO_X_NOWALK
        for (int it = wave_id(); it < total; it += RO_THREADS / ICPMI_WAVE) {       // free cells, mapping.py:135-139
            const int e = items[it];
            const int4 sg = seg[(e >> 8) * ICPMI_WAVE + lane];
            const int k0 = sg.z + (e & 0xff) * RO_STEPS, k1 = min(sg.w, k0 + RO_STEPS);
            Ray r2;
            r2.init(ox, oy, sg.x, sg.y);
            if (k0 < k1) r2.seek(k0);
            // all the block's cells first, then all the neighbour exchanges, then the adds: the exchanges (LDS round trips)
            // are in flight together instead of one per step
            int cid[RO_STEPS], prev[RO_STEPS];
#pragma unroll
            for (int q = 0; q < RO_STEPS; ++q) {
                cid[q] = -1;
                if (k0 + q < k1) {
                    int x, y;
                    r2.cell(k0 + q, x, y);
                    cid[q] = (y - y0) * w + (x - x0);
                    r2.step();
                }
            }
            // runs of equal cells in adjacent lanes become one atomic each: around the origin the 64 neighbouring beams of a
            // wave sit in the same few cells (up to 64 lanes would queue on one LDS word), and further out, where neighbours
            // share a cell two or three at a time, it still wins (plain adds there: 15.8 against 13.4 us for the kernel)
#pragma unroll
            for (int q = 0; q < RO_STEPS; ++q) prev[q] = __shfl_up(cid[q], 1, ICPMI_WAVE);
#pragma unroll
            for (int q = 0; q < RO_STEPS; ++q) {
                const bool leader = lane == 0 || cid[q] != prev[q];
                const unsigned long long lead = __ballot(leader);
                if (leader && cid[q] >= 0) {
                    const unsigned long long above = lane == 63 ? 0ull : (lead >> (lane + 1));
                    const int run = above ? __ffsll((long long)above) : 64 - lane;
                    atomicAdd(&cnt[cid[q]], (uint32_t)run);
                }
            }
        }
#endif
        __syncthreads();                                             // seg and items are written again by the next chunk
#ifdef RO_X_TIMES
        t_walk += __builtin_readcyclecounter() - tb;
#endif
    }
    // this rectangle's cells: H hit adds, then M miss adds, then the clip (mapping.py:129,139,141)
#ifndef RO_X_NOFIN
#pragma unroll
    for (int j = 0; j < FIN; ++j) {
        const int c = j * RO_THREADS + tid;
        if (c >= w * h) break;
        const uint32_t v = cnt[c];
        if (v) f.log_odds[(size_t)(y0 + c / w) * g.nx + (x0 + c % w)] = apply_counts(old[j], v >> 16, v & 0xffffu, f.l_hit, f.l_miss, f.lo32, f.hi32, true);
    }
#endif
#ifdef RO_X_TIMES
    __syncthreads();
    if (tid == 0) {
        const unsigned long long te = __builtin_readcyclecounter();
        dbg[4 * blockIdx.x] = t_sel; dbg[4 * blockIdx.x + 1] = t_walk; dbg[4 * blockIdx.x + 2] = te - tt0; dbg[4 * blockIdx.x + 3] = 2 + (unsigned long long)(w * h);
    }
#endif
}

__global__ void world_to_grid_kernel(const double* __restrict__ w, long long n, double mn, double res, long long* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (long long)floor((w[i] - mn) / res);
}

// One thread per segment, walking in chunks exactly like ray_count_kernel.
__global__ void bresenham_cells_kernel(const int32_t* __restrict__ segs, const long long* __restrict__ cell_off, int n_seg,
                                       int32_t* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    Ray ray;
    ray.init(segs[4 * s], segs[4 * s + 1], segs[4 * s + 2], segs[4 * s + 3]);
    int32_t* o = out + 2 * cell_off[s];
    for (int c0 = 0; c0 < ray.n; c0 += RC_STEPS) {
        ray.seek(c0);
        for (int i = 0; i < RC_STEPS && c0 + i < ray.n; ++i) {
            int x, y;
            ray.cell(c0 + i, x, y);
            o[2 * (size_t)(c0 + i)] = x; o[2 * (size_t)(c0 + i) + 1] = y;
            ray.step();
        }
    }
}

}  // namespace icpmi

// Counter workspace: room for FOUR grids of uint32 counters (+ three bounding-box slots), whatever the group size.
// A scan's counters cover only the box the caller says its rays stay in (icpmi_grid_update_scans_box), so two sets
// (group parity) of up to RC_GROUP_MAX scans fit as long as the box is at most 1/8 of the grid — the usual case: a
// lidar's reach against a map with tens of metres of margin; with larger boxes the groups shrink (two scans per
// group for a box as large as the grid).
// + the cell boxes of the scans of a group, two sets (tile path).
extern "C" size_t icpmi_grid_workspace_bytes(int32_t ny, int32_t nx) {
    if (ny <= 0 || nx <= 0) return 0;
    return 4 * (size_t)ny * (size_t)nx * sizeof(uint32_t) + 256 + 2 * (size_t)icpmi::RC_GROUP_MAX * icpmi::RT_BOXES * sizeof(icpmi::ScanBox);
}

extern "C" int icpmi_world_to_grid(const double* w, int64_t n, double min_w, double resolution, int64_t* out, void* stream) {
    if (!w || !out || n < 0) return ICPMI_ERR_ARG;
    if (n == 0) return ICPMI_OK;
    icpmi::world_to_grid_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, (long long)n, min_w, resolution, (long long*)out);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

extern "C" int icpmi_bresenham_cells(const int32_t* segs, const int64_t* cell_off, int32_t n_seg, int32_t* out_cells, void* stream) {
    if (!segs || !cell_off || !out_cells || n_seg < 0) return ICPMI_ERR_ARG;
    if (n_seg == 0) return ICPMI_OK;
    icpmi::bresenham_cells_kernel<<<(n_seg + 63) / 64, 64, 0, (hipStream_t)stream>>>(segs, (const long long*)cell_off, n_seg, out_cells);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

extern "C" int icpmi_grid_update_scans(float* log_odds, void* counts_ws, int32_t ny, int32_t nx,
                                       double min_x, double min_y, double resolution,
                                       const double* origins, const double* hits, const int32_t* hit_off_host,
                                       int32_t n_scans, double l_hit, double l_miss, double lo, double hi,
                                       int64_t scan_seq, int32_t full_clip, void* stream) {
    return icpmi_grid_update_scans_band(log_odds, counts_ws, ny, nx, min_x, min_y, resolution, origins, hits, hit_off_host,
                                        n_scans, l_hit, l_miss, lo, hi, scan_seq, full_clip, 0, ny, stream);
}

extern "C" int icpmi_grid_update_scans_band(float* log_odds, void* counts_ws, int32_t ny, int32_t nx,
                                            double min_x, double min_y, double resolution,
                                            const double* origins, const double* hits, const int32_t* hit_off_host,
                                            int32_t n_scans, double l_hit, double l_miss, double lo, double hi,
                                            int64_t scan_seq, int32_t full_clip, int32_t row_begin, int32_t row_end,
                                            void* stream) {
    return icpmi_grid_update_scans_box(log_odds, counts_ws, ny, nx, min_x, min_y, resolution, origins, hits, hit_off_host, n_scans,
                                       l_hit, l_miss, lo, hi, scan_seq, full_clip, row_begin, row_end, nullptr, stream);
}

extern "C" int icpmi_grid_update_scans_box(float* log_odds, void* counts_ws, int32_t ny, int32_t nx,
                                           double min_x, double min_y, double resolution,
                                           const double* origins, const double* hits, const int32_t* hit_off_host,
                                           int32_t n_scans, double l_hit, double l_miss, double lo, double hi,
                                           int64_t scan_seq, int32_t full_clip, int32_t row_begin, int32_t row_end,
                                           const int32_t* box_host, void* stream) {
    using namespace icpmi;
    if (!log_odds || !counts_ws || !origins || !hit_off_host || ny <= 0 || nx <= 0 || n_scans < 0) return ICPMI_ERR_ARG;
    if (row_begin < 0 || row_end > ny || row_begin > row_end) return ICPMI_ERR_ARG;
    if (row_begin == row_end) return ICPMI_OK;                  // an empty band: nothing to write
    if (ny > RC_COORD_MAX || nx > RC_COORD_MAX || !(resolution > 0.0)) return ICPMI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // the window: grid (band) cut to the caller's box; one counter region per scan of a group covers it
    int wx0 = 0, wx1 = nx, wy0 = row_begin, wy1 = row_end;
    if (box_host) {
        wx0 = box_host[0] > wx0 ? box_host[0] : wx0; wy0 = box_host[1] > wy0 ? box_host[1] : wy0;
        wx1 = box_host[2] + 1 < wx1 ? box_host[2] + 1 : wx1; wy1 = box_host[3] + 1 < wy1 ? box_host[3] + 1 : wy1;
    }
    const bool empty_window = wx0 >= wx1 || wy0 >= wy1;             // no cell of the band can be touched (a full clip still runs)
    if (empty_window) { wx0 = 0; wx1 = 1; wy0 = row_begin; wy1 = row_begin + 1; }
    const size_t cells = (size_t)(wx1 - wx0) * (size_t)(wy1 - wy0);                 // counters per scan
    const size_t capacity = 4 * (size_t)ny * (size_t)nx;
    int group_max = (int)(capacity / (2 * cells));                                  // >= 2: the window is at most the grid
    group_max = group_max > RC_GROUP_MAX ? RC_GROUP_MAX : group_max;
    uint32_t* set2[2] = {(uint32_t*)counts_ws, (uint32_t*)counts_ws + (size_t)group_max * cells};
    BBox* slots = (BBox*)((unsigned char*)counts_ws + capacity * sizeof(uint32_t));
    // every call starts with empty boxes and ends with empty counter grids (each finalise pass zeroes what it
    // reads), so calls are independent of each other; scan_seq is no longer needed and ignored
    (void)scan_seq;
    int live_scans = 0;
    for (int t = 0; t < n_scans; ++t) live_scans += hit_off_host[t + 1] > hit_off_host[t] ? 1 : 0;
    // one scan with a box from the caller (the live update): the finalise pass walks that box, no slot to clear first
    const bool window_is_box = live_scans == 1 && box_host && !empty_window;
    if (!window_is_box && hipMemsetAsync(slots, 0, 3 * sizeof(BBox), st) != hipSuccess) return ICPMI_ERR_HIP;
    GridDesc g{nx, ny, min_x, min_y, resolution, wx0, wx1, wy0, wy1, wx0, wy0, wx1 - wx0, row_begin, row_end};
    if (empty_window) { g.wx1 = g.wx0; }                              // nothing is counted; the finalise pass only clips
    FinArgs fin{};
    fin.log_odds = log_odds; fin.l_hit = l_hit; fin.l_miss = l_miss; fin.lo32 = (float)lo; fin.hi32 = (float)hi;
    fin.grid_stride = cells; fin.n_grids = 1; fin.use_window = window_is_box ? 1 : 0;
    // The tile path needs the caller's box (the tiles of the window are enumerated; a scan's own box, found on the
    // device, sends the other tiles home at once).  Without a box (a direct caller of the plain entry points, or
    // non-finite coordinates) the window is the whole grid and the per-beam atomic pass runs instead.
    // ICPMI_RAYCAST=atomic forces the latter (experiments, and the parity tests of both passes).
    // A single scan (the live update) also takes the atomic pass: two short launches, 21 us against 32.
    const char* rc_env = option("RAYCAST");
    // ... and so does a replay whose box is far larger than a scan's reach (a long trajectory in one call): every scan
    // would start four workgroups per tile of the box just to find that it does not get there (~1 000 tiles: even).
    const long long window_tiles = (long long)((wx1 - wx0 + 63) / 64) * ((wy1 - wy0 + 63) / 64);
    const bool forced_tiles = rc_env && rc_env[0] == 't';
    const bool tiles_ok = box_host && !empty_window && !(rc_env && rc_env[0] == 'a') && hits &&
                          ((live_scans > 1 && window_tiles <= 768) || forced_tiles);
    // One live scan with a box from the caller: the owner pass, a single launch (RAYCAST = owner forces it for any single
    // scan with a box, atomic / tiles keep the passes above: the parity tests run all three).  It needs no clip of cells
    // the scan does not touch (full_clip) and 16-bit counts.
    if (live_scans == 1 && box_host && !empty_window && !full_clip && hits && (!rc_env || rc_env[0] == 'o')) {
        int s1 = 0;
        while (hit_off_host[s1 + 1] == hit_off_host[s1]) ++s1;
        const int nb1 = hit_off_host[s1 + 1] - hit_off_host[s1];
        const long long tx = (wx1 - wx0 + RT_TILE - 1) / RT_TILE, ty = (wy1 - wy0 + RT_TILE - 1) / RT_TILE;
        // (every workgroup looks at every beam: beyond a few hundred tiles — a box far larger than a lidar's reach — the two
        // passes with counters are the cheaper way)
        if (nb1 > 0 && nb1 <= 65535 && tx * ty <= 320) {
            ray_owner_kernel<<<RO_NEAR + (int)(tx * ty) * (RT_TILE / RO_FAR) * (RT_TILE / RO_FAR), RO_THREADS, 0, st>>>(g, origins + 2 * (size_t)s1, hits + 2 * (size_t)hit_off_host[s1], nb1,
                                                                               fin, (int)tx, (int)ty, (unsigned long long*)counts_ws);
            ICPMI_LAUNCH_CHECK();
            return ICPMI_OK;
        }
    }
    TileArgs ta{};
    ScanBox* box_sets = (ScanBox*)((unsigned char*)counts_ws + capacity * sizeof(uint32_t) + 256);
    ta.tiles_x = (wx1 - wx0 + RT_TILE - 1) / RT_TILE; ta.tiles_y = (wy1 - wy0 + RT_TILE - 1) / RT_TILE;
    const char* wg_env = option("RT_WGS");
    const int rt_wgs = wg_env && atoi(wg_env) > 0 ? atoi(wg_env) : 1536;      // resident workgroups of the tile pass (6 per CU)
    bool pending = false;            // a counted group whose finalisation rides on the next launch
    int64_t q = 0;                   // index of the next group (counter-grid set q&1, box slot q%3)
    int clip_all = full_clip;
    auto blocks_of = [](int nb) {
        const long waves = (long)((nb + ICPMI_WAVE - 1) / ICPMI_WAVE) * RC_SLOTS;
        return (int)((waves * ICPMI_WAVE + RC_THREADS - 1) / RC_THREADS);
    };
    // the group that starts at the first non-empty scan at or after `from`: 1 = built (end = one past its last scan),
    // 0 = none (the replay ends, or a scan too wide for a group comes first), -1 = bad offsets
    auto build_group = [&](int from, ScanGroup& gq, int& end) {
        gq = ScanGroup{};
        int t = from;
        while (t < n_scans && hit_off_host[t + 1] == hit_off_host[t]) ++t;
        end = t;
        if (t >= n_scans) return 0;
        while (t < n_scans && gq.n < group_max) {
            const int nb = hit_off_host[t + 1] - hit_off_host[t];
            if (nb < 0) return -1;
            if (nb > 65535) break;
            if (nb > 0) {
                gq.nb[gq.n] = nb; gq.origin_row[gq.n] = t; gq.hit_row[gq.n] = hit_off_host[t];
                gq.first_block[gq.n + 1] = gq.first_block[gq.n] + blocks_of(nb);
                ++gq.n;
            }
            ++t;
        }
        end = t;
        return gq.n > 0 ? 1 : 0;
    };
    ScanGroup grp{}, nxt{};
    bool have_nxt = false;
    int nxt_end = 0;
    int s = 0;
    while (s < n_scans) {
        const int nb0 = hit_off_host[s + 1] - hit_off_host[s];
        if (nb0 < 0) return ICPMI_ERR_ARG;
        if (nb0 == 0) { ++s; continue; }                        // mapping.py:113-114: silent no-op, no clip
        if (!hits) return ICPMI_ERR_ARG;
        uint32_t* counts = set2[q & 1];
        BBox* cur = slots + (q % 3);
        if (nb0 <= 65535) {
            // a group: the following scans too, while they fit a 16-bit counter (empty scans are skipped over)
            int t = s;
            bool boxes_ready = false;
            if (have_nxt && nxt.origin_row[0] == s) { grp = nxt; t = nxt_end; boxes_ready = true; }
            else if (build_group(s, grp, t) < 0) return ICPMI_ERR_ARG;
            have_nxt = false;
            const int blocks = grp.first_block[grp.n];
            if (tiles_ok) {
                ta.boxes = nullptr;
                if (live_scans > 1) {                               // several scans share the window: each one's own box
                    ScanBox* boxes = box_sets + (size_t)(q & 1) * RC_GROUP_MAX * RT_BOXES;
                    if (!boxes_ready) ray_scan_boxes_kernel<<<grp.n, RT_THREADS, 0, st>>>(g, origins, hits, grp, boxes);
                    ta.boxes = boxes;
                    const int r = build_group(t, nxt, nxt_end);     // the boxes of the group after this one ride on this launch
                    if (r < 0) return ICPMI_ERR_ARG;
                    have_nxt = r > 0;
                }
                const int n_items = grp.n * rt_blocks_per_scan(ta.tiles_x * ta.tiles_y);
                const int n_tiles = n_items < rt_wgs ? n_items : rt_wgs;
                const int n_fin = pending ? RC_FIN_BLOCKS : 0;
                if (!have_nxt) nxt.n = 0;
                ray_tile_step_kernel<<<n_tiles + n_fin + nxt.n, RT_THREADS, 0, st>>>(g, origins, hits, grp, counts, cells, cur, ta, n_items, n_tiles, n_fin,
                                                                                     fin, nxt, box_sets + (size_t)((q + 1) & 1) * RC_GROUP_MAX * RT_BOXES);
            } else if (pending) ray_step_kernel<<<blocks + RC_FIN_BLOCKS, RC_THREADS, 0, st>>>(g, origins, hits, grp, counts, cells, cur, blocks, fin);
            else ray_count_group_kernel<<<blocks, RC_THREADS, 0, st>>>(g, origins, hits, grp, counts, cells, cur);
            // this group's finalisation: reads its own grids and slot, frees the slot two groups ahead
            fin.counts = counts; fin.n_grids = grp.n; fin.bbox = cur; fin.other = slots + ((q + 2) % 3);
            fin.count_kind = 0; fin.clip = 1; fin.full_clip = clip_all;
            pending = true;
            s = t;
        } else {
            // more beams than a 16-bit counter holds: hits and misses in two rounds, no pipelining
            if (pending) { ray_finalize_kernel<<<RC_FIN_BLOCKS, RC_THREADS, 0, st>>>(g, fin); pending = false; }
            const double* h = hits + 2 * (size_t)hit_off_host[s];
            const double* o = origins + 2 * (size_t)s;
            const int blocks = blocks_of(nb0);
            FinArgs w = fin;
            w.counts = counts; w.n_grids = 1; w.bbox = cur; w.full_clip = 0;
            ray_count_kernel<<<blocks, RC_THREADS, 0, st>>>(g, o, h, nb0, counts, cur, RC_DO_HITS);
            w.other = nullptr; w.count_kind = 1; w.clip = 0;
            ray_finalize_kernel<<<RC_FIN_BLOCKS, RC_THREADS, 0, st>>>(g, w);
            ray_count_kernel<<<blocks, RC_THREADS, 0, st>>>(g, o, h, nb0, counts, cur, RC_DO_MISS);
            w.other = slots + ((q + 2) % 3); w.count_kind = 2; w.clip = 1; w.full_clip = clip_all;
            ray_finalize_kernel<<<RC_FIN_BLOCKS, RC_THREADS, 0, st>>>(g, w);
            ++s;
        }
        clip_all = 0;                                           // every cell is inside [lo, hi] after one clipped scan
        ++q;
        ICPMI_LAUNCH_CHECK();
    }
    if (pending) ray_finalize_kernel<<<RC_FIN_BLOCKS, RC_THREADS, 0, st>>>(g, fin);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
