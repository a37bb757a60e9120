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
        const long long k = (num + den - 1) / den;
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
        const bool a = world_to_cell(hits[2 * (size_t)beam], g.min_x, g.res, hx);
        const bool b = world_to_cell(hits[2 * (size_t)beam + 1], g.min_y, g.res, hy);
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
#define ICPMI_RC_GROUP 16
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
extern "C" size_t icpmi_grid_workspace_bytes(int32_t ny, int32_t nx) {
    if (ny <= 0 || nx <= 0) return 0;
    return 4 * (size_t)ny * (size_t)nx * sizeof(uint32_t) + 256;
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
    if (hipMemsetAsync(slots, 0, 3 * sizeof(BBox), st) != hipSuccess) return ICPMI_ERR_HIP;
    GridDesc g{nx, ny, min_x, min_y, resolution, wx0, wx1, wy0, wy1, wx0, wy0, wx1 - wx0, row_begin, row_end};
    if (empty_window) { g.wx1 = g.wx0; }                              // nothing is counted; the finalise pass only clips
    FinArgs fin{};
    fin.log_odds = log_odds; fin.l_hit = l_hit; fin.l_miss = l_miss; fin.lo32 = (float)lo; fin.hi32 = (float)hi;
    fin.grid_stride = cells; fin.n_grids = 1;
    bool pending = false;            // a counted group whose finalisation rides on the next launch
    int64_t q = 0;                   // index of the next group (counter-grid set q&1, box slot q%3)
    int clip_all = full_clip;
    auto blocks_of = [](int nb) {
        const long waves = (long)((nb + ICPMI_WAVE - 1) / ICPMI_WAVE) * RC_SLOTS;
        return (int)((waves * ICPMI_WAVE + RC_THREADS - 1) / RC_THREADS);
    };
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
            ScanGroup grp{};
            int t = s;
            while (t < n_scans && grp.n < group_max) {
                const int nb = hit_off_host[t + 1] - hit_off_host[t];
                if (nb < 0) return ICPMI_ERR_ARG;
                if (nb > 65535) break;
                if (nb > 0) {
                    grp.nb[grp.n] = nb; grp.origin_row[grp.n] = t; grp.hit_row[grp.n] = hit_off_host[t];
                    grp.first_block[grp.n + 1] = grp.first_block[grp.n] + blocks_of(nb);
                    ++grp.n;
                }
                ++t;
            }
            const int blocks = grp.first_block[grp.n];
            if (pending) ray_step_kernel<<<blocks + RC_FIN_BLOCKS, RC_THREADS, 0, st>>>(g, origins, hits, grp, counts, cells, cur, blocks, fin);
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
