// prep.hip — prepare target clouds for the sweep search, and K4 normals on top.
//
// Per target cloud (one workgroup): pick the sort axis, sort the points along
// it in LDS, publish the sorted copy (points, original rows, axis) for the
// fused ICP kernel, and — for point_to_line — estimate_normals_2d (reference
// utilities/icp.py:51-76) by an outward sweep from every point's own sorted
// position, which is an exact k-NN search (see sweep.hpp).
#include "prep_common.hpp"

namespace icpmi {

#ifndef PREP_POLAR32
#define PREP_POLAR32 1          // bearing order by a 32-bit (fixed-point float32 bearing | row) sort; 0: float64 atan2, 64-bit network
#endif
#ifndef PREP_AXIS_STEP
#define PREP_AXIS_STEP 4        // large batches estimate the search axis on every PREP_AXIS_STEP-th point
#endif
#ifndef ICPMI_PREP_WPS
#define ICPMI_PREP_WPS 6      // waves per SIMD the k-NN kernel is compiled for up to KK = 13: three workgroups per CU
#endif
constexpr int PREP_THREADS = 512;
constexpr int PREP_MAXW = PREP_THREADS / ICPMI_WAVE;
constexpr int PREP_MAX_POINTS = 4096;   // sorted copy (20 B/pt) + sort scratch (12 B/pt) stay in LDS

// KK = capacity of the per-query neighbour list (0: no normals); GRID: k-NN through a grid instead of the sweep
// Sort of up to E * PREP_THREADS rows by (key, row) on registers; leaves keys[i] / rows[i] = full sortable key and row of
// the i-th smallest, like bitonic_sort_pairs.  by_row, keys: E * PREP_THREADS slots of LDS each; rows: as many.
// Bearing order (round 4): the key is the float32 bearing in 21 bits of fixed point (steps of 2 pi / 2^21 = 3.0e-6 rad)
// with the row below it — ONE 32-bit word per point, the network of the voxel filter instead of the 64-bit one, atan2f
// instead of the float64 atan2 (2.02 -> see DESIGN section 6).  Any order of the points is a correct order for the exact
// searches; what they need is (a) images that do not decrease along the sorted array — the image IS the quantised key —
// and (b) every image within the searches' key slack of the true bearing: the floor 3.0e-6 + the float32 product in
// front of it (up to an eighth of a step at 2^21) + atan2f and its float32 inputs ~1e-6 + the image's own two roundings
// 7e-7 < 5.3e-6 (SweepFQuery::mu = 1e-5 with the query's own atan2f and the subtraction; the normals' walk compares two
// images: 1.6e-5).  Equal keys keep row order; no fix-up.
constexpr float PREP_POLAR_Q = 6.283185307179586f / 2097152.0f;        // 2 pi / 2^21
__device__ __forceinline__ uint32_t prep_polar_key21(double x, double y) {
    const float t = (atan2f((float)y, (float)x) + 3.14159265f) * (2097152.0f / 6.283185307179586f);
    const int k = (int)t;
    return (uint32_t)(k < 0 ? 0 : (k > 2097151 ? 2097151 : k));        // NaN -> 0: any key is a correct key
}
__device__ __forceinline__ float prep_polar_image(uint32_t k21) { return (float)k21 * PREP_POLAR_Q - 3.14159265f; }

template <int E>
__device__ __forceinline__ void prep_sort_polar32(const double* __restrict__ P, int M, uint64_t* keys, uint32_t* rows) {
    uint32_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * PREP_THREADS + (int)threadIdx.x;            // coalesced; any start order sorts the same
        v[e] = i < M ? (prep_polar_key21(P[2 * i], P[2 * i + 1]) << 11) | (uint32_t)i : 0xffffffffu;
    }
    bitonic_sort_regs_fixed<uint32_t, E, PREP_THREADS>(v, reinterpret_cast<uint32_t*>(keys));
    __syncthreads();                                                   // the network's scratch is the key array written next
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (int)threadIdx.x * E + e;
        rows[i] = v[e] == 0xffffffffu ? 0xffffffffu : (v[e] & 0x7ffu);
        keys[i] = v[e] == 0xffffffffu ? ~0ull : f64_sortable((double)prep_polar_image(v[e] >> 11));   // the image is the key from here on
    }
    __syncthreads();
}

template <int E>
__device__ __forceinline__ void prep_sort_regs(const double* __restrict__ P, int M, int dir, uint64_t* by_row, uint64_t* keys,
                                               uint32_t* rows) {
    if (PREP_POLAR32 && dir == SWEEP_POLAR) { prep_sort_polar32<E>(P, M, keys, rows); return; }
    constexpr uint64_t ROW_MASK = 0x7ff;                               // rows below 2 048
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (int)threadIdx.x * E + e;
        const uint64_t k = i < M ? f64_sortable(dir == SWEEP_POLAR ? polar_key(P[2 * i], P[2 * i + 1]) : proj(dir, P[2 * i], P[2 * i + 1])) : ~0ull;
        by_row[i] = k;
        v[e] = i < M ? ((k & ~ROW_MASK) | (uint64_t)i) : ~0ull;
    }
    bitonic_sort_regs_fixed<uint64_t, E, PREP_THREADS>(v, keys);       // its LDS stages hold barriers: by_row is complete too
    int bad = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (int)threadIdx.x * E + e;
        const uint32_t r = v[e] == ~0ull ? 0xffffffffu : (uint32_t)(v[e] & ROW_MASK);
        rows[i] = r;
        keys[i] = r == 0xffffffffu ? ~0ull : by_row[r];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (int)threadIdx.x * E + e;
        if (i + 1 < M && (keys[i] > keys[i + 1] || (keys[i] == keys[i + 1] && rows[i] > rows[i + 1]))) bad = 1;
    }
    if (__syncthreads_or(bad)) {
        // keys that agree in their upper 53 bits came out in row order: one thread puts such runs in (key, row) order
        if (threadIdx.x == 0)
            for (int i = 1; i < M; ++i) {
                const uint64_t k = keys[i];
                const uint32_t r = rows[i];
                int j = i - 1;
                while (j >= 0 && (keys[j] > k || (keys[j] == k && rows[j] > r))) { keys[j + 1] = keys[j]; rows[j + 1] = rows[j]; --j; }
                keys[j + 1] = k; rows[j + 1] = r;
            }
        __syncthreads();
    }
}

template <int KK, bool GRID>
__global__ __launch_bounds__(PREP_THREADS, (KK <= 13 ? ICPMI_PREP_WPS : (KK <= 16 ? 4 : 2))) void prep_targets_kernel(   // three workgroups per CU up to KK = 13, two up to 16
    const double* __restrict__ pts, const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
    const int32_t* __restrict__ cloud_ids, int k, double2* __restrict__ g_sxy, double2* __restrict__ g_snrm,
    int32_t* __restrict__ g_sorig, float* __restrict__ g_skey, int32_t* __restrict__ g_dir, double* __restrict__ out_normals,
    int lds_points, int split, int polar) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ double dsc[8 * PREP_MAXW];
    __shared__ int hist[6 * PREP_BINS];
    // `split` workgroups share one cloud: each repeats the (cheap) axis choice and sort, then takes its
    // slice of the normal queries — small batches would otherwise leave a cloud's k-NN sweeps to one CU
    const int ci = blockIdx.x / split, part = blockIdx.x % split;
    const int c = cloud_ids ? cloud_ids[ci] : ci;
    const int M = cnt ? cnt[c] : off[c + 1] - off[c];
    if (off[c + 1] - off[c] > PREP_MAX_POINTS) return;          // prepared through global memory (prep_big.hip)
    if (M <= 0 || M > lds_points) { if (threadIdx.x == 0) g_dir[c] = -1; return; }
    const double* P = pts + (size_t)off[c] * 2;
    int npad = 64;
    while (npad < M) npad <<= 1;
    double2* sxy = reinterpret_cast<double2*>(dyn);                                   // lds_points * 16 B
    int32_t* sorig = reinterpret_cast<int32_t*>(dyn + (size_t)lds_points * 16);       // lds_points * 4 B
    float* sth = reinterpret_cast<float*>(dyn + (size_t)lds_points * 20);             // lds_points * 4 B: float32 bearings (polar order)
    // sort scratch (20 B per padded slot, 12 for 4 096 slots): behind the sorted copy when the grid needs it afterwards,
    // otherwise ON the sorted copy (the sorted rows pass through registers) — 40 KB instead of 64 KB for ~1 500 points, so
    // that the k-NN loops of three workgroups instead of two share a CU
    const size_t scratch_at = GRID ? (size_t)lds_points * 24 : 0;
    const bool reg_sort = npad <= 4 * PREP_THREADS;                                    // the network on registers (below)
    const int nsort = reg_sort ? max(npad, PREP_THREADS) : npad;
    uint64_t* by_row = reinterpret_cast<uint64_t*>(dyn + scratch_at);                  // reg_sort: keys by row, nsort * 8 B
    uint64_t* keys = reinterpret_cast<uint64_t*>(dyn + scratch_at + (reg_sort ? (size_t)nsort * 8 : 0));     // sorted keys, 8 B per slot
    uint32_t* rows = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(keys) + (size_t)nsort * 8); // their rows

    double bounds[4];
    const int dir = choose_axis<PREP_THREADS>(P, M, dsc, hist, bounds, polar, (!GRID && M >= 512) ? PREP_AXIS_STEP : 1);   // bounds: the grid's
    // ── sort along the chosen axis (or by bearing) ───────────────────────────
    if (reg_sort) {
        // (key, row) as ONE 64-bit element: the key's low 11 bits give way to the row, the network runs on registers
        // (sort.hpp: thread t owns slots t E .. t E + E - 1), and the rare neighbours whose keys agree in the 53 bits
        // that are left are put right afterwards from the full keys — the order is exactly the pair sort's.
        if (nsort == PREP_THREADS) prep_sort_regs<1>(P, M, dir, by_row, keys, rows);
        else if (nsort == 2 * PREP_THREADS) prep_sort_regs<2>(P, M, dir, by_row, keys, rows);
        else prep_sort_regs<4>(P, M, dir, by_row, keys, rows);
    } else {
        for (int i = threadIdx.x; i < npad; i += PREP_THREADS) {
            keys[i] = i < M ? f64_sortable(dir == SWEEP_POLAR ? polar_key(P[2 * i], P[2 * i + 1]) : proj(dir, P[2 * i], P[2 * i + 1])) : ~0ull;
            rows[i] = i < M ? (uint32_t)i : 0xffffffffu;
        }
        __syncthreads();
        bitonic_sort_pairs(keys, rows, npad);
    }
    double2* o_sxy = g_sxy + off[c];
    int32_t* o_sorig = g_sorig + off[c];
    float* o_skey = g_skey + off[c];
    int my_row[PREP_MAX_POINTS / PREP_THREADS];
    float my_key[PREP_MAX_POINTS / PREP_THREADS];          // float32 image of the sort key (the bearing: one atan2 per point in all)
#pragma unroll
    for (int u = 0; u < PREP_MAX_POINTS / PREP_THREADS; ++u) {
        const int i = u * PREP_THREADS + (int)threadIdx.x;
        my_row[u] = i < M ? (int)rows[i] : 0;
        my_key[u] = i < M ? (float)f64_unsortable(keys[i]) : 0.0f;
    }
    __syncthreads();                                             // the scratch may be the memory written next
#pragma unroll
    for (int u = 0; u < PREP_MAX_POINTS / PREP_THREADS; ++u) {
        const int i = u * PREP_THREADS + (int)threadIdx.x;
        if (i < M) {
            const int row = my_row[u];
            const double2 p = make_double2(P[2 * row], P[2 * row + 1]);
            sxy[i] = p; sorig[i] = row;
            if (KK > 0 && !GRID && dir == SWEEP_POLAR) sth[i] = my_key[u];
            if (part == 0) { o_sxy[i] = p; o_sorig[i] = row; o_skey[i] = my_key[u]; }
        }
    }
    if (threadIdx.x == 0 && part == 0) g_dir[c] = dir;
    __syncthreads();
    if constexpr (KK > 0) {
        const int kc = min(k, M - 1);              // icp.py:61
        const int kk = kc + 1;                     // self included, icp.py:66
        double2* o_snrm = g_snrm + off[c];
        double* o_rows = out_normals ? out_normals + (size_t)off[c] * 2 : nullptr;
        const int per = (M + split - 1) / split;
        if constexpr (GRID) {
            // Few clouds (a single scan pair): the time of the launch is the longest search of any lane, and a
            // grid bounds that far better than a 1-D window (a wall across the sweep axis puts hundreds of
            // points in it).  Built in the LDS the sort has released: cell ends (4 B per cell), then positions
            // by cell.  Same neighbours, same order.
            const int cells_cap = min(4096, 2 * npad);
            uint32_t* cell_end = reinterpret_cast<uint32_t*>(dyn + scratch_at);
            uint16_t* cell_pts = reinterpret_cast<uint16_t*>(cell_end + cells_cap);
            const PrepGrid grid = prep_grid_build(sxy, M, bounds, kk, cell_end, cell_pts, cells_cap, hist);
            prep_normals_grid<KK>(sxy, sorig, M, min(M, part * per), min(M, (part + 1) * per), kk, grid, o_snrm, o_rows);
        } else {
            // Many clouds: every CU is busy, the sum of the work counts, and the sweep has less of it per candidate.
            if (dir == SWEEP_POLAR) prep_normals_polar<KK>(sxy, sorig, sth, M, min(M, part * per), min(M, (part + 1) * per), kk, o_snrm, o_rows);
            else prep_normals<KK>(sxy, sorig, M, min(M, part * per), min(M, (part + 1) * per), dir, kk, o_snrm, o_rows);
        }
    }
}


// estimate_normals_2d for ANY k (the reference accepts every k, icp.py:61; the register lists above stop at 31):
// one wave per query on the sorted copy in global memory.  The k + 1 nearest are drawn one after the other — each
// round every lane scans its share of the cloud for the smallest (distance, row) beyond the last one drawn, and a
// wave reduction picks the overall next — so the neighbours come out in the (distance, row) order every other path
// sums them in: same lists, same arithmetic, same normals.  O(k M / 64) per query: a fallback, not a fast path.
constexpr int ANYK_THREADS = 256;
// first sorted position whose key is >= v (wave-uniform)
__device__ __forceinline__ int anyk_lower_bound(const double2* sxy, int M, int dir, double v) {
    int lo = 0, hi = M;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double2 c = sxy[mid];
        if (proj(dir, c.x, c.y) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(ANYK_THREADS) void normals_anyk_kernel(
    const int32_t* __restrict__ off, const int32_t* __restrict__ cnt, const int32_t* __restrict__ cloud_ids, int k,
    const double2* __restrict__ g_sxy, double2* __restrict__ g_snrm, const int32_t* __restrict__ g_sorig,
    const int32_t* __restrict__ g_dir, double* __restrict__ out_normals, int sel_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    int32_t* sel = reinterpret_cast<int32_t*>(dyn) + (size_t)wave_id() * sel_cap;
    const int c = cloud_ids ? cloud_ids[blockIdx.y] : blockIdx.y;
    const int M = cnt ? min(cnt[c], off[c + 1] - off[c]) : off[c + 1] - off[c];
    const int dir = g_dir[c];
    if (M <= 0 || dir < 0) return;
    const double2* sxy = g_sxy + off[c];
    const int32_t* sorig = g_sorig + off[c];
    double2* o_snrm = g_snrm + off[c];
    double* o_rows = out_normals ? out_normals + (size_t)off[c] * 2 : nullptr;
    const int kk = min(min(k, M - 1) + 1, sel_cap);             // icp.py:61,66: k clamped, self included
    const int lane = lane_id();
    const int waves = gridDim.x * (ANYK_THREADS / ICPMI_WAVE);
    for (int s = blockIdx.x * (ANYK_THREADS / ICPMI_WAVE) + wave_id(); s < M; s += waves) {     // wave-uniform
        const double2 q = sxy[s];
        // Only a window of the sort order can hold the k nearest: the 2 kk + 1 points around the query's own place
        // contain at least kk points, so the kk-th nearest is no farther than the farthest of THEM (R), and every
        // candidate has its key within kappa R of the query's (|du| <= d along x or y, <= sqrt(2) d on the diagonals).
        // A rolling submap of 10 000 points then scans a few hundred per query, not all of them.  (Bearing order: no
        // such bound — the whole cloud, at most 2 048 points.)
        int lo = 0, hi = M;
        if (dir < SWEEP_POLAR) {
            const int a0 = max(0, s - kk), a1 = min(M - 1, s + kk);
            double r2 = 0.0;
            for (int i = a0 + lane; i <= a1; i += ICPMI_WAVE) r2 = fmax(r2, sweep_d2(q.x, q.y, sxy[i]));
            r2 = wave_max(r2);
            if (a1 - a0 + 1 >= kk) {
                const double uq = proj(dir, q.x, q.y);
                const double w = sqrt(r2) * 1.4142135623730951 * 1.000001 + 1e-9 * (1.0 + fabs(uq));
                lo = anyk_lower_bound(sxy, M, dir, uq - w);
                hi = anyk_lower_bound(sxy, M, dir, uq + w + 1e-9 * (1.0 + fabs(uq)));
                hi = min(M, hi + 1);
            }
        }
        double last_d = -1.0;
        int last_row = -1;
        for (int j = 0; j < kk; ++j) {
            double bd = __builtin_inf();
            int brow = 0x7fffffff, bpos = -1;
            for (int i = lo + lane; i < hi; i += ICPMI_WAVE) {
                const double d2 = sweep_d2(q.x, q.y, sxy[i]);
                const int row = sorig[i];
                const bool after = d2 > last_d || (d2 == last_d && row > last_row);
                if (after && (d2 < bd || (d2 == bd && row < brow))) { bd = d2; brow = row; bpos = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double od = __shfl_xor(bd, o, ICPMI_WAVE);
                const int orow = __shfl_xor(brow, o, ICPMI_WAVE), opos = __shfl_xor(bpos, o, ICPMI_WAVE);
                if (od < bd || (od == bd && orow < brow)) { bd = od; brow = orow; bpos = opos; }
            }
            last_d = bd; last_row = brow;
            if (lane == 0) sel[j] = bpos;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) {
            // np.cov over the neighbours in (distance, row) order, eigenvector of the smaller eigenvalue: emit_normal's arithmetic
            double mx = 0.0, my = 0.0;
            for (int j = 0; j < kk; ++j) { const double2 p = sxy[sel[j]]; mx += p.x; my += p.y; }
            mx /= (double)kk; my /= (double)kk;
            double sxx = 0.0, sxy_ = 0.0, syy = 0.0;
            for (int j = 0; j < kk; ++j) {
                const double2 p = sxy[sel[j]];
                const double dx = p.x - mx, dy = p.y - my;
                sxx += dx * dx; sxy_ += dx * dy; syy += dy * dy;
            }
            double vx = 1.0, vy = 0.0;
            if (kk > 1) {
                const double den = (double)(kk - 1);
                smallest_evec_2x2(sxx / den, sxy_ / den, syy / den, vx, vy);
            }
            double nn = sqrt(vx * vx + vy * vy);
            nn = nn < 1e-10 ? 1e-10 : nn;
            const double2 n2 = make_double2(vx / nn, vy / nn);
            o_snrm[s] = n2;
            if (o_rows) { const int row = sorig[s]; o_rows[2 * row] = n2.x; o_rows[2 * row + 1] = n2.y; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace icpmi

namespace icpmi {
size_t prep_big_scratch_bytes(int n);                                                                  // prep_big.hip
int prep_big_cloud(const double* P, const int32_t* cnt_c, int n_cap, int normal_k, double2* o_sxy, double2* o_snrm,
                   int32_t* o_sorig, int32_t* dir_c, double* o_rows, void* scratch, size_t scratch_bytes, hipStream_t st);
}

// layout of a prepared-target buffer: sorted xy | sorted normals | sorted->row map | float32 bearings (bearing order) |
// axis per cloud | scratch for
// the sort of clouds above 4096 rows (max_n = rows of the largest cloud that will be prepared)
static size_t prepared_core_bytes(int32_t total_rows, int32_t n_clouds) {
    return (((size_t)total_rows * (16 + 16 + 4 + 4) + (size_t)n_clouds * 4) + 255) / 256 * 256 + 256;
}

extern "C" size_t icpmi_prepared_bytes(int32_t total_rows, int32_t n_clouds, int32_t max_n) {
    if (total_rows < 0 || n_clouds < 0 || max_n < 0) return 0;
    return prepared_core_bytes(total_rows, n_clouds) + (max_n > icpmi::PREP_MAX_POINTS ? icpmi::prep_big_scratch_bytes(max_n) : 0);
}

extern "C" int icpmi_prepare_targets(const double* pts, const int32_t* off_dev, const int32_t* off_host,
                                     const int32_t* cnt_dev, const int32_t* cloud_ids, const int32_t* cloud_ids_host,
                                     int32_t n_sel, int32_t n_clouds, int32_t total_rows, int32_t max_n,
                                     int32_t normal_k, double* out_normals, void* prepared, size_t prepared_bytes,
                                     void* stream) {
    return icpmi_prepare_targets_ex(pts, off_dev, off_host, cnt_dev, cloud_ids, cloud_ids_host, n_sel, n_clouds, total_rows, max_n,
                                    normal_k, out_normals, prepared, prepared_bytes, 0, stream);
}

extern "C" int icpmi_prepare_targets_ex(const double* pts, const int32_t* off_dev, const int32_t* off_host,
                                        const int32_t* cnt_dev, const int32_t* cloud_ids, const int32_t* cloud_ids_host,
                                        int32_t n_sel, int32_t n_clouds, int32_t total_rows, int32_t max_n,
                                        int32_t normal_k, double* out_normals, void* prepared, size_t prepared_bytes,
                                        int32_t allow_polar, void* stream) {
    using namespace icpmi;
    if (!pts || !off_dev || !prepared || n_sel < 0 || n_clouds < 0 || total_rows < 0 || max_n < 0) return ICPMI_ERR_ARG;
    if (normal_k > 31 && (size_t)(normal_k + 1) * 4 * (ANYK_THREADS / ICPMI_WAVE) > 96 * 1024) return ICPMI_ERR_UNSUPPORTED;   // k beyond 6 143
    if (prepared_bytes < icpmi_prepared_bytes(total_rows, n_clouds, max_n)) return ICPMI_ERR_WORKSPACE;
    if (max_n > PREP_MAX_POINTS && !off_host) return ICPMI_ERR_ARG;       // sizes are needed on the host to route big clouds
    if (cloud_ids && max_n > PREP_MAX_POINTS && !cloud_ids_host) return ICPMI_ERR_ARG;
    if (n_sel == 0 || max_n == 0) return ICPMI_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* b = (unsigned char*)prepared;
    double2* g_sxy = (double2*)b;
    double2* g_snrm = (double2*)(b + (size_t)total_rows * 16);
    int32_t* g_sorig = (int32_t*)(b + (size_t)total_rows * 32);
    float* g_skey = (float*)(b + (size_t)total_rows * 36);
    int32_t* g_dir = (int32_t*)(b + (size_t)total_rows * 40);
    // clouds above the LDS capacity: one by one through global memory (prep_big.hip)
    int small_max = max_n;
    if (max_n > PREP_MAX_POINTS) {
        void* scratch = b + prepared_core_bytes(total_rows, n_clouds);
        const size_t scratch_bytes = prepared_bytes - prepared_core_bytes(total_rows, n_clouds);
        small_max = 0;
        for (int i = 0; i < n_sel; ++i) {
            const int c = cloud_ids ? cloud_ids_host[i] : i;
            if (c < 0 || c >= n_clouds) return ICPMI_ERR_ARG;
            const int n = off_host[c + 1] - off_host[c];
            if (n <= PREP_MAX_POINTS) { small_max = n > small_max ? n : small_max; continue; }
            const size_t o = (size_t)off_host[c];
            const int rc = prep_big_cloud(pts + o * 2, cnt_dev ? cnt_dev + c : nullptr, n, normal_k > 31 ? -1 : normal_k, g_sxy + o, g_snrm + o,
                                          g_sorig + o, g_dir + c, out_normals ? out_normals + o * 2 : nullptr, scratch,
                                          scratch_bytes, st);
            if (rc != ICPMI_OK) return rc;
        }
        if (small_max == 0 && normal_k <= 31) return ICPMI_OK;
    }
    int npad = 64;
    while (npad < small_max) npad <<= 1;
    // sorted copy: 20 B per point, sized to the largest cloud (rounded to 64); sort scratch: 12 B per padded slot.
    // With ~1 500-point clouds this is 53 KB instead of 64 KB: three workgroups per CU instead of two.
    const int lds_points = (small_max + 63) / 64 * 64;
    const size_t sort_bytes = npad <= 4 * PREP_THREADS ? (size_t)(npad > PREP_THREADS ? npad : PREP_THREADS) * 20 : (size_t)npad * 12;
    const size_t lds_sep = (size_t)lds_points * 24 + sort_bytes;                            // grid instantiation
    const size_t lds_alias = (size_t)lds_points * 24 > sort_bytes ? (size_t)lds_points * 24 : sort_bytes;
    // bearing order (sweep.hpp, SWEEP_POLAR) only where every consumer understands it: on request, and only when EVERY
    // selected cloud has at most 2 048 rows — the fused ICP launch picks ONE instantiation for the whole batch from the
    // same max_n, and the ones for larger targets (no float32 images) cannot walk a bearing order: a batch that mixes
    // scans with one rolling submap therefore sorts everything along projections.  Option "POLAR": 0 never, 2 always
    // (tests); the filter being off (option "ICP2_FILTER" = 0) also turns it off.
    int polar = allow_polar && max_n <= 2048 ? 1 : 0;
    if (const char* env = option("POLAR")) polar = polar ? (env[0] == '0' ? 0 : (env[0] == '2' ? 2 : 1)) : 0;
    if (const char* env = option("ICP2_FILTER")) polar = env[0] == '0' ? 0 : polar;
    int split = 256 / n_sel;                 // a workgroup for every CU when the batch is small
    split = split < 1 ? 1 : (split > 16 ? 16 : split);
    // k-NN search of the normals: grid for few clouds that will be sorted along a projection (a wall across the sweep axis
    // puts hundreds of points into a window, and a small launch lasts as long as its longest search), sweep otherwise —
    // in bearing order a scan's windows are short everywhere (round 4, one 2 048-beam scan, k = 12: sweep 0.037 ms, grid
    // 0.076; 64 scans: 0.050 / 0.115).  Where the bearing order is allowed the device may still pick a projection for a
    // cloud that is no scan: the sweep is then slower, never wrong.  Option PREP_KNN = grid | sweep forces one of the two
    // (tests run both on the same inputs)
    int use_grid = split > 1 && !polar;
    if (const char* env = option("PREP_KNN")) use_grid = env[0] == 'g' ? 1 : (env[0] == 's' ? 0 : use_grid);
#define ICPMI_PREP_GO2(KKV, G)                                                                                          \
    do {                                                                                                                \
        const size_t lds = G ? lds_sep : lds_alias;                                                                     \
        if (dyn_lds((const void*)prep_targets_kernel<KKV, G>,                                                           \
                                (int)lds) != hipSuccess) return ICPMI_ERR_HIP;                                          \
        prep_targets_kernel<KKV, G><<<n_sel * (KKV > 0 ? split : 1), PREP_THREADS, lds, st>>>(                          \
            pts, off_dev, cnt_dev, cloud_ids, normal_k, g_sxy, g_snrm, g_sorig, g_skey, g_dir, out_normals, lds_points, \
            KKV > 0 ? split : 1, polar);                                                                                \
    } while (0)
#define ICPMI_PREP_GO(KKV)                                                                                              \
    do { if (use_grid && KKV > 0) ICPMI_PREP_GO2(KKV, true); else ICPMI_PREP_GO2(KKV, false); } while (0)
    // list capacity = k + 1 exactly for the usual k (5, 10 = reference default, 12 = config.yaml), else the next size up;
    // beyond 31 neighbours: sort only, then the any-k kernel on the sorted copy
    if (normal_k > 31) {
        if (small_max > 0) { ICPMI_PREP_GO2(0, false); }
        ICPMI_LAUNCH_CHECK();
        // the list of a query: k + 1 positions, at most the largest cloud
        const int kcap = normal_k + 1 < max_n ? normal_k + 1 : max_n;
        const int sel_cap = (kcap + 63) / 64 * 64;
        const size_t lds_k = (size_t)sel_cap * 4 * (ANYK_THREADS / ICPMI_WAVE);
        if (dyn_lds((const void*)normals_anyk_kernel, lds_k) != hipSuccess) return ICPMI_ERR_HIP;
        int per_cloud = 2048 / n_sel;
        per_cloud = per_cloud < 1 ? 1 : (per_cloud > 256 ? 256 : per_cloud);
        normals_anyk_kernel<<<dim3(per_cloud, n_sel), ANYK_THREADS, lds_k, st>>>(off_dev, cnt_dev, cloud_ids, normal_k, g_sxy, g_snrm, g_sorig,
                                                                                  g_dir, out_normals, sel_cap);
        ICPMI_LAUNCH_CHECK();
        return ICPMI_OK;
    }
    if (normal_k < 0) ICPMI_PREP_GO(0);
    else if (normal_k + 1 <= 6) ICPMI_PREP_GO(6);
    else if (normal_k + 1 <= 8) ICPMI_PREP_GO(8);
    else if (normal_k + 1 <= 11) ICPMI_PREP_GO(11);
    else if (normal_k + 1 <= 13) ICPMI_PREP_GO(13);
    else if (normal_k + 1 <= 16) ICPMI_PREP_GO(16);
    else ICPMI_PREP_GO(32);
#undef ICPMI_PREP_GO2
#undef ICPMI_PREP_GO
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
