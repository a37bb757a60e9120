/*
 * icpmi.h — C ABI of libicpmi.so: the MI355X (gfx950) implementation of the
 * per-scan hot path of DUBSON0/iterative-closest-point-avmi.
 *
 * The reference has no FFI layer: its boundary is the Python module surface of
 * utilities/icp.py and utilities/mapping.py.  Each entry point below names the
 * reference function (file:line under /root/reference) it replaces; the Python
 * modules in iterative-closest-point-avmi_amd/utilities/ bind these with ctypes
 * and keep the reference signatures.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (e.g. torch.Tensor.data_ptr()) unless the
 *    parameter name ends in _host.  No device memory is allocated, freed or
 *    retained by the library; scratch memory is passed in as a workspace.
 *  - Process-wide state, all of it listed under "library state" below: the
 *    ICPMI_* option switches (read from the environment once, at first use) and
 *    up to three side streams per device that launchers use to overlap the
 *    independent launches of one call (made on first use, destroyed by
 *    icpmi_shutdown).  Calls may come from several host threads.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *  - Return value: ICPMI_OK (0) or a negative ICPMI_ERR_* code; no exceptions
 *    cross the boundary.  Launch-time HIP errors are reported as ICPMI_ERR_HIP.
 *  - Point clouds are row-major float64 (n, dim), dim = 2 or 3, like the
 *    reference's NumPy arrays.  A *cloud set* is several clouds packed in one
 *    buffer: `off[c]` is the first row of cloud c (C+1 entries, capacity based),
 *    `cnt[c]` the number of valid rows (device side, so that voxel filtering,
 *    normals and ICP chain without a host round trip).  cnt == NULL means full
 *    capacity (off[c+1]-off[c]).
 */
#ifndef ICPMI_H
#define ICPMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPMI_OK 0
#define ICPMI_ERR_ARG (-1)        /* bad argument (dim, sizes, null pointer)      */
#define ICPMI_ERR_WORKSPACE (-2)  /* workspace too small                          */
#define ICPMI_ERR_HIP (-3)        /* HIP runtime error at launch                  */
#define ICPMI_ERR_UNSUPPORTED (-4)

/* ICP method, reference utilities/icp.py:133 `method=` */
#define ICPMI_POINT_TO_POINT 0
#define ICPMI_POINT_TO_LINE 1

/* per-pair termination status (slot ICPMI_RES_STATUS of a result record) */
#define ICPMI_ST_CONVERGED 1      /* icp.py:216-219                               */
#define ICPMI_ST_MAXITER 2        /* icp.py:222-223                               */
#define ICPMI_ST_FEW_INLIERS 3    /* icp.py:186-187 `break`                       */
#define ICPMI_ST_EMPTY 4          /* a cloud of the pair is empty after filtering */

/* One ICP result = 16 float64: R row-major in [0, dim*dim), t in [9, 9+dim),
 * then error, last |prev_error - error|, iterations executed, status.
 * All-double so that one RCCL all_gather moves a batch of results. */
#define ICPMI_RES_DOUBLES 16
#define ICPMI_RES_R 0
#define ICPMI_RES_T 9
#define ICPMI_RES_ERR 12
#define ICPMI_RES_DELTA 13
#define ICPMI_RES_ITERS 14
#define ICPMI_RES_STATUS 15

typedef struct icpmi_icp_params {
    double error_threshold;   /* icp.py:132 */
    double max_corr_dist;     /* icp.py:134; < 0 means None */
    int32_t max_iterations;   /* icp.py:132 */
    int32_t method;           /* ICPMI_POINT_TO_POINT / ICPMI_POINT_TO_LINE */
    int32_t has_init;         /* 1: R_init AND t_init given (icp.py:153) */
    int32_t dim;              /* 2 or 3 */
} icpmi_icp_params;

const char* icpmi_version(void);
const char* icpmi_strerror(int code);

/* ---- library state --------------------------------------------------------
 * Options are experiment / test switches (none is needed in production); each is
 * read from the environment variable ICPMI_<NAME> ONCE, when the library first
 * looks at an option, and icpmi_set_option overrides it afterwards (value NULL
 * unsets).  Names (with or without the ICPMI_ prefix): ICP2_SIDE (0: no side
 * streams), ICP2_SHAPE ("TxS": workgroup shape of the fused ICP), ICP2_FILTER
 * (0: no float32 filter), ICP2_STAGES (1: one launch, 2: two stages also for
 * point_to_point), ICP2_FAR (mean squared error in m^2 of a pair's first step
 * above which it finishes on the kernel for far queries; default 1, 0: never),
 * POLAR (0 never / 2 always the bearing order), PREP_KNN
 * (grid | sweep), RAYCAST (atomic | tiles | owner: the pass of the occupancy
 * update), RT_WGS (resident workgroups of the tile pass), RS_BATCH (full: the
 * batched rotation search scores every angle; projection: no bearing order).
 * Unknown names: ICPMI_ERR_ARG.  Must not race with running calls.
 * icpmi_shutdown synchronises and destroys the side streams and events the
 * library made (they are made again on demand); call it before unloading. */
int icpmi_set_option(const char* name, const char* value);
int icpmi_shutdown(void);
/* One HIP runtime per process: the streams and device pointers a caller hands over must belong to the runtime the
 * library's launches go through.  A process that maps TWO copies of libamdhip64 (this library's /opt/rocm one, loaded
 * first, and the copy PyTorch bundles, loaded later — or the other way round without the loader finding the first)
 * fails every launch with ICPMI_ERR_HIP and nothing to tell why.  icpmi_runtime_check walks the loaded objects
 * (dl_iterate_phdr): ICPMI_OK when at most one libamdhip64 is mapped, else ICPMI_ERR_HIP with the paths written to
 * msg (NUL-terminated, at most msg_bytes; msg may be NULL).  The reference has no counterpart (pure Python); the
 * Python host layer (icpmi/_lib.py) calls it right after loading the library and whenever a call returns ICPMI_ERR_HIP. */
int icpmi_runtime_check(char* msg, size_t msg_bytes);

/* ---- voxel_downsample, utilities/icp.py:117-129 --------------------------
 * For every cloud c: keys floor((p - min_c) / voxel) per axis, lexicographic
 * unique, per-voxel mean with the sum taken in input order.  out_pts uses the
 * same offsets as the input; out_cnt[c] = number of voxels (or -1 if the key
 * range overflows 64 bits).  off_host mirrors off_dev (the launcher routes
 * clouds larger than 8192 points to the multi-kernel path). */
size_t icpmi_voxel_workspace_bytes(int32_t max_n);
int icpmi_voxel_downsample_batch(const double* pts, const int32_t* off_dev, const int32_t* off_host,
                                 int32_t n_clouds, int32_t dim, double voxel_size,
                                 double* out_pts, int32_t* out_cnt,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- nearest neighbour, icp.py:35-46,173,179 (scipy KDTree.query, k=1) ----
 * Pair b matches every valid row of cloud pair_src[b] against cloud
 * pair_tgt[b] (exhaustive, LDS-tiled, float64 direct differences, lowest index
 * on ties).  out_idx/out_dist are [n_pairs][out_stride]; dist is the Euclidean
 * distance (sqrt), index is relative to the target cloud. */
int icpmi_nn_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                   const int32_t* pair_src, const int32_t* pair_tgt, int32_t n_pairs,
                   int32_t max_src_n, int32_t dim,
                   int32_t* out_idx, double* out_dist, int32_t out_stride, void* stream);

/* ---- estimate_normals_2d, icp.py:51-76 ------------------------------------
 * k+1 nearest neighbours (self included, k clamped to n-1), 2x2 covariance,
 * eigenvector of the smaller eigenvalue, unit length.  Sign is arbitrary, as
 * in the reference (it cancels in the point-to-line solve).  k <= 31 here (the
 * exhaustive companion of icpmi_nn_batch); icpmi_prepare_targets takes any k.
 * One workgroup per selected cloud: cloud_ids[n_sel] lists the clouds to
 * process (NULL = clouds 0..n_sel-1).  max_n bounds the rows of any selected
 * cloud; total_rows = off[C].  The workspace is only needed when max_n > 8192
 * (the x-sorted index then lives in global memory instead of LDS). */
size_t icpmi_normals_workspace_bytes(int32_t total_rows, int32_t max_n);
int icpmi_normals_2d_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                           const int32_t* cloud_ids, int32_t n_sel, int32_t total_rows,
                           int32_t max_n, int32_t k, double* out_normals,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- _point_to_line_solve_2d, icp.py:79-115 --------------------------------
 * One linearised point-to-line step over n_src correspondences. out_Rt = 6
 * doubles: R row-major (4) then t (2). Exactly singular system -> identity. */
int icpmi_p2l_solve_2d(const double* src, int32_t n_src, const double* tgt, const double* normals,
                       const int32_t* nn_idx, double* out_Rt, void* stream);

/* ---- ICP, icp.py:132-223 (after its two voxel_downsample calls) ------------
 * Runs n_pairs independent registrations to completion on the device: NN
 * search, correspondence rejection, point-to-line / point-to-point solve,
 * apply, error, convergence test; one workgroup per pair, no host round trip.
 * normals: same row layout as pts (only rows of target clouds are read; may be
 * NULL for point_to_point).  init: [n_pairs][dim*dim + dim] (R then t) or NULL.
 * results: [n_pairs][ICPMI_RES_DOUBLES].  max_src_n bounds the valid rows of
 * any source cloud, max_tgt_n those of any target cloud, total_rows = off[C].
 *
 * prepared (optional): buffer filled by icpmi_prepare_targets for the target
 * clouds of this batch.  With it, 2-D pairs whose source has at most 4096 rows
 * run on the fast kernel (exact sweep search on the axis-sorted target copy —
 * staged in LDS up to 4096 target rows, read in place through L2 above; pair
 * state in registers; `normals` is then not read, and `workspace` is optional:
 * given, a point_to_line batch of >= 1024 pairs runs in two stages — every pair
 * up to 12 iterations, then the pairs still running, parked there with their
 * moving rows and matches, together in a second launch — and a pair of any
 * batch whose first step leaves a mean squared error above option ICP2_FAR (it
 * starts metres from its target) is parked likewise and finished by the kernel
 * for far queries; both give the same results bit for bit, sooner).  Without
 * `prepared`, or for 3-D / larger
 * sources, the exhaustive LDS-tiled kernel runs and needs `workspace` (and
 * `normals` for point_to_line).  Results agree. */
size_t icpmi_icp_workspace_bytes(int32_t n_pairs, int32_t max_src_n, int32_t dim);
int icpmi_icp_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                    const double* normals, const void* prepared,
                    const int32_t* pair_src, const int32_t* pair_tgt,
                    int32_t n_pairs, int32_t max_src_n, int32_t max_tgt_n, int32_t total_rows,
                    const icpmi_icp_params* params_host, const double* init, double* results,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- prepared targets: axis choice + sort (+ normals) for the sweep search ----
 * For every selected cloud (<= 4096 rows): pick the projection axis (x, y, x+y
 * or x-y) with the smallest expected search window, sort the cloud along it and
 * store the sorted points, the sorted->row map and the axis in `prepared`
 * (icpmi_prepared_bytes(total_rows, n_clouds) bytes).  normal_k >= 0 also
 * computes estimate_normals_2d (icp.py:51-76) with k = normal_k (any k, as the reference:
 * up to 31 from register lists, beyond that a wave per query): stored in sorted order inside
 * `prepared`, and in row order in out_normals if given.
 * normal_k < 0 skips normals (point_to_point).
 *
 * Clouds above 4096 rows (a rolling submap) are prepared through global memory
 * (rocPRIM sort) and later searched in place through L2; for them the launcher
 * needs the sizes on the host: off_host mirrors off_dev and cloud_ids_host
 * mirrors cloud_ids (both may be NULL when max_n <= 4096).  max_n = rows of the
 * largest selected cloud; it also sizes the sort scratch inside `prepared`. */
size_t icpmi_prepared_bytes(int32_t total_rows, int32_t n_clouds, int32_t max_n);
int icpmi_prepare_targets(const double* pts, const int32_t* off_dev, const int32_t* off_host,
                          const int32_t* cnt_dev, const int32_t* cloud_ids, const int32_t* cloud_ids_host,
                          int32_t n_sel, int32_t n_clouds, int32_t total_rows, int32_t max_n,
                          int32_t normal_k, double* out_normals, void* prepared, size_t prepared_bytes,
                          void* stream);

/* The same, with the sort order left to the library: allow_polar != 0 lets clouds of at most 2048 rows be sorted
 * by BEARING about the frame origin instead of along a projection, when the library estimates smaller search
 * windows for it (a lidar scan in its sensor frame has one return per bearing, so the points within a distance B
 * of a query lie in a wedge of a few points, while a projection slab holds every wall that crosses it).  The
 * order only changes how fast icpmi_icp_batch searches — results are the exact nearest neighbours either way.
 * Buffers prepared with allow_polar may only be passed to icpmi_icp_batch (icpmi_nn_prepared_batch walks
 * projections).  Replaces no reference function: the KDTree construction of icp.py:173 is the closest analogue. */
int icpmi_prepare_targets_ex(const double* pts, const int32_t* off_dev, const int32_t* off_host,
                             const int32_t* cnt_dev, const int32_t* cloud_ids, const int32_t* cloud_ids_host,
                             int32_t n_sel, int32_t n_clouds, int32_t total_rows, int32_t max_n,
                             int32_t normal_k, double* out_normals, void* prepared, size_t prepared_bytes,
                             int32_t allow_polar, void* stream);

/* Nearest neighbour on prepared targets (same contract as icpmi_nn_batch, same
 * answers bit for bit, icp.py:179): binary search + outward sweep on the sorted
 * copy instead of the exhaustive scan.  Target clouds of the pairs must have
 * been prepared (<= 4096 rows).  out_second_sq (optional) receives the squared
 * distance of the second nearest target point. */
int icpmi_nn_prepared_batch(const double* pts, const int32_t* off_dev, const int32_t* cnt_dev,
                            const void* prepared, const int32_t* pair_src, const int32_t* pair_tgt,
                            int32_t n_pairs, int32_t max_src_n, int32_t max_tgt_n, int32_t total_rows,
                            int32_t* out_idx, double* out_dist, double* out_second_sq,
                            int32_t out_stride, void* stream);

/* ---- rotation search scoring, utilities/features.py:213-232 (and slam.py:138-159) ----
 * For every angle a (given as cos, sin pairs — computed by the caller with the
 * reference's own NumPy calls): mean over the rows of src_c of the squared
 * nearest-neighbour distance of (src_c @ R(a).T + shift) in tgt, R = [[c,-s],[s,c]].
 * One launch for a whole sweep.  src_c: (n_src, 2), tgt: (n_tgt, 2), row-major. */
int icpmi_rotation_scores(const double* src_c, int32_t n_src, const double* tgt, int32_t n_tgt,
                          const double* cos_sin, int32_t n_angles, double shift_x, double shift_y,
                          double* out_scores, void* stream);

/* The whole correlative search on the device — utilities/features.py:198-232 (voxel filter of both clouds, their
 * means, the coarse sweep, its arg-min, the fine sweep around the winner, its arg-min) and the sweeps of
 * slam.py:146-159 — one chain of launches, nothing returns to the host in between.
 * pts: n_src source rows followed by n_tgt target rows (raw clouds, (n, 2) float64).  coarse_cs: cos, sin of the
 * n_coarse coarse angles.  fine_cs / fine_cnt: for EVERY coarse angle k the fine grid that follows when k wins
 * (fine_cnt[k] <= max_fine angles, cos, sin at fine_cs[(k * max_fine + j) * 2]) — the grids and their cos / sin are
 * the caller's (the reference's own NumPy expressions), so the chosen angle is the reference's bit for bit.
 * centred != 0: the source is centred on its mean and shifted to the target's mean (features.py:205-216);
 * else the rows are rotated as they are and shifted by (shift_x, shift_y) (slam.py:138-143).
 * out_record (device, 12 doubles): voxel counts of source and target, mean of the source (2), shift (2), winning
 * coarse index, its score, length of its fine grid, winning fine index, its score, reserved.  np.argmin semantics
 * (first minimum; first NaN if any).  The voxel-filtered clouds stay in the workspace (256 bytes in: source rows,
 * then target rows, counts as int32 at byte 16) for a following icpmi_nn_batch-style refinement. */
size_t icpmi_rotation_search_workspace_bytes(int32_t n_src, int32_t n_tgt, int32_t n_coarse, int32_t max_fine);
int icpmi_rotation_search(const double* pts, int32_t n_src, int32_t n_tgt, double voxel_size,
                          const double* coarse_cs, int32_t n_coarse,
                          const double* fine_cs, const int32_t* fine_cnt, int32_t max_fine,
                          int32_t centred, double shift_x, double shift_y,
                          double* out_record, void* workspace, size_t workspace_bytes, void* stream);

/* Translation refinement of the submap variant, slam.py:161-181, on the state icpmi_rotation_search left behind
 * (search_workspace: the same buffer, untouched since; record: its out_record; the same angle tables): the filtered
 * source rotated by the winning angle (NumPy's own (n, 2) @ (2, 2) arithmetic) and placed at (pred_x, pred_y), its
 * nearest target rows, np.percentile(d^2, 80) with linear interpolation, and the mean of (matched - rotated) over the
 * rows at or below it — out4 (device): refined t (2; the predicted position when fewer than 5 rows qualify, slam.py:
 * 175-181), the inlier count, the percentile.  n_src / n_tgt: the RAW row counts passed to the search; n_src <= 2048
 * (ICPMI_ERR_UNSUPPORTED above: refine on the host from the filtered clouds). */
size_t icpmi_rotation_refine_workspace_bytes(int32_t n_src);
int icpmi_rotation_refine(const void* search_workspace, int32_t n_src, int32_t n_tgt, const double* record,
                          const double* coarse_cs, const double* fine_cs, int32_t max_fine,
                          double pred_x, double pred_y, double* out4, void* scratch, size_t scratch_bytes, void* stream);

/* ---- batched pre-alignment: rotation_search (utilities/features.py:165-242) for every pair of a batch — the first
 * half of _run_icp_pair (slam.py:53-98), which slam.py:575-579 calls once per loop-closure candidate ----------------
 * pts / off_dev / off_host: a cloud set of RAW 2-D clouds (at most 4096 rows each); pair b searches the rotation of
 * cloud pair_src[b] onto cloud pair_tgt[b].  tgt_ids (device, optional): the distinct target clouds (only they are
 * put in search order; NULL = every cloud).  One chain of launches: voxel filter of every cloud at voxel_size, the
 * means of the filtered clouds, the search order of the targets, then ONE workgroup per pair for both sweeps.  Angle
 * grids as in icpmi_rotation_search (coarse_cs; fine_cs / fine_cnt / max_fine: one fine grid per coarse winner; the
 * caller's NumPy cos / sin; at most 1024 coarse angles and 1024 per fine grid).
 * Only the arg-min of the coarse scores matters to the reference (features.py:223), so a workgroup bounds every
 * coarse score from below with a distance field of its target (one look-up per row), scores angles exactly — same
 * float64 nearest-neighbour distances as icpmi_rotation_search — in order of that bound, and stops at the first angle
 * whose bound exceeds the best exact score: np.argmin over the scored angles (first minimum) is np.argmin over all.
 * out_records [n_pairs][16]: slots 0..10 as icpmi_rotation_search's record (filtered counts, mean of the source (2),
 * mean of the target (2), winning coarse index, its score, length of its fine grid, winning fine index, its score);
 * 11 status — 0 searched; 1 a filtered cloud has fewer than 5 points (features.py:203-204: identity, zeros, inf);
 * 2 a filtered cloud exceeds the on-chip capacity (not searched: use icpmi_rotation_search); 3 the winner's fine grid
 * is empty (np.argmin raises in the reference); 12, 13 coarse / fine angles scored exactly (diagnostic).
 * out_init (optional) [n_pairs][6]: R row-major then t = mu_t - R mu_s (features.py:235-237) — the `init` argument of
 * icpmi_icp_batch, so pre-alignment and ICP chain on one stream with no host round trip; identity for status != 0.
 * max_rows_hint: an upper bound the caller expects for the FILTERED clouds (0: none).  The on-chip copies are sized
 * by min(largest raw cloud, 2048, hint); up to 1024 rows two workgroups share a CU.  A pair beyond it gets status 2. */
size_t icpmi_rotation_search_batch_workspace_bytes(int32_t total_rows, int32_t n_clouds, int32_t max_n);
int icpmi_rotation_search_batch(const double* pts, const int32_t* off_dev, const int32_t* off_host, int32_t n_clouds,
                                const int32_t* tgt_ids, int32_t n_tgt_ids,
                                const int32_t* pair_src, const int32_t* pair_tgt, int32_t n_pairs,
                                double voxel_size, const double* coarse_cs, int32_t n_coarse,
                                const double* fine_cs, const int32_t* fine_cnt, int32_t max_fine,
                                int32_t max_rows_hint, double* out_records, double* out_init,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---- OccupancyGrid2D, utilities/mapping.py ---------------------------------
 * world -> cell index, mapping.py:57-60,94-98: floor((w - min) / res), float64
 * IEEE division, result as int64. */
int icpmi_world_to_grid(const double* w, int64_t n, double min_w, double resolution,
                        int64_t* out, void* stream);

/* Cells of _bresenham(x0,y0,x1,y1), mapping.py:68-89 (start included, end
 * excluded), for n_seg segments {x0,y0,x1,y1} (int32).  Segment s writes its
 * max(|dx|,|dy|) cells (x,y int32 pairs) at out_cells[2*cell_off[s] ...]. This
 * is the device function the ray-cast kernel walks; exposed for parity tests
 * and for the `_bresenham` method of the drop-in class. */
int icpmi_bresenham_cells(const int32_t* segs, const int64_t* cell_off, int32_t n_seg,
                          int32_t* out_cells, void* stream);

/* update_scan, mapping.py:103-141, for n_scans consecutive scans (n_scans = 1
 * is the reference call; more is the _rebuild_map replay of slam.py:271-277).
 * One scan with a cell box from the caller (icpmi_grid_update_scans_box) and no
 * full_clip is ONE launch: workgroups own rectangles of the box, count in LDS and
 * apply the replay and the clip to their own cells; the counter workspace is not
 * touched.
 * log_odds: float32 (ny, nx), updated in place.  Scan s has origin
 * origins[2s..2s+1] and hits rows [hit_off_host[s], hit_off_host[s+1]) of
 * hits (world frame, float64).  Per cell the result equals the reference's
 * sequence: H adds of l_hit, then M adds of l_miss, each rounded to float32
 * from a float64 sum, then one clip to [lo, hi] per scan.
 * counts: workspace of icpmi_grid_workspace_bytes(ny, nx) bytes — room for four grids of uint32 counters plus
 * bounding-box slots and the cell boxes of the scans of two groups — zeroed once by the caller before first use
 * and owned by this grid afterwards.  With
 * n_scans > 1, consecutive scans are counted in ONE launch, each into its own counter region, and finalised
 * together in scan order per cell (so the result is the sequential one bit for bit); the count pass of a group
 * shares its launch with the finalise pass of the previous group (the other set of regions).  A region covers the
 * box the rays stay in (icpmi_grid_update_scans_box; the whole grid otherwise), so a group holds up to 32 scans
 * when the box is at most 1/16 of the grid and at least 2 always: a replay of S scans is about S/32 + 1 launches.
 * Every call leaves the workspace all zero again.
 * scan_seq: ignored (kept for binary compatibility; calls are independent).
 * full_clip != 0 clips every cell of the grid on the first scan (needed only
 * when cells may lie outside [lo, hi] beforehand). */
size_t icpmi_grid_workspace_bytes(int32_t ny, int32_t nx);
int icpmi_grid_update_scans(float* log_odds, void* counts, int32_t ny, int32_t nx,
                            double min_x, double min_y, double resolution,
                            const double* origins, const double* hits, const int32_t* hit_off_host,
                            int32_t n_scans, double l_hit, double l_miss, double lo, double hi,
                            int64_t scan_seq, int32_t full_clip, void* stream);

/* The same update restricted to the rows [row_begin, row_end) of the grid: one
 * rank's band of a sharded map replay (the replay of slam.py:271-277 over a
 * long history).  Cells are independent and each keeps its scan
 * order, so replaying every scan on every rank with disjoint bands and then
 * gathering the bands gives the same grid bit for bit.  log_odds and counts
 * are FULL-size (ny x nx) buffers; only rows of the band are read or written,
 * rays are clipped to the band analytically (no steps are walked outside it),
 * and full_clip clips the band only. */
int icpmi_grid_update_scans_band(float* log_odds, void* counts, int32_t ny, int32_t nx,
                                 double min_x, double min_y, double resolution,
                                 const double* origins, const double* hits, const int32_t* hit_off_host,
                                 int32_t n_scans, double l_hit, double l_miss, double lo, double hi,
                                 int64_t scan_seq, int32_t full_clip, int32_t row_begin, int32_t row_end,
                                 void* stream);

/* The same with a promise about where the rays lie: box_host = {x0, y0, x1, y1}, inclusive CELL bounds (host
 * memory) that contain the origin cell and every hit cell of every scan of the call (Bresenham stays inside the
 * rectangle spanned by its end points; cells outside the grid need not be covered), or NULL for "anywhere".
 * Counters are then kept for that box only, which is what lets 32 scans be counted per launch inside a workspace
 * of four grids (the Python class computes the box from the scans it is given), and a replay of several scans is
 * counted tile by tile in LDS instead of with one device atomic per beam and cell (same counts, ~1.4x the rate).
 * A ray cell outside the box is not counted: the box is a contract, not a clip the reference has. */
int icpmi_grid_update_scans_box(float* log_odds, void* counts, int32_t ny, int32_t nx,
                                double min_x, double min_y, double resolution,
                                const double* origins, const double* hits, const int32_t* hit_off_host,
                                int32_t n_scans, double l_hit, double l_miss, double lo, double hi,
                                int64_t scan_seq, int32_t full_clip, int32_t row_begin, int32_t row_end,
                                const int32_t* box_host, void* stream);

/* ── pose graph: PoseGraph2D.optimize, utilities/pose_graph.py:83-134 ──────────
 * Gauss-Newton on SE(2) over n_nodes poses [x, y, theta] (nodes: device, updated
 * in place) and n_edges constraints (i, j, z_ij [3], Omega [3][3] row-major).
 * edges_ij_host: int32 pairs on the HOST (the library classifies the edges and
 * builds its gather lists from them); edges_z / edges_omega: device.
 * All iterations run in one launch.  Graphs whose consecutive nodes are all
 * joined by an edge (the odometry chain slam.py:543-549 builds) are solved as
 * block-tridiagonal + low-rank; any other graph through the dense 3n x 3n
 * matrix, as the reference does.  fix_node is held by the 1e10 diagonal of
 * pose_graph.py:107-112.
 * info (device, 10 doubles): iterations run, status (0 nothing to do: fewer
 * than two nodes or no edges; 1 converged: step norm < convergence_eps; 2
 * iteration limit; 3 singular system: the nodes keep the values of the previous
 * iteration, pose_graph.py:117-119), norm of the last step; then the time spent
 * per phase in microseconds, summed over the iterations (assembly, chain
 * factorisation, chain sweeps, closure system, its solve, update of dx, apply).
 * workspace: icpmi_pose_graph_workspace_bytes(edges_ij_host, n_nodes, n_edges). */
size_t icpmi_pose_graph_workspace_bytes(const int32_t* edges_ij_host, int32_t n_nodes, int32_t n_edges);
int icpmi_pose_graph_optimize(double* nodes, const int32_t* edges_ij_host, const double* edges_z,
                              const double* edges_omega, int32_t n_nodes, int32_t n_edges,
                              int32_t n_iterations, int32_t fix_node, double convergence_eps,
                              double* info, void* workspace, size_t workspace_bytes, void* stream);

/* total_error, pose_graph.py:189-194: sum over edges of e^T Omega e, in edge
 * order.  Everything on the device; scratch: n_edges doubles; out: 1 double. */
int icpmi_pose_graph_error(const double* nodes, const int32_t* edges_i, const int32_t* edges_j,
                           const double* edges_z, const double* edges_omega, int32_t n_edges,
                           double* scratch, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICPMI_H */
