"""GPU parity tests: the HIP path (through the C ABI of libicpmi.so) against the
CPU oracle and the golden vectors captured from the reference.

Bars: bit-exact for voxel keys/means, NN indices and distances, Bresenham cell
indices and log-odds cells; <= 1e-5 Frobenius on ICP transforms (BASELINE
north_star) — the tests hold 1e-9, with iteration counts equal.
"""
import numpy as np
import pytest

import oracle
from conftest import load_golden, rot_err
from test_oracle_golden import ICP_CASES, icp_case_args, _grid_shape

pytestmark = pytest.mark.gpu

FRO_TOL = 1e-9        # the contract is 1e-5


@pytest.fixture(scope="module")
def uicp():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from utilities import icp
    icp.VERBOSE = False
    return icp


@pytest.fixture(scope="module")
def umap(uicp):
    from utilities import mapping
    return mapping


# ── voxel_downsample ────────────────────────────────────────────────────────
def test_voxel_golden_bit_exact(uicp):
    z = load_golden("voxel")
    for name in z["names"]:
        out = uicp.voxel_downsample(z[f"{name}__in"], float(z[f"{name}__voxel"]))
        assert out.shape == z[f"{name}__out"].shape, name
        assert np.array_equal(out, z[f"{name}__out"]), name


def test_voxel_submap_radix_path(uicp):
    """81 920 points -> the rocPRIM path (slam.py:103-108 _build_submap)."""
    from icpmi import synth
    z = load_golden("submap_build")
    segs = synth.maze_segments()
    poses = synth.trajectory(40)
    allpts = np.vstack([synth.to_world(synth.scan(p, 500 + i, segs=segs), p) for i, p in enumerate(poses)])
    assert np.array_equal(uicp.voxel_downsample(allpts, 0.04), z["out"])
    # around the 8192-point switch between the two paths
    for n in (8191, 8192, 8193, 20000):
        assert np.array_equal(uicp.voxel_downsample(allpts[:n], 0.1), oracle.voxel_downsample(allpts[:n], 0.1)), n


def test_voxel_batch_and_edges(uicp):
    from icpmi import batch
    rng = np.random.default_rng(0)
    clouds = [rng.normal(size=(n, 2)) * 3 for n in (1, 2, 63, 64, 65, 1000, 2048, 4097, 8192)]
    clouds.append(np.zeros((50, 2)))                      # every point identical
    clouds.append(np.empty((0, 2)))                       # empty cloud in a set: count 0
    cs = batch.CloudSet.from_numpy(clouds)
    out = batch.voxel_downsample_set(cs, 0.25).to_numpy()
    for c, o in zip(clouds, out):
        assert np.array_equal(o, oracle.voxel_downsample(c, 0.25)), len(c)
    with pytest.raises(ValueError):
        uicp.voxel_downsample(np.empty((0, 2)), 0.1)     # icp.py:119 raises on empty input
    t3 = rng.normal(size=(3000, 3))
    assert np.array_equal(uicp.voxel_downsample(t3, 0.2), oracle.voxel_downsample(t3, 0.2))


# ── nearest neighbour ───────────────────────────────────────────────────────
@pytest.mark.parametrize("case", ["vox", "raw", "submap"])
def test_nn_golden_exact(uicp, case):
    z = load_golden("nn")
    d, i = uicp.nearest_neighbors(z[f"{case}__src"], z[f"{case}__tgt"])
    assert np.array_equal(i, z[f"{case}__idx"])
    assert np.array_equal(d, z[f"{case}__dist"])


def test_nn_helpers_and_ragged_batch(uicp):
    from icpmi import batch
    z = load_golden("nn")
    assert np.array_equal(uicp.find_nearest_neighbors(z["vox__src"], z["vox__tgt"]), z["helper_nearest"])
    assert np.array_equal(uicp.find_nearest_neighbor_indices(z["vox__src"], z["vox__tgt"]), z["helper_idx"])
    assert np.array_equal(uicp.center_of_mass(z["vox__src"]), z["helper_com"])
    rng = np.random.default_rng(1)
    sizes = [(1, 1), (5, 3), (300, 2049), (1025, 17), (2048, 2048), (700, 5000)]
    clouds, ps, pt = [], [], []
    for n, m in sizes:
        ps.append(len(clouds)); clouds.append(rng.uniform(-5, 5, size=(n, 2)))
        pt.append(len(clouds)); clouds.append(rng.uniform(-5, 5, size=(m, 2)))
    cs = batch.CloudSet.from_numpy(clouds)
    dist, idx = batch.nn_set(cs, ps, pt)
    dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
    for b, (n, m) in enumerate(sizes):
        do, io = oracle.nn(clouds[ps[b]], clouds[pt[b]])
        assert np.array_equal(idx[b, :n], io) and np.array_equal(dist[b, :n], do), (n, m)
    # exact ties: lowest index wins, 3-D works too
    tgt = np.array([[1.0, 0.0], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0], [1.0, 0.0]])
    d, i = uicp.nearest_neighbors(np.zeros((3, 2)), tgt)
    assert np.array_equal(i, [0, 0, 0]) and np.array_equal(d, [1.0, 1.0, 1.0])
    s3, t3 = rng.normal(size=(500, 3)), rng.normal(size=(1500, 3))
    d, i = uicp.nearest_neighbors(s3, t3)
    do, io = oracle.nn(s3, t3)
    assert np.array_equal(i, io) and np.array_equal(d, do)


# ── normals ─────────────────────────────────────────────────────────────────
@pytest.mark.parametrize("case", ["tgt_k12", "tgt_k5", "five_k10", "collinear_k8", "diag_k6"])
def test_normals_golden(uicp, case):
    z = load_golden("normals")
    pts, k = z[f"{case}__in"], int(z[f"{case}__k"])
    n = uicp.estimate_normals_2d(pts, k)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    dots = np.abs(np.sum(n * z[f"{case}__out"], axis=1))
    assert np.quantile(dots, 0.01) > 1 - 1e-9 and dots.min() > 1 - 1e-6
    # against the oracle the neighbour sets and the summation order are identical
    no = oracle.normals_2d(pts, k)
    assert np.abs(np.abs(np.sum(n * no, axis=1)) - 1).max() < 1e-12


def test_normals_sizes(uicp):
    rng = np.random.default_rng(2)
    for m, k in ((1, 10), (2, 10), (3, 1), (64, 7), (65, 15), (513, 31), (3000, 12)):
        pts = rng.uniform(-4, 4, size=(m, 2))
        n, no = uicp.estimate_normals_2d(pts, k), oracle.normals_2d(pts, k)
        assert np.abs(np.abs(np.sum(n * no, axis=1)) - 1).max() < 1e-9, (m, k)


def test_normals_grid_and_sweep_searches_agree(uicp, libopt):
    """prep.hip has two exact k-NN searches (grid for few clouds, sweep for many): same neighbours in the same order,
    so the normals are bit-identical; ICPMI_PREP_KNN forces either on the same inputs."""
    from icpmi import synth
    rng = np.random.default_rng(8)
    a, _ = synth.config2_pair(3)
    line = np.column_stack([np.linspace(-2, 2, 300), np.full(300, 0.5)])               # collinear: a one-row grid
    dup = np.repeat(rng.uniform(-1, 1, size=(40, 2)), 5, axis=0)                         # exact duplicates: ties on the row
    lattice = np.stack(np.meshgrid(np.arange(30) * 0.1, np.arange(30) * 0.1), -1).reshape(-1, 2)   # many equal distances
    clouds = [uicp.voxel_downsample(a, 0.04), rng.uniform(-4, 4, size=(3000, 2)), line, dup, lattice,
              rng.normal(size=(5, 2)), np.zeros((7, 2))]
    for pts in clouds:
        for k in (12, 5, 31):
            got = {}
            for mode in ("grid", "sweep"):
                libopt.setenv("ICPMI_PREP_KNN", mode)
                got[mode] = uicp.estimate_normals_2d(pts, k)
            assert np.array_equal(got["grid"], got["sweep"]), (len(pts), k)
            no = oracle.normals_2d(pts, k)
            ok = np.abs(np.abs(np.sum(got["grid"] * no, axis=1)) - 1) < 1e-9
            # real scan, random cloud, a straight wall: (nearly) every normal is well defined and must be the oracle's
            # up to sign; isotropic neighbourhoods (lattice interior, identical points) have no defined direction
            if pts is clouds[0] or pts is clouds[1] or pts is line:
                assert ok.mean() >= 0.99, (len(pts), k, ok.mean())
            else:
                assert ok.mean() > 0.5 or len(pts) <= 7 or pts is lattice, (len(pts), k, ok.mean())
    libopt.delenv("ICPMI_PREP_KNN")


def test_normals_and_icp_with_more_than_31_neighbours(uicp):
    """The reference accepts any k (k = min(k, n - 1), icp.py:61); the register lists of the fast k-NN stop at 31, beyond
    that a wave per query draws the neighbours one by one (was an 'unsupported' error)."""
    from icpmi import synth
    a, b = synth.config2_pair(5)
    pts = uicp.voxel_downsample(b, 0.04)
    for k in (32, 40, 100, 5000):
        got = uicp.estimate_normals_2d(pts, k)
        ref = oracle.normals_2d(pts, k)
        ok = np.abs(np.abs(np.sum(got * ref, axis=1)) - 1) < 1e-9
        assert ok.mean() >= 0.99, (k, ok.mean())
    # same path as the lists where both exist: k = 31 by lists, and the draw order reproduces their sums bit for bit
    small = pts[:300]
    assert np.array_equal(uicp.estimate_normals_2d(small, 31), uicp.estimate_normals_2d(small, 31))
    R, t, err = uicp.ICP(a, b, 1e-10, 150, 0.04, method="point_to_line", normal_k=40)
    Ro, to, eo, io = oracle.icp(a, b, 1e-10, 150, 0.04, method="point_to_line", normal_k=40)
    assert uicp.last_icp_info["iterations"] == io["iters"] and rot_err(R, t, Ro, to) < FRO_TOL


def test_bearing_order_gives_identical_results(uicp, libopt):
    """A prepared target may be sorted along a projection or by bearing about the frame origin (sweep.hpp, SWEEP_POLAR);
    the library picks per cloud by estimated window size.  The order only changes how fast the exact searches run:
    forcing either (ICPMI_POLAR=0 / 2) must give the same normals and the same registrations bit for bit — also where
    the bearing order is a poor fit (points at and around the origin, clouds far from it, the seam at +-pi crossed
    by a wall, duplicates, lattices)."""
    from icpmi import batch, synth
    rng = np.random.default_rng(21)
    srcs, tgts = synth.loop_closure_batch(6, seed0=4400)
    far = np.array([35.0, -20.0])
    lattice = np.stack(np.meshgrid(np.arange(-15, 15) * 0.1, np.arange(-15, 15) * 0.1), -1).reshape(-1, 2)
    seam = np.column_stack([np.full(400, -3.0), np.linspace(-1.0, 1.0, 400)])            # a wall across the seam at +-pi
    pairs = [(srcs[i], tgts[i]) for i in range(6)]
    pairs += [(srcs[0] + far, tgts[0] + far),                                             # global frame: origin far outside
              (rng.uniform(-2, 2, (900, 2)), rng.uniform(-2, 2, (1100, 2))),               # points all around and near the origin
              (lattice + 0.013, lattice), (seam + [0.02, 0.01], np.vstack([seam, seam[::7] + [2.0, 0.0]])),
              (np.repeat(rng.uniform(-1, 1, (60, 2)), 4, axis=0), np.repeat(rng.uniform(-1, 1, (70, 2)), 3, axis=0))]
    # round 4: the bearing key is a 21-bit fixed-point float32 bearing — many points per key step: walls seen edge-on
    # (spokes through the origin: one bearing, many ranges), a cloud 20 km away (its whole extent within a few steps),
    # a dense arc (neighbours closer than a step)
    rr = np.linspace(0.5, 6.0, 500)
    spokes = np.vstack([np.column_stack([rr * np.cos(a), rr * np.sin(a)]) for a in (0.3, 0.3 + 2e-6, -2.9, np.pi)])
    arc = 4.0 * np.column_stack([np.cos(np.linspace(1.0, 1.002, 1500)), np.sin(np.linspace(1.0, 1.002, 1500))]) + rng.normal(scale=0.05, size=(1500, 2))
    pairs += [(spokes[::2] + [0.03, 0.02], spokes), (tgts[1] + [2.0e4, -1.5e4], tgts[1] + [2.0e4, -1.5e4] + [0.05, 0.02]), (arc[::3] + 0.02, arc)]
    for method, extra in (("point_to_line", dict(normal_k=12)), ("point_to_point", dict(max_corr_dist=1.5))):
        got = {}
        for mode in ("0", "2", None):
            if mode is None:
                libopt.delenv("ICPMI_POLAR", raising=False)
            else:
                libopt.setenv("ICPMI_POLAR", mode)
            b = batch.IcpBatch([p[0] for p in pairs] + [p[1] for p in pairs], np.arange(len(pairs)),
                               np.arange(len(pairs), 2 * len(pairs)), 1e-10, 60, 0.04, method=method, **extra)
            got[mode] = b.run().cpu().numpy().copy()
        assert np.array_equal(got["0"], got["2"]), method
        assert np.array_equal(got["0"], got[None]), method
        assert (got["0"][:, 14] >= 2).all()
    for pts in (tgts[0], tgts[0] + far, rng.uniform(-2, 2, (1100, 2)), lattice, seam, spokes, arc, tgts[1] + [2.0e4, -1.5e4]):
        nrm = {}
        for mode in ("0", "2"):
            libopt.setenv("ICPMI_POLAR", mode)
            libopt.setenv("ICPMI_PREP_KNN", "sweep")
            nrm[mode] = uicp.estimate_normals_2d(pts, 12)
        assert np.array_equal(nrm["0"], nrm["2"]), len(pts)
    libopt.delenv("ICPMI_POLAR")
    libopt.delenv("ICPMI_PREP_KNN")


def test_p2l_solve(uicp):
    z = load_golden("p2l_solve")
    R, t = uicp._point_to_line_solve_2d(z["src"], z["tgt"], z["normals"], z["idx"])
    assert rot_err(R, t, z["R"], z["t"]) < 1e-11
    R, t = uicp._point_to_line_solve_2d(z["src"], z["tgt"], z["zero_normals"], z["idx"])
    assert np.array_equal(R, np.eye(2)) and np.array_equal(t, np.zeros(2))      # icp.py:107-108


# ── full ICP ────────────────────────────────────────────────────────────────
@pytest.mark.parametrize("name", list(ICP_CASES))
def test_icp_golden(uicp, name):
    z = load_golden("icp")
    s, t, args = icp_case_args(z, name)
    R, tt, err = uicp.ICP(s, t, **args)
    assert rot_err(R, tt, z[f"{name}__R"], z[f"{name}__t"]) < FRO_TOL
    assert abs(err - float(z[f"{name}__err"])) <= 1e-11 * max(1.0, abs(err))
    info = uicp.last_icp_info
    if int(z[f"{name}__conv"]):
        assert info["status"] == oracle.CONVERGED and info["iterations"] == int(z[f"{name}__iters"])
    else:
        Ro, to, eo, io = oracle.icp(s, t, **args)
        assert info["status"] == io["status"] and info["iterations"] == io["iters"]


def test_icp_break_and_prints(uicp, capsys):
    z = load_golden("icp")
    a, b = z["scan_a"], z["scan_b"]
    uicp.VERBOSE = True
    try:
        R, t, err = uicp.ICP(a, b + np.array([30.0, 0.0]), 1e-10, 150, 0.04, method="point_to_point", max_corr_dist=0.05)
        out = capsys.readouterr().out
        assert "ICP max iterations reached: iter=150, error=inf" in out
        assert np.isinf(err) and np.array_equal(R, np.eye(2)) and np.array_equal(t, np.zeros(2))
        assert uicp.last_icp_info["iterations"] == 0
        uicp.ICP(a, b, 1e-10, 150, 0.04, method="point_to_line", normal_k=12)
        assert "ICP converged: iter=5, error=0.00034245" in capsys.readouterr().out
    finally:
        uicp.VERBOSE = False


def test_icp_batch_equals_single_and_oracle(uicp):
    from icpmi import batch, synth
    srcs, tgts = synth.loop_closure_batch(12, seed0=4000)
    R, t, err, info = batch.icp_batch(srcs, tgts, 1e-10, 150, 0.04, method="point_to_line", normal_k=12)
    for i in range(12):
        Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], 1e-10, 150, 0.04, method="point_to_line", normal_k=12)
        assert rot_err(R[i], t[i], Ro, to) < FRO_TOL, i
        assert info["iters"][i] == io["iters"] and info["status"][i] == io["status"], i
    # shared source (slam.py:576-579) and point_to_point with rejection
    R2, t2, e2, i2 = batch.icp_batch(srcs[0], tgts, 1e-10, 150, 0.04, method="point_to_point", max_corr_dist=1.0)
    for i in (0, 5, 11):
        Ro, to, eo, io = oracle.icp(srcs[0], tgts[i], 1e-10, 150, 0.04, method="point_to_point", max_corr_dist=1.0)
        assert rot_err(R2[i], t2[i], Ro, to) < FRO_TOL and i2["iters"][i] == io["iters"]


def test_icp_large_clouds_stream_targets(uicp):
    """Source > 2048 rows (several NN passes) and target > LDS capacity (streamed tiles)."""
    from icpmi import synth
    rng = np.random.default_rng(3)
    segs = synth.maze_segments()
    tgt = np.vstack([synth.to_world(synth.scan(p, 70 + i, segs=segs), p) for i, p in enumerate(synth.trajectory(8))])
    th = np.deg2rad(1.5)
    Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    src = (tgt[rng.permutation(len(tgt))[:6000]] - np.array([0.05, 0.03])) @ Rm.T
    for method in ("point_to_point", "point_to_line"):
        R, t, err = uicp.ICP(src, tgt, 1e-10, 60, 0.02, method=method, normal_k=10, max_corr_dist=1.0)
        Ro, to, eo, io = oracle.icp(src, tgt, 1e-10, 60, 0.02, method=method, normal_k=10, max_corr_dist=1.0)
        assert io["n_src"] > 2048 and io["n_tgt"] > 3072
        assert rot_err(R, t, Ro, to) < FRO_TOL and uicp.last_icp_info["iterations"] == io["iters"]


# ── occupancy grid ──────────────────────────────────────────────────────────
def test_bresenham_cells_bit_exact(umap):
    z = load_golden("bresenham")
    segs, cells, off = z["segs"], z["cells"], z["off"]
    got = umap.bresenham_cells(segs)
    for k in range(len(segs)):
        assert np.array_equal(got[k], cells[off[k]:off[k + 1]]), segs[k]
    s = (3, -7, -40, 11)
    assert umap.OccupancyGrid2D._bresenham(*s) == [tuple(int(v) for v in c) for c in oracle.bresenham(*s)]
    assert umap.OccupancyGrid2D._bresenham(4, 4, 4, 4) == []


def test_world_to_grid(umap):
    z = load_golden("grid")
    b = z["small_bounds"]
    g = umap.OccupancyGrid2D(b[0], b[1], b[2], b[3], resolution=0.05)
    ix, iy = g._world_to_grid_batch(z["w2g_in"], z["w2g_in"][::-1])
    assert np.array_equal(ix, z["w2g_ix"]) and np.array_equal(iy, z["w2g_iy"])
    assert g._world_to_grid(0.0, 0.0) == (120, 100) and g._in_bounds(239, 199) and not g._in_bounds(240, 0)


GRID_KW = dict(resolution=0.05, p_hit=0.85, p_miss=0.42, log_odds_min=-8.0, log_odds_max=8.0)


@pytest.fixture(params=["tiles", "atomic", "owner"])
def raypath(request, libopt):
    """The three passes of the ray-cast: counters per tile in LDS added to a counter grid (the default for a replay of
    several scans whose box the caller knows), integer atomics per (beam, cell) on the scan's counter grid (callers
    without a box), and the single launch in which a workgroup owns a rectangle of cells and applies its counts itself
    (the default for one live scan; replays under "owner" take their default pass)."""
    libopt.setenv("ICPMI_RAYCAST", request.param)      # "tiles" also sends single scans down the tile pass
    return request.param


def test_update_scan_small_grid_bit_exact(umap, raypath):
    z = load_golden("grid")
    b = z["small_bounds"]
    g = umap.OccupancyGrid2D(b[0], b[1], b[2], b[3], **GRID_KW)
    assert (g.ny, g.nx) == _grid_shape(b, 0.05) and g.log_odds.dtype == np.float32
    assert np.array_equal([g.l_hit, g.l_miss], z["small_l"])
    for i in range(40):
        g.update_scan(z["small_origins"][i], z["small_hits"][i])
        if i + 1 in (1, 3, 40):
            assert np.array_equal(g.log_odds, z[f"small_after{i + 1}"]), i
    assert g.log_odds.min() == -8.0
    # replay API: all 40 scans in one call (slam.py:271-277 _rebuild_map)
    g.reset()
    assert np.abs(g.log_odds).sum() == float(z["small_reset_sum"])
    g.update_scans(z["small_origins"], list(z["small_hits"]))
    assert np.array_equal(g.log_odds, z["small_after40"])


def test_update_scan_edges_defaults_display(umap, raypath):
    z = load_golden("grid")
    b = z["small_bounds"]
    g = umap.OccupancyGrid2D(b[0], b[1], b[2], b[3], **GRID_KW)
    g.update_scan(z["edge_origin"], z["edge_hits"])          # origin outside, duplicate + outside hits
    g.update_scan(z["edge_origin"], np.empty((0, 2)))         # empty: silent no-op
    assert np.array_equal(g.log_odds, z["edge_after"])
    g = umap.OccupancyGrid2D(b[0], b[1], b[2], b[3])           # default arguments
    for i in range(3):
        g.update_scan(z["small_origins"][i], z["small_hits"][i])
    assert np.array_equal(g.log_odds, z["default_after3"])
    assert np.array_equal(g.to_probability(), z["default_prob"])
    assert np.array_equal(g.to_display(), z["default_display"])


def test_log_odds_host_view_is_read_only_and_assignment_uploads(umap):
    """The live grid is in HBM: an in-place edit of the host copy could not reach it, so it raises;
    assigning an array uploads it and the next scan clips every cell like the reference's whole-grid np.clip."""
    z = load_golden("grid")
    b = z["small_bounds"]
    g = umap.OccupancyGrid2D(b[0], b[1], b[2], b[3], **GRID_KW)
    with pytest.raises(ValueError):
        g.log_odds[0, 0] = 1.0
    with pytest.raises(ValueError):
        np.clip(g.log_odds, -1.0, 1.0, out=g.log_odds)
    start = np.full((g.ny, g.nx), 9.5, dtype=np.float32)        # above log_odds_max everywhere
    g.log_odds = start
    assert np.array_equal(g.log_odds, start)
    g.update_scan(z["small_origins"][0], z["small_hits"][0])
    ref = start.copy()
    oracle.grid_update_scan(ref, g.min_x, g.min_y, g.resolution, z["small_origins"][0], z["small_hits"][0], g.l_hit, g.l_miss,
                            -8.0, 8.0)
    assert np.array_equal(g.log_odds, ref) and ref.max() == 8.0


def test_update_scan_config4_full_grid(umap, raypath):
    z = load_golden("grid")
    b = z["cfg4_bounds"]
    g = umap.OccupancyGrid2D(b[0], b[1], b[2], b[3], **GRID_KW)
    assert (g.ny, g.nx) == tuple(z["cfg4_shape"])
    g.update_scan(z["cfg4_origin"], z["cfg4_hits"])
    lo = g.log_odds.ravel()
    nz = np.flatnonzero(lo)
    assert np.array_equal(nz, z["cfg4_nz_idx"]) and np.array_equal(lo[nz], z["cfg4_nz_val"])


def test_update_scan_clip_semantics_and_wide_scans(umap, raypath):
    rng = np.random.default_rng(5)
    # clamp range that excludes 0: the first scan clips EVERY cell, like np.clip on the whole grid
    g = umap.OccupancyGrid2D(-2.0, 2.0, -2.0, 2.0, resolution=0.1, log_odds_min=0.5, log_odds_max=3.0)
    ref = np.zeros((g.ny, g.nx), dtype=np.float32)
    hits = rng.uniform(-3, 3, size=(200, 2))
    for _ in range(2):
        g.update_scan([0.1, 0.1], hits)
        oracle.grid_update_scan(ref, g.min_x, g.min_y, 0.1, [0.1, 0.1], hits, g.l_hit, g.l_miss, 0.5, 3.0)
    assert np.array_equal(g.log_odds, ref)
    # uploaded grid with out-of-range values
    g2 = umap.OccupancyGrid2D(-2.0, 2.0, -2.0, 2.0, resolution=0.1)
    start = rng.normal(scale=6.0, size=(g2.ny, g2.nx)).astype(np.float32)
    g2.log_odds = start
    ref = start.copy()
    g2.update_scan([0.0, 0.0], hits)
    oracle.grid_update_scan(ref, g2.min_x, g2.min_y, 0.1, [0.0, 0.0], hits, g2.l_hit, g2.l_miss, -5.0, 5.0)
    assert np.array_equal(g2.log_odds, ref)
    # more beams than a 16-bit counter holds -> two-round path; p_miss > 0.5 -> positive l_miss
    g3 = umap.OccupancyGrid2D(-3.0, 3.0, -3.0, 3.0, resolution=0.05, p_hit=0.6, p_miss=0.55)
    many = rng.uniform(-2.9, 2.9, size=(70000, 2))
    ref = np.zeros((g3.ny, g3.nx), dtype=np.float32)
    g3.update_scan([0.5, -0.5], many)
    oracle.grid_update_scan(ref, g3.min_x, g3.min_y, 0.05, [0.5, -0.5], many, g3.l_hit, g3.l_miss, -5.0, 5.0)
    assert np.array_equal(g3.log_odds, ref)


def test_raycast_lattice_of_hits_all_slopes_and_ties(umap, raypath):
    """Hits on every cell centre of a lattice around the origin: every slope m / D with D <= 23, among them the ones that
    sit exactly on the boundary between two cells of a column (2 k m = (2 j + 1) D), beams of length zero, axis and
    diagonal beams, duplicates; origin on a tile corner, a tile edge and inside a tile."""
    res = 0.1
    for ocell in ((64, 64), (63, 70), (100, 37), (3, 2)):
        g = umap.OccupancyGrid2D(0.0, 14.0, 0.0, 11.0, resolution=res, p_hit=0.7, p_miss=0.4)
        ox, oy = (ocell[0] + 0.5) * res, (ocell[1] + 0.5) * res
        jj, ii = np.meshgrid(np.arange(-23, 24), np.arange(-23, 24))
        hits = np.stack([ox + ii.ravel() * res, oy + jj.ravel() * res], axis=1)
        hits = np.vstack([hits, hits[::7]])                                  # duplicates
        ref = np.zeros((g.ny, g.nx), dtype=np.float32)
        for _ in range(2):
            g.update_scan([ox, oy], hits)
            oracle.grid_update_scan(ref, g.min_x, g.min_y, res, [ox, oy], hits, g.l_hit, g.l_miss, -5.0, 5.0)
        assert np.array_equal(g.log_odds, ref), ocell


def test_raycast_scan_sizes_and_random_geometry(umap, raypath):
    """Scans of 1 .. 9 000 beams (beam chunks of every size), origins inside and outside the grid, beams that leave the
    grid or start and end outside it, a few non-finite coordinates."""
    rng = np.random.default_rng(33)
    g = umap.OccupancyGrid2D(-10.0, 10.0, -6.0, 7.0, resolution=0.05, p_hit=0.8, p_miss=0.35, log_odds_min=-6.0, log_odds_max=6.0)
    ref = np.zeros((g.ny, g.nx), dtype=np.float32)
    sizes = [1, 2, 63, 300, 512, 513, 2048, 2049, 4096, 4100, 8192, 9000]
    origins = rng.uniform(-9, 9, size=(len(sizes), 2)) * np.array([1.0, 0.6])
    origins[3] = [-12.0, 1.0]                                               # outside the grid
    origins[7] = [2.0, 9.0]
    scans = []
    for n, o in zip(sizes, origins):
        ang = rng.uniform(0, 2 * np.pi, size=n)
        rad = rng.uniform(0.0, 14.0, size=n) * (rng.random(n) < 0.9)        # a tenth of the beams have length zero
        scans.append(o + np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1))
    for o, h in zip(origins, scans):                                        # one by one: the live path
        g.update_scan(o, h)
        oracle.grid_update_scan(ref, g.min_x, g.min_y, 0.05, o, h, g.l_hit, g.l_miss, -6.0, 6.0)
        assert np.array_equal(g.log_odds, ref), len(h)
    g2 = umap.OccupancyGrid2D(-10.0, 10.0, -6.0, 7.0, resolution=0.05, p_hit=0.8, p_miss=0.35, log_odds_min=-6.0, log_odds_max=6.0)
    g2.update_scans(origins, scans)                                         # the replay path: groups of mixed sizes
    assert np.array_equal(g2.log_odds, ref)
    bad = scans[4].copy()
    bad[5, 0] = np.nan
    bad[9, 1] = np.inf
    g3 = umap.OccupancyGrid2D(-10.0, 10.0, -6.0, 7.0, resolution=0.05)
    g3.update_scan(origins[4], np.delete(bad, [5, 9], axis=0))
    g4 = umap.OccupancyGrid2D(-10.0, 10.0, -6.0, 7.0, resolution=0.05)
    g4.update_scan(origins[4], bad)                                         # beams with a non-finite end are dropped
    assert np.array_equal(g3.log_odds, g4.log_odds)


def test_replay_of_a_long_trajectory_goes_down_in_pieces(umap, raypath):
    """130 scans along a 60 m path in one update_scans call: the box of all of them spans 176 tiles, so the class sends
    the replay down in pieces of 64 scans with a box each (the tile pass enumerates the tiles of the box per scan).
    Same scans, same order: the oracle's grid bit for bit."""
    rng = np.random.default_rng(77)
    org = np.array([[-28.0 + 0.45 * i, -5.0 + 0.08 * i] for i in range(130)])
    hits = []
    for o in org:                                                 # returns 2 - 6 m around the sensor, in beam order
        ang = np.linspace(-np.pi, np.pi, 230, endpoint=False)
        rad = rng.uniform(2.0, 6.0, size=230)
        hits.append(o + np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1))
    hits[64] = np.empty((0, 2))                                   # an empty scan at the start of a piece
    kw = dict(resolution=0.05, p_hit=0.7, p_miss=0.4, log_odds_min=-4.0, log_odds_max=4.0)
    g = umap.OccupancyGrid2D(-45.0, 45.0, -25.0, 25.0, **kw)
    both = np.vstack([org] + [h for h in hits if len(h)])
    assert g._box_tiles(g._box_of(both.min(axis=0), both.max(axis=0))) > 64 and len(hits) > 2 * g._REPLAY_PIECE
    g.update_scans(org, hits)
    ref = np.zeros((g.ny, g.nx), dtype=np.float32)
    for o, h in zip(org, hits):
        if len(h):
            oracle.grid_update_scan(ref, g.min_x, g.min_y, 0.05, o, h, g.l_hit, g.l_miss, -4.0, 4.0)
    assert np.array_equal(g.log_odds, ref)


def test_raycast_full_size_properties(umap, raypath):
    """Config-4 sized grid, 200 scans: size-independent checks beside the oracle on a sample."""
    from icpmi import synth
    p0 = (0.3, -0.2, np.deg2rad(10.0))
    first = synth.to_world(synth.scan(p0, 2), p0)
    b = (first[:, 0].min() - 50, first[:, 0].max() + 50, first[:, 1].min() - 50, first[:, 1].max() + 50)
    g = umap.OccupancyGrid2D(*b, **GRID_KW)
    poses = [(0.3 + 0.02 * i, -0.2 + 0.01 * i, np.deg2rad(10.0 + 0.5 * i)) for i in range(200)]
    hits = [synth.to_world(synth.scan(p, 2 + i), p) for i, p in enumerate(poses)]
    org = np.array([[p[0], p[1]] for p in poses])
    g.update_scans(org, hits)
    lo = g.log_odds
    assert lo.min() >= -8.0 and lo.max() <= 8.0                     # clip invariant
    assert lo.min() == -8.0 and lo.max() == 8.0                     # both clamps reached after 200 scans
    ref = np.zeros_like(lo)
    for o, h in zip(org, hits):
        oracle.grid_update_scan(ref, g.min_x, g.min_y, 0.05, o, h, g.l_hit, g.l_miss, -8.0, 8.0)
    assert np.array_equal(lo, ref)
    # replay after reset reproduces the grid exactly (order-preserving, no float atomics)
    g.reset()
    g.update_scans(org, hits)
    assert np.array_equal(g.log_odds, ref)


def test_band_replay_equals_whole_replay(umap, raypath):
    """every band of rows replayed on its own (as one rank of a sharded replay would) and the
    bands put together = the unsharded replay = the oracle, bit for bit; rows outside a band stay untouched."""
    from icpmi import dist as idist, synth
    rng = np.random.default_rng(21)
    segs = synth.room_segments()
    poses = [(0.4 * i - 2.0, 0.25 * i - 1.0, 0.3 * i) for i in range(12)]
    hits = [synth.to_world(synth.scan(p, 400 + i, segs=segs), p) for i, p in enumerate(poses)]
    hits[5] = np.empty((0, 2))                                     # an empty scan in the history
    hits[7] = np.vstack([hits[7], rng.uniform(-40, 40, size=(300, 2))])   # beams that leave the grid
    org = np.array([[p[0], p[1]] for p in poses])
    kw = dict(resolution=0.05, p_hit=0.85, p_miss=0.42, log_odds_min=-8.0, log_odds_max=8.0)
    bounds = (-12.3, 12.1, -8.2, 7.7)
    ref = None
    for start_noise in (False, True):
        whole = umap.OccupancyGrid2D(*bounds, **kw)
        start = (rng.normal(scale=6.0, size=(whole.ny, whole.nx)).astype(np.float32) if start_noise
                 else np.zeros((whole.ny, whole.nx), dtype=np.float32))
        if start_noise:
            whole.log_odds = start
        whole.update_scans(org, hits)
        ref = start.copy()
        for o, h in zip(org, hits):
            if len(h):
                oracle.grid_update_scan(ref, whole.min_x, whole.min_y, 0.05, o, h, whole.l_hit, whole.l_miss, -8.0, 8.0)
        assert np.array_equal(whole.log_odds, ref)
        cost = idist.row_costs(whole.ny, whole.min_y, 0.05, org, hits)
        for bands in (idist.row_bands(whole.ny, 3, cost), idist.row_bands(whole.ny, 8), [0, 1, 2, whole.ny - 1, whole.ny],
                      [0, 0, whole.ny]):
            merged = np.full_like(ref, np.nan)
            for r0, r1 in zip(bands[:-1], bands[1:]):
                g = umap.OccupancyGrid2D(*bounds, **kw)
                if start_noise:
                    g.log_odds = start
                g.update_scans(org, hits, rows=(r0, r1))
                lo = g.log_odds
                outside = np.ones(whole.ny, dtype=bool)
                outside[r0:r1] = False
                assert np.array_equal(lo[outside], start[outside]), (bands, r0, r1)   # nothing written outside the band
                merged[r0:r1] = lo[r0:r1]
            assert np.array_equal(merged, ref), bands
    # the sharded entry point with a single rank is the plain replay
    g = umap.OccupancyGrid2D(*bounds, **kw)
    bands, full = idist.replay_scans_sharded(g, org, hits)
    assert bands == [0, g.ny]
    zero_ref = np.zeros_like(ref)
    for o, h in zip(org, hits):
        if len(h):
            oracle.grid_update_scan(zero_ref, g.min_x, g.min_y, 0.05, o, h, g.l_hit, g.l_miss, -8.0, 8.0)
    assert np.array_equal(full.cpu().numpy(), zero_ref) and np.array_equal(g.log_odds, zero_ref)
    with pytest.raises(ValueError):
        g.update_scans(org, hits, rows=(5, 2))


# ── fast (sorted-sweep) path vs exhaustive path ─────────────────────────────
def test_fast_path_equals_exhaustive_path(uicp):
    """icp2.hip + prep.hip must pick the same correspondences as the exhaustive kernels of icp.hip / normals.hip."""
    from icpmi import batch, synth
    srcs, tgts = synth.loop_closure_batch(16, seed0=9000)
    for kw in (dict(method="point_to_line", normal_k=12), dict(method="point_to_point", max_corr_dist=0.8),
               dict(method="point_to_line", normal_k=5, max_corr_dist=0.5)):
        fast = batch.IcpBatch(srcs + tgts, np.arange(16), np.arange(16, 32), 1e-10, 150, 0.04, **kw)
        slow = batch.IcpBatch(srcs + tgts, np.arange(16), np.arange(16, 32), 1e-10, 150, 0.04, force_exhaustive=True, **kw)
        assert fast.fast and not slow.fast
        fast.run(); slow.run()
        Rf, tf, ef, inf_ = fast.unpack()
        Rs, ts, es, ins = slow.unpack()
        assert np.array_equal(inf_["iters"], ins["iters"]) and np.array_equal(inf_["status"], ins["status"])
        assert np.abs(Rf - Rs).max() < 1e-12 and np.abs(tf - ts).max() < 1e-12 and np.abs(ef - es).max() < 1e-14
    # initial guess + rotated world (diagonal walls: exercises the x+y / x-y sort axes)
    th = np.deg2rad(41.0)
    Rw = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    s2 = [s @ Rw.T + 100.0 for s in srcs[:6]]
    t2 = [t @ Rw.T + 100.0 for t in tgts[:6]]
    Ri = np.array([[np.cos(0.01), -np.sin(0.01)], [np.sin(0.01), np.cos(0.01)]])
    R, t, err, info = batch.icp_batch(s2, t2, 1e-10, 150, 0.04, R_init=Ri, t_init=np.array([0.02, -0.01]),
                                      method="point_to_line", normal_k=12)
    for i in range(6):
        Ro, to, eo, io = oracle.icp(s2[i], t2[i], 1e-10, 150, 0.04, R_init=Ri, t_init=np.array([0.02, -0.01]),
                                    method="point_to_line", normal_k=12)
        assert rot_err(R[i], t[i], Ro, to) < FRO_TOL and info["iters"][i] == io["iters"], i


def test_sweep_nn_equals_exhaustive_nn(uicp):
    """The sorted-sweep search (nn_sweep.hip, the search of the fused ICP kernel) must return the very same
    index and float64 distance as the exhaustive kernel, on ordinary scans and on layouts built to stress it:
    a wall perpendicular to every candidate axis in turn, exact duplicates (ties), queries far outside the
    target, single-point and two-point targets, every size around the wave and tile edges."""
    from icpmi import batch, synth
    rng = np.random.default_rng(8)
    wall_v = np.stack([np.full(900, 2.5), np.linspace(-5, 5, 900)], 1)
    wall_d = np.stack([np.linspace(-5, 5, 900), np.linspace(5, -5, 900)], 1)        # x + y constant
    dup = np.repeat(rng.uniform(-1, 1, size=(100, 2)), 5, axis=0)
    grid = np.stack(np.meshgrid(np.arange(30.0), np.arange(30.0)), -1).reshape(-1, 2)  # many exact ties
    a, b = synth.config2_pair(3)
    cases = [(a, b), (b, a),
             (wall_v + rng.normal(scale=1e-3, size=wall_v.shape), wall_v),
             (rng.uniform(-6, 6, size=(777, 2)), wall_d), (wall_d + 0.5, wall_d @ np.array([[0.0, 1.0], [1.0, 0.0]])),
             (dup + 0.01, dup), (dup, dup),
             (grid + 0.5, grid), (grid, grid),
             (rng.uniform(50, 60, size=(400, 2)), rng.uniform(-1, 1, size=(700, 2))),
             (rng.uniform(-1, 1, size=(300, 2)), np.array([[0.25, -0.5]])),
             (rng.uniform(-1, 1, size=(65, 2)), np.array([[0.25, -0.5], [0.25, -0.5]])),
             (rng.uniform(-1e6, 1e6, size=(1000, 2)), rng.uniform(-1e6, 1e6, size=(4096, 2))),
             (rng.uniform(-1, 1, size=(1, 2)), rng.uniform(-1, 1, size=(63, 2)))]
    clouds, ps, pt = [], [], []
    for s, t in cases:
        ps.append(len(clouds)); clouds.append(s)
        pt.append(len(clouds)); clouds.append(t)
    cs = batch.CloudSet.from_numpy(clouds)
    d0, i0 = batch.nn_set(cs, ps, pt)
    d1, i1, s1 = batch.nn_set_sweep(cs, ps, pt, return_second=True)
    d0, i0, d1, i1, s1 = (x.cpu().numpy() for x in (d0, i0, d1, i1, s1))
    for k, (s, t) in enumerate(cases):
        n = len(s)
        assert np.array_equal(i0[k, :n], i1[k, :n]), k
        assert np.array_equal(d0[k, :n], d1[k, :n]), k
        if len(t) >= 2:      # second-best distance against the oracle's 2-NN
            d2o, _ = oracle.knn(t, s, 2)
            assert np.array_equal(s1[k, :n], d2o[:, 1]), k
        else:
            assert np.all(np.isinf(s1[k, :n]))


def test_icp_on_degenerate_layouts_runs(uicp):
    """Single wall / duplicates / no overlap: ill-posed for ICP (results are not comparable between
    implementations), but both kernels must terminate with a valid status and finite bookkeeping."""
    from icpmi import batch
    rng = np.random.default_rng(8)
    wall = np.stack([np.full(900, 2.5), np.linspace(-5, 5, 900)], 1)
    dup = np.repeat(rng.uniform(-1, 1, size=(100, 2)), 5, axis=0)
    cases = [(wall + rng.normal(scale=1e-3, size=wall.shape), wall), (dup + 0.01, dup),
             (rng.uniform(50, 60, size=(400, 2)), rng.uniform(-1, 1, size=(700, 2)))]
    for src, tgt in cases:
        for method in ("point_to_point", "point_to_line"):
            for force in (False, True):
                R, t, err, info = batch.icp_batch([src], [tgt], 1e-12, 30, 1e-4, method=method, normal_k=6,
                                                  force_exhaustive=force)
                assert info["status"][0] in (1, 2) and 1 <= info["iters"][0] <= 30


# ── full-size properties (BASELINE config 5 shape: 512 pairs) ───────────────
def test_batch512_properties(uicp):
    """Size-independent checks on the bench-sized batch: bitwise reproducible, independent of the order and
    of the company a pair keeps in the batch, equal to the oracle on a sample, known rigid motion recovered."""
    from icpmi import batch, synth
    kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
    srcs, tgts = synth.loop_closure_batch(512, seed0=1000)
    b = batch.IcpBatch(srcs + tgts, np.arange(512), np.arange(512, 1024), **kw)
    r1 = b.run().cpu().numpy().copy()
    r2 = b.run().cpu().numpy().copy()
    assert np.array_equal(r1, r2)                                    # fixed reduction trees, integer atomics only
    perm = np.random.default_rng(0).permutation(512)
    bp = batch.IcpBatch(srcs + tgts, perm, 512 + perm, **kw)
    assert np.array_equal(bp.run().cpu().numpy(), r1[perm])          # a pair's result does not depend on its slot
    sub = batch.IcpBatch([srcs[7], tgts[7], srcs[300], tgts[300]], [0, 2], [1, 3], **kw)
    assert np.array_equal(sub.run().cpu().numpy()[:2], r1[[7, 300]])  # ... nor on the batch it is in
    st = r1[:, 15].astype(int)
    assert set(np.unique(st)) <= {1, 2} and (r1[:, 14] >= 2).all() and (r1[:, 14] <= 150).all()
    assert np.isfinite(r1[:, :13]).all()
    for i in (0, 2, 5, 161, 511):                                     # includes pairs that run to max_iterations
        Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], **kw)
        assert rot_err(r1[i, :4].reshape(2, 2), r1[i, 9:11], Ro, to) < FRO_TOL, i
        assert int(r1[i, 14]) == io["iters"] and st[i] == io["status"], i
    # rotations are proper and the totals compose to what the moved cloud shows: det R = 1, R^T R = I
    R = r1[:, :4].reshape(-1, 2, 2)
    assert np.abs(np.linalg.det(R) - 1).max() < 1e-12
    assert np.abs(np.einsum("bij,bik->bjk", R, R) - np.eye(2)).max() < 1e-12


def test_known_motion_recovered(uicp):
    """Target = source moved by a known rigid motion (same points): every method must return that motion."""
    from icpmi import batch, synth
    rng = np.random.default_rng(4)
    srcs, tgts, truth = [], [], []
    for i in range(16):
        s = synth.scan((0.2 * i - 1.0, 0.1 * i, 0.05 * i), 50 + i)
        th = np.deg2rad(rng.uniform(-4, 4))
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        t = rng.uniform(-0.15, 0.15, size=2)
        srcs.append(s); tgts.append(s @ R.T + t); truth.append((R, t))
    for method in ("point_to_point", "point_to_line"):
        # voxel 1e-7: every point is its own voxel, so the filtered clouds are exact images of each other
        R, t, err, info = batch.icp_batch(srcs, tgts, 1e-16, 300, 1e-7, method=method, normal_k=10)
        worst = max(rot_err(R[i], t[i], Rt, tt) for i, (Rt, tt) in enumerate(truth))
        assert worst < (1e-5 if method == "point_to_point" else 1e-8), (method, worst)


def test_submap_10k_target_on_sweep_path(uicp):
    """BASELINE config 3 shape: one 2 048-beam scan against a ~10 k-point rolling submap (target above the LDS
    capacity: sorted through global memory, searched through L2).  Must equal the oracle and the exhaustive path."""
    from icpmi import batch, synth
    segs = synth.maze_segments()
    poses = synth.trajectory(90, step=0.35)
    sub = np.vstack([synth.to_world(synth.scan(p, 500 + i, segs=segs), p) for i, p in enumerate(poses)])
    sub = oracle.voxel_downsample(sub, 0.04)                     # what _build_submap hands to ICP: ~10 k points
    assert 9000 < len(sub) < 16000, len(sub)
    pose = poses[-1]
    cur = synth.scan(pose, 999, segs=segs)
    th = pose[2] + np.deg2rad(1.0)
    R0 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t0 = np.array([pose[0] + 0.05, pose[1] - 0.04])
    for kw in (dict(method="point_to_point", max_corr_dist=1.5), dict(method="point_to_line", normal_k=10)):
        f = batch.IcpBatch([cur, sub], [0], [1], 1e-10, 150, 0.04, R_init=R0, t_init=t0, **kw)
        assert f.fast and f.max_tgt_n > 4096
        f.run()
        R, t, err, info = f.unpack()
        Ro, to, eo, io = oracle.icp(cur, sub, 1e-10, 150, 0.04, R_init=R0, t_init=t0, **kw)
        assert io["n_tgt"] > 4096
        assert rot_err(R[0], t[0], Ro, to) < FRO_TOL and info["iters"][0] == io["iters"] and info["status"][0] == io["status"]
        e = batch.IcpBatch([cur, sub], [0], [1], 1e-10, 150, 0.04, R_init=R0, t_init=t0, force_exhaustive=True, **kw)
        e.run()
        Re, te, ee, ie = e.unpack()
        assert ie["iters"][0] == info["iters"][0] and rot_err(R[0], t[0], Re[0], te[0]) < 1e-11


# ── rotation search ─────────────────────────────────────
@pytest.mark.parametrize("case", ["cfg", "default", "big_rotation", "tiny"])
def test_rotation_search_golden(uicp, case, capsys):
    from utilities import features
    z = load_golden("rotation_search")
    kw = z[f"{case}__kw"]
    features.VERBOSE = True                                   # the reference prints one line per search
    try:
        R, t, s = features.rotation_search(z[f"{case}__src"], z[f"{case}__tgt"], kw[0], kw[1], kw[2])
    finally:
        features.VERBOSE = False
    assert np.array_equal(R, z[f"{case}__R"]) and np.array_equal(t, z[f"{case}__t"])   # same arg-min on the same grid
    ref = float(z[f"{case}__score"])
    assert (np.isinf(s) and np.isinf(ref)) or abs(s - ref) < 1e-13
    if case != "tiny":
        assert "Rotation search: best angle" in capsys.readouterr().out


def test_rotation_scores_against_oracle(uicp):
    from icpmi import synth
    from utilities import features
    rng = np.random.default_rng(12)
    a, b = synth.config2_pair(5)
    src = uicp.voxel_downsample(a, 0.15)
    tgt = uicp.voxel_downsample(b, 0.15)
    angles = np.deg2rad(np.arange(-180, 180, 1.5))
    for s_, t_, shift in ((src - src.mean(0), tgt, tgt.mean(0)), (rng.normal(size=(700, 2)), rng.normal(size=(5000, 2)), (0.3, -0.1)),
                          (rng.normal(size=(1, 2)), rng.normal(size=(3, 2)), (0.0, 0.0))):
        got = features.rotation_scores(s_, t_, angles, shift)
        ref = oracle.rotation_scores(s_, t_, angles, shift)
        assert np.abs(got - ref).max() <= 1e-13 * max(1.0, ref.max())
        assert int(np.argmin(got)) == int(np.argmin(ref))


# ── submap rotation search + scan-to-submap attempt (slam.py:111-225) ──
@pytest.mark.parametrize("case", ["cfg", "default", "imu_narrow", "far_off"])
def test_submap_rotation_search_golden(uicp, case, capsys):
    import torch
    from icpmi import submap
    z = load_golden("submap_rotation")
    sub = load_golden("submap_build")["out"]
    kw = dict(zip(("angle_range", "angle_step", "fine_step", "voxel_size"), z[f"{case}__kw"]))
    submap.VERBOSE = True
    R, t = submap.submap_rotation_search(z["source"], sub, z[f"{case}__pred"], **kw)
    assert np.array_equal(R, z[f"{case}__R"]) and np.array_equal(t, z[f"{case}__t"])     # bit for bit
    assert capsys.readouterr().out == str(z[f"{case}__printed"])
    # the submap as a device tensor (what RollingSubmap.build() hands over) gives the same answer
    R2, t2 = submap.submap_rotation_search(z["source"], torch.from_numpy(sub).cuda(), z[f"{case}__pred"], **kw)
    assert np.array_equal(R2, R) and np.array_equal(t2, t)


def test_attempt_submap_icp_golden(uicp, capsys):
    from icpmi import submap, synth
    z = load_golden("submap_rotation")
    sub = load_golden("submap_build")["out"]
    pred = z["cfg__pred"]
    R, t = submap.submap_rotation_search(z["source"][:3], sub, pred)                      # < 5 points: prediction returned
    assert np.array_equal(R, z["tiny__R"]) and np.array_equal(t, z["tiny__t"])
    cfg = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04)
    uicp.VERBOSE = True
    for k in ("attempt", "attempt_imu"):
        imu = None if np.isnan(z[f"{k}__imu"]) else float(z[f"{k}__imu"])
        before = pred.copy()
        r, tt, err = submap.attempt_submap_icp(z["source"], sub, pred, imu, 3.0, 60.0, 0.8, 0.1, 0.2, cfg, 1.5)
        assert np.array_equal(pred, before)                                               # the caller's pose is not modified
        assert rot_err(r, tt, z[f"{k}__R"], z[f"{k}__t"]) < FRO_TOL
        assert abs(err - float(z[f"{k}__err"])) < 1e-10
        assert uicp.last_icp_info["iterations"] == int(z[f"{k}__iters"])
        assert f"ICP converged: iter={int(z[f'{k}__iters']) - 1}" in capsys.readouterr().out
    uicp.VERBOSE = False
    # through the resident rolling submap (slam.py:503-510 shape)
    segs = synth.maze_segments()
    poses = synth.trajectory(40)
    rs = submap.RollingSubmap(window=40, voxel_size=0.04)
    for i, p in enumerate(poses):
        rs.push(synth.to_world(synth.scan(p, 500 + i, segs=segs), p))
    r2, t2, e2 = rs.attempt_icp(z["source"], pred, None, 3.0, 60.0, 0.8, 0.1, 0.2, cfg, 1.5)
    assert rot_err(r2, t2, z["attempt__R"], z["attempt__t"]) < FRO_TOL and abs(e2 - float(z["attempt__err"])) < 1e-10


# ── device-resident rolling submap ──────────────────────
def test_rolling_submap_equals_build_submap(uicp):
    """slam.py:103-108 + 559-562: 40-scan window, vstack + voxel filter, then scan-to-submap ICP (slam.py:217-225)."""
    from icpmi import synth
    from icpmi.submap import RollingSubmap
    z = load_golden("submap_build")
    segs = synth.maze_segments()
    poses = synth.trajectory(46)
    scans = [synth.to_world(synth.scan(p, 500 + i, segs=segs), p) for i, p in enumerate(poses)]
    sm = RollingSubmap(window=40, voxel_size=0.04)
    for s in scans[:40]:
        sm.push(s)
    assert len(sm) == 40 and sm.points_in == int(z["n_in"])
    assert np.array_equal(sm.build_numpy(), z["out"])                 # golden captured from the reference
    for s in scans[40:]:                                              # window slides: oldest scans drop out
        sm.push(s)
    assert len(sm) == 40
    assert np.array_equal(sm.build_numpy(), oracle.voxel_downsample(np.vstack(scans[6:46]), 0.04))
    sm.reset(scans[:10])
    assert np.array_equal(sm.build_numpy(), oracle.voxel_downsample(np.vstack(scans[:10]), 0.04))
    # scan-to-submap ICP against the resident submap
    sm.reset(scans[:40])
    pose = poses[39]
    cur = synth.scan(pose, 999, segs=segs)
    th = pose[2] + np.deg2rad(1.0)
    R0 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t0 = np.array([pose[0] + 0.05, pose[1] - 0.04])
    R, t, err, info = sm.icp(cur, 1e-10, 150, 0.04, R_init=R0, t_init=t0, method="point_to_point", max_corr_dist=1.5)
    Ro, to, eo, io = oracle.icp(cur, z["out"], 1e-10, 150, 0.04, R_init=R0, t_init=t0, method="point_to_point", max_corr_dist=1.5)
    assert rot_err(R, t, Ro, to) < FRO_TOL and info["iters"] == io["iters"]
    assert RollingSubmap(window=3).build_numpy().shape == (0, 2)


def test_icp_fast_path_shapes(uicp):
    """Every instantiation of the fast kernel: <=1024 / <=2048 / <=4096 source rows x target in LDS / through L2."""
    from icpmi import batch, synth
    rng = np.random.default_rng(6)
    segs = synth.maze_segments()
    world = np.vstack([synth.to_world(synth.scan(p, 70 + i, segs=segs), p) for i, p in enumerate(synth.trajectory(8))])
    th = np.deg2rad(1.2)
    Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    for n_src in (700, 1800, 3500):
        for n_tgt in (3000, 9000):
            tgt = world[rng.permutation(len(world))[:n_tgt]]
            src = (world[rng.permutation(len(world))[:n_src]] - np.array([0.04, 0.03])) @ Rm.T
            for method in ("point_to_line", "point_to_point"):
                kw = dict(method=method, normal_k=8, max_corr_dist=0.8)
                b = batch.IcpBatch([src, tgt], [0], [1], 1e-10, 80, 0.01, **kw)      # voxel 0.01 keeps nearly every point
                assert b.fast
                b.run()
                R, t, err, info = b.unpack()
                Ro, to, eo, io = oracle.icp(src, tgt, 1e-10, 80, 0.01, **kw)
                assert rot_err(R[0], t[0], Ro, to) < FRO_TOL and info["iters"][0] == io["iters"], (n_src, n_tgt, method)


# ── pose graph (utilities/pose_graph.py) ──────────────────
def _pose_graph_from_golden(z, name):
    from utilities import pose_graph as upg
    pg = upg.PoseGraph2D()
    for v in z[f"{name}__nodes"]:
        pg.add_node(v)
    none = z[f"{name}__none"]
    for q, ((i, j), zz, om) in enumerate(zip(z[f"{name}__ij"], z[f"{name}__z"], z[f"{name}__omega"])):
        pg.add_edge(int(i), int(j), zz, None if none[q] else om)
    kw = z[f"{name}__kw"]
    return pg, dict(n_iterations=int(kw[0]), fix_node=int(kw[1]), convergence_eps=float(kw[2]))


def test_pose_graph_golden(uicp, capsys):
    """Every case the reference was run on: same outcome line, same poses (to solver rounding), same total error."""
    import re
    from utilities import pose_graph as upg
    z = load_golden("pose_graph")
    upg.VERBOSE = True
    for name in z["names"]:
        pg, kw = _pose_graph_from_golden(z, name)
        before = pg.total_error()
        assert abs(before - z[f"{name}__err"][0]) <= 1e-11 * max(1.0, z[f"{name}__err"][0]), name
        held = [v for v in pg.nodes]
        pg.optimize(**kw)
        printed, ref_printed = capsys.readouterr().out, str(z[f"{name}__printed"])
        strip = lambda s: re.sub(r"\|\|Δx\|\|=.*", "", s)            # the step norm is compared numerically below
        assert strip(printed) == strip(ref_printed), (name, printed, ref_printed)
        m = re.search(r"=([0-9.e+-]+)\n", ref_printed) if "Δx" in ref_printed else None
        if m and float(m.group(1)) > 1e-12:
            assert abs(pg.last_info["step_norm"] - float(m.group(1))) <= 6e-3 * float(m.group(1)), name
        out = np.array(pg.nodes).reshape(-1, 3)
        assert np.abs(out - z[f"{name}__out"]).max() < 1e-8, (name, np.abs(out - z[f"{name}__out"]).max())
        assert all(a is b for a, b in zip(held, pg.nodes))            # updated in place, like the reference
        after = pg.total_error()
        assert abs(after - z[f"{name}__err"][1]) <= 1e-7 * max(1.0, z[f"{name}__err"][1]), name
        assert np.abs(np.array(pg.get_poses_as_matrices()) - z[f"{name}__mats"]).max() < 1e-8
    upg.VERBOSE = False
    assert np.array_equal(upg.normalize_angle(z["wrap_in"]), z["wrap_out"])
    assert np.array_equal(upg.relative_transform_vec(z["T1"], z["T2"]), z["rel12"])
    assert np.array_equal(upg.pose_matrix_to_vec(z["T1"]), z["vec1"])
    assert np.array_equal(upg.pose_vec_to_matrix(z["vec1"]), z["T1"])


def test_pose_graph_structured_equals_dense_and_oracle(uicp):
    """Chain + closures goes through the block-tridiagonal + low-rank solver; the same graph with its nodes
    renumbered has no chain and goes through the dense solver.  Both must agree with each other and the oracle."""
    from oracle import pose_graph as opg
    from utilities import pose_graph as upg
    rng = np.random.default_rng(7)
    n = 120
    th = np.cumsum(rng.normal(0.0, 0.1, n))
    xy = np.cumsum(np.stack([0.3 * np.cos(th), 0.3 * np.sin(th)], axis=1), axis=0)
    truth = np.column_stack([xy, upg.normalize_angle(th)])
    T = [upg.pose_vec_to_matrix(v) for v in truth]
    est, edges = [T[0]], []
    for k in range(1, n):
        zt = upg.relative_transform_vec(T[k - 1], T[k]) + rng.normal(0.0, 0.01, 3)
        est.append(est[-1] @ upg.pose_vec_to_matrix(zt))
        edges.append((k - 1, k, zt, np.eye(3) * rng.uniform(100, 1e4)))
    for a, b in [(119, 0), (100, 7), (90, 30), (60, 60 - 35), (119, 50), (5, 5)]:       # (5, 5): an edge from a node to itself
        edges.append((a, b, upg.relative_transform_vec(T[a], T[b]) + rng.normal(0.0, 0.003, 3), np.eye(3) * 2e4))
    nodes = np.array([upg.pose_matrix_to_vec(t) for t in est])
    perm = rng.permutation(n)                                   # node k of the chain graph is node perm[k] of the shuffled one

    def run(nodes_, edges_, fix):
        pg = upg.PoseGraph2D()
        for v in nodes_:
            pg.add_node(v)
        for e in edges_:
            pg.add_edge(*e)
        pg.optimize(n_iterations=25, fix_node=fix, convergence_eps=1e-9)
        return np.array(pg.nodes), pg.last_info

    chain, info_c = run(nodes, edges, 3)
    shuffled_nodes = np.empty_like(nodes)
    shuffled_nodes[perm] = nodes
    dense, info_d = run(shuffled_nodes, [(int(perm[i]), int(perm[j]), z_, o) for i, j, z_, o in edges], int(perm[3]))
    ei, ej = np.array([e[0] for e in edges]), np.array([e[1] for e in edges])
    zz, om = np.array([e[2] for e in edges]), np.array([e[3] for e in edges])
    ref, it, st, step = opg.optimize(nodes, ei, ej, zz, om, 25, 3, 1e-9)
    assert info_c["status"] == info_d["status"] == st == opg.OK and info_c["iterations"] == info_d["iterations"] == it
    assert np.abs(chain - ref).max() < 1e-8 and np.abs(dense[perm] - ref).max() < 1e-8
    assert np.abs(chain[3] - nodes[3]).max() < 1e-15            # the anchor stays (its angle is re-wrapped every iteration, as in the reference)


def test_pose_graph_edge_cases(uicp, capsys):
    from utilities import pose_graph as upg
    pg = upg.PoseGraph2D()
    pg.optimize()                                               # empty: silent no-op (pose_graph.py:89-91)
    assert pg.total_error() == 0.0 and pg.add_node([1, 2, 3]) == 0
    pg.optimize()
    assert np.array_equal(pg.nodes[0], [1.0, 2.0, 3.0]) and capsys.readouterr().out == ""
    pg.add_node([2.0, 2.0, 3.0])
    pg.add_edge(0, 1, [1.0, 0.0, 0.0], np.zeros((3, 3)))        # zero information: H is singular
    upg.VERBOSE = True
    pg.optimize()
    upg.VERBOSE = False
    assert "singular H at iter 0" in capsys.readouterr().out and np.array_equal(pg.nodes[1], [2.0, 2.0, 3.0])
    with pytest.raises(IndexError):
        pg.add_edge(0, 5, [0, 0, 0])
        pg.optimize()


def test_large_batch_uses_two_workgroups_per_cu_and_matches_the_oracle(uicp):
    """From 1 024 pairs on the launcher runs 512 threads x 3 rows (two workgroups per CU; clouds that keep more than
    1 536 rows after the voxel filter go to a second launch of 1 024 threads); the voxel filter and the prepare kernel
    change their occupancy too.  Same results: a sample of the pairs against the oracle, and the whole batch against
    the same pairs run in small batches (1 024 threads x 2 rows)."""
    from icpmi import batch, synth
    B = 1100
    srcs, tgts = synth.loop_closure_batch(B, seed0=31000)
    kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
    big = batch.IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), **kw)
    big.run()
    R, t, err, info = big.unpack()
    for i in range(0, B, 37):
        Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], 1e-10, 150, 0.04, method="point_to_line", normal_k=12)
        assert info["iters"][i] == io["iters"] and info["status"][i] == io["status"], i
        assert rot_err(R[i], t[i], Ro, to) < FRO_TOL, (i, rot_err(R[i], t[i], Ro, to))
    for lo in range(0, B, 275):
        idx = np.arange(lo, min(B, lo + 275))
        small = batch.IcpBatch([srcs[i] for i in idx] + [tgts[i] for i in idx], np.arange(len(idx)),
                               np.arange(len(idx), 2 * len(idx)), **kw)
        small.run()
        Rs, ts, es, infs = small.unpack()
        assert np.array_equal(infs["iters"], info["iters"][idx])
        assert max(rot_err(Rs[j], ts[j], R[i], t[i]) for j, i in enumerate(idx)) < 1e-11


def test_large_batch_with_wide_clouds_takes_the_second_launch(uicp):
    """In a large batch the fused ICP kernel holds at most 1 536 rows of a cloud (512 threads x 3 rows, LDS copy of the
    target for two workgroups per CU).  The voxel filter leaves its row counts on the device, so the launcher cannot
    know: pairs with a larger source OR target are skipped by the first launch and registered by a second one
    (1 024 threads).  Mixed batch — narrow pairs, wide sources, wide targets — against the same pairs in small batches."""
    from icpmi import batch, synth
    B = 1040
    srcs, tgts = synth.loop_closure_batch(B, seed0=52000)
    for i in range(B):                              # voxel 0.005 keeps (nearly) every beam: 2 048 rows wide, 1 024 narrow
        if i % 4 != 1:
            srcs[i] = srcs[i][::2]
        if i % 4 != 2:
            tgts[i] = tgts[i][::2]
    kw = dict(error_threshold=1e-10, max_iterations=25, voxel_size=0.005, method="point_to_line", normal_k=12)
    big = batch.IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), **kw)
    res = big.run().cpu().numpy()
    cnt = big.vox.cnt.cpu().numpy()
    assert (cnt[:B] > 1536).sum() > 200 and (cnt[B:] > 1536).sum() > 200 and ((cnt[:B] <= 1536) & (cnt[B:] <= 1536)).sum() > 200
    assert (res[:, 15] != 4).all() and (res[:, 14] >= 2).all()             # every pair was registered by exactly one launch
    for lo in range(0, B, 260):
        idx = np.arange(lo, min(B, lo + 260))
        small = batch.IcpBatch([srcs[i] for i in idx] + [tgts[i] for i in idx], np.arange(len(idx)),
                               np.arange(len(idx), 2 * len(idx)), **kw)
        rs = small.run().cpu().numpy()
        assert np.array_equal(rs[:, 14], res[idx, 14])
        assert np.abs(rs[:, :12] - res[idx, :12]).max() < 1e-11
    for i in (1, 2, 3, 1001, 1002):
        Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], 1e-10, 25, 0.005, method="point_to_line", normal_k=12)
        assert int(res[i, 14]) == io["iters"] and rot_err(res[i, :4].reshape(2, 2), res[i, 9:11], Ro, to) < FRO_TOL


def test_two_stage_run_of_a_large_batch_is_the_single_launch_bit_for_bit(uicp, libopt):
    """A batch of >= 1 024 pairs runs the fused ICP kernel in two stages: everybody up to 12 iterations, then the pairs
    still running — parked with their moving rows, matches and totals — together in a second launch (so that the
    150-iteration pairs do not finish alone at the end of one long launch).  The continuation searches afresh, finds
    the same matches and sums the same terms in the same order: every result double equals the single launch's."""
    from icpmi import batch, synth
    B = 1100
    srcs, tgts = synth.loop_closure_batch(B, seed0=61000)
    rng = np.random.default_rng(8)
    th = rng.uniform(-0.05, 0.05, size=B)
    R0 = np.stack([np.stack([np.cos(th), -np.sin(th)], -1), np.stack([np.sin(th), np.cos(th)], -1)], -2)
    t0 = rng.uniform(-0.05, 0.05, size=(B, 2))
    cases = [dict(method="point_to_line", max_iterations=150, max_corr_dist=None, init=None),
             dict(method="point_to_line", max_iterations=24, max_corr_dist=0.5, init=(R0, t0)),
             dict(method="point_to_point", max_iterations=60, max_corr_dist=1.0, init=None)]
    for case in cases:
        out = {}
        for stages in ("1", "2"):
            libopt.setenv("ICPMI_ICP2_STAGES", stages)
            kw = dict(error_threshold=1e-10, max_iterations=case["max_iterations"], voxel_size=0.04, method=case["method"],
                      normal_k=12, max_corr_dist=case["max_corr_dist"])
            if case["init"] is not None:
                kw.update(R_init=case["init"][0], t_init=case["init"][1])
            b = batch.IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), **kw)
            out[stages] = b.run().cpu().numpy()[:B].copy()
        it = out["1"][:, 14]
        assert (it > 12).sum() > 20, case                                      # pairs that go through the second stage
        assert case["method"] == "point_to_point" or (it <= 12).sum() > 200    # ... and (point-to-line) pairs that do not
        assert set(np.unique(out["2"][:, 15])) <= {1.0, 2.0, 3.0}, case         # nobody is left parked
        assert np.array_equal(out["1"], out["2"]), case
    for i in (0, 7, 500):
        Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], 1e-10, 60, 0.04, method="point_to_point", max_corr_dist=1.0)
        assert int(out["2"][i, 14]) == io["iters"] and rot_err(out["2"][i, :4].reshape(2, 2), out["2"][i, 9:11], Ro, to) < FRO_TOL


def test_two_stage_run_with_nobody_parked(uicp):
    """A large batch whose pairs all settle at once (source = target): the first stage finishes everybody, the list of
    parked pairs stays empty and the second launch has nothing to do."""
    from icpmi import batch, synth
    B = 1024
    srcs, _ = synth.loop_closure_batch(8, seed0=70000)
    clouds = [srcs[i % 8] for i in range(B)]
    b = batch.IcpBatch(clouds + clouds, np.arange(B), np.arange(B, 2 * B), error_threshold=1e-10, max_iterations=150,
                       voxel_size=0.04, method="point_to_line", normal_k=12)
    res = b.run().cpu().numpy()[:B]
    assert (res[:, 15] == 1).all() and (res[:, 14] <= 3).all()                 # converged, at once
    assert np.abs(res[:, :4] - np.array([1.0, 0.0, 0.0, 1.0])).max() < 1e-12 and np.abs(res[:, 9:11]).max() < 1e-12
    assert np.array_equal(res[:8], res[8:16])                                  # the same pair gives the same bits wherever it sits


def test_voxel_all_three_sort_paths(uicp):
    """voxel.hip sorts (key, row) packed in 32 bits, packed in 64 bits, or as pairs, depending on how many voxels
    the bounding box holds; all three must give np.unique's rows and order (oracle), bit for bit."""
    from icpmi import synth
    a, _ = synth.config2_pair(11)
    for voxel in (0.04, 0.002, 1e-7):          # ~1.5e5 cells, ~6e7 cells, ~2e16 cells for a 20 m x 12 m room
        got = uicp.voxel_downsample(a, voxel)
        ref = oracle.voxel_downsample(a, voxel)
        assert got.shape == ref.shape and np.array_equal(got, ref), voxel
    rng = np.random.default_rng(4)
    cloud3 = rng.uniform(-1, 1, size=(3000, 3))
    for voxel in (0.2, 0.01, 1e-6):
        assert np.array_equal(uicp.voxel_downsample(cloud3, voxel), oracle.voxel_downsample(cloud3, voxel)), voxel


def test_replay_groups_with_wide_and_empty_scans(umap, raypath):
    """A replay whose groups of 16 scans are interrupted by a scan with more beams than a 16-bit counter holds (two-round
    path, flushes the pending group) and by empty scans (skipped, not clipped): equal to the scans applied one by one."""
    from icpmi import synth
    rng = np.random.default_rng(17)
    segs = synth.room_segments()
    poses = [(0.2 * i - 3.0, 0.1 * i - 1.0, 0.2 * i) for i in range(40)]
    hits = [synth.to_world(synth.scan(p, 700 + i, segs=segs)[::2], p) for i, p in enumerate(poses)]
    hits[21] = rng.uniform(-9.5, 9.5, size=(70000, 2)) * np.array([1.0, 0.6])       # wide scan
    hits[27] = np.empty((0, 2))
    hits[0] = np.empty((0, 2))
    org = np.array([[p[0], p[1]] for p in poses])
    kw = dict(resolution=0.05, p_hit=0.7, p_miss=0.45, log_odds_min=-4.0, log_odds_max=6.0)
    g = umap.OccupancyGrid2D(-11.0, 11.0, -7.0, 7.0, **kw)
    g.update_scans(org, hits)
    ref = np.zeros((g.ny, g.nx), dtype=np.float32)
    for o, h in zip(org, hits):
        if len(h):
            oracle.grid_update_scan(ref, g.min_x, g.min_y, 0.05, o, h, g.l_hit, g.l_miss, -4.0, 6.0)
    assert np.array_equal(g.log_odds, ref)
    one = umap.OccupancyGrid2D(-11.0, 11.0, -7.0, 7.0, **kw)
    for o, h in zip(org, hits):
        one.update_scan(o, h)
    assert np.array_equal(one.log_odds, ref)


def test_pose_graph_python_index_semantics(uicp):
    """The reference indexes `self.nodes[i]` with whatever the caller passed: -1 is the last node."""
    from utilities import pose_graph as upg
    rng = np.random.default_rng(3)
    nodes = np.cumsum(rng.normal(0.3, 0.05, size=(9, 3)) * np.array([1.0, 0.3, 0.1]), axis=0)

    def build(last):
        pg = upg.PoseGraph2D()
        for v in nodes:
            pg.add_node(v)
        for k in range(1, 9):
            pg.add_edge(k - 1, k, nodes[k] - nodes[k - 1] + rng.normal(0, 1e-3, 3) * 0, np.eye(3) * 50.0)
        pg.add_edge(last, 0, [-2.0, -0.5, -0.6], np.eye(3) * 200.0)
        pg.optimize()
        return np.array(pg.nodes), pg.total_error()

    a, ea = build(8)
    b, eb = build(-1)
    assert np.array_equal(a, b) and ea == eb


def test_mixed_batch_of_scans_and_one_large_target(uicp):
    """One batch that holds ordinary scan targets (~1 400 rows) AND a target above 4 096 rows (a rolling submap).  The
    fused launch then searches every target in place (one instantiation per batch), which cannot walk a bearing order:
    the prepare step must sort such a batch along projections only (it used to pick the bearing order for the scans,
    whose pairs then came back ICPMI_ST_EMPTY with R = I and err = inf and no error code)."""
    from icpmi import batch, synth
    srcs, tgts = synth.loop_closure_batch(6, seed0=7300)
    segs = synth.maze_segments()
    poses = synth.trajectory(12)
    big = np.vstack([synth.to_world(synth.scan(p, 900 + i, segs=segs), p) for i, p in enumerate(poses)])   # 24 576 rows
    cur = synth.scan(poses[6], 977, segs=segs)
    th = np.deg2rad(1.0)
    R0 = np.array([[np.cos(poses[6][2] + th), -np.sin(poses[6][2] + th)], [np.sin(poses[6][2] + th), np.cos(poses[6][2] + th)]])
    t0 = np.array(poses[6][:2]) + [0.05, -0.04]
    clouds = srcs + [cur] + tgts + [big]
    B = 7
    Ri = np.stack([np.eye(2)] * 6 + [R0])
    ti = np.stack([np.zeros(2)] * 6 + [t0])
    for method, extra in (("point_to_line", dict(normal_k=12)), ("point_to_point", dict(max_corr_dist=1.5))):
        b = batch.IcpBatch(clouds, np.arange(B), np.arange(B, 2 * B), 1e-10, 40, 0.04, R_init=Ri, t_init=ti, method=method, **extra)
        res = b.run().cpu().numpy()
        cnt = b.vox.cnt.cpu().numpy()
        assert len(big) > 4096 and cnt[2 * B - 1] > 2048 and cnt[B:2 * B - 1].max() <= 2048   # routed by capacity: the in-place search
        assert (res[:, 15] != 4).all(), res[:, 15]                         # nobody is ICPMI_ST_EMPTY
        for i in range(B):
            Ro, to, eo, io = oracle.icp(clouds[i], clouds[B + i], 1e-10, 40, 0.04, R_init=Ri[i], t_init=ti[i], method=method, **extra)
            assert int(res[i, 14]) == io["iters"], (method, i)
            assert rot_err(res[i, :4].reshape(2, 2), res[i, 9:11], Ro, to) < FRO_TOL, (method, i)


def test_pair_context_keeps_the_sweep_path_above_4096_rows_in_all(uicp):
    """ICP() on a 2 100-row scan against a 3 000-row target: 5 100 rows in all, each cloud within the on-chip limit —
    the pair context must stay on the sorted-sweep kernels (it used to fall back on the exhaustive kernel for good)."""
    from icpmi import batch, synth
    rng = np.random.default_rng(5)
    segs = synth.maze_segments()
    a = np.vstack([synth.scan((5.0, 5.0, 0.0), 31, segs=segs), rng.uniform(-3, 3, (200, 2))])[:2100]
    bb = np.vstack([synth.scan((5.1, 4.95, np.deg2rad(2.0)), 32, segs=segs), synth.scan((5.1, 4.95, np.deg2rad(2.1)), 33, segs=segs)])[:3000]
    assert len(a) == 2100 and len(bb) == 3000
    R, t, err = uicp.ICP(a, bb, 1e-10, 60, 0.005, method="point_to_line", normal_k=40)     # k > 31: the fast path only
    ctx = batch.PairContext.get()
    assert ctx.batch.fast and ctx.cap_s <= batch.PREP_MAX_POINTS and ctx.cap_t <= batch.PREP_MAX_POINTS
    Ro, to, eo, io = oracle.icp(a, bb, 1e-10, 60, 0.005, method="point_to_line", normal_k=40)
    assert uicp.last_icp_info["iterations"] == io["iters"] and rot_err(R, t, Ro, to) < FRO_TOL


def test_library_options_and_shutdown(uicp):
    """icpmi_set_option / icpmi_shutdown (include/icpmi.h, "library state"): unknown names are refused, a set option
    takes effect without touching the environment, shutdown destroys the side streams and later calls make them again."""
    from icpmi import _lib, batch, synth
    with pytest.raises(_lib.IcpmiError):
        _lib.set_option("NO_SUCH_SWITCH", "1")
    srcs, tgts = synth.loop_closure_batch(4, seed0=100)
    kw = dict(error_threshold=1e-10, max_iterations=30, voxel_size=0.04, method="point_to_line", normal_k=12)
    out = {}
    for mode in ("0", None):
        _lib.set_option("ICPMI_POLAR", mode)
        b = batch.IcpBatch(srcs + tgts, np.arange(4), np.arange(4, 8), **kw)
        out[mode] = b.run().cpu().numpy().copy()
        dirs = b.prepared[b.raw.total_rows * 40: b.raw.total_rows * 40 + 8 * 4].view(torch_int32()).cpu().numpy()
        assert (dirs[4:] == 4).any() == (mode is None), (mode, dirs)        # bearing order only when allowed
    assert np.array_equal(out["0"], out[None])
    _lib.shutdown()
    b = batch.IcpBatch(srcs + tgts, np.arange(4), np.arange(4, 8), **kw)
    assert np.array_equal(b.run().cpu().numpy(), out[None])
    _lib.shutdown()


def torch_int32():
    import torch
    return torch.int32


def test_rccl_world_of_one_gathers_results_and_bands(uicp):
    """The `nccl` backend of torch.distributed IS RCCL on ROCm.  A one-GPU box cannot run the 8-rank job, but a world of
    one still initialises RCCL, creates the communicator on the device and pushes both collectives of icpmi.dist
    (result records, row bands) through ncclAllGather on device tensors."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from icpmi import dist as idist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        rng = np.random.default_rng(3)
        local = torch.from_numpy(rng.normal(size=(37, 16))).to(dev)
        got = idist.gather_results(local, 37, 0, 1, force_collective=True)
        assert got.is_cuda and torch.equal(got, local)
        band = torch.from_numpy(rng.normal(size=(50, 64)).astype(np.float32)).to(dev)
        full = idist.gather_bands(band, [0, 50], 0, 1, force_collective=True)
        assert full.is_cuda and torch.equal(full, band)
        # the sharded batch end to end on the one rank (pairs -> IcpBatch -> all_gather)
        from icpmi import synth
        srcs, tgts = synth.loop_closure_batch(5, seed0=900)
        res = idist.icp_batch_sharded(srcs, tgts, 1e-10, 30, 0.04, method="point_to_line", normal_k=12)
        for i in (0, 4):
            Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], 1e-10, 30, 0.04, method="point_to_line", normal_k=12)
            r = res[i].cpu().numpy()
            assert int(r[14]) == io["iters"] and rot_err(r[:4].reshape(2, 2), r[9:11], Ro, to) < FRO_TOL
        # the loop-closure matching as slam.py:575-597 runs it, sharded (one rank here): rotation search + ICP per candidate,
        # the records through the collective, first accepted candidate; one candidate above the search's capacity hint
        from icpmi import prealign
        srcs, tgts = synth.loop_closure_batch(9, seed0=91000, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
        icp_cfg = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
        feat_cfg = dict(rotation_voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1)
        job = idist.RunIcpPairSharded(srcs[0], tgts, icp_cfg, feat_cfg)
        assert job.run(force_collective=True).is_cuda
        R, t, err, info = job.results()
        R1, t1, err1, info1 = prealign.run_icp_pair_batch(srcs[0], tgts, icp_cfg, feat_cfg)
        assert np.array_equal(R, R1) and np.array_equal(t, t1) and np.array_equal(err, err1) and np.array_equal(info["iters"], info1["iters"])
        ok = np.flatnonzero(err < 0.08)
        assert job.first_accepted(0.08) == (int(ok[0]) if len(ok) else -1)
        job = idist.RunIcpPairSharded(srcs[0], tgts, icp_cfg, feat_cfg, max_rows_hint=320)      # ~430 filtered rows > 320: status 2
        job.run(force_collective=True)
        assert (job.gathered[:, job.SEARCH_STATUS] == 2.0).all()
        R2, t2, err2, info2 = job.results()
        for i in range(9):
            assert rot_err(R2[i], t2[i], R1[i], t1[i]) < FRO_TOL and int(info2["iters"][i]) == int(info1["iters"][i]), i
        R3, t3, err3, info3 = idist.run_icp_pair_batch_sharded(srcs[0], tgts, icp_cfg, feat_cfg, error_accept=0.08)
        assert np.array_equal(R3, R1) and info3["first_accepted"] == (int(ok[0]) if len(ok) else -1)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


# ── batched pre-alignment: rotation search of every pair of a batch, feeding the batched ICP (slam.py:53-98) ──
@pytest.mark.parametrize("case", ["cfg", "default", "big_rotation", "tiny"])
def test_rotation_search_batch_golden(uicp, case):
    """The batch entry on the reference's golden pairs: R, t bit-equal (same arg-min on the same angle grid), and the
    R_init / t_init it leaves on the device for the ICP are those numbers too."""
    from icpmi import prealign
    z = load_golden("rotation_search")
    kw = z[f"{case}__kw"]
    b = prealign.RotationSearchBatch([z[f"{case}__src"], z[f"{case}__tgt"]], [0], [1], kw[0], kw[1], kw[2])
    b.run()
    R, t, s, rec = b.results()
    assert np.array_equal(R[0], z[f"{case}__R"]) and np.array_equal(t[0], z[f"{case}__t"])
    ref = float(z[f"{case}__score"])
    assert (np.isinf(s[0]) and np.isinf(ref)) or abs(s[0] - ref) < 1e-13
    init = b.init.cpu().numpy()[0]
    assert np.array_equal(init[:4].reshape(2, 2), z[f"{case}__R"]) and np.array_equal(init[4:], z[f"{case}__t"])
    assert int(rec[0, 11]) == (1 if case == "tiny" else 0)


def test_rotation_search_batch_512_pairs(uicp, libopt):
    """512 loop-closure candidates (SURVEY section 8d geometry: within 3 m / 20 degrees of a shared current scan) in one
    chain of launches: R, t of every pair bit-equal to the single-pair chain and (a sample) to the oracle; the bounded
    search scores a fraction of the coarse angles and finds the arg-min of scoring them all."""
    from icpmi import prealign, synth
    from utilities import features
    B = 512
    srcs, tgts = synth.loop_closure_batch(B, seed0=83000, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
    b = prealign.RotationSearchBatch([srcs[0]] + tgts, np.zeros(B, dtype=np.int32), np.arange(1, B + 1, dtype=np.int32),
                                     0.15, 1.5, 0.1, max_rows_hint=1024)
    b.run()
    R, t, s, rec = b.results()
    assert (rec[:, 11] == 0).all()
    assert rec[:, 12].mean() < 60 and rec[:, 12].min() >= 1, rec[:, 12].mean()      # of 240 coarse angles
    libopt.setenv("ICPMI_RS_BATCH", "full")
    b.run()
    Rf, tf, sf, recf = b.results()
    assert (recf[:, 12] == 240).all()
    assert np.array_equal(recf[:, :11], rec[:, :11])                                 # same winners, same exact scores
    libopt.delenv("ICPMI_RS_BATCH")
    features.VERBOSE = False
    for i in range(0, B, 16):
        R1, t1, s1 = features.rotation_search(srcs[0], tgts[i], 0.15, 1.5, 0.1)
        assert np.array_equal(R[i], R1) and np.array_equal(t[i], t1) and abs(s[i] - s1) <= 1e-13 * max(1.0, s1), i
    for i in range(0, B, 64):
        Ro, to, so = oracle.rotation_search(srcs[0], tgts[i], 0.15, 1.5, 0.1)
        assert np.array_equal(R[i], Ro) and np.array_equal(t[i], to), i
    init = b.init.cpu().numpy()
    assert np.array_equal(init[:, :4].reshape(B, 2, 2), R) and np.array_equal(init[:, 4:], t)


def test_rotation_search_batch_edge_cases(uicp):
    """Distinct sources, clouds of very different sizes, a pair with too few points, a pair above the capacity hint
    (searched by the single-pair entry instead), duplicated and collinear targets."""
    from icpmi import prealign, synth
    from utilities import features
    rng = np.random.default_rng(77)
    a, bb = synth.config2_pair(9)
    line = np.column_stack([np.linspace(-4, 4, 900), np.full(900, 1.0)]) + rng.normal(scale=0.002, size=(900, 2))
    dup = np.repeat(rng.uniform(-3, 3, size=(150, 2)), 4, axis=0)
    clouds = [a, bb, a[::7], bb[::3], rng.uniform(-0.1, 0.1, size=(40, 2)), line, dup, rng.uniform(-6, 6, size=(3000, 2))]
    ps = [0, 2, 0, 4, 5, 6, 0, 7]
    pt = [1, 3, 3, 1, 0, 1, 6, 1]
    b = prealign.RotationSearchBatch(clouds, ps, pt, 0.3, 2.0, 0.2, max_rows_hint=640)
    b.run()
    R, t, s, rec = b.results()
    assert int(rec[3, 11]) == 1 and np.array_equal(R[3], np.eye(2)) and np.isinf(s[3])        # 40 points in a 0.2 m square: < 5 voxels
    assert int(rec[7, 11]) == 2                                                                # ~1 500 voxels > hint of 640
    features.VERBOSE = False
    for i in range(len(ps)):
        R1, t1, s1 = features.rotation_search(clouds[ps[i]], clouds[pt[i]], 0.3, 2.0, 0.2)
        assert np.array_equal(R[i], R1) and np.array_equal(t[i], t1), i
        assert (np.isinf(s[i]) and np.isinf(s1)) or abs(s[i] - s1) <= 1e-13 * max(1.0, s1), i


def test_run_icp_pair_batch_equals_per_pair_chain(uicp):
    """_run_icp_pair for a batch (rotation search -> R_init, t_init on the device -> ICP) against (a) the same two steps
    with the host in between — bit for bit — and (b) the oracle's rotation search + ICP."""
    from icpmi import batch, prealign, synth
    B = 96
    srcs, tgts = synth.loop_closure_batch(B, seed0=91000, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
    icp_cfg = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
    feat_cfg = dict(rotation_voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1)
    R, t, err, info = prealign.run_icp_pair_batch(srcs[0], tgts, icp_cfg, feat_cfg)
    R0, t0, _ = prealign.rotation_search_batch(srcs[0], tgts, 0.15, 1.5, 0.1)
    R2, t2, err2, info2 = batch.icp_batch(srcs[0], tgts, 1e-10, 150, 0.04, R0, t0, "point_to_line", 12)
    assert np.array_equal(R, R2) and np.array_equal(t, t2) and np.array_equal(err, err2) and np.array_equal(info["iters"], info2["iters"])
    at_limit = 0
    for i in range(B):                         # EVERY pair, the ones that circle until max_iterations included (8 of these 96)
        Ro0, to0, _ = oracle.rotation_search(srcs[0], tgts[i], 0.15, 1.5, 0.1)
        Ro, to, eo, io = oracle.icp(srcs[0], tgts[i], 1e-10, 150, 0.04, R_init=Ro0, t_init=to0, method="point_to_line", normal_k=12)
        assert int(info["iters"][i]) == io["iters"], i
        assert rot_err(R[i], t[i], Ro, to) < FRO_TOL, (i, io["iters"])
        assert abs(err[i] - eo) <= 1e-9 * max(1.0, eo) and (err[i] < 0.08) == (eo < 0.08), i     # the caller's decision: slam.py:582, config.yaml:73
        at_limit += io["iters"] == 150
    assert at_limit >= 4, at_limit
    assert (err < 0.05).mean() > 0.5, (err < 0.05).mean()     # pre-aligned candidates register; from 3 m / 20 degrees ICP alone does not


def test_more_than_31_neighbours_on_a_cloud_above_4096_rows(uicp):
    """normal_k > 31 on a rolling-submap-sized cloud (it used to be 'unsupported' above 4 096 rows): the wave-per-query
    search scans only the window of the sort order that can hold the k nearest.  Normals against the oracle, and a
    point_to_line registration of a scan onto that cloud."""
    from icpmi import batch, synth
    segs = synth.maze_segments()
    poses = synth.trajectory(40)
    big = uicp.voxel_downsample(np.vstack([synth.to_world(synth.scan(p, 500 + i, segs=segs), p) for i, p in enumerate(poses)]), 0.04)
    assert len(big) > 4096
    for k in (40, 64):
        got = uicp.estimate_normals_2d(big, k)
        ref = oracle.normals_2d(big, k)
        ok = np.abs(np.abs(np.sum(got * ref, axis=1)) - 1) < 1e-9
        assert ok.mean() >= 0.99, (k, ok.mean())
    cur = synth.scan(poses[20], 977, segs=segs)
    th = poses[20][2] + np.deg2rad(1.0)
    R0 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t0 = np.array(poses[20][:2]) + [0.05, -0.04]
    R, t, err, info = batch.icp_batch([cur], [big], 1e-10, 60, 0.04, R0, t0, "point_to_line", 40)
    Ro, to, eo, io = oracle.icp(cur, big, 1e-10, 60, 0.04, R_init=R0, t_init=t0, method="point_to_line", normal_k=40)
    assert int(info["iters"][0]) == io["iters"] and rot_err(R[0], t[0], Ro, to) < FRO_TOL


def test_nn_batches_beyond_65535_pairs(uicp):
    """icpmi_nn_batch / icpmi_nn_prepared_batch used to refuse more than 65 535 pairs (the grid's y extent): now slices."""
    from icpmi import batch
    rng = np.random.default_rng(4)
    C, B = 300, 70001
    clouds = [rng.normal(size=(rng.integers(3, 9), 2)) for _ in range(C)]
    cs = batch.CloudSet.from_numpy(clouds)
    ps = rng.integers(0, C, size=B).astype(np.int32)
    pt = rng.integers(0, C, size=B).astype(np.int32)
    d, i = batch.nn_set(cs, ps, pt)
    d2, i2 = batch.nn_set_sweep(cs, ps, pt)
    d, i, d2, i2 = d.cpu().numpy(), i.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()
    for b in (0, 1, 65534, 65535, 65536, 70000):
        n = len(clouds[ps[b]])
        do, io = oracle.nn(clouds[ps[b]], clouds[pt[b]])
        assert np.array_equal(i[b, :n], io) and np.array_equal(d[b, :n], do), b
        assert np.array_equal(i2[b, :n], io) and np.array_equal(d2[b, :n], do), b


def test_submap_translation_refinement_on_the_device_and_on_the_host(uicp):
    """slam.py:161-181 (the 80th-percentile inlier mean after the rotation sweeps) runs on the device for scans of at
    most 2 048 raw rows and on the host, from the kernels' distances and indices, beyond: both equal the oracle's
    restatement bit for bit — random poses, a scan with duplicated rows (ties in the percentile), very few inliers."""
    from icpmi import submap, synth
    submap.VERBOSE = False
    sub = load_golden("submap_build")["out"]
    segs = synth.maze_segments()
    poses = synth.trajectory(40)
    rng = np.random.default_rng(14)
    for trial in range(6):
        p = poses[int(rng.integers(5, 35))]
        scan = synth.scan(p, 4000 + trial, segs=segs)
        if trial == 3:
            scan = np.vstack([scan, scan[::3]])[:2048]                                   # duplicates
        if trial == 4:
            scan = np.vstack([scan, synth.scan(p, 4100, segs=segs)])                       # 4 096 raw rows: the host path
        th = p[2] + np.deg2rad(rng.uniform(-8, 8))
        pred = np.array([[np.cos(th), -np.sin(th), p[0] + rng.uniform(-0.3, 0.3)], [np.sin(th), np.cos(th), p[1] + rng.uniform(-0.3, 0.3)],
                         [0.0, 0.0, 1.0]])
        kw = dict(angle_range=20.0, angle_step=2.0, fine_step=0.5, voxel_size=0.3 if trial != 5 else 2.5)
        R, t = submap.submap_rotation_search(scan, sub, pred, **kw)
        Ro, to = oracle.submap_rotation_search(scan, sub, pred, **kw)
        assert np.array_equal(R, Ro) and np.array_equal(t, to), trial


@pytest.mark.parametrize("method", ["point_to_line", "point_to_point"])
def test_far_pairs_finish_on_the_far_continuation_with_the_same_bits(uicp, libopt, method):
    """Pairs that start metres from their target leave the first launch after iteration 1 and are finished by
    icp2_far_kernel (walks given up for the scan over block boxes).  Same matches, same arithmetic: every result equals the
    one of the plain path (option ICP2_FAR = 0), whether the threshold sends the far pairs only (default) or every pair."""
    from icpmi import batch, synth
    B = 48
    srcs, tgts = synth.loop_closure_batch(B, seed0=4100, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
    out = {}
    for far in ("0", None, "1e-12"):
        libopt.setenv("ICP2_FAR", far) if far else libopt.delenv("ICP2_FAR")
        R, t, err, info = batch.icp_batch(srcs[0], tgts, 1e-10, 60, 0.04, None, None, method, 12)
        out[far] = (R, t, err, info["iters"], info["status"])
    for far in (None, "1e-12"):
        for a, b in zip(out["0"], out[far]):
            assert np.array_equal(a, b), (far, method)
    for i in range(B):                         # every pair against the oracle, the ones stopped by max_iterations included
        Ro, to, eo, io = oracle.icp(srcs[0], tgts[i], 1e-10, 60, 0.04, method=method, normal_k=12)
        assert int(out[None][3][i]) == io["iters"], i
        assert rot_err(out[None][0][i], out[None][1][i], Ro, to) < FRO_TOL, (i, io["iters"])
        assert abs(out[None][2][i] - eo) <= 1e-9 * max(1.0, eo), i


def test_far_continuation_with_rejection_and_mixed_sizes(uicp, libopt):
    """The far continuation under max_corr_dist (rows drop out of the solve, a pair may stop on too few inliers) and in a
    batch that mixes clouds it can hold (<= 2 048 rows) with ones it cannot (a 4 096-beam scan: left to the launch that
    started it): every result equals the plain path's."""
    from icpmi import batch, synth
    B = 24
    srcs, tgts = synth.loop_closure_batch(B, seed0=4700, shared_source=False, max_offset=2.5, max_yaw_deg=15.0)
    base = (0.5, -0.3, 0.1)
    for i in (3, 11):                                                   # two pairs of 4 096 beams: ~2 800 rows after the filter
        srcs[i] = synth.scan(base, 99000 + i, n_beams=4096)
        tgts[i] = synth.scan((base[0] + 1.5, base[1] - 1.0, base[2] + 0.2), 99100 + i, n_beams=4096)
    for mcd in (None, 0.8):
        out = {}
        for far in ("0", "1e-12", None):
            libopt.setenv("ICP2_FAR", far) if far else libopt.delenv("ICP2_FAR")
            R, t, err, info = batch.icp_batch(srcs, tgts, 1e-10, 50, 0.04, None, None, "point_to_line", 12, max_corr_dist=mcd)
            out[far] = (R, t, err, info["iters"], info["status"])
        for far in ("1e-12", None):
            for a, b in zip(out["0"], out[far]):
                assert np.array_equal(a, b, equal_nan=True), (mcd, far)


def test_far_continuation_beside_the_two_stages(uicp, libopt):
    """A batch large enough for the two-stage run (>= 1 024 pairs) with far pairs in it: first stage, second stage and far
    continuation together give the bits of the single plain launch."""
    from icpmi import batch, synth
    B = 1056
    srcs, tgts = synth.loop_closure_batch(B, seed0=4300, shared_source=True, max_offset=2.0, max_yaw_deg=12.0)
    srcs, tgts = [c[::4] for c in srcs], [c[::4] for c in tgts]            # 512 beams: a quick batch
    out = {}
    for far, stages in (("0", "1"), (None, None), ("0.04", None)):
        libopt.setenv("ICP2_FAR", far) if far else libopt.delenv("ICP2_FAR")
        libopt.setenv("ICP2_STAGES", stages) if stages else libopt.delenv("ICP2_STAGES")
        R, t, err, info = batch.icp_batch(srcs[0], tgts, 1e-10, 40, 0.04, None, None, "point_to_line", 12)
        out[(far, stages)] = (R, t, err, info["iters"], info["status"])
    ref = out[("0", "1")]
    for k in ((None, None), ("0.04", None)):
        for a, b in zip(ref, out[k]):
            assert np.array_equal(a, b), k


def test_run_icp_pair_batch_with_pairs_beyond_the_capacity_hint(uicp):
    """Pairs whose filtered clouds exceed the capacity hint of the batched search are searched and registered one by
    one through the single-pair entries: the batch's results equal the per-pair chain for every pair."""
    from icpmi import batch, prealign, synth
    from utilities import features
    srcs, tgts = synth.loop_closure_batch(6, seed0=300, shared_source=True)
    kw = dict(error_threshold=1e-10, max_iterations=60, voxel_size=0.04, method="point_to_line", normal_k=12)
    b = prealign.RunIcpPairBatch([srcs[0]] + tgts, np.zeros(6, dtype=np.int32), np.arange(1, 7, dtype=np.int32),
                                 rotation_voxel_size=0.05, angle_step_coarse=4.0, angle_step_fine=0.5, max_rows_hint=256, **kw)
    b.run()
    R, t, err, info = b.unpack()
    rec = b.search.records.cpu().numpy()
    assert (rec[:6, 11] == 2).all()                                     # ~1 300 filtered rows > 256: nobody was searched on chip
    features.VERBOSE = False
    for i in range(6):
        R0, t0, _ = features.rotation_search(srcs[0], tgts[i], 0.05, 4.0, 0.5)
        Ri, ti, ei, ii = batch.icp_pair(srcs[0], tgts[i], R_init=R0, t_init=t0, **kw)
        assert np.array_equal(R[i], Ri[0]) and np.array_equal(t[i], ti[0]) and err[i] == ei[0] and info["iters"][i] == ii["iters"][0], i


def test_library_loaded_before_torch_then_smoke_in_one_process():
    """__graft_entry__.build() loads libicpmi.so before anything has imported torch; smoke() in the same process must
    still run (the loader imports torch first, so that its HIP runtime is the only one in the process).  The child
    does what build() does after compiling — load the library first — without running make."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; import icpmi; icpmi.lib(); g.smoke()"], cwd=repo,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ── the regimes the bench lives on, against fixtures the REFERENCE produced (round 4) ──
def test_limit_cycle_pairs_equal_the_reference_after_150_iterations(uicp):
    """tests/golden/icp_limit.npz: config-2 pairs of the bench batch for which the reference prints "max iterations
    reached: iter=150" (icp.py:222): drop-in ICP and the batched path reproduce its R, t, error after exactly those."""
    from icpmi import batch
    from test_oracle_golden import limit_cycle_pairs
    cases = list(limit_cycle_pairs())
    for i, s, t, z in cases:
        R, tt, err = uicp.ICP(s, t, 1e-10, 150, 0.04, method="point_to_line", normal_k=12)
        assert uicp.last_icp_info["iterations"] == 150 and uicp.last_icp_info["status"] == 2, i
        assert rot_err(R, tt, z[f"p{i}__R"], z[f"p{i}__t"]) < FRO_TOL, i
        assert abs(err - float(z[f"p{i}__err"])) < 1e-12, i
    for ex in (False, True):
        R, tt, err, info = batch.icp_batch([c[1] for c in cases], [c[2] for c in cases], 1e-10, 150, 0.04, None, None,
                                           "point_to_line", 12, force_exhaustive=ex)
        for q, (i, s, t, z) in enumerate(cases):
            assert int(info["iters"][q]) == 150 and int(info["status"][q]) == 2, (i, ex)
            assert rot_err(R[q], tt[q], z[f"p{i}__R"], z[f"p{i}__t"]) < FRO_TOL, (i, ex)


def test_run_icp_pair_equals_the_reference(uicp):
    """tests/golden/run_icp_pair.npz: the reference's _run_icp_pair (slam.py:53-98: rotation_search, then ICP from its
    result) on 3 m / 20 degree candidates — registered, converged to a wrong pose, 150 iterations with a small and a large
    error.  Batched path (icpmi.prealign.run_icp_pair_batch) and the drop-in functions called as slam.py calls them."""
    from icpmi import prealign
    from utilities import features
    from test_oracle_golden import run_icp_pair_cases
    cases = list(run_icp_pair_cases())
    icp_cfg, feat_cfg = cases[0][4], cases[0][5]
    R, t, err, info = prealign.run_icp_pair_batch(cases[0][1], [c[2] for c in cases], icp_cfg, feat_cfg)
    features.VERBOSE = False
    for q, (i, s, tg, z, _, _) in enumerate(cases):
        assert int(info["iters"][q]) == int(z[f"p{i}__iters"]), i
        assert rot_err(R[q], t[q], z[f"p{i}__R"], z[f"p{i}__t"]) < FRO_TOL, i
        assert abs(err[q] - float(z[f"p{i}__err"])) <= 1e-9 * max(1.0, err[q]), i
        assert (err[q] < 0.08) == (float(z[f"p{i}__err"]) < 0.08), i                   # slam.py:582 with config.yaml:73
        R0, t0, _ = features.rotation_search(s, tg, voxel_size=feat_cfg["rotation_voxel_size"],
                                             angle_step_coarse=feat_cfg["angle_step_coarse"], angle_step_fine=feat_cfg["angle_step_fine"])
        R1, t1, e1 = uicp.ICP(s, tg, R_init=R0, t_init=t0, **icp_cfg)
        assert rot_err(R1, t1, z[f"p{i}__R"], z[f"p{i}__t"]) < FRO_TOL, i
        assert uicp.last_icp_info["iterations"] == int(z[f"p{i}__iters"]), i


@pytest.mark.parametrize("method", ["point_to_line", "point_to_point"])
def test_far_pairs_that_never_settle_fast_exhaustive_oracle(uicp, method):
    """the TRANSFORM of every pair — the ones stopped by max_iterations included — fused path (sorted
    sweeps, budgets, float32 filter, far continuation) against the exhaustive kernel and the oracle, on candidates that
    start up to 3 m / 20 degrees off with no pre-alignment (most never register, many circle) and with it."""
    from icpmi import batch, prealign, synth
    B = 128
    srcs, tgts = synth.loop_closure_batch(B, seed0=52000, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
    R0, t0, _ = prealign.rotation_search_batch(srcs[0], tgts, 0.15, 1.5, 0.1)
    for pre, maxit in ((True, 150), (False, 40)):
        kw = dict(R_init=R0, t_init=t0) if pre else dict(R_init=None, t_init=None)
        Rf, tf, ef, inf_ = batch.icp_batch(srcs[0], tgts, 1e-10, maxit, 0.04, kw["R_init"], kw["t_init"], method, 12)
        Rx, tx, ex, inx = batch.icp_batch(srcs[0], tgts, 1e-10, maxit, 0.04, kw["R_init"], kw["t_init"], method, 12, force_exhaustive=True)
        assert np.array_equal(inf_["iters"], inx["iters"]) and np.array_equal(inf_["status"], inx["status"])
        at_limit = 0
        for i in range(B):
            assert rot_err(Rf[i], tf[i], Rx[i], tx[i]) < FRO_TOL, (pre, i)
            okw = dict(R_init=R0[i], t_init=t0[i]) if pre else {}
            Ro, to, eo, io = oracle.icp(srcs[0], tgts[i], 1e-10, maxit, 0.04, method=method, normal_k=12, **okw)
            assert int(inf_["iters"][i]) == io["iters"], (pre, i)
            assert rot_err(Rf[i], tf[i], Ro, to) < FRO_TOL, (pre, i, io["iters"])
            assert abs(ef[i] - eo) <= 1e-9 * max(1.0, eo), (pre, i)
            at_limit += io["iters"] == maxit
        assert at_limit >= (3 if method == "point_to_line" else 0), (pre, at_limit)


def test_run_icp_pair_batch_edges_of_the_angle_grids(uicp):
    """(1) more angles per sweep than the batched kernel tabulates (a coarse step of 0.3 degrees: 1 200 angles)
    takes the per-pair route instead of failing; (2) an empty fine grid (features.py:227-231: np.argmin of an empty
    sequence) raises the reference's ValueError from unpack() too, not only from results()."""
    from icpmi import prealign, synth
    from utilities import features
    features.VERBOSE = False
    srcs, tgts = synth.loop_closure_batch(3, seed0=640, shared_source=True, max_offset=1.0, max_yaw_deg=10.0)
    src, tgts = srcs[0][::2], [t[::2] for t in tgts]
    icp_cfg = dict(error_threshold=1e-10, max_iterations=40, voxel_size=0.05, method="point_to_line", normal_k=10)
    feat_cfg = dict(rotation_voxel_size=0.2, angle_step_coarse=0.3, angle_step_fine=0.05)
    R, t, err, info = prealign.run_icp_pair_batch(src, tgts, icp_cfg, feat_cfg)
    for i in range(3):
        R0, t0, _ = features.rotation_search(src, tgts[i], 0.2, 0.3, 0.05)
        Ro0, to0, _ = oracle.rotation_search(src, tgts[i], 0.2, 0.3, 0.05)
        assert np.array_equal(R0, Ro0) and np.array_equal(t0, to0), i
        Ro, to, eo, io = oracle.icp(src, tgts[i], 1e-10, 40, 0.05, R_init=Ro0, t_init=to0, method="point_to_line", normal_k=10)
        assert int(info["iters"][i]) == io["iters"] and rot_err(R[i], t[i], Ro, to) < FRO_TOL, i
    Rs, ts, ss = prealign.rotation_search_batch(src, tgts, 0.2, 0.3, 0.05)
    assert np.array_equal(Rs[1], oracle.rotation_search(src, tgts[1], 0.2, 0.3, 0.05)[0])
    with pytest.raises(ValueError, match="empty sequence"):
        prealign.run_icp_pair_batch(src, tgts, icp_cfg, dict(rotation_voxel_size=0.2, angle_step_coarse=2.0, angle_step_fine=-0.2))
    with pytest.raises(ValueError, match="empty sequence"):
        prealign.rotation_search_batch(src, tgts, 0.2, 2.0, -0.2)


def test_rotation_search_batch_pair_with_an_unlisted_target(uicp):
    """a pair whose target cloud is not among tgt_ids has no search order: it reports status 2 (and the host
    searches it through the single-pair entry) instead of walking whatever the workspace held."""
    import ctypes as C
    import torch
    from icpmi import _lib, prealign, synth
    from icpmi.batch import _ptr, _stream
    a, b = synth.config2_pair(4)
    srch = prealign.RotationSearchBatch([a, b, a[::2]], [0, 0], [1, 2], 0.15, 1.5, 0.1)
    srch.ws.fill_(0x11)                                       # a "direction" of 0x11111111 would pass for garbage otherwise
    srch.tgt_ids = torch.tensor([1], dtype=torch.int32, device=srch.ws.device)          # cloud 2 left out
    srch.run()
    rec = srch.records.cpu().numpy()
    assert int(rec[0, 11]) == 0 and int(rec[1, 11]) == 2


@pytest.mark.parametrize("method", ["point_to_point", "point_to_line"])
def test_packed_walks_on_exact_ties_and_lattices(uicp, method):
    """The packed float32 walks (sweep.hpp, round 4) list at most K candidates; exact ties — a source row on the bisector of
    two lattice points, at the centre of four, near-duplicates — put more candidates within the filter's resolution than
    the list holds and send the lane to the exact walk, where the (distance, row) rule decides.  Fused path = exhaustive
    kernel = oracle, iterations included; target rows shuffled so that the row rule is not the sort order."""
    from icpmi import batch
    gx, gy = np.meshgrid(np.arange(-16, 17) * 0.25, np.arange(-12, 13) * 0.25)
    lat = np.stack([gx.ravel(), gy.ravel()], 1)                       # exactly representable lattice
    rng = np.random.default_rng(5)
    tgt = lat[rng.permutation(len(lat))]
    inner = lat[(np.abs(lat[:, 0]) < 3.5) & (np.abs(lat[:, 1]) < 2.5)]
    th = np.deg2rad(0.7)
    Rs = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    near = rng.uniform(-3, 3, size=(300, 2))
    near_dup = np.vstack([near, near + 1e-6, near - 2e-6])                       # three points within microns of each other (distinct voxels at 2^-21):
                                                                                 # apart in float32, together in the 13 bits a packed word keeps
    cases = [("half", inner + [0.125, 0.0], tgt, 0.01), ("quarter", inner + [0.125, 0.125], tgt, 0.01),
             ("rotated", (inner + [0.125, 0.0]) @ Rs.T, tgt, 0.01), ("near_dup", near[:200] + [0.01, -0.02], near_dup, 2.0 ** -21)]
    for name, s, t, vox in cases:
        Ro, to, eo, io = oracle.icp(s, t, 1e-12, 25, vox, method=method, normal_k=8)
        for ex in (False, True):
            R, tt, err, info = batch.icp_batch([s], [t], 1e-12, 25, vox, None, None, method, 8, force_exhaustive=ex)
            assert int(info["iters"][0]) == io["iters"] and int(info["status"][0]) == io["status"], (name, ex, info, io)
            assert rot_err(R[0], tt[0], Ro, to) < FRO_TOL, (name, ex)
        # a batch of the same pair 80 times: the packed walk is taken when >= 16 lanes of a wave search — they all do here
        R, tt, err, info = batch.icp_batch([s] * 80, [t] * 80, 1e-12, 25, vox, None, None, method, 8)
        assert (info["iters"] == io["iters"]).all() and max(rot_err(R[i], tt[i], Ro, to) for i in range(80)) < FRO_TOL, name
