"""Pin the CPU oracle (oracle/icp_oracle.c) to golden vectors captured by running
the reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from conftest import load_golden, rot_err


def test_voxel_downsample_bit_exact():
    z = load_golden("voxel")
    for name in z["names"]:
        out = oracle.voxel_downsample(z[f"{name}__in"], float(z[f"{name}__voxel"]))
        ref = z[f"{name}__out"]
        assert out.shape == ref.shape, name
        assert np.array_equal(out, ref), name      # same keys, same order, same sums


@pytest.mark.parametrize("case", ["vox", "raw", "submap"])
@pytest.mark.parametrize("kd", [False, True])
def test_nn_index_and_distance_exact(case, kd):
    z = load_golden("nn")
    d, i = oracle.nn(z[f"{case}__src"], z[f"{case}__tgt"], kdtree=kd)
    assert np.array_equal(i, z[f"{case}__idx"])
    assert np.array_equal(d, z[f"{case}__dist"])   # sqrt(dx*dx + dy*dy), no FMA: bitwise


@pytest.mark.parametrize("case", ["tgt_k12", "tgt_k5", "five_k10", "collinear_k8", "diag_k6"])
def test_normals_up_to_sign(case):
    z = load_golden("normals")
    n = oracle.normals_2d(z[f"{case}__in"], int(z[f"{case}__k"]))
    ref = z[f"{case}__out"]
    dots = np.abs(np.sum(n * ref, axis=1))
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)
    # eigenvectors of nearly isotropic neighbourhoods are ill conditioned: allow a few
    assert np.quantile(dots, 0.01) > 1 - 1e-9
    assert dots.min() > 1 - 1e-6


def test_p2l_solve():
    z = load_golden("p2l_solve")
    R, t = oracle.p2l_solve_2d(z["src"], z["tgt"], z["normals"], z["idx"])
    assert rot_err(R, t, z["R"], z["t"]) < 1e-11
    # exactly singular system -> identity (icp.py:105-108)
    R, t = oracle.p2l_solve_2d(z["src"], z["tgt"], z["zero_normals"], z["idx"])
    assert np.array_equal(R, z["R_zero"]) and np.array_equal(t, z["t_zero"])
    assert np.array_equal(R, np.eye(2))


ICP_CASES = {
    # name: (src, tgt, kwargs)
    "p2l": ("scan_a", "scan_b", dict(method="point_to_line", normal_k=12)),
    "p2p": ("scan_a", "scan_b", dict(method="point_to_point")),
    "p2l_fine": ("scan_a", "scan_b", dict(method="point_to_line", normal_k=12, voxel_size=0.005)),
    "p2p_fine": ("scan_a", "scan_b", dict(method="point_to_point", voxel_size=0.005)),
    "p2l_init": ("scan_a", "scan_b", dict(method="point_to_line", normal_k=12, init="init")),
    "p2p_Ronly": ("scan_a", "scan_b", dict(method="point_to_point", init="Ronly")),
    "p2p_corr": ("scan_a", "scan_b", dict(method="point_to_point", max_corr_dist=0.5)),
    "p2l_corr": ("scan_a", "scan_b", dict(method="point_to_line", normal_k=12, max_corr_dist=0.3)),
    "p2p_maxit": ("scan_a", "scan_b", dict(method="point_to_point", max_iterations=5)),
    "p2l_maxit1": ("scan_a", "scan_b", dict(method="point_to_line", normal_k=12, max_iterations=1)),
    "p2p_breakN": ("break_src", "break_tgt", dict(method="point_to_point", voxel_size=0.005, max_corr_dist=0.055)),
    "teapot": ("teapot_moved", "teapot", dict(method="point_to_point", error_threshold=1e-12,
                                               max_iterations=300, voxel_size=0.005)),
    "teapot_p2l": ("teapot_moved", "teapot", dict(method="point_to_line", error_threshold=1e-12,
                                                   max_iterations=300, voxel_size=0.005)),
    "submap": ("sub_cur", "sub_map", dict(method="point_to_point", max_corr_dist=1.5, init="sub")),
}


def icp_case_args(z, name):
    src, tgt, kw = ICP_CASES[name]
    kw = dict(kw)
    init = kw.pop("init", None)
    args = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04)
    args.update(kw)
    if init == "init":
        args.update(R_init=z["init_R"], t_init=z["init_t"])
    elif init == "Ronly":
        args.update(R_init=z["init_R"])
    elif init == "sub":
        args.update(R_init=z["sub_R0"], t_init=z["sub_t0"])
    return z[src], z[tgt], args


@pytest.mark.parametrize("name", list(ICP_CASES))
@pytest.mark.parametrize("kd", [True, False])
def test_icp_matches_reference(name, kd):
    z = load_golden("icp")
    if name == "submap" and not kd:
        pytest.skip("brute-force search on the submap is slow; covered by the k-d tree run")
    s, t, args = icp_case_args(z, name)
    R, tt, err, info = oracle.icp(s, t, kdtree=kd, **args)
    assert rot_err(R, tt, z[f"{name}__R"], z[f"{name}__t"]) < 1e-9
    assert abs(err - float(z[f"{name}__err"])) <= 1e-12 * max(1.0, abs(err))
    if int(z[f"{name}__conv"]):
        assert info["status"] == oracle.CONVERGED
        assert info["iters"] == int(z[f"{name}__iters"])
    else:
        assert info["status"] in (oracle.MAXITER, oracle.FEW_INLIERS)


def test_icp_break_semantics():
    z = load_golden("icp")
    a, b = z["scan_a"], z["scan_b"]
    R, t, err, info = oracle.icp(a, b + np.array([30.0, 0.0]), 1e-10, 150, 0.04,
                                 method="point_to_point", max_corr_dist=0.05)
    assert info["status"] == oracle.FEW_INLIERS and info["iters"] == 0
    assert np.isinf(err) and np.isinf(float(z["p2p_break0__err"]))
    assert np.array_equal(R, z["p2p_break0__R"]) and np.array_equal(t, z["p2p_break0__t"])
    s, tg, args = icp_case_args(z, "p2p_breakN")
    R, t, err, info = oracle.icp(s, tg, **args)
    assert info["status"] == oracle.FEW_INLIERS and info["iters"] == 1


def test_teapot_known_answer():
    """demos/teapot_icp_demo.py:38-65: ICP must undo Ry(25 deg), t=[0.25,0.05,0]."""
    z = load_golden("icp")
    R, t, err, info = oracle.icp(z["teapot_moved"], z["teapot"], 1e-12, 300, 0.005)
    Ry, tr = z["teapot_Ry"], z["teapot_tr"]
    assert np.allclose(R, Ry.T, atol=1e-9)
    assert np.allclose(t, -Ry.T @ tr, atol=1e-9)
    assert info["iters"] == 15


def test_bresenham_cells_exact():
    z = load_golden("bresenham")
    segs, cells, off = z["segs"], z["cells"], z["off"]
    for k, s in enumerate(segs):
        c = oracle.bresenham(*[int(v) for v in s])
        assert np.array_equal(c, cells[off[k]:off[k + 1]]), s


def test_world_to_grid():
    z = load_golden("grid")
    b = z["small_bounds"]
    assert np.array_equal(oracle.world_to_grid(z["w2g_in"], b[0], 0.05), z["w2g_ix"])
    assert np.array_equal(oracle.world_to_grid(z["w2g_in"][::-1], b[2], 0.05), z["w2g_iy"])


def _grid_shape(b, res):
    return int(np.ceil((b[3] - b[2]) / res)), int(np.ceil((b[1] - b[0]) / res))


def test_update_scan_small_grid_bit_exact():
    z = load_golden("grid")
    b = z["small_bounds"]
    l_hit, l_miss = z["small_l"]
    g = np.zeros(_grid_shape(b, 0.05), dtype=np.float32)
    for i in range(40):
        oracle.grid_update_scan(g, b[0], b[2], 0.05, z["small_origins"][i], z["small_hits"][i],
                                l_hit, l_miss, -8.0, 8.0)
        if i + 1 in (1, 3, 40):
            assert np.array_equal(g, z[f"small_after{i + 1}"]), i
    assert g.min() == -8.0            # saturation reached: the clip path is exercised


def test_update_scan_edges_and_defaults():
    z = load_golden("grid")
    b = z["small_bounds"]
    l_hit, l_miss = z["small_l"]
    g = np.zeros(_grid_shape(b, 0.05), dtype=np.float32)
    n = oracle.grid_update_scan(g, b[0], b[2], 0.05, z["edge_origin"], z["edge_hits"], l_hit, l_miss, -8.0, 8.0)
    assert n > 0
    assert oracle.grid_update_scan(g, b[0], b[2], 0.05, z["edge_origin"], np.empty((0, 2)), l_hit, l_miss, -8, 8) == 0
    assert np.array_equal(g, z["edge_after"])
    lh, lm = z["default_l"]
    g = np.zeros(_grid_shape(b, 0.1), dtype=np.float32)
    for i in range(3):
        oracle.grid_update_scan(g, b[0], b[2], 0.1, z["small_origins"][i], z["small_hits"][i], lh, lm, -5.0, 5.0)
    assert np.array_equal(g, z["default_after3"])


def test_update_scan_config4_sparse():
    z = load_golden("grid")
    b = z["cfg4_bounds"]
    l_hit, l_miss = z["small_l"]
    ny, nx = _grid_shape(b, 0.05)
    assert (ny, nx) == tuple(z["cfg4_shape"])
    g = np.zeros((ny, nx), dtype=np.float32)
    n = oracle.grid_update_scan(g, b[0], b[2], 0.05, z["cfg4_origin"], z["cfg4_hits"], l_hit, l_miss, -8.0, 8.0)
    nz = np.flatnonzero(g.ravel())
    assert np.array_equal(nz, z["cfg4_nz_idx"])
    assert np.array_equal(g.ravel()[nz], z["cfg4_nz_val"])
    assert n > 200000


def test_build_submap_voxel():
    """slam.py:103-108 is vstack + voxel_downsample on ~82k points."""
    from icpmi import synth
    z = load_golden("submap_build")
    segs = synth.maze_segments()
    poses = synth.trajectory(40)
    allpts = np.vstack([synth.to_world(synth.scan(p, 500 + i, segs=segs), p) for i, p in enumerate(poses)])
    assert len(allpts) == int(z["n_in"]) and np.array_equal(allpts[:8], z["in_head"])
    out = oracle.voxel_downsample(allpts, 0.04)
    assert np.array_equal(out, z["out"])


@pytest.mark.parametrize("case", ["cfg", "default", "big_rotation", "tiny"])
def test_rotation_search_matches_reference(case):
    """features.py:165-242: same angle grid, same arg-min, so R and t are bit-equal; score to rounding."""
    z = load_golden("rotation_search")
    kw = z[f"{case}__kw"]
    R, t, s = oracle.rotation_search(z[f"{case}__src"], z[f"{case}__tgt"], kw[0], kw[1], kw[2])
    assert np.array_equal(R, z[f"{case}__R"]) and np.array_equal(t, z[f"{case}__t"])
    ref = float(z[f"{case}__score"])
    assert (np.isinf(s) and np.isinf(ref)) or abs(s - ref) < 1e-14


def _submap_rotation_inputs():
    z = load_golden("submap_rotation")
    sub = load_golden("submap_build")["out"]                 # the 40-scan submap the golden was computed on
    assert len(sub) == int(z["submap_n"]) and np.array_equal(sub.sum(axis=0), z["submap_sum"])
    return z, sub


@pytest.mark.parametrize("case", ["cfg", "default", "imu_narrow", "far_off"])
def test_submap_rotation_search_matches_reference(case):
    """slam.py:111-183: same grids and arg-min -> R bit-equal; the refined translation is a NumPy mean over the
    same matches -> bit-equal too."""
    z, sub = _submap_rotation_inputs()
    kw = z[f"{case}__kw"]
    R, t = oracle.submap_rotation_search(z["source"], sub, z[f"{case}__pred"], kw[0], kw[1], kw[2], kw[3])
    assert np.array_equal(R, z[f"{case}__R"])
    assert np.array_equal(t, z[f"{case}__t"])


def test_submap_rotation_search_small_input_and_attempt():
    z, sub = _submap_rotation_inputs()
    pred = z["cfg__pred"]
    R, t = oracle.submap_rotation_search(z["source"][:3], sub, pred)
    assert np.array_equal(R, z["tiny__R"]) and np.array_equal(t, z["tiny__t"])
    cfg = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04)
    for k in ("attempt", "attempt_imu"):
        imu = None if np.isnan(z[f"{k}__imu"]) else float(z[f"{k}__imu"])
        r, tt, err, info = oracle.attempt_submap_icp(z["source"], sub, pred, imu, 3.0, 60.0, 0.8, 0.1, 0.2, cfg, 1.5)
        assert info["iters"] == int(z[f"{k}__iters"])
        assert np.linalg.norm(r - z[f"{k}__R"]) + np.linalg.norm(tt - z[f"{k}__t"]) < 1e-9
        assert abs(err - float(z[f"{k}__err"])) < 1e-12


# ── pose graph (utilities/pose_graph.py) ──────────────────
def pose_graph_case(z, name):
    ij = z[f"{name}__ij"]
    kw = z[f"{name}__kw"]
    return (z[f"{name}__nodes"], ij[:, 0], ij[:, 1], z[f"{name}__z"], z[f"{name}__omega"],
            dict(n_iterations=int(kw[0]), fix_node=int(kw[1]), convergence_eps=float(kw[2])))


def expected_pose_graph_outcome(printed):
    """(status, iterations run) from the line the reference printed."""
    import re
    from oracle import pose_graph as opg
    m = re.search(r"converged: iter=(\d+)", printed)
    if m:
        return opg.OK, int(m.group(1)) + 1
    m = re.search(r"max iterations: iter=(\d+)", printed)
    if m:
        return opg.MAX_ITERATIONS, int(m.group(1))
    m = re.search(r"singular H at iter (\d+)", printed)
    if m:
        return opg.SINGULAR, int(m.group(1))
    return opg.NOTHING_TO_DO, 0


def test_pose_graph_oracle_matches_reference():
    from oracle import pose_graph as opg
    z = load_golden("pose_graph")
    for name in z["names"]:
        nodes, ei, ej, zz, om, kw = pose_graph_case(z, name)
        assert abs(opg.total_error(nodes, ei, ej, zz, om) - z[f"{name}__err"][0]) <= 1e-12 * max(1.0, z[f"{name}__err"][0])
        out, iters, status, step = opg.optimize(nodes, ei, ej, zz, om, **kw)
        assert (status, iters) == expected_pose_graph_outcome(str(z[f"{name}__printed"])), name
        assert np.abs(out - z[f"{name}__out"]).max() < 1e-10, (name, np.abs(out - z[f"{name}__out"]).max())
        assert abs(opg.total_error(out, ei, ej, zz, om) - z[f"{name}__err"][1]) <= 1e-9 * max(1.0, z[f"{name}__err"][1])
    assert np.array_equal(opg.wrap(z["wrap_in"]), z["wrap_out"])


# ── the two regimes the bench lives on, as the REFERENCE runs them (golden: make_golden.py gold_icp_limit / gold_run_icp_pair) ──
def limit_cycle_pairs():
    """(id, source, target, golden dict) of the config-2 loop-closure pairs that run all 150 iterations in the reference."""
    from icpmi import synth
    z = load_golden("icp_limit")
    srcs, tgts = synth.loop_closure_batch(int(z["n_pairs"]), seed0=int(z["seed0"]))
    for i in z["ids"]:
        i = int(i)
        assert np.array_equal(srcs[i].sum(axis=0), z[f"p{i}__src_sum"]) and np.array_equal(tgts[i].sum(axis=0), z[f"p{i}__tgt_sum"])
        yield i, srcs[i], tgts[i], z


def run_icp_pair_cases():
    """(id, source, target, golden dict, icp_cfg, feat_cfg) of the 3 m / 20 degree candidates whose reference
    _run_icp_pair result (slam.py:53-98) the fixture holds: registered, converged wrong, 150 iterations."""
    from icpmi import synth
    z = load_golden("run_icp_pair")
    srcs, tgts = synth.loop_closure_batch(int(z["n_pairs"]), seed0=int(z["seed0"]), shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
    assert np.array_equal(srcs[0].sum(axis=0), z["src_sum"])
    e, m, v, k = z["icp_cfg"]
    icp_cfg = dict(error_threshold=float(e), max_iterations=int(m), voxel_size=float(v), method="point_to_line", normal_k=int(k))
    rv, ac, af = z["feat_cfg"]
    feat_cfg = dict(rotation_voxel_size=float(rv), angle_step_coarse=float(ac), angle_step_fine=float(af))
    for i in z["ids"]:
        i = int(i)
        assert np.array_equal(tgts[i].sum(axis=0), z[f"p{i}__tgt_sum"])
        yield i, srcs[0], tgts[i], z, icp_cfg, feat_cfg


def test_icp_pairs_that_circle_until_max_iterations():
    """150 iterations of a limit cycle of the point-to-line step (65 % of the bench's counted iterations): the
    reference's transform after exactly those, icp.py:177-223."""
    n = 0
    for i, s, t, z in limit_cycle_pairs():
        R, tt, err, info = oracle.icp(s, t, 1e-10, 150, 0.04, method="point_to_line", normal_k=12)
        assert info["iters"] == 150 and info["status"] == oracle.MAXITER, i
        assert "max iterations reached: iter=150" in str(z[f"p{i}__printed"])
        assert rot_err(R, tt, z[f"p{i}__R"], z[f"p{i}__t"]) < 1e-9, i
        assert abs(err - float(z[f"p{i}__err"])) < 1e-12, i
        n += 1
    assert n >= 3


def test_run_icp_pair_rotation_search_then_icp():
    """slam.py:53-98 with alignment_method "rotation_search": oracle.rotation_search -> oracle.icp from its result."""
    seen = set()
    for i, s, t, z, icp_cfg, feat_cfg in run_icp_pair_cases():
        R0, t0, _ = oracle.rotation_search(s, t, feat_cfg["rotation_voxel_size"], feat_cfg["angle_step_coarse"], feat_cfg["angle_step_fine"])
        R, tt, err, info = oracle.icp(s, t, icp_cfg["error_threshold"], icp_cfg["max_iterations"], icp_cfg["voxel_size"],
                                      R_init=R0, t_init=t0, method="point_to_line", normal_k=icp_cfg["normal_k"])
        assert info["iters"] == int(z[f"p{i}__iters"]), i
        assert rot_err(R, tt, z[f"p{i}__R"], z[f"p{i}__t"]) < 1e-9, i
        assert abs(err - float(z[f"p{i}__err"])) <= 1e-9 * max(1.0, err), i
        seen.add((info["iters"] == 150, err < 0.08))
    assert seen == {(False, True), (False, False), (True, True), (True, False)}       # every regime is in the fixture
