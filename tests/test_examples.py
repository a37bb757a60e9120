"""The SLAM-loop harness of examples/ (SURVEY §8f rank 3): wire format on CPU, the loop itself on the GPU."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import REPO


def _load():
    spec = importlib.util.spec_from_file_location("slam_loop", os.path.join(REPO, "examples", "slam_loop.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_lidar_wire_format_round_trip(tmp_path):
    """services/lidar_service.py:5-19: `timestamp;x;y;z;...` lines, all-zero triples dropped, z slice kept."""
    import torch
    if not torch.cuda.is_available():
        pytest.importorskip("icpmi")       # the harness imports the drop-in modules; they import without a GPU
    m = _load()
    rng = np.random.default_rng(0)
    scans = [rng.normal(size=(n, 2)) for n in (5, 1, 17)]
    scans[0][2] = 0.0                                   # a (0, 0) return survives: z = 1 keeps the triple non-zero
    path = tmp_path / "lidar.csv"
    m.write_lidar_log(str(path), scans)
    back = list(m.read_lidar_log(str(path)))
    assert [t for t, _ in back] == [1000000, 1100000, 1200000]
    for (_, got), ref in zip(back, scans):
        assert np.array_equal(got, ref)                 # repr() round-trips float64 exactly
    with open(path, "a") as f:
        f.write("1300000;0;0;0;1.5;2.5;0.1;3.0;4.0;1.0\n")   # zero triple dropped, z = 0.1 outside the slice
    assert np.array_equal(list(m.read_lidar_log(str(path)))[-1][1], [[3.0, 4.0]])


@pytest.mark.gpu
def test_slam_loop_tracks_the_drive(tmp_path):
    m = _load()
    out = m.run(36, log_path=str(tmp_path / "drive.csv"), verbose=False)
    assert out["drift"].max() < 0.25, out["drift"].max()          # metres, over ~6.5 m driven with 1 cm range noise
    assert out["occupied"] > 1000 and out["free"] > 50000
    assert out["closures"] == [] or all(np.isfinite(c[2]) for c in out["closures"])


def test_imu_wire_format_round_trip(tmp_path):
    """services/imu_service.py:1-38: `timestamp;qx;qy;qz;qw` lines -> yaw about z, nearest reading by timestamp."""
    m = _load()
    yaws = [0.0, 0.1, 0.25, -0.4, 3.0, -3.1]
    path = tmp_path / "imu.csv"
    m.write_imu_log(str(path), yaws, per_scan=2, noise=0.0)
    log = m.ImuLog(str(path))
    assert len(log.ts) == 12 and log.ts[0] == 1000000 and log.ts[2] == 1100000
    for i, y in enumerate(yaws):
        assert abs(log.yaw_at(1000000 + 100000 * i) - y) < 1e-12
    assert abs(log.yaw_at(1000000 + 100000 * 2 + 20000) - 0.25) < 1e-12            # nearest, not interpolated
    assert abs(log.delta_yaw(1000000 + 400000, 1000000 + 500000) - (2 * np.pi - 6.1)) < 1e-12   # wrapped to (-pi, pi]
    with open(path, "a") as f:
        f.write("bad;line\n")
    assert len(m.ImuLog(str(path)).ts) == 12


@pytest.mark.gpu
def test_slam_loop_with_imu(tmp_path):
    m = _load()
    out = m.run(36, log_path=str(tmp_path / "drive.csv"), verbose=False, imu_path=str(tmp_path / "imu.csv"))
    assert out["drift"].max() < 0.25, out["drift"].max()
    assert out["occupied"] > 1000 and out["free"] > 50000


@pytest.mark.gpu
def test_slam_loop_closes_the_loop():
    """A circuit of 1.2 laps: closures are found, the pose graph is optimised, poses are rewritten and the map is
    rebuilt by replaying every scan (slam.py:566-620)."""
    m = _load()
    out = m.run(81, verbose=False, loop=True)
    assert len(out["accepted"]) >= 1, out["closures"]
    for scan, matched, moved, info in out["accepted"]:
        assert scan - matched >= 20 and info["status"] == 1 and info["iterations"] <= 20
    assert out["drift"][-1] < 0.4, out["drift"][-1]
    g = out["graph"]
    assert len(g.nodes) == 81 and len(g.edges) == 80 + len(out["accepted"])
    # the rebuilt map is exactly the replay of every scan at its corrected pose
    from utilities.mapping import OccupancyGrid2D
    mp = out["mapper"]
    last = out["accepted"][-1][0]
    ref = OccupancyGrid2D(mp.min_x, mp.max_x, mp.min_y, mp.max_y, resolution=0.05, p_hit=0.85, p_miss=0.42,
                          log_odds_min=-8.0, log_odds_max=8.0)
    assert (ref.ny, ref.nx) == (mp.ny, mp.nx)
    # the map was rebuilt at the last closure (scans 0..last at their corrected poses, in order) and scans after it
    # were added one by one at poses nothing rewrote since: the same as every scan of the history replayed in order
    assert 30 <= last <= 80, out["accepted"]
    for pts, T in out["history"]:
        ref.update_scan(T[:2, 2], pts @ T[:2, :2].T + T[:2, 2])
    assert np.array_equal(ref.log_odds, mp.log_odds)


# ── oracle twin of the loop: the same decisions with the CPU oracle behind every geometric call ──
def _oracle_backend():
    """Test infrastructure: the CPU oracle behind the harness' backend interface (never part of the product)."""
    import oracle
    from oracle import pose_graph as opg

    class Submap:
        def __init__(self, window=40, voxel_size=0.04):
            self.window, self.voxel_size, self.scans = window, voxel_size, []

        def __len__(self):
            return len(self.scans)

        def push(self, pts):
            self.scans.append(np.array(pts, dtype=np.float64))
            if len(self.scans) > self.window:
                self.scans.pop(0)

        def reset(self, scans=()):
            self.scans = [np.array(s, dtype=np.float64) for s in list(scans)[-self.window:]]

        def attempt_icp(self, source, pose, imu_yaw, imu_narrow, rot_range, rot_step, rot_fine, rot_voxel, icp_cfg, corr):
            sub = oracle.voxel_downsample(np.vstack(self.scans), self.voxel_size)          # _build_submap, slam.py:103-108
            r, t, e, _ = oracle.attempt_submap_icp(source, sub, pose, imu_yaw, imu_narrow, rot_range, rot_step, rot_fine,
                                                   rot_voxel, icp_cfg, corr)
            return r, t, e

    class Grid:
        def __init__(self, min_x, max_x, min_y, max_y, resolution, p_hit, p_miss, log_odds_min, log_odds_max):
            self.min_x, self.max_x, self.min_y, self.max_y, self.resolution = min_x, max_x, min_y, max_y, resolution
            self.nx = int(np.ceil((max_x - min_x) / resolution))
            self.ny = int(np.ceil((max_y - min_y) / resolution))
            self.l_hit, self.l_miss = np.log(p_hit / (1 - p_hit)), np.log(p_miss / (1 - p_miss))
            self.lo, self.hi = log_odds_min, log_odds_max
            self.log_odds = np.zeros((self.ny, self.nx), dtype=np.float32)

        def update_scan(self, origin, hits):
            oracle.grid_update_scan(self.log_odds, self.min_x, self.min_y, self.resolution, origin, hits, self.l_hit, self.l_miss,
                                    self.lo, self.hi)

        def update_scans(self, origins, hits):
            for o, h in zip(origins, hits):
                self.update_scan(o, h)

        def reset(self):
            self.log_odds[:] = 0.0

    class Graph:
        def __init__(self):
            self.nodes, self.edges, self.last_info = [], [], {}

        def add_node(self, v):
            self.nodes.append(np.asarray(v, dtype=float).copy())
            return len(self.nodes) - 1

        def add_edge(self, i, j, z, omega=None):
            self.edges.append((i, j, np.asarray(z, dtype=float), np.eye(3) if omega is None else np.asarray(omega, dtype=float)))

        def optimize(self, n_iterations=20, fix_node=0, convergence_eps=1e-6):
            ei, ej = [e[0] for e in self.edges], [e[1] for e in self.edges]
            nodes, it, st, step = opg.optimize(np.array(self.nodes), ei, ej, np.array([e[2] for e in self.edges]),
                                               np.array([e[3] for e in self.edges]), n_iterations, fix_node, convergence_eps)
            self.nodes = [v.copy() for v in nodes]
            self.last_info = dict(iterations=it, status=st, step=step)

    class Backend:
        rotation_search = staticmethod(oracle.rotation_search)

        @staticmethod
        def ICP(s, t, **kw):
            return oracle.icp(s, t, **kw)[:3]

        @staticmethod
        def run_icp_pairs(source, targets, feat_cfg, icp_cfg):
            out = []
            for tg in targets:                                        # slam.py:53-98, candidate by candidate
                R0, t0, _ = oracle.rotation_search(source, tg, feat_cfg["rotation_voxel_size"], feat_cfg["angle_step_coarse"],
                                                   feat_cfg["angle_step_fine"])
                out.append(oracle.icp(source, tg, R_init=R0, t_init=t0, **icp_cfg))
            return (np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out]),
                    np.array([o[3]["iters"] for o in out]))

    Backend.Submap, Backend.Grid, Backend.Graph = Submap, Grid, Graph
    return Backend


@pytest.mark.gpu
def test_slam_loop_equals_its_oracle_twin():
    """The composed per-scan pipeline (slam.py:466-620: scan-to-scan ICP -> submap attempt -> map update -> closure
    candidates, first accepted -> pose graph -> pose rewrite -> submap and map rebuild) run twice on the same synthetic
    circuit: on the MI355X drop-ins and with the CPU oracle behind every geometric call.  Same decisions (rejections,
    closure candidates tried, closures accepted), poses within 1e-6, final occupancy grid bit for bit."""
    m = _load()
    n = 41                                                            # closure checks at scans 30 and 40
    gpu = m.run(n, verbose=False, loop=True)
    ref = m.run(n, verbose=False, loop=True, backend=_oracle_backend())
    assert gpu["rejected"] == ref["rejected"]
    assert [c[:2] for c in gpu["closures"]] == [c[:2] for c in ref["closures"]] and len(gpu["closures"]) >= 1
    assert [c[3] for c in gpu["closures"]] == [c[3] for c in ref["closures"]]           # same iteration counts
    assert np.allclose([c[2] for c in gpu["closures"]], [c[2] for c in ref["closures"]], rtol=0, atol=1e-9)
    assert [a[:2] for a in gpu["accepted"]] == [a[:2] for a in ref["accepted"]]
    for (_, Tg), (_, Tr) in zip(gpu["history"], ref["history"]):
        assert np.abs(Tg - Tr).max() < 1e-6
    assert np.array_equal(gpu["mapper"].log_odds, ref["mapper"].log_odds)
