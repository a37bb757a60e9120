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
    assert last == 80, out["accepted"]                                # closed on the last scan: nothing was added after the rebuild
    if last == 80:
        for pts, T in out["history"]:
            ref.update_scan(T[:2, 2], pts @ T[:2, :2].T + T[:2, 2])
        assert np.array_equal(ref.log_odds, mp.log_odds)
