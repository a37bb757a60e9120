#!/usr/bin/env python3
"""A minimal scan-matching + mapping loop on the drop-in API.

Not a port of the reference's slam.py (its pose graph, display and services are out of scope): a
fresh harness with the same per-scan shape (slam.py:377-640) that shows the accelerated path working
end to end on a synthetic drive:

    scan-to-scan:   rotation_search -> ICP(point_to_line), or the IMU     slam.py:53-98, 466-483
                    yaw step as the initial guess when an IMU log exists
    pose update:    global_pose @ inverse(T)                              slam.py:38-43, 494
    submap:         RollingSubmap.attempt_icp: rotation search about the  slam.py:111-225, 505-536
                    predicted pose (narrow about the IMU yaw) + p2p ICP
    mapping:        OccupancyGrid2D.update_scan                           slam.py:552-557
    loop closure:   every candidate goes through _run_icp_pair (rotation    slam.py:566-620, 53-98
                    search, then ICP from its result) — all candidates in
                    ONE batch, nothing returning to the host between the
                    search and the ICP (icpmi.prealign) — and, as in
                    the reference, the FIRST candidate in order whose error
                    is below the gate is accepted: it adds a pose-graph
                    edge, the graph is optimised, poses are rewritten, the
                    submap buffer and the occupancy grid are rebuilt
                    (replay of all scans)

The geometry behind the loop is a `backend` (rotation_search, ICP, run_icp_pairs, Submap, Grid, Graph): the MI355X
drop-ins by default; the tests inject the CPU oracle there to check the whole composition scan by scan.

It also writes and re-reads the drive in the reference's wire formats: lidar lines
`timestamp_us;x1;y1;z1;x2;...` (services/lidar_service.py:5-19) and IMU lines
`timestamp_us;qx;qy;qz;qw` (services/imu_service.py:1-38).

    python examples/slam_loop.py [n_scans] [--imu] [--loop]      (--loop: a closed circuit, 1.2 laps)
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))

from icpmi import batch, prealign, synth  # noqa: E402
from icpmi.submap import RollingSubmap  # noqa: E402
from utilities import features  # noqa: E402
from utilities import icp as uicp  # noqa: E402
from utilities.mapping import OccupancyGrid2D  # noqa: E402
from utilities.pose_graph import PoseGraph2D, pose_matrix_to_vec, pose_vec_to_matrix, relative_transform_vec  # noqa: E402
from utilities import pose_graph as upg  # noqa: E402


# ── reference wire format ────────────────────────────────────────────────────
def write_lidar_log(path, scans, dt_us=100000, z=1.0):
    """One line per scan: timestamp_us;x;y;z;x;y;z;... (sensor frame)."""
    with open(path, "w") as f:
        for i, s in enumerate(scans):
            xyz = np.column_stack([s, np.full(len(s), z)])
            f.write(str(1000000 + i * dt_us) + ";" + ";".join(repr(float(v)) for v in xyz.ravel()) + "\n")


def read_lidar_log(path, z_min=0.2, z_max=2.0):
    """Yield (timestamp_us, points_xy): all-zero triples dropped, z-slice kept (slam.py:24-27)."""
    with open(path) as f:
        for line in f:
            el = line.strip().replace(";", " ").split()
            ts = int(el[0])
            p = np.array(el[1:], dtype=np.float64).reshape(-1, 3)
            p = p[~np.all(p == 0, axis=1)]
            keep = (p[:, 2] >= z_min) & (p[:, 2] <= z_max)
            yield ts, np.ascontiguousarray(p[keep, :2])


def write_imu_log(path, yaws, dt_us=100000, per_scan=4, seed=11, noise=0.002):
    """Orientation quaternions about z, `per_scan` readings per scan interval, yaw noise in radians."""
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for i in range(len(yaws)):
            nxt = yaws[min(i + 1, len(yaws) - 1)]
            for j in range(per_scan):
                yaw = yaws[i] + (nxt - yaws[i]) * j / per_scan + rng.normal(0.0, noise)
                f.write(f"{1000000 + i * dt_us + j * dt_us // per_scan};0.0;0.0;{float(np.sin(yaw / 2))!r};{float(np.cos(yaw / 2))!r}\n")


class ImuLog:
    """Yaw by timestamp from an orientation log: yaw = atan2(2(qw qz + qx qy), 1 - 2(qy^2 + qz^2)), nearest reading."""

    def __init__(self, path):
        ts, yaw = [], []
        with open(path) as f:
            for line in f:
                el = line.strip().split(";")
                if len(el) < 5:
                    continue
                qx, qy, qz, qw = (float(v) for v in el[1:5])
                ts.append(int(el[0]))
                yaw.append(np.arctan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz)))
        self.ts, self.yaw = np.array(ts, dtype=np.int64), np.array(yaw)

    def yaw_at(self, t_us):
        i = int(np.clip(np.searchsorted(self.ts, t_us), 1, len(self.ts) - 1))
        return self.yaw[i - 1] if t_us - self.ts[i - 1] <= self.ts[i] - t_us else self.yaw[i]

    def delta_yaw(self, t0_us, t1_us):
        return (self.yaw_at(t1_us) - self.yaw_at(t0_us) + np.pi) % (2 * np.pi) - np.pi


def pose_matrix(x, y, th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, x], [s, c, y], [0.0, 0.0, 1.0]])


class GpuBackend:
    """The accelerated drop-ins (the product path)."""
    rotation_search = staticmethod(features.rotation_search)
    ICP = staticmethod(uicp.ICP)
    Submap = RollingSubmap
    Grid = OccupancyGrid2D
    Graph = PoseGraph2D

    @staticmethod
    def run_icp_pairs(source, targets, feat_cfg, icp_cfg):
        """_run_icp_pair(source, target) for every candidate (slam.py:575-579 -> 53-98) -> (R [B,2,2], t [B,2], err [B],
        iterations [B]): rotation searches and ICPs of all candidates as one chain of launches."""
        R, t, err, info = prealign.run_icp_pair_batch(source, targets, icp_cfg, feat_cfg)
        return R, t, err, info["iters"]


def run(n_scans=60, log_path=None, verbose=True, imu_path=None, loop=False, use_submap=True, lc_error_threshold=0.05,
        backend=None, max_candidates=5):
    from icpmi import submap as submap_mod
    uicp.VERBOSE = features.VERBOSE = submap_mod.VERBOSE = upg.VERBOSE = False
    be = backend or GpuBackend
    segs = synth.maze_segments()
    truth = synth.loop_trajectory(n_scans) if loop else synth.trajectory(n_scans, step=0.18)
    scans = [synth.scan(p, 9000 + i, segs=segs) for i, p in enumerate(truth)]
    stamps = [1000000 + 100000 * i for i in range(n_scans)]
    if log_path:
        write_lidar_log(log_path, scans)
        stamps, scans = (list(v) for v in zip(*read_lidar_log(log_path)))
    imu = None
    if imu_path:
        write_imu_log(imu_path, [p[2] for p in truth])
        imu = ImuLog(imu_path)
        yaw_offset = imu.yaw_at(stamps[0]) - truth[0][2]          # calibrated so that the first scan has the start yaw
    icp_kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
    icp_cfg = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04)
    pose = pose_matrix(*truth[0])                       # start at the true pose; everything after is estimated
    submap = be.Submap(window=40, voxel_size=0.04)
    history, mapper, timing = [], None, {"s2s": 0.0, "submap": 0.0, "map": 0.0, "loop": 0.0}
    closures, rejected, accepted = [], [], []
    graph = be.Graph()
    prev = None
    for i, cur in enumerate(scans):
        if prev is not None:
            t0 = time.perf_counter()
            # ICP(prev -> cur) maps the previous scan into the current sensor frame (slam.py:481-483)
            imu_yaw = None
            if imu is not None:                                               # slam.py:455-479: IMU yaw step as the guess
                imu_yaw = (imu.yaw_at(stamps[i]) - yaw_offset + np.pi) % (2 * np.pi) - np.pi
                d = imu.delta_yaw(stamps[i - 1], stamps[i])
                R0, t0v = np.array([[np.cos(d), np.sin(d)], [-np.sin(d), np.cos(d)]]), np.zeros(2)   # prev -> cur frame
            else:
                R0, t0v, _ = be.rotation_search(prev, cur, voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1)
            r, t, err = be.ICP(prev, cur, R_init=R0, t_init=t0v, **icp_kw)
            if err <= 0.15:                                                   # error_reject_threshold, slam.py:485-490
                T_inv = np.eye(3)
                T_inv[:2, :2] = r.T
                T_inv[:2, 2] = -r.T @ t
                pose = pose @ T_inv                                           # slam.py:38-43
            else:
                rejected.append(i)                                            # keep the pose; the submap step may still fix it
            timing["s2s"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            if use_submap and len(submap) >= 5:                               # slam.py:505-536
                Rs, ts, es = submap.attempt_icp(cur, pose, imu_yaw, 3.0, 60.0, 0.8, 0.1, 0.2, icp_cfg, 1.5)
                dpos = np.linalg.norm(ts - pose[:2, 2])
                dyaw = abs((np.arctan2(Rs[1, 0], Rs[0, 0]) - np.arctan2(pose[1, 0], pose[0, 0]) + np.pi) % (2 * np.pi) - np.pi)
                if es <= 0.15 and dpos < 1.5 and dyaw < np.deg2rad(15):          # slam.py:512-531
                    pose[:2, :2], pose[:2, 2] = Rs, ts
            timing["submap"] += time.perf_counter() - t0
        world = cur @ pose[:2, :2].T + pose[:2, 2]
        t0 = time.perf_counter()
        if mapper is None:                                                    # slam.py:398-408
            mapper = be.Grid(world[:, 0].min() - 50, world[:, 0].max() + 50, world[:, 1].min() - 50,
                             world[:, 1].max() + 50, resolution=0.05, p_hit=0.85, p_miss=0.42,
                             log_odds_min=-8.0, log_odds_max=8.0)
        mapper.update_scan(pose[:2, 2], world)
        timing["map"] += time.perf_counter() - t0
        submap.push(world)
        history.append((cur, pose.copy()))
        node = graph.add_node(pose_matrix_to_vec(pose))                      # slam.py:543-549: node + odometry edge
        if node > 0:
            odo_err = err if err <= 0.15 else 0.15
            graph.add_edge(node - 1, node, relative_transform_vec(history[-2][1], pose), np.eye(3) / max(odo_err, 1e-6))
        # loop closure candidates: old scans near the current position (slam.py:230-268), tried in order; each is
        # pre-aligned and registered as _run_icp_pair does (slam.py:53-98) — all of them in one batch, searches and ICPs
        # chained on the device — and the FIRST whose error is below the gate wins (slam.py:582-597)
        t0 = time.perf_counter()
        if i >= 30 and i % 10 == 0:
            cands = [k for k, (_, pk) in enumerate(history[:-20]) if np.linalg.norm(pk[:2, 2] - pose[:2, 2]) < 3.0][:max_candidates]
            if cands:
                R, t, err, its = be.run_icp_pairs(cur, [history[k][0] for k in cands],
                                                  dict(rotation_voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1), icp_kw)
                ok = np.flatnonzero(np.asarray(err) < lc_error_threshold)
                first = int(ok[0]) if len(ok) else -1
                shown = first if first >= 0 else int(np.argmin(err))
                closures.append((i, cands[shown], float(err[shown]), int(its[shown])))
                if first >= 0:                                               # slam.py:582-620
                    best = first
                    T_lc = np.eye(3)
                    T_lc[:2, :2], T_lc[:2, 2] = R[best], t[best]              # cur -> candidate frame, so z = T_lc^-1
                    graph.add_edge(node, cands[best], pose_matrix_to_vec(np.linalg.inv(T_lc)),
                                   np.eye(3) * 10.0 / max(float(err[best]), 1e-6))
                    before = pose[:2, 2].copy()
                    graph.optimize(n_iterations=20, fix_node=0)
                    history = [(pts, pose_vec_to_matrix(v)) for (pts, _), v in zip(history, graph.nodes)]
                    pose = history[-1][1].copy()
                    worlds = [pts @ T[:2, :2].T + T[:2, 2] for pts, T in history]
                    submap.reset(worlds[-submap.window:])
                    mapper.reset()                                            # _rebuild_map, slam.py:271-277
                    mapper.update_scans(np.array([T[:2, 2] for _, T in history]), worlds)
                    accepted.append((i, cands[best], float(np.linalg.norm(pose[:2, 2] - before)), dict(graph.last_info)))
        timing["loop"] += time.perf_counter() - t0
        prev = cur
    est = np.array([p[:2, 2] for _, p in history])
    gt = np.array([[p[0], p[1]] for p in truth])
    drift = np.linalg.norm(est - gt, axis=1)
    occupied = int((mapper.log_odds > 0).sum())
    free = int((mapper.log_odds < 0).sum())
    if verbose:
        print(f"{n_scans} scans: final position error {drift[-1]:.3f} m (max {drift.max():.3f} m over {np.sum(np.linalg.norm(np.diff(gt, axis=0), axis=1)):.1f} m driven)")
        print(f"map: {occupied} occupied / {free} free cells; scan-to-scan rejections at {rejected}; loop-closure checks: {closures}")
        print("wall ms per scan: " + ", ".join(f"{k} {v / n_scans * 1e3:.2f}" for k, v in timing.items()))
        if accepted:
            print("closures accepted (scan, matched scan, pose moved by [m], iterations): "
                  + ", ".join(f"({a}, {b}, {d:.3f}, {inf['iterations']})" for a, b, d, inf in accepted))
    return dict(drift=drift, occupied=occupied, free=free, closures=closures, rejected=rejected, timing=timing,
                accepted=accepted, graph=graph, mapper=mapper, history=history)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    run(int(args[0]) if args else 60, log_path="/tmp/icpmi_demo_lidar.csv",
        imu_path="/tmp/icpmi_demo_imu.csv" if "--imu" in sys.argv else None, loop="--loop" in sys.argv,
        use_submap="--no-submap" not in sys.argv)
