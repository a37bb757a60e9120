"""Synthetic 2-D LiDAR scans of a polygonal world.

The reference ships no sensor logs (``data/*.csv`` is git-ignored there), so
every measurement and parity input in this repo comes from this generator:
a world of wall segments, a 2 048-beam scanner, Gaussian range noise from
``np.random.default_rng(seed)``.  Pure NumPy, host side only.
"""
import numpy as np

# Outer room plus four boxes: (x0, y0, x1, y1) axis-aligned rectangles.
ROOM = (-10.0, -6.0, 10.0, 6.0)
BOXES = (
    (2.0, 1.0, 4.0, 3.0),
    (-6.0, -4.0, -4.0, -1.5),
    (-3.0, 3.0, -1.0, 4.5),
    (5.0, -4.0, 7.5, -3.0),
)


def _rect_segments(r):
    x0, y0, x1, y1 = r
    c = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
    return [(c[i], c[(i + 1) % 4]) for i in range(4)]


def room_segments():
    """(S, 4) array of wall segments ``[ax, ay, bx, by]`` of the single room."""
    segs = _rect_segments(ROOM)
    for b in BOXES:
        segs += _rect_segments(b)
    return np.array([[a[0], a[1], b[0], b[1]] for a, b in segs], dtype=np.float64)


def maze_segments(nx=6, ny=4, cell=10.0, door=2.0, seed=7):
    """A larger multi-room world (≈ ``nx*ny`` rooms of ``cell`` metres) used to
    grow a rolling submap to ~10 k points after voxel filtering (config 3)."""
    rng = np.random.default_rng(seed)
    segs = []
    W, H = nx * cell, ny * cell
    segs += _rect_segments((0.0, 0.0, W, H))
    for i in range(1, nx):            # vertical walls with a door per room
        x = i * cell
        for j in range(ny):
            y0, y1 = j * cell, (j + 1) * cell
            d = y0 + 1.0 + rng.uniform(0, cell - 2.0 - door)
            segs.append(((x, y0), (x, d)))
            segs.append(((x, d + door), (x, y1)))
    for j in range(1, ny):            # horizontal walls with a door per room
        y = j * cell
        for i in range(nx):
            x0, x1 = i * cell, (i + 1) * cell
            d = x0 + 1.0 + rng.uniform(0, cell - 2.0 - door)
            segs.append(((x0, y), (d, y)))
            segs.append(((d + door, y), (x1, y)))
    for i in range(nx):               # one pillar per room
        for j in range(ny):
            cx = i * cell + rng.uniform(2.5, cell - 2.5)
            cy = j * cell + rng.uniform(2.5, cell - 2.5)
            s = rng.uniform(0.4, 1.2)
            segs += _rect_segments((cx - s, cy - s, cx + s, cy + s))
    return np.array([[a[0], a[1], b[0], b[1]] for a, b in segs], dtype=np.float64)


def cast_ranges(segs, pose, n_beams=2048):
    """Exact ranges from ``pose=(x, y, theta)`` to the nearest wall per beam.

    Beam angles are ``theta + linspace(-pi, pi, n_beams, endpoint=False)``.
    Beams that hit nothing get ``inf``.
    """
    x, y, th = pose
    ang = th + np.linspace(-np.pi, np.pi, n_beams, endpoint=False)
    dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]          # (B,1)
    ax, ay = segs[None, :, 0] - x, segs[None, :, 1] - y           # (1,S)
    ex, ey = segs[None, :, 2] - segs[None, :, 0], segs[None, :, 3] - segs[None, :, 1]
    den = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (ax * ey - ay * ex) / den                             # along the beam
        u = (ax * dy - ay * dx) / den                             # along the wall
    ok = (np.abs(den) > 1e-12) & (t > 1e-9) & (u >= 0.0) & (u <= 1.0)
    t = np.where(ok, t, np.inf)
    return t.min(axis=1), ang


def scan(pose, seed, n_beams=2048, noise=0.01, segs=None):
    """One scan in the SENSOR frame: ``(n_beams, 2) float64``."""
    if segs is None:
        segs = room_segments()
    rng = np.random.default_rng(seed)
    r, ang = cast_ranges(segs, pose, n_beams)
    r = r + rng.normal(0.0, noise, size=n_beams)
    keep = np.isfinite(r)
    a_local = ang - pose[2]
    pts = np.stack([r * np.cos(a_local), r * np.sin(a_local)], axis=1)
    return np.ascontiguousarray(pts[keep])


def to_world(pts, pose):
    """Sensor-frame points -> world frame for ``pose=(x, y, theta)``."""
    c, s = np.cos(pose[2]), np.sin(pose[2])
    R = np.array([[c, -s], [s, c]])
    return pts @ R.T + np.array([pose[0], pose[1]])


def config2_pair(seed=0):
    """BASELINE config 2: scans at (0,0,0) and (0.15,-0.08,3 deg)."""
    a = scan((0.0, 0.0, 0.0), seed)
    b = scan((0.15, -0.08, np.deg2rad(3.0)), seed + 1)
    return a, b


def _free_pose(x, y, clearance=0.25):
    """Inside the room and not inside (or within `clearance` of) one of its boxes."""
    if not (ROOM[0] + clearance < x < ROOM[2] - clearance and ROOM[1] + clearance < y < ROOM[3] - clearance):
        return False
    return not any(b[0] - clearance < x < b[2] + clearance and b[1] - clearance < y < b[3] + clearance for b in BOXES)


def loop_closure_batch(n_pairs, seed0=1000, shared_source=False, max_offset=0.6, max_yaw_deg=6.0):
    """Candidate scan pairs of a loop closure (BASELINE config 5): the target pose lies within ``max_offset`` metres
    (uniform distance, uniform direction) and ``max_yaw_deg`` degrees (uniform) of the source pose.

    The defaults, 0.6 m / 6 deg, are offsets ICP converges from WITHOUT pre-alignment (the `ICP iterations/s` batches of
    bench.py).  SURVEY section 8d / config.yaml:70 describe the candidates the reference gates — within 3 m / 20 deg:
    ``max_offset=3.0, max_yaw_deg=20.0``; those are only reachable through the rotation search of _run_icp_pair
    (icpmi.prealign), which is exactly why the reference pre-aligns."""
    srcs, tgts = [], []
    base = (0.5, -0.3, 0.1)
    src_shared = scan(base, seed0 - 1)
    for i in range(n_pairs):
        rng = np.random.default_rng(seed0 + i)
        while True:                          # a sensor cannot stand inside a box or a wall: such draws are taken again
            d = rng.uniform(0.0, max_offset)
            a = rng.uniform(-np.pi, np.pi)
            th = np.deg2rad(rng.uniform(-max_yaw_deg, max_yaw_deg))
            pose_t = (base[0] + d * np.cos(a), base[1] + d * np.sin(a), base[2] + th)
            if _free_pose(pose_t[0], pose_t[1]):
                break
        srcs.append(src_shared if shared_source else scan(base, seed0 + 7919 * (i + 1)))
        tgts.append(scan(pose_t, seed0 + i))
    return srcs, tgts


def trajectory(n, start=(5.0, 5.0, 0.0), step=0.25, segs=None, seed=3):
    """A smooth drive through the maze world: list of poses."""
    rng = np.random.default_rng(seed)
    poses = []
    x, y, th = start
    for _ in range(n):
        poses.append((x, y, th))
        th += rng.normal(0.0, 0.03)
        x += step * np.cos(th)
        y += step * np.sin(th)
    return poses


def loop_trajectory(n, center=(25.0, 15.0), radius=3.2, laps=1.2):
    """A closed circuit (counter-clockwise circle, heading along the tangent) that overlaps itself after one
    lap, for loop closures.  The default sits in a room of ``maze_segments()`` clear of its pillar."""
    poses = []
    for k in range(n):
        a = 2.0 * np.pi * laps * k / max(n - 1, 1)
        poses.append((center[0] + radius * np.cos(a), center[1] + radius * np.sin(a), a + np.pi / 2.0))
    return poses
