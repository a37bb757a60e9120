"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the
header declares; the product path refuses to run without a GPU (no CPU
fallback); the multi-rank sharding/gather logic works under gloo with
world_size 2 (the oracle is injected as the local solver — test only)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, REPO

HEADER = os.path.join(REPO, "include", "icpmi.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(icpmi_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    import ctypes
    import icpmi
    path = icpmi.build()
    lib = icpmi.lib()                     # torch first, then the library: ONE HIP runtime in the process (icpmi_runtime_check)
    L = ctypes.CDLL(path)                 # the same mapping, seen through a plain handle
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/icpmi.h but not exported"
    # and the Python binding knows every one of them
    from icpmi import _lib
    assert sorted(_lib.EXPORTS) == syms
    assert _lib.runtime_problem() == ""
    assert lib.icpmi_version().decode().startswith("icpmi")
    assert lib.icpmi_strerror(-2).decode() == "workspace missing or too small"
    # host-only size queries (no GPU touched)
    assert lib.icpmi_grid_workspace_bytes(10, 20) == 4 * 10 * 20 * 4 + 256 + 2 * 32 * 17 * 16   # four grids of counters (was 32) + box slots + per-scan cell boxes
    assert lib.icpmi_icp_workspace_bytes(2, 100, 2) == 2 * 100 * (16 + 8 + 4) + 256
    assert lib.icpmi_voxel_workspace_bytes(2048) == 256
    assert lib.icpmi_voxel_workspace_bytes(100000) > 100000 * 24


def test_two_hip_runtimes_in_one_process_are_named():
    """libicpmi.so loaded BEFORE torch maps /opt/rocm's libamdhip64 and torch then maps its own — every
    launch would fail with a bare ICPMI_ERR_HIP.  icpmi_runtime_check names the two copies; the Python loader raises with
    that message instead of handing out a library that cannot launch."""
    code = (f"import ctypes, sys; sys.path.insert(0, {PKG!r}); L = ctypes.CDLL({os.path.join(PKG, 'lib', 'libicpmi.so')!r});"
            "buf = ctypes.create_string_buffer(1024); L.icpmi_runtime_check.argtypes = [ctypes.c_char_p, ctypes.c_size_t];"
            "assert L.icpmi_runtime_check(buf, 1024) == 0, buf.value;"          # the library alone: one runtime
            "import torch;"
            "rc = L.icpmi_runtime_check(buf, 1024); print('RC', rc, buf.value.decode());"
            "from icpmi import _lib\n"
            "try:\n    _lib.lib(); print('LOADED')\nexcept _lib.IcpmiError as e:\n    print('RAISED', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    if "RC 0" in r.stdout:                # torch resolved to the runtime already mapped (same SONAME): nothing to diagnose
        assert "LOADED" in r.stdout
    else:
        assert "RC -3 two HIP runtimes in one process" in r.stdout and "RAISED two HIP runtimes" in r.stdout, r.stdout


def test_struct_layout_matches_header():
    import ctypes
    from icpmi._lib import IcpParams
    assert ctypes.sizeof(IcpParams) == 32
    assert IcpParams.max_iterations.offset == 16 and IcpParams.dim.offset == 28


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from icpmi import IcpmiError
    from utilities import icp as uicp
    from utilities.mapping import OccupancyGrid2D
    pts = np.random.default_rng(0).normal(size=(100, 2))
    with pytest.raises(IcpmiError):
        uicp.ICP(pts, pts, 1e-6, 10, 0.1)
    with pytest.raises(IcpmiError):
        uicp.voxel_downsample(pts, 0.1)
    with pytest.raises(IcpmiError):
        OccupancyGrid2D(-1, 1, -1, 1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package or bench's GPU legs may import it."""
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import oracle|from oracle)", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_shard_and_gather_single_process():
    import torch
    from icpmi import dist as idist
    assert list(idist.shard(10, 1, 4)) == [1, 5, 9]
    assert idist.slots_per_rank(10, 4) == 3
    loc = torch.arange(5 * 16, dtype=torch.float64).reshape(5, 16)
    out = idist.gather_results(loc, 5, 0, 1)
    assert torch.equal(out, loc)
    res = torch.zeros((4, 16), dtype=torch.float64)
    res[:, 12] = torch.tensor([0.5, 0.2, 0.01, 0.001])
    assert idist.first_accepted(res, 0.05) == 2 and idist.first_accepted(res, 1e-6) == -1


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, {pkg!r})
import oracle
from icpmi import dist as idist, synth, _lib

def solver(src, tgt, Ri, ti):
    out = np.zeros((len(tgt), _lib.RES_DOUBLES))
    for i, (s, t) in enumerate(zip(src, tgt)):
        R, tt, err, info = oracle.icp(s, t, 1e-10, 60, 0.1, method="point_to_line", normal_k=8)
        out[i, :4] = R.ravel(); out[i, 9:11] = tt; out[i, 12] = err
        out[i, 14] = info["iters"]; out[i, 15] = info["status"]
    return torch.from_numpy(out)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
srcs, tgts = synth.loop_closure_batch(5, seed0=77)
srcs = [s[::4] for s in srcs]; tgts = [t[::4] for t in tgts]
res = idist.icp_batch_sharded(srcs, tgts, 1e-10, 60, 0.1, method="point_to_line", normal_k=8, solver=solver)
full = solver(srcs, tgts, None, None)            # every pair locally, in order
assert res.shape == (5, _lib.RES_DOUBLES)
assert torch.equal(res, full), (rank, (res - full).abs().max())
shared = idist.icp_batch_sharded(srcs[0], tgts, 1e-10, 60, 0.1, solver=solver)
assert torch.equal(shared, solver([srcs[0]] * 5, tgts, None, None))
dist.barrier()
if rank == 0:
    print("GLOO_OK", world)
dist.destroy_process_group()
'''


def test_sharded_batch_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO, pkg=PKG))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29713", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GLOO_OK 2" in r.stdout


RUN_ICP_PAIR_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, {pkg!r})
import oracle
from icpmi import dist as idist, synth, _lib

ICP = dict(error_threshold=1e-10, max_iterations=40, voxel_size=0.1, method="point_to_line", normal_k=8)
FEAT = dict(rotation_voxel_size=0.3, angle_step_coarse=6.0, angle_step_fine=1.0)

def solver(src, tgts):          # slam.py:53-98 with the oracle: rotation search, then ICP from its result
    out = np.zeros((len(tgts), _lib.RES_DOUBLES))
    for i, t in enumerate(tgts):
        R0, t0, _ = oracle.rotation_search(src, t, FEAT["rotation_voxel_size"], FEAT["angle_step_coarse"], FEAT["angle_step_fine"])
        R, tt, err, info = oracle.icp(src, t, 1e-10, 40, 0.1, R_init=R0, t_init=t0, method="point_to_line", normal_k=8)
        out[i, :4] = R.ravel(); out[i, 9:11] = tt; out[i, 12] = err
        out[i, 13] = info["delta"]; out[i, 14] = info["iters"]; out[i, 15] = info["status"]
    return torch.from_numpy(out)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
srcs, tgts = synth.loop_closure_batch(7, seed0=311, shared_source=True, max_offset=2.0, max_yaw_deg=15.0)
src = srcs[0][::4]; tgts = [t[::4] for t in tgts]
job = idist.RunIcpPairSharded(src, tgts, ICP, FEAT, solver=solver)
assert list(job.mine) == list(range(rank, 7, world))
res = job.run()
full = solver(src, tgts)                                   # every candidate locally, in candidate order
assert res.shape == (7, _lib.RES_DOUBLES) and torch.equal(res, full), rank
err = full[:, 12].numpy()
gate = float(np.sort(err)[2]) * 1.0000001                  # three candidates pass: the FIRST in candidate order wins
want = int(np.flatnonzero(err < gate)[0])
assert job.first_accepted(gate) == want and job.first_accepted(0.0) == -1
R, t, e, info = idist.run_icp_pair_batch_sharded(src, tgts, ICP, FEAT, error_accept=gate, solver=solver)
assert info["first_accepted"] == want and np.array_equal(e, err) and np.array_equal(R.reshape(7, 4), full[:, :4].numpy())
assert np.array_equal(info["iters"], full[:, 14].numpy().astype(np.int64))
dist.barrier()
if rank == 0:
    print("GLOO_RUN_ICP_PAIR_OK", world)
dist.destroy_process_group()
'''


def test_sharded_run_icp_pair_gloo_world2(tmp_path):
    """icpmi.dist.RunIcpPairSharded / run_icp_pair_batch_sharded (slam.py:575-597 over ranks): candidates interleaved over two
    gloo ranks, the oracle injected as the local solver; gathered records in candidate order, first accepted candidate."""
    script = tmp_path / "worker.py"
    script.write_text(RUN_ICP_PAIR_WORKER.format(repo=REPO, pkg=PKG))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29733", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GLOO_RUN_ICP_PAIR_OK 2" in r.stdout


def test_row_bands_cover_and_balance():
    import torch
    from icpmi import dist as idist, synth
    assert idist.row_bands(10, 4) == [0, 2, 5, 7, 10]
    assert idist.row_bands(7, 1) == [0, 7]
    segs = synth.room_segments()
    poses = [(0.0, 0.0, 0.0), (1.0, -0.5, 0.3), (-2.0, 1.0, -0.2)]
    hits = [synth.to_world(synth.scan(p, 50 + i, segs=segs), p) for i, p in enumerate(poses)]
    org = np.array([[p[0], p[1]] for p in poses])
    ny = 800
    cost = idist.row_costs(ny, -20.0, 0.05, org, hits)
    assert cost.shape == (ny,) and int(cost.min()) >= 0 and int(cost.sum()) > 0
    for world in (2, 3, 8):
        b = idist.row_bands(ny, world, cost)
        assert b[0] == 0 and b[-1] == ny and all(b[i] <= b[i + 1] for i in range(world))
        work = [int(cost[b[i]:b[i + 1]].sum()) for i in range(world)]
        assert max(work) < 1.5 * sum(work) / world, work           # near-equal shares of the estimated work
    # same numbers from torch tensors as from numpy arrays (every rank must agree)
    cost_t = idist.row_costs(ny, -20.0, 0.05, torch.from_numpy(org), [torch.from_numpy(h) for h in hits])
    assert torch.equal(cost, cost_t)


REPLAY_WORKER = r"""
import os, sys, types
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, {pkg!r})
import oracle
from icpmi import dist as idist, synth

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
segs = synth.room_segments()
poses = [(0.3 * i, -0.1 * i, 0.05 * i) for i in range(4)]
hits = [synth.to_world(synth.scan(p, 300 + i, segs=segs), p)[::8] for i, p in enumerate(poses)]
org = np.array([[p[0], p[1]] for p in poses])
g = types.SimpleNamespace(ny=300, nx=420, min_x=-10.5, min_y=-7.5, resolution=0.05)
l_hit, l_miss = float(np.log(0.85 / 0.15)), float(np.log(0.42 / 0.58))

def whole():
    lo = np.zeros((g.ny, g.nx), dtype=np.float32)
    for o, h in zip(org, hits):
        oracle.grid_update_scan(lo, g.min_x, g.min_y, g.resolution, o, h, l_hit, l_miss, -8.0, 8.0)
    return lo

def replay(r0, r1):          # stand-in for the GPU band replay: cells are independent, so a band is a slice
    return torch.from_numpy(whole()[r0:r1].copy())

bands, full = idist.replay_scans_sharded(g, org, hits, replay=replay)
assert len(bands) == world + 1 and bands[0] == 0 and bands[-1] == g.ny
assert 0 < bands[1] < g.ny                       # both ranks own rows the scans touch
assert np.array_equal(full.numpy(), whole())
bands2, full2 = idist.replay_scans_sharded(g, org, hits, replay=replay, balance=False)
assert bands2 == [0, 150, 300] and np.array_equal(full2.numpy(), whole())
dist.barrier()
if rank == 0:
    print("REPLAY_OK", world, bands)
dist.destroy_process_group()
"""


def test_sharded_replay_gloo_world2(tmp_path):
    script = tmp_path / "replay_worker.py"
    script.write_text(REPLAY_WORKER.format(repo=REPO, pkg=PKG))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29717", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "REPLAY_OK 2" in r.stdout


def test_bench_self_launch_decision():
    """`python bench.py --gpus 2` without WORLD_SIZE (how the driver starts it) must decide to spawn child ranks
    under torch.distributed.run BEFORE torch is imported in the parent (never re-exec a process that touched the GPU)."""
    import json
    code = ("import sys, runpy\n"
            f"sys.argv = [{os.path.join(REPO, 'bench.py')!r}, '--gpus', '2', '--steps', '3', '--warmup', '1']\n"
            "try:\n"
            f"    runpy.run_path({os.path.join(REPO, 'bench.py')!r}, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules, 'the launching parent imported torch'\n"
            "print('PARENT_CLEAN')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ICPMI_BENCH_DRYLAUNCH"] = "1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PARENT_CLEAN" in r.stdout
    cmd = json.loads(r.stdout.splitlines()[0])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    # under a launcher (WORLD_SIZE set) the same command line does not spawn again: it goes on to need a GPU
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                        env=env2, timeout=300)
    assert "launch" not in r2.stdout
    import torch
    if not torch.cuda.is_available():
        assert r2.returncode != 0 and "no CPU fallback" in (r2.stdout + r2.stderr)


def test_bench_refuses_stale_pmc_traffic(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 --pmc summary; one collected on other kernel sources is refused."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sig = bench.csrc_signature()
    assert re.fullmatch(r"[0-9a-f]{16}", sig)
    doc = {"csrc_sha256": sig, "kernels": {"void icpmi::icp2_fused_kernel<512, 2, true, true>(icpmi::Icp2Args) [64 workgroups]": {"hbm_bytes_per_launch": 123},
                                           "void icpmi::icp2_resume_rest_kernel<512, 2, true, true>(icpmi::Icp2Args, int) [256 workgroups]": {"hbm_bytes_per_launch": 7}}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "PMC_TRAFFIC", str(f))
    assert bench.pmc_traffic(64)[0] == 123
    assert bench.pmc_traffic(65)[0] is None
    doc["csrc_sha256"] = "0" * 16
    f.write_text(json.dumps(doc))
    t, why = bench.pmc_traffic(64)
    assert t is None and "stale" in why
    # the vector-issue roofline of a batch comes from the committed SQ-counter summary the same way
    k = "void icpmi::icp2_fused_kernel<512, 2, true, true> [64 workgroups]"
    mix = {"csrc_sha256": sig, "kernels": {k: {"SQ_ACTIVE_INST_VALU": 6.4e6, "SQ_BUSY_CYCLES": 1.0e6,
                                               "derived": {"valu_issue_frac": 0.8, "active_lanes_per_valu_instruction": 44.0}}}}
    g = tmp_path / "mix.json"
    g.write_text(json.dumps(mix))
    monkeypatch.setattr(bench, "PMC_MIX", str(g))
    issue, src = bench.pmc_issue(64)
    assert abs(issue["frac"] - 6.4e6 * 4 / (1024 * 1.0e6 / 32)) < 1e-4 and "same csrc" in src
    mix["csrc_sha256"] = "0" * 16
    g.write_text(json.dumps(mix))
    assert bench.pmc_issue(64)[0] is None and "stale" in bench.pmc_issue(64)[1]
    assert os.path.basename(bench.latest_profile("pmc_traffic.json")).startswith("r")


def test_arange_rows_is_numpys_arange():
    """icpmi.prealign.arange_rows tabulates np.arange(lo[k], hi[k], step) for every k at once (the fine grids that can
    follow each coarse winner of a rotation search, features.py:227-229): the same numbers and lengths bit for bit."""
    from icpmi.prealign import arange_rows
    for cstep, fstep in ((2.0, 0.2), (1.5, 0.1), (2.0, 0.5), (7.0, 0.33), (1.0, 2.5)):
        coarse = np.deg2rad(np.arange(-180, 180, cstep))
        lo, hi = coarse - np.deg2rad(cstep), coarse + np.deg2rad(cstep)
        vals, n = arange_rows(lo, hi, np.deg2rad(fstep))
        for k in range(len(coarse)):
            ref = np.arange(lo[k], hi[k], np.deg2rad(fstep))
            assert n[k] == len(ref) and np.array_equal(vals[k, :n[k]], ref), (cstep, fstep, k)
    vals, n = arange_rows([1.0, 2.0], [1.0, 1.5], 0.1)                  # empty grids
    assert list(n) == [0, 0]


def test_candidate_poses_stay_out_of_obstacles():
    """synth.loop_closure_batch: the defaults are unchanged by the rejection of poses inside boxes (none of them is
    within reach at 0.6 m), the 3 m / 20 degree candidates all stand in free space."""
    from icpmi import synth
    rng_poses = []
    for i in range(64):
        rng = np.random.default_rng(500 + i)
        d, a, th = rng.uniform(0.0, 0.6), rng.uniform(-np.pi, np.pi), np.deg2rad(rng.uniform(-6.0, 6.0))
        rng_poses.append((0.5 + d * np.cos(a), -0.3 + d * np.sin(a)))
    assert all(synth._free_pose(x, y) for x, y in rng_poses)            # first draws are always accepted at 0.6 m
    assert not synth._free_pose(3.0, 2.0) and not synth._free_pose(9.9, 0.0) and synth._free_pose(0.5, -0.3)
    srcs, tgts = synth.loop_closure_batch(3, seed0=500, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
    assert srcs[0] is srcs[1] and all(len(t) == 2048 for t in tgts)     # a scan from free space sees a wall on every beam


def test_pair_lists_of_the_batched_run_icp_pair():
    from icpmi.prealign import _pair_lists
    a, b, c = np.zeros((5, 2)), np.ones((6, 2)), np.ones((7, 2))
    clouds, ps, pt = _pair_lists(a, [b, c])                              # one source shared by every pair (slam.py:576-579)
    assert len(clouds) == 3 and list(ps) == [0, 0] and list(pt) == [1, 2]
    clouds, ps, pt = _pair_lists([a, b], [b, c])
    assert len(clouds) == 4 and list(ps) == [0, 1] and list(pt) == [2, 3]
    with pytest.raises(ValueError):
        _pair_lists([a], [b, c])


def test_cell_box_of_a_scan_matches_the_array_expression():
    """The live update_scan takes the cell box of a scan on the host: min / max of the rows on a transposed copy and the
    box in scalar arithmetic.  Both must equal the plain NumPy expressions they replace (same IEEE operations)."""
    import types
    from utilities import mapping
    rng = np.random.default_rng(5)
    grid = types.SimpleNamespace(min_x=-56.05, min_y=-60.0, resolution=0.05)
    for trial in range(300):
        scale = 10.0 ** rng.integers(-3, 11)
        rows = rng.normal(size=(int(rng.integers(1, 400)), 2)) * scale
        if trial % 50 == 7:
            rows[rng.integers(0, len(rows)), rng.integers(0, 2)] = rng.choice([np.nan, np.inf, -np.inf])
        lo, hi = mapping._minmax_rows(rows)
        assert np.array_equal(lo, rows.min(axis=0), equal_nan=True) and np.array_equal(hi, rows.max(axis=0), equal_nan=True)
        got = mapping.OccupancyGrid2D._box_of(grid, lo, hi)
        lo_hi = np.array([rows.min(axis=0), rows.max(axis=0)])
        if not np.isfinite(lo_hi).all():
            assert got is None
            continue
        c = np.clip(np.floor((lo_hi - np.array([grid.min_x, grid.min_y])) / grid.resolution), -2.0 ** 29, 2.0 ** 29)
        want = np.array([c[0, 0] - 1, c[0, 1] - 1, c[1, 0] + 1, c[1, 1] + 1], dtype=np.int32)
        assert got.dtype == np.int32 and np.array_equal(got, want), trial
