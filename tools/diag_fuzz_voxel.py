#!/usr/bin/env python3
"""Diagnostic: the voxel filter on random clouds of many sizes and extents against the oracle, bit for bit — single clouds
(1 024 threads: the register path for 1 025..2 048 and 2 049..4 096 rows) and sets of > 256 clouds (512 threads: the register
path for 513..1 024 and 1 025..2 048 rows); 2-D and 3-D; lattices, duplicates, one voxel.  usage: diag_fuzz_voxel.py [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np
import oracle
from icpmi import batch

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def cloud(n, dim):
    kind = rng.integers(0, 5)
    ext = 10.0 ** rng.uniform(-2, 2.5)
    if kind == 0:
        p = rng.uniform(-ext, ext, size=(n, dim))
    elif kind == 1:
        p = rng.normal(scale=ext, size=(n, dim)) + rng.uniform(-1e3, 1e3, size=dim)
    elif kind == 2:                                            # lattice: points on cell boundaries
        p = rng.integers(-40, 40, size=(n, dim)) * 0.25
    elif kind == 3:                                            # duplicates
        p = np.repeat(rng.uniform(-ext, ext, size=(max(n // 5, 1), dim)), 5, axis=0)[:n]
        if len(p) < n:
            p = np.vstack([p, p[: n - len(p)]])
    else:                                                      # a wall and noise
        t = rng.uniform(-ext, ext, size=n)
        p = np.stack([t, 0.3 * t + rng.normal(scale=0.01, size=n)] + ([rng.normal(scale=0.1, size=n)] if dim == 3 else []), axis=1)
    return np.ascontiguousarray(p)


bad = 0
sizes = [1, 2, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 1500, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 8191, 8192]
for dim in (2, 3):
    for n in sizes:
        for rep in range(3):
            p = cloud(n, dim)
            v = float(10.0 ** rng.uniform(-2.2, 0.3)) if rep else 0.25
            got = batch.voxel_downsample_set(batch.CloudSet.from_numpy([p]), v).to_numpy()[0]
            ref = oracle.voxel_downsample(p, v)
            if got.shape != ref.shape or not np.array_equal(got, ref):
                bad += 1
                print("MISMATCH single", dim, n, v, got.shape, ref.shape)
    # a set of 300 clouds (512-thread workgroups)
    clouds = [cloud(int(rng.choice([300, 600, 1000, 1024, 1025, 1600, 2048, 2049, 3000])), dim) for _ in range(300)]
    v = 0.07
    out = batch.voxel_downsample_set(batch.CloudSet.from_numpy(clouds), v).to_numpy()
    for i, (c, o) in enumerate(zip(clouds, out)):
        ref = oracle.voxel_downsample(c, v)
        if o.shape != ref.shape or not np.array_equal(o, ref):
            bad += 1
            print("MISMATCH set", dim, i, len(c))
print("voxel fuzz mismatches:", bad)
sys.exit(1 if bad else 0)
