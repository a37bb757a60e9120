#!/usr/bin/env python3
"""Diagnostic: where the host time of the NumPy-in / NumPy-out calls goes (cProfile over repeated calls)."""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
from icpmi import synth
from utilities import icp as uicp, features
uicp.VERBOSE = False; features.VERBOSE = False
a, b = synth.config2_pair(0)
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
def icp_call():
    return uicp.ICP(a, b, kw["error_threshold"], kw["max_iterations"], kw["voxel_size"], method=kw["method"], normal_k=kw["normal_k"])
def pair_call():
    R0, t0, _ = features.rotation_search(a, b, 0.15, 1.5, 0.1)
    return uicp.ICP(a, b, kw["error_threshold"], kw["max_iterations"], kw["voxel_size"], R_init=R0, t_init=t0, method=kw["method"], normal_k=kw["normal_k"])
for name, fn in (("ICP()", icp_call), ("rotation_search + ICP", pair_call)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
