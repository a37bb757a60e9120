#!/usr/bin/env python3
"""Diagnostic: a large differential sweep of the fused path against the exhaustive kernel (and the oracle on a sample): B pairs
per geometry and method, transforms / errors / iterations of EVERY pair compared.  usage: diag_sweep.py [B]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np
import oracle
from icpmi import batch, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bad_total = 0
for (off, yaw, seed0) in ((0.6, 6.0, 300000), (1.5, 12.0, 310000), (3.0, 20.0, 320000)):
    srcs, tgts = synth.loop_closure_batch(B, seed0=seed0, max_offset=off, max_yaw_deg=yaw)
    for method, mcd, maxit in (("point_to_line", None, 150), ("point_to_point", None, 60), ("point_to_point", 0.6, 60), ("point_to_line", 0.4, 60)):
        t0 = time.time()
        Rf, tf, ef, i_f = batch.icp_batch(srcs, tgts, 1e-10, maxit, 0.04, None, None, method, 12, mcd)
        Rx, tx, ex, i_x = batch.icp_batch(srcs, tgts, 1e-10, maxit, 0.04, None, None, method, 12, mcd, force_exhaustive=True)
        d = np.sqrt(((Rf - Rx) ** 2).sum(axis=(1, 2)) + ((tf - tx) ** 2).sum(axis=1))
        it_bad = np.flatnonzero((i_f["iters"] != i_x["iters"]) | (i_f["status"] != i_x["status"]))
        e_bad = np.flatnonzero(~((np.abs(ef - ex) <= 1e-9 * np.maximum(1.0, np.abs(ex))) | (np.isinf(ef) & np.isinf(ex))))
        bad = np.union1d(np.union1d(np.flatnonzero(d > 1e-9), it_bad), e_bad)
        # the oracle on a sample
        ob = 0
        for i in range(0, B, max(1, B // 24)):
            Ro, to, eo, io = oracle.icp(srcs[i], tgts[i], 1e-10, maxit, 0.04, method=method, normal_k=12, max_corr_dist=mcd)
            if int(i_f["iters"][i]) != io["iters"] or np.sqrt(((Rf[i] - Ro) ** 2).sum() + ((tf[i] - to) ** 2).sum()) > 1e-9:
                ob += 1
        print(f"offset {off} yaw {yaw} {method} max_corr {mcd}: {B} pairs, max |fast - exhaustive| {d.max():.2e}, differing pairs {len(bad)} {bad[:8]}, "
              f"oracle sample mismatches {ob}, at the limit {(i_f['iters'] == maxit).sum()}, {time.time() - t0:.1f} s", flush=True)
        bad_total += len(bad) + ob
print("TOTAL differing:", bad_total)
sys.exit(1 if bad_total else 0)
