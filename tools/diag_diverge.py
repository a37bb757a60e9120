#!/usr/bin/env python3
"""Diagnostic: every pair of the pre-aligned 3 m / 20 deg batches — fused path, exhaustive path and the oracle — compared on
the TRANSFORM, the pairs that reach max_iterations included; a pair that differs is bisected over max_iterations to the
first iteration whose error differs (results are deterministic).  Usage: diag_diverge.py [B] [seed0] [max_iterations]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np
import oracle
from icpmi import batch, prealign, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 91000
MAXIT = int(sys.argv[3]) if len(sys.argv) > 3 else 150
METHOD = sys.argv[4] if len(sys.argv) > 4 else "point_to_line"
PRE = (sys.argv[5] if len(sys.argv) > 5 else "1") == "1"


def fro(R, t, Ro, to):
    return float(np.sqrt(((R - Ro) ** 2).sum() + ((t - to) ** 2).sum()))


srcs, tgts = synth.loop_closure_batch(B, seed0=seed0, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
if PRE:
    R0, t0, _ = prealign.rotation_search_batch(srcs[0], tgts, 0.15, 1.5, 0.1)
else:
    R0, t0 = None, None


def gpu(maxit, exhaustive, sel=None):
    tg = tgts if sel is None else [tgts[i] for i in sel]
    r0 = R0 if (sel is None or R0 is None) else R0[sel]
    tt0 = t0 if (sel is None or t0 is None) else t0[sel]
    return batch.icp_batch(srcs[0], tg, 1e-10, maxit, 0.04, r0, tt0, METHOD, 12, force_exhaustive=exhaustive)


def orc(i, maxit):
    kw = dict(R_init=R0[i], t_init=t0[i]) if PRE else {}
    return oracle.icp(srcs[0], tgts[i], 1e-10, maxit, 0.04, method=METHOD, normal_k=12, **kw)


Rf, tf, ef, inf_ = gpu(MAXIT, False)
Rx, tx, ex, inx = gpu(MAXIT, True)
bad = []
nmax = 0
for i in range(B):
    Ro, to, eo, io = orc(i, MAXIT)
    df, dx = fro(Rf[i], tf[i], Ro, to), fro(Rx[i], tx[i], Ro, to)
    at_max = io["iters"] == MAXIT
    nmax += at_max
    flag = df > 1e-9 or dx > 1e-9 or int(inf_["iters"][i]) != io["iters"] or int(inx["iters"][i]) != io["iters"]
    if flag or at_max:
        print(f"pair {i}: iters oracle/fast/exh = {io['iters']}/{int(inf_['iters'][i])}/{int(inx['iters'][i])} err={eo:.6f} "
              f"fro fast={df:.3e} exh={dx:.3e}" + ("   <-- DIFFERS" if flag else ""))
    if flag:
        bad.append(i)
print(f"{B} pairs, {nmax} at max_iterations, {len(bad)} differ: {bad}")
for i in bad[:4]:
    lo, hi = 1, MAXIT          # first k at which fast != oracle in err (1e-12 relative)
    def differs(k):
        R, t, e, info = gpu(k, False, [i])
        R2, t2, e2, info2 = gpu(k, True, [i])
        Ro, to, eo, io = orc(i, k)
        return (abs(e[0] - eo) > 1e-12 * max(1.0, abs(eo)) or fro(R[0], t[0], Ro, to) > 1e-9,
                abs(e2[0] - eo) > 1e-12 * max(1.0, abs(eo)) or fro(R2[0], t2[0], Ro, to) > 1e-9, e[0], e2[0], eo)
    while lo < hi:
        mid = (lo + hi) // 2
        if differs(mid)[0]:
            hi = mid
        else:
            lo = mid + 1
    for k in (lo - 1, lo, lo + 1):
        if k >= 1:
            print(f"  pair {i} max_iterations={k}: fast differs={differs(k)[0]} exhaustive differs={differs(k)[1]} "
                  f"err fast/exh/oracle = {differs(k)[2]!r} {differs(k)[3]!r} {differs(k)[4]!r}")
