#!/usr/bin/env python3
"""Diagnostic: time of batches made ONLY of pairs that run all 150 iterations (the limit-cycling pairs of the bench batch),
N pairs at a time: N = 256 is one workgroup per CU, 512 two per CU, ... — what a tail iteration costs alone and shared.
usage: [ICPMI_ICP2_SHAPE=768x2] time_tail.py N [N ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import _lib
if os.environ.get("ICPMI_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["ICPMI_LIB"])
from icpmi import synth
from icpmi.batch import IcpBatch
Ns = [int(a) for a in sys.argv[1:]] or [256, 512]
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
P = 4096
srcs, tgts = synth.loop_closure_batch(P, seed0=1000)
b = IcpBatch(srcs + tgts, np.arange(P), np.arange(P, 2 * P), **kw)
b.run(); torch.cuda.synchronize()
it = b.results.cpu().numpy()[:P, 14]
long_ids = np.flatnonzero(it == 150)
print(f"{len(long_ids)} of {P} pairs run 150 iterations")
for N in Ns:
    ids = np.resize(long_ids, N)
    bb = IcpBatch([srcs[i] for i in ids] + [tgts[i] for i in ids], np.arange(N), np.arange(N, 2 * N), **kw)
    for _ in range(2):
        bb.run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for e in ev:
        bb.run(events=e)
    torch.cuda.synchronize()
    t = np.mean([x.elapsed_time(y) for x, y in ev])
    its = bb.results.cpu().numpy()[:N, 14]
    print(f"N={N}: icp {t:.3f} ms, {t * 1e3 / 150:.2f} us per iteration of the batch, iterations min={int(its.min())}")
