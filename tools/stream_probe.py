#!/usr/bin/env python3
"""Diagnostic: do k torch streams run k small ICP batches concurrently? (HIP maps streams onto a few hardware queues)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import synth
from icpmi.batch import IcpBatch
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
B = 170
srcs, tgts = synth.loop_closure_batch(6 * B, seed0=5000)
bs = [IcpBatch(srcs[k * B:(k + 1) * B] + tgts[k * B:(k + 1) * B], np.arange(B), np.arange(B, 2 * B), **kw) for k in range(6)]
for kind in ("torch.cuda.Stream()", "priority=-1", "every second of 8"):
    if kind == "torch.cuda.Stream()":
        pool = [torch.cuda.Stream() for _ in range(6)]
    elif kind == "priority=-1":
        pool = [torch.cuda.Stream(priority=-1) for _ in range(6)]
    else:
        pool = [torch.cuda.Stream() for _ in range(12)][::2]
    for n in (1, 2, 3, 4, 6):
        for _ in range(3):
            for k in range(n):
                with torch.cuda.stream(pool[k]):
                    bs[k].run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            for k in range(n):
                with torch.cuda.stream(pool[k]):
                    bs[k].run()
        torch.cuda.synchronize()
        print(f"{kind}: {n} batches of {B} pairs on {n} streams: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per round")
