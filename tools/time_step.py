#!/usr/bin/env python3
"""Diagnostic: where a step of B pairs spends its time — HIP events between the launches of IcpBatch.run(), and the
wall time of back-to-back steps.  usage: time_step.py B [B...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import synth, batch as B_
from icpmi.batch import IcpBatch
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
Bs = [int(a) for a in sys.argv[1:]] or [512]
srcs, tgts = synth.loop_closure_batch(max(Bs), seed0=5000)
for B in Bs:
    b = IcpBatch(srcs[:B] + tgts[:B], np.arange(B), np.arange(B, 2 * B), **kw)
    for _ in range(3):
        b.run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    # events: before voxel | (events arg of run: around the ICP launch)
    tot = []
    for _ in range(10):
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        e0.record()
        b.run(events=(e1, e2))
        e3.record()
        torch.cuda.synchronize()
        tot.append((e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)))
    tot = np.array(tot)
    t0 = time.perf_counter()
    for _ in range(20):
        b.run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        b.run()
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"B={B}: voxel+prep {tot[:,0].mean():.3f} ms, icp {tot[:,1].mean():.3f} ms, tail {tot[:,2].mean():.3f} ms; "
          f"back-to-back step {wall*1e3:.3f} ms; host enqueue per step {host*1e3:.3f} ms")
