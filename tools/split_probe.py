#!/usr/bin/env python3
"""Diagnostic: one batch of B pairs against the same pairs as P parts on P streams (each part: voxel -> prepare -> ICP). usage: split_probe.py B [P...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import synth
from icpmi.batch import IcpBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
parts = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
srcs, tgts = synth.loop_closure_batch(B, seed0=1000)
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
ref = None
for P in parts:
    edges = [B * i // P for i in range(P + 1)]
    bs = [IcpBatch(srcs[a:b] + tgts[a:b], np.arange(b - a), np.arange(b - a, 2 * (b - a)), **kw) for a, b in zip(edges, edges[1:])]
    streams = [torch.cuda.Stream(priority=-1) for _ in range(P)] if P > 1 else [torch.cuda.current_stream()]
    def step():
        if P == 1:
            bs[0].run(); return
        ev = torch.cuda.Event(); ev.record()
        for b, s in zip(bs, streams):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                b.run()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    res = np.vstack([b.results.cpu().numpy()[:b.B] for b in bs])
    if ref is None: ref = res
    print(f"B={B} in {P} part(s): {ms:.3f} ms per step; results equal to one batch: {np.array_equal(ref, res)}")
