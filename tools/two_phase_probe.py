#!/usr/bin/env python3
"""Diagnostic: would splitting the fused ICP launch into "everybody up to K iterations" + "the unfinished pairs to the
end" shorten it?  Times both parts with the existing kernel (the second part redoes the first K iterations, so it is
an upper bound).  usage: two_phase_probe.py [B] [K]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import synth
from icpmi.batch import IcpBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
srcs, tgts = synth.loop_closure_batch(B, seed0=1000)
def kernel_ms(b, n=5):
    for _ in range(2):
        b.run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e in ev:
        b.run(events=e)
    torch.cuda.synchronize()
    return float(np.mean([x.elapsed_time(y) for x, y in ev]))
kw = dict(error_threshold=1e-10, voxel_size=0.04, method="point_to_line", normal_k=12)
full = IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), max_iterations=150, **kw)
t_full = kernel_ms(full)
it = full.results.cpu().numpy()[:B, 14]
print(f"B={B}: one launch {t_full:.3f} ms; iterations: sum {int(it.sum())}, pairs at 150: {(it == 150).sum()}")
for K in [int(a) for a in sys.argv[2:]] or [10, 12, 16, 24]:
    first = IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), max_iterations=K, **kw)
    t1 = kernel_ms(first)
    left = np.flatnonzero(it > K)
    rest = IcpBatch([srcs[i] for i in left] + [tgts[i] for i in left], np.arange(len(left)), np.arange(len(left), 2 * len(left)),
                    max_iterations=150, **kw)
    t2 = kernel_ms(rest)
    print(f"  K={K}: all pairs to {K} iterations {t1:.3f} ms + {len(left)} unfinished pairs to the end {t2:.3f} ms = {t1 + t2:.3f} ms")
