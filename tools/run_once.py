#!/usr/bin/env python3
"""Diagnostic: the hot path (voxel x2 + prepare + fused ICP) on B synthetic pairs, a few times — the command to
put under `rocprofv3 --kernel-trace --pmc ...` when only the three kernels of a step are wanted.
usage: run_once.py B [repeats]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import _lib
if os.environ.get("ICPMI_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["ICPMI_LIB"])
from icpmi import synth
from icpmi.batch import IcpBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = dict(error_threshold=1e-10, max_iterations=int(os.environ.get("MAXIT", "150")), voxel_size=0.04, method="point_to_line", normal_k=12)
srcs, tgts = synth.loop_closure_batch(B, seed0=1000)
b = IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), **kw)
for _ in range(reps):
    b.run()
torch.cuda.synchronize()
print("iterations", int(b.results.cpu().numpy()[:B, 14].sum()))
