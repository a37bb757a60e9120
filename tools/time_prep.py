#!/usr/bin/env python3
"""Diagnostic: prepare_targets kernel time with and without normals, and the voxel kernel. usage: time_prep.py B [B...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import synth, _lib
if os.environ.get("ICPMI_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["ICPMI_LIB"])
from icpmi.batch import IcpBatch, _ptr, _stream, voxel_downsample_set
L = _lib.lib()
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
Bs = [int(a) for a in sys.argv[1:]] or [1]
srcs, tgts = synth.loop_closure_batch(max(Bs), seed0=1000)
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in Bs:
    b = IcpBatch(srcs[:B] + tgts[:B], np.arange(B), np.arange(B, 2 * B), **kw)
    b.run(); torch.cuda.synchronize()
    def prep(k):
        _lib.check(L.icpmi_prepare_targets_ex(_ptr(b.vox.pts), _ptr(b.vox.off), None, _ptr(b.vox.cnt), _ptr(b.tgt_ids_dev), None,
                   len(b.tgt_ids), b.raw.n_clouds, b.raw.total_rows, b.max_tgt_n, k, None, _ptr(b.prepared), b.prepared.numel(),
                   1, _stream()), "prep")     # sort order as IcpBatch.run asks for it (ICPMI_POLAR=0: projections only)
    print(f"B={B}: voxel {timed(lambda: voxel_downsample_set(b.raw, 0.04, out=b.vox, workspace=b.vox_ws)):.3f} ms  "
          f"prep(no normals) {timed(lambda: prep(-1)):.3f} ms  prep(k=5) {timed(lambda: prep(5)):.3f} ms  prep(k=12) {timed(lambda: prep(12)):.3f} ms")
