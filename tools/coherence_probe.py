#!/usr/bin/env python3
"""Diagnostic: does the fused ICP kernel run faster when the source rows of a pair come in bearing order (neighbouring
lanes then search neighbouring parts of the target) instead of the voxel filter's key order?  Permutes the filtered
source rows on the device between the prepare and the ICP launch.  usage: coherence_probe.py [B]"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import synth, _lib
from icpmi.batch import IcpBatch, _ptr, _stream, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
srcs, tgts = synth.loop_closure_batch(B, seed0=1000)
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
b = IcpBatch(srcs + tgts, np.arange(B), np.arange(B, 2 * B), **kw)
L = _lib.lib()
def icp_only(n=5):
    def go():
        check(L.icpmi_icp_batch(_ptr(b.vox.pts), _ptr(b.vox.off), _ptr(b.vox.cnt), _ptr(b.normals), _ptr(b.prepared),
                                _ptr(b.pair_src), _ptr(b.pair_tgt), b.B, b.max_src_n, b.max_tgt_n, b.raw.total_rows,
                                C.byref(b.params), _ptr(b.init), _ptr(b.results), _ptr(b.icp_ws),
                                b.icp_ws.numel() if b.icp_ws is not None else 0, _stream()), "ICP")
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        go()
    e1.record(); torch.cuda.synchronize()
    it = b.results.cpu().numpy()[:B, 14]
    return e0.elapsed_time(e1) / n, int(it.sum())
b.run(); torch.cuda.synchronize()
print("voxel-key order   : %.3f ms, iterations %d" % icp_only())
pts = b.vox.pts.view(-1, 2)
off = b.vox.off.long(); cnt = b.vox.cnt.long()
n_clouds = cnt.numel()
rows = torch.arange(pts.shape[0], device=pts.device)
cloud = torch.searchsorted(off[1:].contiguous(), rows, right=True).clamp(max=n_clouds - 1)
valid = rows - off[cloud] < cnt[cloud]
for name, keyf in (("bearing order", lambda p: (torch.atan2(p[:, 1], p[:, 0]) + np.pi) / (2 * np.pi + 1e-9)),
                   ("range order", lambda p: torch.clamp(torch.hypot(p[:, 0], p[:, 1]) / 100.0, max=0.999))):
    key = cloud.double() * 4 + torch.where(valid, keyf(pts), torch.full_like(pts[:, 0], 2.0))
    perm = torch.argsort(key)
    b.vox.pts.view(-1, 2).copy_(pts[perm].clone())
    print("%-18s: %.3f ms, iterations %d" % ((name,) + icp_only()))
    b.run(); torch.cuda.synchronize()      # restore the filter's own order
    pts = b.vox.pts.view(-1, 2)
