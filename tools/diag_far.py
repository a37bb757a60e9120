#!/usr/bin/env python3
"""Diagnostic (needs `make diag`): per-phase cycles of the slowest pairs of a pre-aligned 3 m / 20 deg batch."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
from icpmi import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libicpmi.so", "libicpmi_diag.so")
import numpy as np, torch
from icpmi import synth
from icpmi.prealign import RunIcpPairBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
srcs, tgts = synth.loop_closure_batch(B, seed0=7000, shared_source=True, max_offset=3.0, max_yaw_deg=20.0)
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
b = RunIcpPairBatch([srcs[0]] + tgts, np.zeros(B, dtype=np.int32), np.arange(1, B + 1, dtype=np.int32),
                    rotation_voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1, max_rows_hint=1024, **kw)
b.run(); torch.cuda.synchronize()
import ctypes
raw = ctypes.CDLL(_lib.LIB_PATH)
dbg = (ctypes.c_ulonglong * 16)()
raw.icpmi_diag_read(dbg)
b.run(); torch.cuda.synchronize()
raw.icpmi_diag_read(dbg)
d = list(dbg)
print(f"sweepf_top2_far over the batch: {d[5]} calls, {d[6]} far scans; cycles per call: query setup {d[0]/max(d[5],1):.0f}, walk {d[1]/max(d[5],1):.0f}; "
      f"per far scan: {d[3]/max(d[6],1):.0f}")
r = b.icp.results.cpu().numpy()[:B]
cnt = b.icp.vox.cnt.cpu().numpy()
N = cnt[0]
dirs = b.icp.prepared[b.icp.raw.total_rows * 40:].view(torch.int32)[:B + 1].cpu().numpy()
it = np.maximum(r[:, 14], 1)
tot = r[:, 4] + r[:, 5] + r[:, 6] + r[:, 7]
for i in np.argsort(-tot)[:10]:
    print(f"pair {i}: iters={int(r[i,14])} err={r[i,12]:.3f} total Mcycles={tot[i]/1e6:.2f} per-iter search={r[i,4]/it[i]:.0f} partials={r[i,5]/it[i]:.0f} lead={r[i,6]/it[i]:.0f} "
          f"apply={r[i,7]/it[i]:.0f} searches/row-iter={r[i,8]/(it[i]*N):.3f} far scans/search={(r[i,11] // 2**32)/max(r[i,8],1):.3f} "
          f"dir={int(dirs[1+i])} M={cnt[1+i]}")
bad = r[:, 12] >= 0.05
print("bad pairs:", bad.sum(), "median total Mcycles bad", np.median(tot[bad]) / 1e6, "good", np.median(tot[~bad]) / 1e6)
