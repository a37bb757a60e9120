#!/usr/bin/env python3
"""Diagnostic: batched _run_icp_pair (rotation search + ICP) on B loop-closure candidates. usage: time_prealign.py B [max_offset max_yaw_deg]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
import numpy as np, torch
from icpmi import _lib
if os.environ.get("ICPMI_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["ICPMI_LIB"])
from icpmi import synth
from icpmi.prealign import RunIcpPairBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
off = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
yaw = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
srcs, tgts = synth.loop_closure_batch(B, seed0=7000, shared_source=True, max_offset=off, max_yaw_deg=yaw)
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
b = RunIcpPairBatch([srcs[0]] + tgts, np.zeros(B, dtype=np.int32), np.arange(1, B + 1, dtype=np.int32),
                    rotation_voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1, max_rows_hint=1024, **kw)
for _ in range(3):
    b.run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(); b.search.run(); e[1].record(); b.icp.run(events=(e[2], e[3])); torch.cuda.synchronize()
    ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])))
ts = np.array(ts)
res = b.icp.results.cpu().numpy()[:B]
rec = b.search.records.cpu().numpy()[:B]
print(f"B={B} offset<={off} yaw<={yaw}: search {ts[:,0].mean():.3f} ms, voxel+prepare {ts[:,1].mean():.3f} ms, icp {ts[:,2].mean():.3f} ms; "
      f"registered {(res[:,12] < 0.05).mean():.3f}, iterations {int(res[:,14].sum())} (n150 {(res[:,14] == 150).sum()}), "
      f"exact angles coarse {rec[:,12].mean():.1f} (max {int(rec[:,12].max())}, >40: {(rec[:,12] > 40).sum()}) fine {rec[:,13].mean():.1f}, fine score median {np.median(rec[:,10]):.4f}")
if os.environ.get("RSB_TIMES"):
    print("cycles per phase (stage, field, bounds, order, coarse, fine): median", np.median(rec[:, :6], axis=0).astype(int), "max", rec[:, :6].max(axis=0).astype(int))
if os.environ.get("SAVE_REC"):
    np.save(os.environ["SAVE_REC"], np.hstack([rec, res]))
