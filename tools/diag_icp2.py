#!/usr/bin/env python3
"""Diagnostic only: per-phase cycle shares of one fused-ICP iteration.

Needs the diagnostic library (`make -C iterative-closest-point-avmi_amd/csrc diag`), whose
kernel stores s_memtime sums in spare result slots.  Not part of the product or the tests.
usage: python tools/diag_icp2.py [n_pairs]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd"))
from icpmi import _lib  # noqa: E402

_lib.LIB_PATH = _lib.LIB_PATH.replace("libicpmi.so", "libicpmi_diag.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from icpmi import synth  # noqa: E402
from icpmi.batch import IcpBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
srcs, tgts = synth.loop_closure_batch(max(B, 8), seed0=1000)
kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)
b = IcpBatch(srcs[:B] + tgts[:B], np.arange(B), np.arange(B, 2 * B), **kw)
b.run()
torch.cuda.synchronize()
r = b.results.cpu().numpy()[:B]
it = np.maximum(r[:, 14], 1)
for i in range(min(B, 8)):
    print(f"pair {i}: iters={int(r[i, 14])} cycles/iter search={r[i, 4] / it[i]:.0f} partials={r[i, 5] / it[i]:.0f} "
          f"lead(combine+solve)={r[i, 6] / it[i]:.0f} apply={r[i, 7] / it[i]:.0f}")
Nrows = b.vox.cnt.cpu().numpy()[:B]
print("searches per row-iteration (all pairs): %.4f; in the 150-iteration pairs: %s" % (
    (r[:, 8] / (it * Nrows)).mean(), np.round((r[:, 8] / (it * Nrows))[r[:, 14] == 150][:8], 4)))
print("mean cycles/iter: search=%.0f partials=%.0f lead=%.0f apply=%.0f" % tuple((r[:, k] / it).mean() for k in (4, 5, 6, 7)))
tot = r[:, 4] + r[:, 5] + r[:, 6] + r[:, 7]
order = np.argsort(-tot)[:8]
for i in order:
    print(f"slowest pair {i}: iters={int(r[i, 14])} total Mcycles={tot[i] / 1e6:.2f} per-iter search={r[i, 4] / it[i]:.0f} "
          f"partials={r[i, 5] / it[i]:.0f} lead={r[i, 6] / it[i]:.0f} apply={r[i, 7] / it[i]:.0f}")
