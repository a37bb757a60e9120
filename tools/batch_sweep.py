#!/usr/bin/env python3
"""Run bench.py at several per-GPU batch sizes and print value / step / kernel time / roofline fraction.
usage: python tools/batch_sweep.py 512 1024 2048 ..."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096]:
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-raycast", "--no-extras",
                          "--pairs-per-gpu", str(b)], capture_output=True, text=True)
    if out.returncode != 0:
        print(b, "failed:", out.stderr[-500:])
        continue
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    print(f"pairs/GPU {b}: {d['value']:.0f} it/s, step {d['ms_per_step']:.3f} ms, fused ICP {r['kernel_ms']:.3f} ms, "
          f"roofline frac {r['frac']:.4f}, iterations/step {d['config']['iterations_per_step']:.0f}", flush=True)
