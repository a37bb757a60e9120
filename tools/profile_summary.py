#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the summaries kept under profiles/.

  by-grid table:   python tools/profile_summary.py trace <dir>/<prefix>_kernel_trace.csv > profiles/rNN_bench_by_grid.md
  PMC traffic:     python tools/profile_summary.py pmc <fetch>_counter_collection.csv <write>_counter_collection.csv > profiles/rNN_pmc_traffic.json
                   (records the signature of the kernel sources, bench.csrc_signature(): bench.py quotes the summary as
                   `roofline.traffic` only while the library it runs was built from the same sources)

The PMC passes are separate rocprofv3 runs (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with --kernel-trace only).
Counters are KiB per dispatch; FETCH_SIZE is doubled for the 16-B-per-lane coalesced reads of these kernels, as
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes; WRITE_SIZE is taken as read.
"""
import csv
import json
import sys
from collections import defaultdict


def by_grid(path):
    rows = defaultdict(list)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "icpmi" not in name and "rocprim" not in name:
                continue
            wg = int(r["Workgroup_Size_X"])
            key = (name[:60], int(r["Grid_Size_X"]) // max(wg, 1), wg)
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[key] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    print("| kernel | workgroups | wg size | VGPR | LDS B | scratch B | calls | avg us | min us | max us | total ms |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        v, l, s = meta[key]
        print(f"| {key[0]} | {key[1]} | {key[2]} | {v} | {l} | {s} | {len(d)} | {sum(d) / len(d):.1f} | {min(d):.1f} | {max(d):.1f} | {sum(d) / 1e3:.2f} |")


def pmc(fetch_csv, write_csv):
    def collect(path, counter):
        acc = defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter or "icpmi" not in r["Kernel_Name"]:
                    continue
                wg = int(r["Workgroup_Size"])
                acc[f'{r["Kernel_Name"]} [{int(r["Grid_Size"]) // max(wg, 1)} workgroups]'].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in acc.items()}

    fe, wr = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
    import os
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except OSError:
        commit = ""
    out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py "
                  "--steps 5 --warmup 1 --no-cpu-baseline; counters are KiB per dispatch, averaged over the "
                  "dispatches of a kernel at one grid size; FETCH_SIZE doubled for 16-B-per-lane coalesced reads as "
                  "MI355X_MICROARCH.md (HBM) prescribes; WRITE_SIZE as read",
           "csrc_sha256": bench.csrc_signature(), "summarised_at_commit": commit,
           "kernels": {}}
    for k in sorted(set(fe) & set(wr)):
        out["kernels"][k] = {"FETCH_SIZE_KiB": round(fe[k], 1), "WRITE_SIZE_KiB": round(wr[k], 1),
                             "hbm_bytes_per_launch": int(round((2.0 * fe[k] + wr[k]) * 1024))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "trace":
        by_grid(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
