#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (SQ instruction counters) per kernel and grid size.

  python tools/pmc_instruction_mix.py <passA>_counter_collection.csv <passB>_counter_collection.csv > profiles/rNN_pmc_instruction_mix.json

pass A: SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
pass B: SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES
(each with --kernel-trace only, on `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`).
"""
import csv
import json
import sys
from collections import defaultdict


def collect(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            if "icpmi" not in r["Kernel_Name"]:
                continue
            wg = int(r["Workgroup_Size"])
            key = f'{r["Kernel_Name"].split("(")[0]} [{int(r["Grid_Size"]) // max(wg, 1)} workgroups]'
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main(a_csv, b_csv, keep):
    a, b = collect(a_csv), collect(b_csv)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {"how": "rocprofv3 --kernel-trace --pmc <6 SQ counters> (two passes) -- python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline; values per dispatch, averaged over the dispatches of a kernel at one grid size; "
                  "instruction counters count wave-level instructions.  derived.valu_issue_frac = SQ_ACTIVE_INST_VALU "
                  "(quad-cycles in which a SIMD issued a vector instruction, summed over the chip) x 4 / (1024 SIMDs x "
                  "SQ_BUSY_CYCLES / 32 shader engines): the share of the chip's vector issue cycles the kernel used",
           "csrc_sha256": bench.csrc_signature(),
           "kernels": {}}
    for k in sorted(a):
        if keep and not any(t in k for t in keep):
            continue
        d = dict(a[k])
        d.update(b.get(k, {}))
        valu, salu, lds, br = (d.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"))
        e = {c: round(v, 1) for c, v in d.items()}
        tot = valu + salu + lds
        e["derived"] = {
            "valu_share_of_valu_salu_lds": round(valu / tot, 3) if tot else None,
            "salu_share": round(salu / tot, 3) if tot else None,
            "branches_per_100_valu": round(100 * br / valu, 1) if valu else None,
            "active_lanes_per_valu_instruction": round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"], 1)
            if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_THREAD_CYCLES_VALU") else None,
            "valu_issue_frac": round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * d["SQ_BUSY_CYCLES"] / 32.0), 4)
            if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_BUSY_CYCLES") else None,
            "fp64_add_mul_share_of_valu": round((d.get("SQ_INSTS_VALU_ADD_F64", 0) + d.get("SQ_INSTS_VALU_MUL_F64", 0)) / valu, 3)
            if valu else None}
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
