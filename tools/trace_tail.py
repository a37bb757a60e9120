#!/usr/bin/env python3
"""Print the last N kernel dispatches of a rocprofv3 kernel trace CSV: start/end relative to the first of them (us), queue, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f}  q{r.get("Queue_Id", "?"):>3} s{r.get("Stream_Id", "?"):>3} grid {int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1):5d}  {r["Kernel_Name"][:60]}')
