#!/usr/bin/env python3
"""Developer: from a rocprofv3 kernel trace, print start/end of the big-dense kernels of a few block columns."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_big" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[40:64]:
    print("%-14s q%-3s start %9.1f us  dur %7.1f us  grid %s" % (r["Kernel_Name"].split("(")[0][4:], r.get("Queue_Id", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
