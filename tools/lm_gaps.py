#!/usr/bin/env python3
"""Developer: from a rocprofv3 kernel trace of LM steps (tools/trace_lm.sh), the timeline of ONE step: every kernel with its
duration and the idle gap in front of it.  usage: lm_gaps.py <kernel_trace.csv> [step index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_build<" in r["Kernel_Name"]]
# the LM steps are the builds that are followed by a k_backsub before the next build
steps = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    if any("k_backsub" in r["Kernel_Name"] for r in rows[a:b]):
        steps.append((a, b))
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 2
a, b = steps[k]
prev_end = int(rows[a - 1]["End_Timestamp"]) if a else int(rows[a]["Start_Timestamp"])
tot_busy = tot_gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    print("%-28s dur %7.1f us   gap before %6.1f us" % (r["Kernel_Name"].split("(")[0].replace("void ba::", "").replace("ba::", "")[:28], (e - s) / 1e3, gap / 1e3))
    tot_busy += e - s; tot_gap += max(gap, 0); prev_end = max(prev_end, e)
print("step %d of %d: kernels %.1f us, gaps %.1f us" % (k, len(steps), tot_busy / 1e3, tot_gap / 1e3))
