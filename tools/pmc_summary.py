"""developer: mean per dispatch of every counter of one kernel, from rocprofv3 --pmc CSV output.

usage: python tools/pmc_summary.py <kernel name fragment> <out.json> <dir or *_counter_collection.csv> [...]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    frag, out = sys.argv[1], sys.argv[2]
    files = []
    for p in sys.argv[3:]:
        files += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(p) else [p]
    acc = collections.defaultdict(list)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if frag in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for k, v in sorted(acc.items())}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
