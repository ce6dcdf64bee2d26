#!/bin/bash
# kernel trace of LM steps with the blocked dense solver forced (Ladybug shape, 10 326 unknowns): gpurun_out/prof_big/
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_big
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_big -o big -- python3 tools/big_dense_time.py > gpurun_out/prof_big.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_big/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:56].ljust(56), r['Calls'].rjust(6), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(9), ('%.2f ms' % (float(r['TotalDurationNs']) / 1e6)).rjust(12), r['Percentage'].rjust(7))
PY
tail -1 gpurun_out/prof_big.log
