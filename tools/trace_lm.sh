#!/bin/bash
# kernel trace of a few LM steps on the bench workload: per-kernel averages (gpurun_out/prof_lm/)
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_lm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lm -o lm -- python3 bench.py --steps 5 --warmup 2 --lm-steps 10 --no-cpu-baseline --no-other-configs --no-live-traffic > gpurun_out/prof_lm.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_lm/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:56].ljust(56), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(9), r['Percentage'].rjust(7))
PY
grep '^{"metric"' gpurun_out/prof_lm.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g obs/s, ms_per_step %.4f, lm_iters_per_sec %.1f' % (d['value'], d['ms_per_step'], d['lm_iters_per_sec']))"
