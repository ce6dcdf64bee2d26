export TMPDIR=/tmp; mkdir -p gpurun_out/r03e
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03e/orbit -o orbit -- python3 tools/prof_run.py orbit1024 lm=6 > gpurun_out/r03e/orbit.log 2>&1
grep "build pass" gpurun_out/r03e/orbit.log
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/r03e/orbit/orbit_kernel_stats.csv")))[:8]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(9))
PY
