#!/bin/bash
# developer: LDS counters of k_build for the bench workload
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 2 --lm-steps 0 --no-cpu-baseline --no-other-configs --no-live-traffic"
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lds -o pmc -- python3 $ARGS > gpurun_out/pmc_lds.log 2>&1 || exit 1
python3 - <<PY
import csv,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_lds/pmc_counter_collection.csv')):
    if 'k_build<' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(x)/len(x)/1e6,1) for k,x in acc.items()})
PY
