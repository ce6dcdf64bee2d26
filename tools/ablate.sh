#!/bin/bash
# developer timing: fused-kernel time with parts of k_build switched off (results invalid)
for a in 0 1 8; do
  echo -n "BA_ABLATE=$a  "
  BA_ABLATE=$a timeout -k 10 120 python -u bench.py --steps 5 --warmup 1 --lm-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])" || exit 1
done
