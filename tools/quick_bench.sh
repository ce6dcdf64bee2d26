#!/bin/bash
# developer: kernel time only
timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --lm-steps ${1:-0} --no-cpu-baseline --no-other-configs --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'], 'value %.3e' % d['value'], 'lm/s', d['lm_iters_per_sec'], 'pcg', d['lm_pcg_iterations'])"
