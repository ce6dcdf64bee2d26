#!/bin/bash
# developer: time of the dense Cholesky kernel on the Temple-shaped problem
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_dense -o t -- python3 tools/solve_config.py ${1:-temple} --verbose 0 > gpurun_out/trace_dense.log 2>&1 || exit 1
grep "k_dense_cholesky" gpurun_out/trace_dense/t_kernel_stats.csv | cut -d, -f2-4
grep "status" gpurun_out/trace_dense.log | cut -c1-160
