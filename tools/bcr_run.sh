#!/bin/bash
# correctness + timing sweep of tools/bcr_bench (built beforehand)
set -e
mkdir -p gpurun_out
for args in "40 3 0" "40 3 1" "100 5 0" "101 5 1" "300 15 0" "301 16 1" "64 16 1" "1776 15 0" "1776 16 0" "1777 16 1" "1776 20 1" "1776 8 0" "1776 2 0" "10000 16 0" "300 7 0 3 9" "1776 10 0 5 4" "120 5 1 3 7"; do
  timeout -k 10 60 tools/bcr_bench $args || { echo "FAILED: $args (exit $?)"; exit 1; }
done
