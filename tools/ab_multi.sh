#!/bin/bash
# Developer: k_build / LM rate of several builds of the library (tools/lib_<name>.so) on one box, alternating
cd "$(dirname "$0")/.."
cp structure_from_motion_amd/libba_hip.so /tmp/keep.so
for rep in 1 2; do
  for f in tools/lib_*.so; do
    cp $f structure_from_motion_amd/libba_hip.so
    echo -n "$(basename $f): "; bash tools/quick_bench.sh ${1:-0} || exit 1
  done
done
cp /tmp/keep.so structure_from_motion_amd/libba_hip.so
