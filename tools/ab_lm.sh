#!/bin/bash
# Developer: LM iterations per second of two builds of the library on the same box, alternating (libba_old.so = the build to compare with)
cd "$(dirname "$0")/.."
cp structure_from_motion_amd/libba_hip.so /tmp/new.so
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp structure_from_motion_amd/libba_old.so structure_from_motion_amd/libba_hip.so; else cp /tmp/new.so structure_from_motion_amd/libba_hip.so; fi
    echo -n "$which: "; bash tools/quick_bench.sh 20 || exit 1
  done
done
cp /tmp/new.so structure_from_motion_amd/libba_hip.so
