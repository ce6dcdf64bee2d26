#!/bin/bash
# Developer: time k_build for ablation builds of the library (libba_abl_<X>.so; timing only, results are wrong by construction)
cd "$(dirname "$0")/.."
cp structure_from_motion_amd/libba_hip.so /tmp/keep.so
for rep in 1 2; do
for v in base CAM PT SCHUR FLUSH; do
  if [ $v = base ]; then cp /tmp/keep.so structure_from_motion_amd/libba_hip.so; else cp structure_from_motion_amd/libba_abl_$v.so structure_from_motion_amd/libba_hip.so; fi
  echo -n "$v: "; BA_NO_WARM_UP=1 timeout -k 10 120 python tools/build_time.py venice1778 2>&1 | tail -1 | sed 's/.*pass/pass/'
done; done
cp /tmp/keep.so structure_from_motion_amd/libba_hip.so
