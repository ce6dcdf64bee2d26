cd /root/repo
cp structure_from_motion_amd/libba_hip.so /tmp/new.so
for rep in 1 2; do for which in old new; do
  if [ $which = old ]; then cp structure_from_motion_amd/libba_old.so structure_from_motion_amd/libba_hip.so; else cp /tmp/new.so structure_from_motion_amd/libba_hip.so; fi
  echo -n "$which det: "; timeout -k 10 200 python tools/build_time.py venice1778 deterministic=1 | tail -1
  echo -n "$which rows orbit: "; timeout -k 10 200 python tools/build_time.py orbit1024 | tail -1
done; done
cp /tmp/new.so structure_from_motion_amd/libba_hip.so
