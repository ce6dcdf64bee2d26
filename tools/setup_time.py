#!/usr/bin/env python3
"""Developer: where ba_create spends its time on a large problem."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from structure_from_motion_amd import _capi, synthetic
p = synthetic.make_config(sys.argv[1] if len(sys.argv) > 1 else "venice1778")
arr = _capi.Arrays.from_dict(p)
for rep in range(3):
    t0 = time.perf_counter()
    h = _capi.Handle(arr); h.synchronize()
    t1 = time.perf_counter()
    h.build(1e-4); h.synchronize()
    t2 = time.perf_counter()
    s = h.run(1)
    print(f"ba_create {1e3*(t1-t0):.1f} ms, first build {1e3*(t2-t1):.1f} ms (t_setup_ms {s['t_setup_ms']:.1f})")
    h.close()
