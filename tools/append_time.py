#!/usr/bin/env python3
"""Developer: ba_append against a new handle on the incremental driver's growth (Fountain / Temple shaped, H4 visibility)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from structure_from_motion_amd import _capi, synthetic
for name, ncam0 in (("fountain_sfm", 9), ("temple_sfm", 45)):
    p = synthetic.make_config(name)
    m = p["obs_cam"] < ncam0
    npts = int(p["obs_pt"][m].max()) + 1
    first = dict(cam_pose=p["cam_pose"][:ncam0], cam_intr=p["cam_intr"][:ncam0], cam_fixed=p["cam_fixed"][:ncam0], points=p["points"][:npts],
                 obs_cam=p["obs_cam"][m], obs_pt=p["obs_pt"][m], obs_xy=p["obs_xy"][m])
    a_first, a_full = _capi.Arrays.from_dict(first), _capi.Arrays.from_dict(p)
    best_app, best_new = 1e9, 1e9
    for rep in range(5):
        with _capi.Handle(a_first, loss="huber", f_scale=1.5, max_iter=150) as h:
            h.build(1e-4); h.synchronize()
            t0 = time.perf_counter(); ok = h.append(a_full); h.synchronize(); best_app = min(best_app, time.perf_counter() - t0)
        t0 = time.perf_counter()
        h2 = _capi.Handle(a_full, loss="huber", f_scale=1.5, max_iter=150); h2.synchronize(); best_new = min(best_new, time.perf_counter() - t0)
        h2.close()
    print(f"{name}: {int(m.sum())} -> {len(m)} observations, appended={ok}: ba_append {best_app*1e3:.2f} ms, new handle {best_new*1e3:.2f} ms")

# dense tracks, Ladybug-1723 shape: the last tenth of the street (cameras and the points that end there) arrives later
full = synthetic.make_config("ladybug1723")
nc = len(full["cam_pose"]); nc0 = int(nc * 0.9)
last = np.zeros(len(full["points"]), np.int64); np.maximum.at(last, full["obs_pt"], full["obs_cam"])
order = np.argsort(last, kind="stable"); inv = np.argsort(order)
pts = full["points"][order]; opt_ = inv[full["obs_pt"]]
n1 = int((last[order] < nc0).sum()); m1 = opt_ < n1
first = dict(cam_pose=full["cam_pose"][:nc0], cam_intr=full["cam_intr"][:nc0], cam_fixed=full["cam_fixed"][:nc0], points=pts[:n1],
             obs_cam=full["obs_cam"][m1], obs_pt=opt_[m1].astype(np.int32), obs_xy=full["obs_xy"][m1])
grown = dict(cam_pose=full["cam_pose"], cam_intr=full["cam_intr"], cam_fixed=full["cam_fixed"], points=pts,
             obs_cam=np.concatenate([full["obs_cam"][m1], full["obs_cam"][~m1]]),
             obs_pt=np.concatenate([opt_[m1], opt_[~m1]]).astype(np.int32),
             obs_xy=np.concatenate([full["obs_xy"][m1], full["obs_xy"][~m1]]))
a_first, a_full = _capi.Arrays.from_dict(first), _capi.Arrays.from_dict(grown)
best_app, best_new = 1e9, 1e9
for rep in range(3):
    with _capi.Handle(a_first) as h:
        h.build(1e-4); h.synchronize()
        t0 = time.perf_counter(); ok = h.append(a_full); h.synchronize(); best_app = min(best_app, time.perf_counter() - t0)
    t0 = time.perf_counter(); h2 = _capi.Handle(a_full); h2.synchronize(); best_new = min(best_new, time.perf_counter() - t0); h2.close()
print(f"ladybug street: {int(m1.sum())} -> {len(m1)} observations, appended={ok}: ba_append {best_app*1e3:.1f} ms, new handle {best_new*1e3:.1f} ms")
