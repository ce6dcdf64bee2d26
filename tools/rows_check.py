import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from structure_from_motion_amd import _capi, synthetic
from oracle import ba_oracle as O
def rel(a,b): return np.linalg.norm(np.asarray(a)-np.asarray(b))/max(np.linalg.norm(b),1e-300)
probs = {
 "street": synthetic.make_street(12, 300, 1300, seed=8, outlier_frac=0.05),
 "random": synthetic.make_ring(90, 1500, seed=78, K=(1200.0,1200.0,960.0,540.0), focal_jitter=0.05, noise=0.5, outlier_frac=0.02, visibility="random", mean_track=7.0, max_track=20, n_fixed=2, radius=6.0),
 "heavy": synthetic.make_ring(270, 110, seed=3, mean_track=30.0, max_track=60, n_landmarks=3, outlier_frac=0.05, n_fixed=2, n_unobserved=2),
}
base = synthetic.make_street(6, 40, 160, seed=2); p = dict(base)
p["obs_cam"] = np.concatenate([base["obs_cam"], base["obs_cam"][:7]]); p["obs_pt"] = np.concatenate([base["obs_pt"], base["obs_pt"][:7]]); p["obs_xy"] = np.concatenate([base["obs_xy"], base["obs_xy"][:7] + 0.3])
probs["twice"] = p
for name, p in probs.items():
    lam = 1e-3
    nb = O.normal_blocks(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"], 1.0, "huber", floor=1.0)
    S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
    free = p["cam_fixed"] == 0
    with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0, schur_mode="rows") as h:
        h.build(lam)
        r = h.reduced_system()
        print(name, "U", rel(r["U"][free], nb["U"][free]), "gc", rel(r["gc"][free], nb["gc"][free]), "V", rel(r["V"], nb["V"]), "gp", rel(r["gp"], nb["gp"]),
              "S", rel(r["S"], S), "gred", rel(r["gred"], gred), "cost", abs(h.cost()-nb["F"])/nb["F"], flush=True)
        s = h.run(20)
    a = _capi.Arrays.from_dict(p)
    s2 = _capi.solve(a, loss="huber", f_scale=1.0, max_iter=20, schur_mode="chunks")
    print("   LM rows", s["iterations"], s["final_cost"], "chunks", s2["iterations"], s2["final_cost"], flush=True)
