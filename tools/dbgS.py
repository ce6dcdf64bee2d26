import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, problem_of
from oracle import ba_oracle as O
from structure_from_motion_amd import _capi
g = load_golden(sys.argv[1] if len(sys.argv) > 1 else "ring16_dense"); p = problem_of(g)
lam = 1e-4
nb = O.normal_blocks(p["cam_pose"], p["cam_intr"], p["points"], p["obs_cam"], p["obs_pt"], p["obs_xy"], 1.0, "huber", floor=1.0)
S, gred, _, _, _ = O.reduced_system(nb, p["cam_fixed"], p["obs_cam"], p["obs_pt"], lam)
with _capi.Handle(_capi.Arrays.from_dict(p), loss="huber", f_scale=1.0) as h:
    h.build(lam); r = h.reduced_system()
D = r["S"] - S
nf = S.shape[0] // 6
E = np.zeros((nf, nf)); 
for i in range(nf):
    for j in range(nf):
        E[i, j] = np.abs(D[6*i:6*i+6, 6*j:6*j+6]).max() / max(np.abs(S[6*i:6*i+6, 6*j:6*j+6]).max(), 1e-300)
np.set_printoptions(linewidth=250, precision=1)
print((E > 1e-6).astype(int))
i, j = np.unravel_index(np.argmax(E), E.shape)
print("worst block", i, j, E[i, j])
print(D[6*i:6*i+6, 6*j:6*j+6] / np.abs(S[6*i:6*i+6, 6*j:6*j+6]).max())
print("gred rel", np.abs(r["gred"] - gred).max() / np.abs(gred).max())
