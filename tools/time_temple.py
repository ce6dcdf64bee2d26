import sys, time
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import _capi, synthetic
p = synthetic.make_config("temple")
for rep in range(3):
    arr = _capi.Arrays.from_dict(p)
    t0 = time.perf_counter(); h = _capi.Handle(arr, loss="huber", f_scale=1.0, max_iter=300)
    t1 = time.perf_counter(); s = h.run()
    t2 = time.perf_counter(); cp, pts = h.state()
    t3 = time.perf_counter(); h.close()
    t4 = time.perf_counter()
    print("create %.2f run %.2f state %.2f close %.2f ms (setup %.2f solve %.2f)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, s["t_setup_ms"], s["t_solve_ms"]))
