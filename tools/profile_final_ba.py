"""Developer: cProfile of a warm final_ba call through the dict API and through a BASession (Temple fixture shape)."""
import contextlib, cProfile, io, pstats, sys, time
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import BundleAdjustment as BA, synthetic
from structure_from_motion_amd.session import BASession
p = synthetic.make_config("temple")
ids, image_data, pts = synthetic.to_image_data(p)
ses = BASession()
def call(fn):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(dict(image_data), pts, ids, verbose=False)
for name, fn in (("dict API", BA.final_ba), ("session", ses.final_ba)):
    for _ in range(3):
        call(fn)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); call(fn); ts.append((time.perf_counter() - t) * 1e3)
    print("%s: %s ms" % (name, " ".join("%.2f" % x for x in ts)))
    pr = cProfile.Profile(); pr.enable(); call(fn); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
