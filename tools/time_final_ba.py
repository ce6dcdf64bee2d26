"""final_ba host to host on Temple-shaped image_data: module function and BASession, several calls each."""
import contextlib, io, sys, time
sys.path.insert(0, '/root/repo')
from structure_from_motion_amd import synthetic, BundleAdjustment as BA
from structure_from_motion_amd.session import BASession
p = synthetic.make_config("temple")
ids, image_data, pts = synthetic.to_image_data(p)
sess = BASession()
for name, fn in (("module", lambda d: BA.final_ba(d, pts, ids, verbose=False)), ("session", lambda d: sess.final_ba(d, pts, ids, verbose=False)),
                 ("module", lambda d: BA.final_ba(d, pts, ids, verbose=False)), ("fresh session", lambda d: BASession().final_ba(d, pts, ids, verbose=False))):
    for rep in range(3):
        d = dict(image_data)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            _pts, _d, ok = fn(d)
        print("%-14s call %d: %.1f ms ok=%s" % (name, rep, (time.perf_counter() - t0) * 1e3, ok))
