"""Array-native marshalling for the ``inter_ba`` / ``final_ba`` protocol.

The reference rebuilds its observation dicts and the parameter vector with per-element Python loops on
every call (``BundleAdjustment.py:243-295, 46-65``) and BA is invoked every second image on a growing
problem (``Reconstruction.py:253-255``).  :class:`BASession` goes from the driver's ``image_data`` tuples
straight to the ``BaProblem`` arrays with NumPy, caches what cannot change between calls (the keypoint
coordinates of an image), and writes results back vectorised.  Semantics -- camera order, observation
order, validity rules, gauge, return values, in-place update of ``image_data`` -- are those of the module
functions in ``BundleAdjustment.py`` of this package, which the tests compare it against.

Device-resident reuse: the session keeps the engine handle of its last solve.  A call whose structure
(camera set, gauge, (camera, point) of every observation) is the one the handle was analysed for goes through
``ba_set_values`` -- values re-uploaded, no symbolic analysis, no allocation.  A call whose problem GREW (the
incremental driver's usual case: images registered, points triangulated since the last call) goes through
``ba_append``: when the new observations only see new points -- what the reference's driver records -- only those are
analysed and their chunks are appended to the resident arrays; otherwise the structure is rebuilt inside the handle.
The C side compares the structures exactly, so a stale handle can never be used for another problem.  ``close()``
(or dropping the session) frees the handle.
"""
import numpy as np

from . import BundleAdjustment as _ba
from . import _capi
from .geometry import matrix_to_rvec, rvec_to_matrix


class BASession:
    def __init__(self, reuse_device=True):
        self._kp_xy = {}          # img_id -> (the keypoint list, its length, (k, 2) float64 coordinates)
        self.last_summary = None
        self.reuse_device = reuse_device
        self._handle = None       # engine handle of the last solve (structure + device buffers)
        self.handle_reuses = 0        # calls that went through ba_set_values / ba_append instead of a new handle
        self.handle_appends = 0       # ... of which the resident structure was extended in place (ba_append, mode 1)
        self._resident = None     # what the handle's device state corresponds to on the host (see compute_ba_reprojection_errors)
        self.resident_error_calls = 0

    def close(self):
        self._resident = None
        if self._handle is not None:
            self._handle.close()
            self._handle = None

    def _run(self, arr, **opts):
        """ba_solve's sequence (create | set_values, run, get_state) on the session's handle; arr is updated in place."""
        h = self._handle
        if h is not None:
            old = (h.arr.struct.n_cams, h.arr.struct.n_points, h.arr.struct.n_obs)
            new = (arr.struct.n_cams, arr.struct.n_points, arr.struct.n_obs)
            try:
                if old == new:
                    h.set_values(arr, **opts)            # same structure (checked exactly on the C side): values only
                    self.handle_reuses += 1
                elif all(n >= o for n, o in zip(new, old)):
                    # the incremental driver's case: images were registered and points triangulated since the last call
                    self.handle_appends += bool(h.append(arr, **opts))
                    self.handle_reuses += 1
                else:
                    self.close()
            except _capi.BaError as e:
                # Whatever went wrong, this handle is not kept: after BA_ERR_INVALID (another problem altogether) a new
                # one is made below; after anything else (out of memory, a HIP error half way through an append) the
                # error goes to the caller, and the next call starts from a fresh handle as well.
                self.close()
                if e.code != _capi.BA_ERR_INVALID:
                    raise
        if self._handle is None:
            self._handle = _capi.Handle(arr, **opts)
        h = self._handle
        try:
            s = h.run()
            cp, pts = h.state()
        except Exception:
            self.close()
            raise
        arr.cam_pose[...] = cp
        arr.points[...] = pts
        if not self.reuse_device:
            self.close()
        return s

    # ------------------------------------------------------------------ marshalling
    def _keypoints(self, img_id, kp):
        # The cache entry holds the keypoint LIST ITSELF and is valid for that very object only (`is`: an id() could be
        # recycled by a new list of the same length).  The first and last coordinates are re-checked as a cheap guard
        # against in-place edits; an edit in the middle of a list needs `invalidate(img_id)`.
        hit = self._kp_xy.get(img_id)
        if hit is not None and hit[0] is kp and hit[1] == len(kp):
            xy = hit[2]
            if len(kp) == 0 or (tuple(xy[0]) == tuple(map(float, kp[0].pt)) and tuple(xy[-1]) == tuple(map(float, kp[-1].pt))):
                return xy
        # (through the module's C walk when it is there: 82 k keypoints took 60 ms as a list comprehension, the whole first call of a session)
        xy = _ba._keypoint_xy(kp, np.arange(len(kp), dtype=np.int64)) if len(kp) else np.empty((0, 2), dtype=np.float64)
        self._kp_xy[img_id] = (kp, len(kp), xy)
        return xy

    def invalidate(self, img_id=None):
        """Forget the cached keypoint coordinates of one image (or of all): needed after editing ``kp[k].pt`` in place."""
        self._resident = None
        if img_id is None:
            self._kp_xy.clear()
        else:
            self._kp_xy.pop(img_id, None)

    def marshal(self, image_data, all_points_3D, visited_ids):
        """-> (Arrays without gauge, cam_ids, colours (Np, 3) or None).  Same camera / observation order as
        ``prepare_cameras_from_image_data`` + ``prepare_observations_from_image_data`` + ``_to_arrays``; the keypoint
        coordinates of an image are converted once per session."""
        return _ba._image_data_to_arrays(image_data, all_points_3D, visited_ids, keypoints=self._keypoints)

    # ------------------------------------------------------------------ solves
    def _solve(self, image_data, all_points_3D, visited_ids, f_scale, max_nfev, verbose, min_obs):
        self._resident = None
        new_points, ok, cam_ids, _fixed, _att = _ba._ba_from_image_data(image_data, all_points_3D, visited_ids, f_scale, max_nfev,
                                                                       verbose, min_obs=min_obs, keypoints=self._keypoints,
                                                                       run=self._run)
        self.last_summary = _ba.last_summary
        if ok and self._handle is not None:
            # the handle's device state IS what was just written back: remember the very objects, so that a
            # following compute_ba_reprojection_errors can tell that nothing was replaced in between
            self._resident = dict(cam_ids=list(cam_ids), points=new_points, n_points=len(new_points),
                                  R={c: image_data[c][0] for c in cam_ids}, t={c: image_data[c][1] for c in cam_ids},
                                  ref={c: (image_data[c][3], len(image_data[c][3])) for c in cam_ids},
                                  kp={c: (image_data[c][5], len(image_data[c][5])) for c in cam_ids})
        return new_points, image_data, ok

    def inter_ba(self, image_data, all_points_3D, visited_ids, ba_interval=10, verbose=True):
        """As ``inter_ba`` (``BundleAdjustment.py:309-350``): huber, f_scale 1.5, max_nfev 150."""
        if len(visited_ids) % ba_interval != 0 or len(visited_ids) < 3:
            return all_points_3D, image_data, False
        return self._solve(image_data, all_points_3D, visited_ids, 1.5, 150, verbose, min_obs=20)

    def final_ba(self, image_data, all_points_3D, visited_ids, verbose=True):
        """As ``final_ba`` (``BundleAdjustment.py:353-385``): huber, f_scale 1.0, max_nfev 300."""
        return self._solve(image_data, all_points_3D, visited_ids, 1.0, 300, verbose, min_obs=None)

    def compute_ba_reprojection_errors(self, image_data, all_points_3D, visited_ids):
        """As ``compute_ba_reprojection_errors`` (``BundleAdjustment.py:388-423``), which the reference's driver calls
        right after every successful BA (``Reconstruction.py:263``) -- SURVEY.md 8(f) f2: when the arguments are exactly
        what the last solve of this session returned (same point list object and length, same R / t / reference /
        keypoint objects per image), the errors come from the handle's RESIDENT state: one kernel over the observations
        already in HBM, nothing re-marshalled, only the 21 camera numbers (R, t, K) per image uploaded.  Anything else (no handle, other images, a
        grown point list) takes the module function's path.  In-place edits of a kept object are not detected --
        ``invalidate()`` / ``close()`` after such edits."""
        r = self._resident
        if r is not None and self._handle is not None and all_points_3D is r["points"] and len(all_points_3D) == r["n_points"]:
            usable = list(dict.fromkeys(i for i in visited_ids if i in image_data and len(image_data[i]) >= 6))
            if usable == r["cam_ids"] and all(
                    image_data[c][0] is r["R"][c] and image_data[c][1] is r["t"][c]
                    and image_data[c][3] is r["ref"][c][0] and len(image_data[c][3]) == r["ref"][c][1]
                    and image_data[c][5] is r["kp"][c][0] and len(image_data[c][5]) == r["kp"][c][1] for c in usable):
                Ks = np.array([np.asarray(image_data[c][2], dtype=np.float64).reshape(9) for c in usable])
                Rt = np.array([np.concatenate([np.asarray(image_data[c][0], dtype=np.float64).reshape(9),
                                               np.asarray(image_data[c][1], dtype=np.float64).reshape(3)]) for c in usable])
                err = self._handle.reprojection_errors(Ks, Rt)
                self.resident_error_calls += 1
                return {cid: float(err[i]) for i, cid in enumerate(usable) if not np.isnan(err[i])}
        return _ba.compute_ba_reprojection_errors(image_data, all_points_3D, visited_ids)

    # ------------------------------------------------------------------ opt-in step next to the path (8(f) f4)
    def cull_points(self, image_data, all_points_3D, visited_ids, max_error_px, min_track=1):
        """Post-BA filtering (NOT reference behaviour -- the reference keeps every point; opt-in only):
        un-reference every point whose worst reprojection error exceeds ``max_error_px`` or that is seen by
        fewer than ``min_track`` registered images.  Point indices stay valid: the point row is kept, its
        entries in the images' reference arrays become -1, so later BA calls pass it through untouched.
        Errors come from the engine's residual kernel (``ba_eval_residuals``).  Returns the culled indices."""
        arr, cam_ids, _ = self.marshal(image_data, all_points_3D, visited_ids)
        if len(arr.obs_cam) == 0:
            return np.zeros(0, np.int64)
        err = np.linalg.norm(_capi.eval_residuals(arr).reshape(-1, 2), axis=1)
        n_p = len(arr.points)
        worst = np.zeros(n_p)
        np.maximum.at(worst, arr.obs_pt, err)
        track = np.bincount(arr.obs_pt, minlength=n_p)
        bad = np.flatnonzero((track > 0) & ((worst > max_error_px) | (track < min_track)))
        if len(bad):
            self._resident = None                    # (the reference arrays are edited in place below)
            is_bad = np.zeros(n_p, dtype=bool)
            is_bad[bad] = True
            for cid in cam_ids:
                ref = image_data[cid][3]
                for k, r in enumerate(ref):
                    if isinstance(r, int) and 0 <= r < n_p and is_bad[r]:
                        ref[k] = -1
        return bad
