"""ctypes binding of ``libba_hip.so`` (C ABI declared in ``include/ba_hip.h``).

There is NO CPU fallback: if the library is missing or no MI355X is visible, every
compute entry point raises (``BaError``).  Loading the library itself works without a GPU
(the CPU-only tests check the exported symbols).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libba_hip.so")

BA_OK = 0
BA_ERR_INVALID = -1
LOSS = {"linear": 0, "soft_l1": 1, "huber": 2, "cauchy": 3, "arctan": 4}
SOLVER = {"auto": 0, "pcg": 1, "dense_cholesky": 2, "banded_cholesky": 3, "cyclic_reduction": 4}
MEM_HOST, MEM_DEVICE = 0, 1
REDUCE_SUM, REDUCE_MAX, BROADCAST0 = 0, 1, 2

EXPORTED = [
    "ba_version", "ba_last_error", "ba_default_options", "ba_solve", "ba_eval_residuals", "ba_eval_cost",
    "ba_eval_jacobian_blocks", "ba_eval_reprojection_errors", "ba_analyze", "ba_create", "ba_destroy",
    "ba_get_pattern", "ba_set_pattern", "ba_set_collective", "ba_build", "ba_build_bench", "ba_run",
    "ba_get_state", "ba_get_cost", "ba_get_reduced_system", "ba_get_reduced_blocks", "ba_synchronize",
    "ba_set_values", "ba_set_collective_mode", "ba_get_stream", "ba_eval_cost_noalloc",
    "ba_eval_reprojection_errors_rt", "ba_get_reprojection_errors", "ba_rccl_unique_id", "ba_rccl_init", "ba_append",
    "ba_bcr_plan", "ba_bcr_plan_rest", "ba_warm_up",
]


class BaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ba_hip error {code}: {msg}")
        self.code = code


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int64),
                ("cam_pose", C.c_void_p), ("cam_intr", C.c_void_p), ("cam_fixed", C.c_void_p),
                ("points", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p), ("obs_xy", C.c_void_p),
                ("memory", C.c_int32), ("reserved", C.c_int32)]


class BaOptions(C.Structure):
    _fields_ = [("loss", C.c_int32), ("max_iter", C.c_int32), ("f_scale", C.c_double),
                ("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double),
                ("lambda0", C.c_double), ("pcg_tol", C.c_double), ("pcg_max_iter", C.c_int32),
                ("linear_solver", C.c_int32), ("verbose", C.c_int32), ("device", C.c_int32),
                ("deterministic", C.c_int32), ("schur_mode", C.c_int32), ("curv_floor", C.c_double)]


class BaSummary(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_int32), ("accepted", C.c_int32),
                ("pcg_iterations", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("grad_inf", C.c_double), ("residual_sum", C.c_double), ("lambda_final", C.c_double),
                ("t_setup_ms", C.c_double), ("t_solve_ms", C.c_double), ("n_chunks", C.c_int64),
                ("nnz_blocks", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BaAnalysis(C.Structure):
    _fields_ = [("n_chunks", C.c_int64), ("n_pair_entries", C.c_int64), ("nnz_blocks_upper", C.c_int64),
                ("nnz_blocks_full", C.c_int64), ("n_direct_pairs", C.c_int64), ("n_direct_cam_obs", C.c_int64),
                ("n_free_cams", C.c_int32), ("n_observed_points", C.c_int32), ("max_track", C.c_int32),
                ("max_chunk_cams", C.c_int32), ("max_chunk_pairs", C.c_int32), ("n_heavy_points", C.c_int32),
                ("mean_track", C.c_double), ("n_groups", C.c_int64), ("auto_rows_mode", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)

_lib = None


def load():
    """Load libba_hip.so (built in-tree by ``csrc/build.sh`` / ``__graft_entry__.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BaError(-2, f"{LIB_PATH} is missing: build it with structure_from_motion_amd/csrc/build.sh "
                          "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.ba_version.restype = C.c_int
    lib.ba_last_error.restype = C.c_char_p
    lib.ba_default_options.argtypes = [P(BaOptions)]
    lib.ba_default_options.restype = None
    lib.ba_solve.argtypes = [P(BaProblem), P(BaOptions), P(BaSummary)]
    lib.ba_eval_residuals.argtypes = [P(BaProblem), C.c_void_p]
    lib.ba_eval_cost.argtypes = [P(BaProblem), C.c_int32, C.c_double, P(C.c_double)]
    lib.ba_eval_cost_noalloc.argtypes = [P(BaProblem), C.c_int32, C.c_double, C.c_void_p, C.c_void_p]
    lib.ba_eval_jacobian_blocks.argtypes = [P(BaProblem), C.c_void_p, C.c_void_p]
    lib.ba_eval_reprojection_errors.argtypes = [P(BaProblem), C.c_void_p, C.c_void_p]
    lib.ba_eval_reprojection_errors_rt.argtypes = [P(BaProblem), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ba_get_reprojection_errors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ba_rccl_unique_id.argtypes = [C.c_void_p]
    lib.ba_bcr_plan.argtypes = [C.c_int32, P(C.c_int32), C.c_int64, P(C.c_int32), C.c_int64, P(C.c_int64)]
    lib.ba_bcr_plan_rest.argtypes = [C.c_int32, P(C.c_int32), C.c_int64, P(C.c_int32), C.c_int64, P(C.c_int64)]
    lib.ba_rccl_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.ba_analyze.argtypes = [P(BaProblem), P(BaAnalysis)]
    lib.ba_create.argtypes = [P(BaProblem), P(BaOptions)]
    lib.ba_create.restype = C.c_void_p
    lib.ba_destroy.argtypes = [C.c_void_p]
    lib.ba_destroy.restype = None
    lib.ba_get_pattern.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, P(C.c_int64)]
    lib.ba_set_pattern.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ba_set_collective.argtypes = [C.c_void_p, COLLECTIVE_FN, C.c_void_p, C.c_int32, C.c_int32]
    lib.ba_build.argtypes = [C.c_void_p, C.c_double]
    lib.ba_build_bench.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_int32, P(C.c_double), P(C.c_double)]
    lib.ba_run.argtypes = [C.c_void_p, C.c_int32, P(BaSummary)]
    lib.ba_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ba_get_cost.argtypes = [C.c_void_p, P(C.c_double)]
    lib.ba_get_reduced_system.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.ba_get_reduced_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, P(C.c_int64)]
    lib.ba_synchronize.argtypes = [C.c_void_p]
    lib.ba_set_collective_mode.argtypes = [C.c_void_p, C.c_int32]
    lib.ba_get_stream.argtypes = [C.c_void_p, P(C.c_void_p)]
    lib.ba_set_values.argtypes = [C.c_void_p, P(BaProblem), P(BaOptions)]
    lib.ba_append.argtypes = [C.c_void_p, P(BaProblem), P(BaOptions), P(C.c_int32)]
    for name in EXPORTED:
        if name not in ("ba_last_error", "ba_default_options", "ba_create", "ba_destroy"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def warm_up():
    """One tiny solve (``ba_warm_up``): pays the per-process one-time costs now instead of on the first BA call.
    Returns False -- never raises -- when the library or a GPU is missing."""
    try:
        return load().ba_warm_up() == BA_OK
    except Exception:
        return False


def check(code):
    if code != BA_OK:
        raise BaError(code, load().ba_last_error().decode("utf-8", "replace"))


def bcr_plan(n_free, pairs, allow_rest=False):
    """The block-cyclic-reduction schedule for a pattern (host only, no GPU): ``pairs`` = (n, 2) free-camera index pairs
    of the upper blocks of S, diagonal included.  Returns None when the pattern has no schedule, else a dict of the
    tables of ``csrc/ba_bcr.h`` (``src`` refers to the pairs in sorted unique order = the block order of S).
    ``allow_rest``: accept a chain / ring plus a remainder (``ba_bcr_plan_rest``; ``n_rest`` = blocks left out)."""
    lib = load()
    fn = lib.ba_bcr_plan_rest if allow_rest else lib.ba_bcr_plan
    pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    need = C.c_int64(0)
    ptr = pr.ctypes.data_as(C.POINTER(C.c_int32))
    check(fn(C.c_int32(n_free), ptr, C.c_int64(len(pr)), None, C.c_int64(0), C.byref(need)))
    w = np.zeros(need.value, dtype=np.int32)
    check(fn(C.c_int32(n_free), ptr, C.c_int64(len(pr)), w.ctypes.data_as(C.POINTER(C.c_int32)),
             C.c_int64(len(w)), C.byref(need)))
    ok, cyclic, m, nt, nb, n_slots, n_init, levels = (int(v) for v in w[:8])
    if not ok:
        return None
    at = [8]

    def take(n):
        out = w[at[0]:at[0] + n]
        at[0] += n
        return out
    cbt = nb // 6
    plan = dict(cyclic=bool(cyclic & 1), reordered=bool(cyclic & 2), m=m, nt=nt, nb=nb, n_slots=n_slots, n_init_slots=n_init,
                levels=levels)
    plan["blk_start"] = take(m + 1)
    plan["pos_f"] = take(m * cbt).reshape(m, cbt)
    plan["src"] = take((m + n_init) * cbt * cbt).reshape(m + n_init, cbt, cbt)
    plan["lvl_elim"] = take(levels + 1)
    plan["lvl_op"] = take(levels + 1)
    plan["elims"] = take(5 * int(plan["lvl_elim"][-1])).reshape(-1, 5)
    plan["ops"] = take(9 * int(plan["lvl_op"][-1])).reshape(-1, 9)
    plan["n_rest"] = int(take(1)[0])
    plan["pend_ptr"] = take(m + 1)
    plan["pend"] = take(2 * int(plan["pend_ptr"][-1])).reshape(-1, 2)
    plan["lvl_fwd"] = take(levels + 1)
    plan["fwd"] = take(6 * int(plan["lvl_fwd"][-1])).reshape(-1, 6)
    assert at[0] == len(w)
    return plan


def default_options(**kw):
    lib = load()
    o = BaOptions()
    lib.ba_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "loss" and isinstance(v, str):
            v = LOSS[v]
        if k == "linear_solver" and isinstance(v, str):
            v = SOLVER[v]
        if k == "schur_mode" and isinstance(v, str):
            v = {"auto": 0, "chunks": 1, "rows": 2}[v]
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k!r}")
        setattr(o, k, v)
    return o


class Arrays:
    """Keeps contiguous, correctly typed host copies of a SoA problem alive next to the
    ``BaProblem`` struct that points into them."""

    def __init__(self, cam_pose, cam_intr, cam_fixed, points, obs_cam, obs_pt, obs_xy):
        self.cam_pose = np.ascontiguousarray(cam_pose, dtype=np.float64).reshape(-1, 6).copy()
        self.cam_intr = np.ascontiguousarray(cam_intr, dtype=np.float64).reshape(-1, 4)
        self.cam_fixed = np.ascontiguousarray(np.asarray(cam_fixed).astype(np.uint8))
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3).copy()
        self.obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32).reshape(-1)
        self.obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32).reshape(-1)
        self.obs_xy = np.ascontiguousarray(obs_xy, dtype=np.float64).reshape(-1, 2)
        n_c, n_p, n_o = len(self.cam_pose), len(self.points), len(self.obs_cam)
        if len(self.cam_intr) != n_c or len(self.cam_fixed) != n_c:
            raise ValueError("camera arrays disagree in length")
        if len(self.obs_pt) != n_o or len(self.obs_xy) != n_o:
            raise ValueError("observation arrays disagree in length")
        p = BaProblem()
        p.n_cams, p.n_points, p.n_obs = n_c, n_p, n_o
        p.cam_pose = self.cam_pose.ctypes.data
        p.cam_intr = self.cam_intr.ctypes.data
        p.cam_fixed = self.cam_fixed.ctypes.data
        p.points = self.points.ctypes.data
        p.obs_cam = self.obs_cam.ctypes.data
        p.obs_pt = self.obs_pt.ctypes.data
        p.obs_xy = self.obs_xy.ctypes.data
        p.memory = MEM_HOST
        self.struct = p

    @classmethod
    def from_dict(cls, d):
        return cls(d["cam_pose"], d["cam_intr"], d["cam_fixed"], d["points"], d["obs_cam"], d["obs_pt"], d["obs_xy"])


# ---- thin functional wrappers ----------------------------------------------------------------
def eval_residuals(arr):
    out = np.empty(2 * arr.struct.n_obs, dtype=np.float64)
    check(load().ba_eval_residuals(C.byref(arr.struct), out.ctypes.data))
    return out


def eval_cost(arr, loss="huber", f_scale=1.0):
    c = C.c_double()
    check(load().ba_eval_cost(C.byref(arr.struct), LOSS[loss] if isinstance(loss, str) else loss, float(f_scale), C.byref(c)))
    return c.value


def eval_jacobian_blocks(arr):
    n = arr.struct.n_obs
    jc = np.empty((n, 2, 6), dtype=np.float64)
    jp = np.empty((n, 2, 3), dtype=np.float64)
    check(load().ba_eval_jacobian_blocks(C.byref(arr.struct), jc.ctypes.data, jp.ctypes.data))
    return jc, jp


def eval_reprojection_errors(arr, cam_k):
    cam_k = np.ascontiguousarray(cam_k, dtype=np.float64).reshape(-1, 9)
    out = np.empty(arr.struct.n_cams, dtype=np.float64)
    check(load().ba_eval_reprojection_errors(C.byref(arr.struct), cam_k.ctypes.data, out.ctypes.data))
    return out


def eval_reprojection_errors_rt(arr, cam_rt, cam_k):
    """Per-camera mean reprojection error with the poses as rotation MATRICES, used as given: cam_rt (Nc, 12) = R
    row-major, t; cam_k (Nc, 9).  ``arr.cam_pose`` / ``arr.cam_intr`` are not read."""
    cam_rt = np.ascontiguousarray(cam_rt, dtype=np.float64).reshape(-1, 12)
    cam_k = np.ascontiguousarray(cam_k, dtype=np.float64).reshape(-1, 9)
    if len(cam_rt) != arr.struct.n_cams or len(cam_k) != arr.struct.n_cams:
        raise ValueError("cam_rt / cam_k need one row per camera")
    out = np.empty(arr.struct.n_cams, dtype=np.float64)
    check(load().ba_eval_reprojection_errors_rt(C.byref(arr.struct), cam_rt.ctypes.data, cam_k.ctypes.data, out.ctypes.data))
    return out


def analyze(arr):
    a = BaAnalysis()
    check(load().ba_analyze(C.byref(arr.struct), C.byref(a)))
    return a.as_dict()


def solve(arr, **opts):
    """One-shot solve; ``arr.cam_pose`` / ``arr.points`` are overwritten on success."""
    o = default_options(**opts)
    s = BaSummary()
    check(load().ba_solve(C.byref(arr.struct), C.byref(o), C.byref(s)))
    return s.as_dict()


def rccl_unique_id():
    """128-byte id for ``Handle.rccl_init`` (rank 0 makes it, every rank uses the same one)."""
    buf = C.create_string_buffer(128)
    check(load().ba_rccl_unique_id(buf))
    return buf.raw


class Handle:
    """Device-resident problem (``ba_create`` ... ``ba_destroy``)."""

    def __init__(self, arr, **opts):
        self._lib = load()
        self.arr = arr
        self.opts = default_options(**opts)
        self.h = self._lib.ba_create(C.byref(arr.struct), C.byref(self.opts))
        if not self.h:
            raise BaError(-1, self._lib.ba_last_error().decode("utf-8", "replace"))
        self._cb = None

    def close(self):
        if self.h:
            self._lib.ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_values(self, arr, **opts):
        """New poses / intrinsics / points / observations for the SAME structure (``ba_set_values``): no symbolic
        analysis, no allocation; the LM state starts afresh.  ``opts`` (if any) replace the handle's options."""
        if opts:
            self.opts = default_options(**opts)
        check(self._lib.ba_set_values(self.h, C.byref(arr.struct), C.byref(self.opts) if opts else None))
        self.arr = arr

    def append(self, arr, **opts):
        """The handle's problem GROWN at the end of its arrays (``ba_append``): new cameras / points / observations,
        all values from ``arr``.  Returns True when the new part was appended to the resident structure, False when
        the structure had to be rebuilt inside the handle (same result either way)."""
        if opts:
            self.opts = default_options(**opts)
        mode = C.c_int32(0)
        check(self._lib.ba_append(self.h, C.byref(arr.struct), C.byref(self.opts) if opts else None, C.byref(mode)))
        self.arr = arr
        return mode.value == 1

    def build(self, lam=1e-4):
        check(self._lib.ba_build(self.h, float(lam)))

    def build_bench(self, lam=1e-4, warmup=2, reps=10):
        a, b = C.c_double(), C.c_double()
        check(self._lib.ba_build_bench(self.h, float(lam), int(warmup), int(reps), C.byref(a), C.byref(b)))
        return a.value, b.value

    def run(self, max_steps=None):
        s = BaSummary()
        check(self._lib.ba_run(self.h, int(self.opts.max_iter if max_steps is None else max_steps), C.byref(s)))
        return s.as_dict()

    def state(self):
        cp = np.empty((self.arr.struct.n_cams, 6), dtype=np.float64)
        pts = np.empty((self.arr.struct.n_points, 3), dtype=np.float64)
        check(self._lib.ba_get_state(self.h, cp.ctypes.data, pts.ctypes.data))
        return cp, pts

    def cost(self):
        c = C.c_double()
        check(self._lib.ba_get_cost(self.h, C.byref(c)))
        return c.value

    def reprojection_errors(self, cam_k=None, cam_rt=None):
        """Per-camera mean reprojection error of the RESIDENT points / observations (``ba_get_reprojection_errors``);
        cam_rt (Nc, 12) = R row-major, t as the caller holds them, or None for the handle's current poses; cam_k
        (Nc, 9), or None for K from the handle's fx, fy, cx, cy."""
        n = self.arr.struct.n_cams
        out = np.empty(n, dtype=np.float64)
        ptr = []
        for a, w in ((cam_rt, 12), (cam_k, 9)):
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, w)
                if len(a) != n:
                    raise ValueError("cam_rt / cam_k need one row per camera")
            ptr.append(a)
        check(self._lib.ba_get_reprojection_errors(self.h, None if ptr[0] is None else ptr[0].ctypes.data,
                                                   None if ptr[1] is None else ptr[1].ctypes.data, out.ctypes.data))
        return out

    def synchronize(self):
        check(self._lib.ba_synchronize(self.h))

    def debug_bcr_apply(self):
        """Diagnostic: relative difference between the cyclic reduction's direct solve of S x = -g and its preconditioner
        path (explicit inverses + sweeps) on the same right-hand side (``ba_debug_bcr_apply``, not part of the header)."""
        rel = C.c_double()
        self._lib.ba_debug_bcr_apply.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._lib.ba_debug_bcr_apply.restype = C.c_int
        check(self._lib.ba_debug_bcr_apply(self.h, C.byref(rel)))
        return rel.value

    def pattern(self):
        n = C.c_int64()
        check(self._lib.ba_get_pattern(self.h, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), dtype=np.int32)
        check(self._lib.ba_get_pattern(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def set_pattern(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        check(self._lib.ba_set_pattern(self.h, pairs.ctypes.data, len(pairs)))

    def stream(self):
        """The hipStream_t (as an integer) every kernel of this handle runs on."""
        p = C.c_void_p()
        check(self._lib.ba_get_stream(self.h, C.byref(p)))
        return p.value or 0

    def set_collective(self, fn, rank, world, stream_ordered=False):
        """fn(dev_ptr:int, count:int, op:int) -> int (0 = ok).  ``stream_ordered``: fn enqueues the collective on
        ``self.stream()`` and the library does not synchronise around the call (``ba_set_collective_mode``)."""
        check(self._lib.ba_set_collective_mode(self.h, 1 if stream_ordered else 0))
        def tramp(ptr, count, op, _user):
            try:
                return int(fn(ptr, count, op) or 0)
            except Exception as exc:  # never let an exception cross the C boundary
                print(f"[ba_hip] collective callback failed: {exc!r}")
                return 1
        self._cb = COLLECTIVE_FN(tramp)
        check(self._lib.ba_set_collective(self.h, self._cb, None, int(rank), int(world)))

    def rccl_init(self, unique_id, rank, world):
        """Built-in RCCL collectives on this handle's stream (``ba_rccl_init``); ``unique_id``: the 128 bytes of
        :func:`rccl_unique_id`, the same on every rank.  Collective over the ranks."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.ba_rccl_init(self.h, buf, int(rank), int(world)))
        self._cb = None

    def reduced_system(self, want_points=True, want_dense=True):
        nf = int((self.arr.cam_fixed == 0).sum())
        nc, npnt = self.arr.struct.n_cams, self.arr.struct.n_points
        S = np.empty((6 * nf, 6 * nf)) if want_dense else None
        g = np.empty(6 * nf)
        U = np.empty((nc, 6, 6)); gc = np.empty((nc, 6))
        V = np.empty((npnt, 3, 3)) if want_points else None
        gp = np.empty((npnt, 3)) if want_points else None
        check(self._lib.ba_get_reduced_system(self.h, S.ctypes.data if want_dense else None, g.ctypes.data,
                                              U.ctypes.data, gc.ctypes.data,
                                              V.ctypes.data if want_points else None,
                                              gp.ctypes.data if want_points else None))
        return dict(S=S, gred=g, U=U, gc=gc, V=V, gp=gp)

    def reduced_blocks(self):
        """Upper 6x6 blocks of the reduced camera system as stored: (i, j, blocks (n, 6, 6)), i <= j."""
        n = C.c_int64(0)
        check(self._lib.ba_get_reduced_blocks(self.h, None, None, 0, C.byref(n)))
        keys = np.empty(n.value, dtype=np.int64); blocks = np.empty((n.value, 6, 6))
        check(self._lib.ba_get_reduced_blocks(self.h, keys.ctypes.data, blocks.ctypes.data, n.value, C.byref(n)))
        return (keys >> 32).astype(np.int64), (keys & 0xffffffff).astype(np.int64), blocks

    def schur_matvec(self, v):
        """S v from the stored blocks (host-side product; parity probe for sizes where dense S does not fit)."""
        i, j, blk = self.reduced_blocks()
        x = np.asarray(v, dtype=np.float64).reshape(-1, 6)
        out = np.zeros_like(x)
        np.add.at(out, i, np.einsum("nab,nb->na", blk, x[j]))
        off = i != j
        np.add.at(out, j[off], np.einsum("nab,na->nb", blk[off], x[i[off]]))
        return out.reshape(-1)
