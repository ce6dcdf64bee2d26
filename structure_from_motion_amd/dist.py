"""Point-sharded multi-GPU bundle adjustment (one process per GPU, RCCL over xGMI).

Every residual couples one camera and one point (``BundleAdjustment.py:185-196``), so a rank
that owns a disjoint set of points together with ALL their observations (cameras replicated)
builds its partial reduced camera system without communication.  Per LM step the ranks then
exchange:

  1. one all-reduce (sum) of the contiguous buffer [S blocks | U | g_c | g_red | cost | one slot per rank]
     (the slots carry each rank's point-gradient norm; their maximum is taken locally);
  2. the replicated PCG solve (no communication), then a broadcast of rank 0's camera step so
     that every rank holds bit-identical cameras;
  3. one 4-double all-reduce of the step statistics (model reduction, trial cost, norms) so that
     every rank takes the same accept / reject decision.

The collectives are ``torch.distributed`` calls (backend "nccl" = RCCL on ROCm) on tensors that
alias the library's device buffers; the library calls back into :func:`make_collective`.
"""
import numpy as np

from . import _capi


class _DevArray:
    """Minimal ``__cuda_array_interface__`` holder so torch can alias a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def make_collective(group=None, stream=None):
    """Collective callback for ``Handle.set_collective`` backed by torch.distributed.

    ``stream`` (a hipStream_t as an integer, ``Handle.stream()``), nccl backend only: the collectives are enqueued in
    stream order on that stream -- torch orders RCCL's own stream behind it and it behind RCCL's -- and the callback
    returns without waiting, so a build pass or an LM step has no host round trip around its all-reduce
    (``set_collective(..., stream_ordered=True)``).  Without it the callback blocks until the collective is done."""
    import torch
    import torch.distributed as dist

    host_staged = dist.get_backend(group) != "nccl"  # e.g. gloo in the single-GPU tests
    views = {}  # (pointer, count) -> tensor aliasing the library's buffer (the buffers live as long as the handle)
    ext = torch.cuda.ExternalStream(stream) if (stream and not host_staged) else None

    def coll(ptr, count, op):
        dev = views.get((ptr, count))
        if dev is None:
            dev = views[(ptr, count)] = torch.as_tensor(_DevArray(ptr, count), device="cuda")
        if ext is not None:
            with torch.cuda.stream(ext):
                _run(dev, op)
            return 0
        t = dev.cpu() if host_staged else dev
        _run(t, op)
        if host_staged:
            dev.copy_(t)
        torch.cuda.current_stream().synchronize()
        return 0

    # (torch's `src` is a GLOBAL rank: the group's first rank, which is rank 0 of the library's numbering)
    src0 = dist.get_global_rank(group, 0) if group is not None else 0

    def _run(t, op):
        if op == _capi.REDUCE_SUM:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        elif op == _capi.REDUCE_MAX:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        else:
            dist.broadcast(t, src=src0, group=group)
    coll.stream_ordered = ext is not None
    return coll


def union_pattern(local_pairs, all_gather_object):
    """Union of the ranks' co-visibility pairs: every rank must use ONE block pattern so the
    reduced-system buffers line up element by element for the all-reduce."""
    gathered = all_gather_object(np.ascontiguousarray(local_pairs, dtype=np.int32))
    allp = np.concatenate([np.asarray(g, dtype=np.int32).reshape(-1, 2) for g in gathered], axis=0)
    return np.unique(allp, axis=0)


def shard_points(prob, rank, world):
    """Contiguous point shards balanced by observation count; a rank gets its points (re-indexed)
    and all their observations, cameras are replicated.  Returns (shard, point_index)."""
    n_p = len(prob["points"])
    cnt = np.bincount(prob["obs_pt"], minlength=n_p)
    csum = np.concatenate([[0], np.cumsum(cnt)])
    total = csum[-1]
    bounds = [int(np.searchsorted(csum, total * r / world, side="left")) for r in range(world)] + [n_p]
    bounds[0] = 0
    lo, hi = bounds[rank], bounds[rank + 1]
    m = (prob["obs_pt"] >= lo) & (prob["obs_pt"] < hi)
    shard = dict(cam_pose=prob["cam_pose"], cam_intr=prob["cam_intr"], cam_fixed=prob["cam_fixed"],
                 points=prob["points"][lo:hi], obs_cam=prob["obs_cam"][m],
                 obs_pt=(prob["obs_pt"][m] - lo).astype(np.int32), obs_xy=prob["obs_xy"][m])
    return shard, np.arange(lo, hi)


def create_sharded_handle(shard, rank, world, group=None, stream_ordered=True, force_collectives=False, native=None, **opts):
    """Handle for this rank's shard with the shared pattern and the collectives installed.

    ``native`` (default: on the nccl backend): the library's own RCCL binding (``ba_rccl_init``) -- ncclAllReduce /
    ncclBroadcast enqueued on the handle's stream from C, no Python between the kernels and the collectives of an LM
    step; torch.distributed only carries the pattern union and the 128-byte RCCL id, once.  Otherwise (gloo in the
    tests, or ``native=False``) the collectives are torch.distributed calls behind the callback hook, stream-ordered on
    the nccl backend unless ``stream_ordered=False``.  ``force_collectives`` installs them for a world of one as well
    (tests: the whole sharded path on a single GPU)."""
    import torch.distributed as dist
    h = _capi.Handle(_capi.Arrays.from_dict(shard), **opts)
    if world > 1 or force_collectives:
        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj, group=group)
            return out
        h.set_pattern(union_pattern(h.pattern(), gather))
        if native is None:
            native = dist.get_backend(group) == "nccl"
        if native:
            # (every rank must take the same transport: the ranks first agree that the RCCL binding can be had at all)
            try:
                uid, err = _capi.rccl_unique_id(), None
            except Exception as e:   # librccl.so not loadable on this rank
                uid, err = None, str(e)
            errs = [e for e in gather(err) if e]
            if errs:
                import warnings
                warnings.warn("RCCL binding unavailable (%s): collectives through torch.distributed instead" % errs[0])
                native = False
        if native:
            box = [uid if rank == 0 else None]
            # (`rank` is the rank inside `group`; torch's `src` is a global rank)
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            h.rccl_init(box[0], rank, world)
        else:
            coll = make_collective(group, h.stream() if stream_ordered else None)
            h.set_collective(coll, rank, world, stream_ordered=coll.stream_ordered)
    return h


def append_sharded(h, grown_shard, world, group=None, **opts):
    """``ba_append`` on a sharded handle (the incremental driver's growth, ``Reconstruction.py:244-255``, on more than one
    GPU): every rank appends the growth of ITS shard -- new points and their observations at the end of its arrays; the
    cameras are replicated, so every rank passes the same grown camera arrays -- and the ranks then agree on one block
    pattern again (collective: every rank must call this).  Returns True when this rank's structure was extended in
    place."""
    import torch.distributed as dist
    appended = h.append(_capi.Arrays.from_dict(grown_shard), **opts)

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj, group=group)
        return out
    h.set_pattern(union_pattern(h.pattern(), gather))
    return appended
