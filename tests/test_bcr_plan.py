"""CPU: the block-cyclic-reduction schedule of ``csrc/ba_bcr.h`` (``ba_bcr_plan``, host only), EXECUTED with NumPy.

The GPU kernels run whatever the planner's tables say -- which block is eliminated on which level, next to which
neighbours, from which coupling slot, with which updates, and where every 6x6 block of S lands.  This test runs the same
tables through plain NumPy linear algebra on random positive definite chain / ring systems and compares with a dense
solve, so a wrong table shows up without a GPU: odd and even block counts, ragged last blocks, rings that shrink to
three / four / five survivors, every block size (2, 4, 6, 8 tiles).  (The reference has no counterpart: it hands the
whole least-squares problem to scipy, ``BundleAdjustment.py:210-220``.)
"""
import numpy as np
import pytest

from structure_from_motion_amd import _capi

TRANSPOSED, SYMMETRISE, MASK = 1 << 30, 1 << 29, (1 << 29) - 1


def _system(nf, band, ring, seed, shuffle=False):
    """Random SPD block system with the co-visibility of a chain (|i - j| <= band) or a closed ring; ``shuffle``: the
    cameras are numbered at random (the chain is still there, in another order)."""
    rng = np.random.default_rng(seed)
    name = rng.permutation(nf) if shuffle else np.arange(nf)
    pairs = [(min(name[i], name[j]), max(name[i], name[j])) for i in range(nf) for j in range(i, nf)
             if j - i <= band or (ring and nf - (j - i) <= band)]
    pairs = np.array(sorted(set(pairs)), dtype=np.int32)
    blocks = rng.uniform(-1.0, 1.0, size=(len(pairs), 6, 6))
    A = np.zeros((6 * nf, 6 * nf))
    for k, (i, j) in enumerate(pairs):
        if i == j:
            blocks[k] = 0.5 * (blocks[k] + blocks[k].T) + (6.0 * (2 * band + 1) + 4.0) * np.eye(6)
            A[6 * i:6 * i + 6, 6 * i:6 * i + 6] = blocks[k]
        else:
            A[6 * i:6 * i + 6, 6 * j:6 * j + 6] = blocks[k]
            A[6 * j:6 * j + 6, 6 * i:6 * i + 6] = blocks[k].T
    return pairs, blocks, A, rng.uniform(-1.0, 1.0, size=6 * nf)


def _gather(plan, blocks):
    """k_bcr_gather: every diagonal block / initial coupling slot from the src table."""
    nb, cbt = plan["nb"], plan["nb"] // 6
    out = np.zeros((len(plan["src"]), nb, nb))
    for blk, table in enumerate(plan["src"]):
        if blk < plan["m"]:
            for o in range(6 * cbt, nb):
                out[blk, o, o] = 1.0
        for ro in range(cbt):
            for co in range(cbt):
                sc = int(table[ro, co])
                if sc == -2:
                    v = np.eye(6)
                elif sc < 0:
                    continue
                else:
                    b = blocks[sc & MASK]
                    v = 0.5 * (b + b.T) if sc & SYMMETRISE else (b.T if sc & TRANSPOSED else b)
                out[blk, 6 * ro:6 * ro + 6, 6 * co:6 * co + 6] = v
    return out


def _execute(plan, blocks, g, keep=None):
    """The factorisation + solve as the fused launch sequence does it; ``keep`` (a dict) receives the factors."""
    m, nb = plan["m"], plan["nb"]
    init = _gather(plan, blocks)
    D = [np.tril(init[b]) + np.tril(init[b], -1).T for b in range(m)]          # the kernels read the lower triangle
    G = {s: init[m + s].copy() for s in range(plan["n_init_slots"])}
    bv = np.zeros((m, nb))
    for b in range(m):
        for o, f in enumerate(plan["pos_f"][b]):
            if f >= 0:
                bv[b, 6 * o:6 * o + 6] = -g[6 * f:6 * f + 6]
    L = {}
    for lv in range(plan["levels"]):
        for k, gu, gv, u, v in plan["elims"][plan["lvl_elim"][lv]:plan["lvl_elim"][lv + 1]]:
            L[k] = np.linalg.cholesky(D[k])
            for slot in (gu, gv):
                if slot >= 0:
                    G[slot] = np.linalg.solve(L[k], G[slot].T).T                   # W = A(nb, k) L^-T
            bv[k] = np.linalg.solve(L[k], bv[k])
        for kind, dst, a0, b0, a1, b1, k0, k1, init_flag in plan["ops"][plan["lvl_op"][lv]:plan["lvl_op"][lv + 1]]:
            if kind == 0:
                for a, ky in ((a0, k0), (a1, k1)):
                    if a >= 0:
                        D[dst] = D[dst] - G[a] @ G[a].T
                        bv[dst] = bv[dst] - G[a] @ bv[ky]
            else:
                acc = np.zeros((nb, nb)) if init_flag else G[dst].copy()
                for a, b in ((a0, b0), (a1, b1)):
                    if a >= 0:
                        acc -= G[a] @ G[b].T
                G[dst] = acc
    x = np.zeros(len(g))
    for lv in range(plan["levels"] - 1, -1, -1):
        for k, gu, gv, u, v in plan["elims"][plan["lvl_elim"][lv]:plan["lvl_elim"][lv + 1]]:
            r = bv[k].copy()
            if gu >= 0:
                r -= G[gu].T @ bv[u]
            if gv >= 0:
                r -= G[gv].T @ bv[v]
            bv[k] = np.linalg.solve(L[k].T, r)
            for o, f in enumerate(plan["pos_f"][k]):
                if f >= 0:
                    x[6 * f:6 * f + 6] = bv[k][6 * o:6 * o + 6]
    if keep is not None:
        keep.update(L=L, G=G)
    return x


def _apply(plan, keep, r):
    """z = M^-1 r through the stored factors, as k_bcr_fwd / k_bcr_back do it: forward with the plan's pending lists (a
    block subtracts every update it has received when it is eliminated), then backward level by level."""
    m, nb = plan["m"], plan["nb"]
    L, G = keep["L"], keep["G"]
    bv = np.zeros((m, nb))
    for lv in range(plan["levels"]):
        for k, gu, gv, u, v in plan["elims"][plan["lvl_elim"][lv]:plan["lvl_elim"][lv + 1]]:
            b = np.zeros(nb)
            for o, f in enumerate(plan["pos_f"][k]):
                if f >= 0:
                    b[6 * o:6 * o + 6] = r[6 * f:6 * f + 6]
            for slot, j in plan["pend"][plan["pend_ptr"][k]:plan["pend_ptr"][k + 1]]:
                b -= G[slot] @ bv[j]
            bv[k] = np.linalg.solve(L[k], b)
    # the same forward sweep as the GPU runs it: per level the task list (running sums per block, solve when eliminated)
    bv2, acc = np.zeros((m, nb)), np.full((m, nb), np.nan)
    for lv in range(plan["levels"]):
        done_now = {}
        for d, a0, k0, a1, k1, fl in plan["fwd"][plan["lvl_fwd"][lv]:plan["lvl_fwd"][lv + 1]]:
            s = np.zeros(nb)
            for a, ky in ((a0, k0), (a1, k1)):
                if a >= 0:
                    s += G[a] @ bv2[ky]
            if not (fl & 1) and not (fl & 4):
                s += acc[d]
            if not (fl & 4) and not (fl & 2):
                acc[d] = s
            if fl & 2:
                b = np.zeros(nb)
                for o, f in enumerate(plan["pos_f"][d]):
                    if f >= 0:
                        b[6 * o:6 * o + 6] = r[6 * f:6 * f + 6]
                done_now[d] = np.linalg.solve(L[d], b - (0.0 if fl & 4 else s))
        for d, y in done_now.items():          # (a level's results become visible to the NEXT level's tasks)
            bv2[d] = y
    np.testing.assert_allclose(bv2, bv, rtol=0, atol=1e-12 * max(1.0, np.abs(bv).max()))
    z = np.zeros(len(r))
    for lv in range(plan["levels"] - 1, -1, -1):
        for k, gu, gv, u, v in plan["elims"][plan["lvl_elim"][lv]:plan["lvl_elim"][lv + 1]]:
            t = bv[k].copy()
            if gu >= 0:
                t -= G[gu].T @ bv[u]
            if gv >= 0:
                t -= G[gv].T @ bv[v]
            bv[k] = np.linalg.solve(L[k].T, t)
            for o, f in enumerate(plan["pos_f"][k]):
                if f >= 0:
                    z[6 * f:6 * f + 6] = bv[k][6 * o:6 * o + 6]
    return z


CASES = [(12, 2, 0), (40, 3, 0), (41, 5, 0), (64, 5, 1), (37, 4, 1), (100, 8, 0), (90, 9, 1), (70, 15, 0), (113, 16, 0),
         (48, 16, 1), (64, 16, 1), (83, 16, 1), (170, 17, 1), (420, 20, 1), (150, 21, 0), (33, 10, 1), (55, 10, 1), (7, 1, 0)]


@pytest.mark.parametrize("nf,band,ring", CASES)
def test_schedule_solves_the_system(nf, band, ring):
    pairs, blocks, A, g = _system(nf, band, bool(ring), seed=nf * 100 + band)
    plan = _capi.bcr_plan(nf, pairs)
    assert plan is not None, "a chain / ring of this bandwidth has a schedule"
    assert plan["cyclic"] == bool(ring) and plan["nb"] == 16 * plan["nt"] and plan["nt"] in (2, 4, 6, 8)
    assert not plan["reordered"]
    assert sorted(plan["pos_f"][plan["pos_f"] >= 0].tolist()) == list(range(nf))
    sizes = np.diff(plan["blk_start"])
    assert plan["blk_start"][0] == 0 and plan["blk_start"][-1] == nf and sizes.min() >= 1 and 6 * sizes.max() <= plan["nb"]
    if ring:
        assert sizes.min() >= band and plan["m"] >= 3          # every block of a ring separates its neighbours
    # every block is eliminated exactly once; the last level is the root alone
    ks = plan["elims"][:, 0]
    assert sorted(ks.tolist()) == list(range(plan["m"]))
    assert plan["lvl_elim"][-1] - plan["lvl_elim"][-2] == 1 and tuple(plan["elims"][-1, 1:]) == (-1, -1, -1, -1)
    assert plan["levels"] <= int(np.ceil(np.log2(plan["m"]))) + 1
    keep = {}
    x = _execute(plan, blocks, g, keep)
    want = np.linalg.solve(A, -g)
    np.testing.assert_allclose(x, want, rtol=0, atol=1e-13 * max(1.0, np.abs(want).max()) * 10)
    # a NEW right-hand side through the stored factors (the preconditioner path: forward by pending lists, then backward)
    r = np.random.default_rng(nf).uniform(-1.0, 1.0, size=6 * nf)
    assert plan["n_rest"] == 0
    np.testing.assert_allclose(_apply(plan, keep, r), np.linalg.solve(A, r), rtol=0, atol=1e-12)


@pytest.mark.parametrize("nf,band", [(60, 3), (150, 7), (200, 10), (97, 2)])
def test_shuffled_camera_numbering_is_reordered(nf, band):
    """The chain is there but the cameras are numbered at random (registration order of an incremental reconstruction):
    in index order the pattern has no bandwidth, reverse Cuthill-McKee finds the chain again."""
    pairs, blocks, A, g = _system(nf, band, False, seed=nf + band, shuffle=True)
    plan = _capi.bcr_plan(nf, pairs)
    assert plan is not None and plan["reordered"] and not plan["cyclic"]
    assert sorted(plan["pos_f"][plan["pos_f"] >= 0].tolist()) == list(range(nf))
    x = _execute(plan, blocks, g)
    want = np.linalg.solve(A, -g)
    np.testing.assert_allclose(x, want, rtol=0, atol=1e-12 * max(1.0, np.abs(want).max()))


def test_loop_closure_inside_a_chain():
    """A chain with one long-range pair (cameras 10 and 40 see the same landmarks: a revisit) has no bandwidth in index
    order; reordered, the loop 10..40 folds onto itself and the bandwidth doubles at most."""
    nf = 60
    rng = np.random.default_rng(4)
    pairs = np.array(sorted([(i, i) for i in range(nf)] + [(i, i + 1) for i in range(nf - 1)] + [(i, i + 2) for i in range(nf - 2)]
                            + [(10, 40), (10, 41), (11, 40)]), dtype=np.int32)
    blocks = rng.uniform(-1.0, 1.0, size=(len(pairs), 6, 6))
    A = np.zeros((6 * nf, 6 * nf))
    for k, (i, j) in enumerate(pairs):
        if i == j:
            blocks[k] = 0.5 * (blocks[k] + blocks[k].T) + 40.0 * np.eye(6)
            A[6 * i:6 * i + 6, 6 * i:6 * i + 6] = blocks[k]
        else:
            A[6 * i:6 * i + 6, 6 * j:6 * j + 6] = blocks[k]
            A[6 * j:6 * j + 6, 6 * i:6 * i + 6] = blocks[k].T
    g = rng.uniform(-1.0, 1.0, size=6 * nf)
    plan = _capi.bcr_plan(nf, pairs)
    assert plan is not None and plan["reordered"]
    np.testing.assert_allclose(_execute(plan, blocks, g), np.linalg.solve(A, -g), rtol=0, atol=1e-12)


def test_patterns_without_a_schedule():
    nf = 60
    dense = np.array([(i, j) for i in range(nf) for j in range(i, nf)], dtype=np.int32)
    assert _capi.bcr_plan(nf, dense) is None                       # everybody sees everybody
    wide = np.array([(i, j) for i in range(nf) for j in range(i, nf) if j - i <= 25], dtype=np.int32)
    assert _capi.bcr_plan(nf, wide) is None                        # band beyond 21 cameras (8 tiles of 16 scalars)
    star = np.array([(i, i) for i in range(nf)] + [(0, i) for i in range(1, nf)], dtype=np.int32)
    assert _capi.bcr_plan(nf, star) is None                        # one camera sees what every other sees: no order helps
    assert _capi.bcr_plan(1, np.array([(0, 0)], dtype=np.int32)) is None
    with pytest.raises(_capi.BaError):
        _capi.bcr_plan(4, np.array([(0, 7)], dtype=np.int32))       # index out of range


@pytest.mark.parametrize("nf,band,ring,n_far", [(120, 6, 0, 15), (300, 12, 1, 60), (500, 20, 1, 200)])
def test_chain_plus_remainder_gets_a_preconditioner(nf, band, ring, n_far):
    """A chain / ring plus far-apart pairs (revisits): no exact schedule, but ``ba_bcr_plan_rest`` plans the in-band part.
    Executed with NumPy, the plan factors exactly the in-band matrix M (the far blocks appear in no table), and its
    forward / backward substitution applies M^-1 to any right-hand side; CG on the whole matrix preconditioned by it
    converges in a handful of iterations where unpreconditioned CG needs many."""
    pairs, blocks, A, g = _system(nf, band, bool(ring), seed=7 * nf + band)
    rng = np.random.default_rng(nf)
    have = {(int(i), int(j)) for i, j in pairs}
    far = set()
    while len(far) < n_far:
        i = int(rng.integers(0, nf)); j = int(rng.integers(0, nf))
        i, j = min(i, j), max(i, j)
        if min(j - i, nf - (j - i)) > 25 and (i, j) not in have:
            far.add((i, j))
    far = sorted(far)
    all_pairs = np.array(sorted(have | set(far)), dtype=np.int32)
    index_of = {(int(i), int(j)): k for k, (i, j) in enumerate(all_pairs)}
    all_blocks = np.zeros((len(all_pairs), 6, 6))
    for k, (i, j) in enumerate(pairs):
        all_blocks[index_of[(int(i), int(j))]] = blocks[k]
    S = A.copy()
    for (i, j) in far:
        b = rng.uniform(-0.3, 0.3, size=(6, 6))
        all_blocks[index_of[(i, j)]] = b
        S[6 * i:6 * i + 6, 6 * j:6 * j + 6] = b
        S[6 * j:6 * j + 6, 6 * i:6 * i + 6] = b.T
    assert _capi.bcr_plan(nf, all_pairs) is None                    # no exact schedule
    plan = _capi.bcr_plan(nf, all_pairs, allow_rest=True)
    assert plan is not None and plan["cyclic"] == bool(ring)
    used = set((plan["src"][plan["src"] >= 0] & MASK).tolist())
    left_out = [p for p in far if index_of[p] not in used]          # (a far pair may still fall into neighbouring blocks)
    assert plan["n_rest"] == len(left_out) >= 0.9 * len(far)
    assert used | {index_of[p] for p in left_out} == set(range(len(all_pairs)))      # every other block is in the tables
    M = S.copy()                                                    # the in-band matrix: S without the blocks left out
    for (i, j) in left_out:
        M[6 * i:6 * i + 6, 6 * j:6 * j + 6] = 0.0
        M[6 * j:6 * j + 6, 6 * i:6 * i + 6] = 0.0
    keep = {}
    x = _execute(plan, all_blocks, g, keep)
    np.testing.assert_allclose(x, np.linalg.solve(M, -g), rtol=0, atol=1e-12)        # the plan factors M, exactly
    r = rng.uniform(-1.0, 1.0, size=6 * nf)
    np.testing.assert_allclose(_apply(plan, keep, r), np.linalg.solve(M, r), rtol=0, atol=1e-12)

    def pcg(apply_m):
        xk = np.zeros(6 * nf); rk = -g.copy(); zk = apply_m(rk); pk = zk.copy(); rz = rk @ zk
        for it in range(1, 500):
            q = S @ pk
            alpha = rz / (pk @ q)
            xk += alpha * pk; rk -= alpha * q
            if np.linalg.norm(rk) <= 1e-10 * np.linalg.norm(g):
                return xk, it
            zk = apply_m(rk); rz_new = rk @ zk
            pk = zk + (rz_new / rz) * pk; rz = rz_new
        return xk, 500
    x_pre, it_pre = pcg(lambda v: _apply(plan, keep, v))
    np.testing.assert_allclose(x_pre, np.linalg.solve(S, -g), rtol=0, atol=1e-9)
    assert it_pre <= 12, it_pre


def test_too_much_remainder_is_no_plan():
    nf = 200
    rng = np.random.default_rng(3)
    pairs = {(i, i) for i in range(nf)} | {(i, i + 1) for i in range(nf - 1)}
    while len(pairs) < nf + 10 * (nf - 1):                            # nine times as many far pairs as chain pairs
        i, j = sorted(int(v) for v in rng.integers(0, nf, size=2))
        if j - i > 30:
            pairs.add((i, j))
    assert _capi.bcr_plan(nf, np.array(sorted(pairs), dtype=np.int32), allow_rest=True) is None
