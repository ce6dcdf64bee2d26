// Block cyclic reduction: a direct solve of the reduced camera system for chain- and ring-shaped co-visibility that
// uses the whole chip -- the block-banded Cholesky of ba_banded.h is a serial chain of Nf block columns on two
// workgroups (1.7 ms of a 2.6 ms LM step on the Venice shape), and a ring with loop closure is not banded at all.
//
// The free cameras are cut into m consecutive blocks of at most CB cameras (CB >= the half bandwidth), so S is block
// tridiagonal -- plus one corner block when the first and the last cameras share landmarks (a closed ring).  It is
// factored as a Cholesky factorisation in ODD-EVEN (nested dissection) order: per level every second remaining block
// is eliminated, all of them independently:
//      L_k L_k^T = D_k,   W_u = A(u,k) L_k^-T,  W_v = A(v,k) L_k^-T,  y_k = L_k^-1 b_k        (k_bcr_factor)
//      D_u -= W_u W_u^T,  b_u -= W_u y_k,  (same for v),  A(u,v) -= W_u W_v^T               (k_bcr_update)
// and the survivors form a block tridiagonal (or cyclic) system of half the size: ceil(log2 m) + 1 levels instead of
// Nf serial columns.  Backward, level by level:  x_k = L_k^-T (y_k - W_u^T x_u - W_v^T x_v)    (k_bcr_back).
// Same arithmetic as a Cholesky factorisation of P S P^T (S is positive definite: no pivoting), so the step agrees with
// the banded / dense solvers to rounding.
//
// Kernels (NB = 16 NT scalars per block, NT = 2, 4, 6 or 8 MFMA tiles):
//   k_bcr_gather  every element of the diagonal blocks and the initial couplings from its 6x6 block of S (a host-built
//                 table), the right-hand side and the pivot flag: one launch, no memset, no scatter.
//   k_bcr_factor  two workgroups per eliminated block (one per neighbour, both factor D_k): D_k lives in LDS and is
//                 factored right-looking in 16-column steps.  ONE wave owns the serial chain: it factors the 16x16
//                 diagonal tile in registers (a row per lane, readlane broadcasts, the tile's inverse in the same sweep),
//                 computes the next diagonal tile's panel and update itself and goes on factoring while three more waves
//                 apply the step to the rest of D_k on the fp64 matrix cores (v_mfma_f64_16x16x4_f64); the neighbour's
//                 coupling block rides along as NT 16-row strips, one wave each, whose tiles never leave the
//                 accumulators (a strip visits LDS only to turn an accumulator tile into an MFMA operand); the
//                 right-hand side is one more row of D_k.
//   k_bcr_update  one wave (= one workgroup) per 16x16 output tile, operands straight from L2: lane (row, l4) loads
//                 pieces of its row such that four lanes read 64 contiguous bytes (the k-index of an MFMA lane is free
//                 as long as both operands agree); no LDS.  Workgroups are renumbered so that one update stays on one XCD.
//   k_bcr_back    one workgroup per block of the level, every global load up front, blocked substitution out of LDS.
// The schedule (which block, which neighbours, which coupling slot, per level) is planned on the host once per pattern
// (bcr_make_plan) and executed from tables.  The planner takes the cameras in index order, else along a chain coordinate
// computed from the co-visibility graph, else in reverse Cuthill-McKee order: the numbering of the cameras does not
// have to follow the trajectory.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <queue>
#include <type_traits>
#include <utility>
#include <vector>

#include "ba_banded.h"  // band_readlane, band_rsqrt, band_wave_sync
#include "ba_device.h"  // lds_f64
#include "ba_device.h"  // wave_sum_dpp
#include "ba_dense.h"   // mfma_d4
#include "ba_tile16.h" // bcr_potrf16

namespace ba {

struct BcrElim { int32_t k, gu, gv, u, v; };                       // block, coupling slots (-1: none), neighbour blocks
struct BcrOp { int32_t kind, dst, a0, b0, a1, b1, k0, k1, init; };  // kind 0: D[dst] -= W[a0] W[a0]^T (+ a1), b[dst] -= W[a0] y[k0] (+ ..)
                                                                    // kind 1: G[dst] (init: = 0) -= W[a0] W[b0]^T (+ W[a1] W[b1]^T)
constexpr int32_t BCR_SRC_TRANSPOSED = 1 << 30, BCR_SRC_SYMMETRISE = 1 << 29, BCR_SRC_MASK = (1 << 29) - 1;
constexpr int32_t BCR_SRC_ZERO = -1, BCR_SRC_UNIT = -2;

struct BcrPlan {
    bool ok = false, cyclic = false;
    int nf = 0, m = 0, nt = 0, nb = 0, band = 0, n_slots = 0, n_init_slots = 0;
    bool reordered = false;          // the cameras are not taken in index order (bcr_chain_order / bcr_rcm_order)
    std::vector<int32_t> blk_start;  // [m + 1] first POSITION (in the elimination order of the cameras) of each block
    std::vector<int32_t> pos_f;      // [m * (nb / 6)] free camera at (block, offset), -1 = padding
    std::vector<int32_t> cam_blk;    // [nf] block << 8 | offset inside the block, per free camera
    std::vector<int32_t> edge_slot;  // [m] slot of the coupling between block b and b + 1 (entry m - 1: the ring's corner
                                     // block, between 0 and m - 1) | bit 30: rows belong to the higher block; -1: none
    std::vector<int32_t> src;        // [(m + n_init_slots) * cbt * cbt], cbt = nb / 6: per 6x6 block of the diagonal blocks and
                                     // the initial coupling slots, where it comes from: slot of S | BCR_SRC_*, or BCR_SRC_ZERO / _UNIT
    std::vector<int32_t> lvl_elim, lvl_op;  // [levels + 1] offsets into elims / ops
    std::vector<BcrElim> elims;
    std::vector<BcrOp> ops;
    // A plan "with a remainder": some blocks of S lie outside the chain / ring (a revisit, a few long tracks).  The
    // schedule then factors the IN-BAND part of S only, and that factorisation is used as the preconditioner of a CG
    // iteration on the whole S (do_bcr_pcg in ba_engine.hip); n_rest = upper blocks left out.
    int64_t n_rest = 0;
    // forward substitution of a NEW right-hand side through the stored factors (k_bcr_fwd): per block the updates it
    // has received by the time it is eliminated, b_k -= W[slot] y_j, as (slot, j) pairs in schedule order
    std::vector<int32_t> pend_ptr, pend;   // [m + 1], [2 * n]
    // the same sweep as the GPU runs it (k_bcr_fwd), one launch per level: tasks of level l = the blocks eliminated at l
    // (solve y = X (r - acc)) and the other survivors of level l - 1's eliminations, all of which first add the updates
    // of level l - 1 to their running sum acc.  6 words per task: block, slot a0, source k0, slot a1 (-1: none), source
    // k1, flags (1: first update of the block -- acc is written, not added to; 2: eliminated at this level -- solve;
    // 4: no update at all -- acc counts as zero)
    std::vector<int32_t> lvl_fwd, fwd;     // [levels + 1] offsets (in tasks), [6 * n_tasks]
    int levels() const { return (int)lvl_elim.size() - 1; }
};

// Reverse Cuthill-McKee order of the camera graph (an edge per off-diagonal block of S): BFS from a pseudo-peripheral
// camera, neighbours by increasing degree.  A chain whose cameras are NUMBERED in any other order than along the chain
// (registration order of an incremental reconstruction, shuffled image lists) gets its small bandwidth back.
inline void bcr_rcm_order(int nf, const int64_t *keys, size_t nk, std::vector<int32_t> &order) {
    std::vector<int32_t> deg(nf + 1, 0);
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        if (i != j) { deg[i]++; deg[j]++; }
    }
    std::vector<int32_t> row(nf + 1, 0);
    for (int c = 0; c < nf; ++c) row[c + 1] = row[c] + deg[c];
    std::vector<int32_t> adj(row[nf]), fill(row.begin(), row.end() - 1);
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        if (i != j) { adj[fill[i]++] = j; adj[fill[j]++] = i; }
    }
    for (int c = 0; c < nf; ++c)
        std::sort(adj.begin() + row[c], adj.begin() + row[c + 1], [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
    std::vector<int32_t> level(nf, -1), mark(nf, -1);
    order.clear();
    order.reserve(nf);
    // breadth-first search from `root` inside the not yet ordered part; returns the last camera reached (smallest degree
    // of the last level) and appends the visit order to `out`
    auto bfs = [&](int root, int stamp, std::vector<int32_t> &out) {
        const size_t first = out.size();
        out.push_back(root);
        mark[root] = stamp;
        level[root] = 0;
        for (size_t h = first; h < out.size(); ++h) {
            const int c = out[h];
            for (int e = row[c]; e < row[c + 1]; ++e) {
                const int n = adj[e];
                if (mark[n] == stamp || mark[n] == -2) continue;
                mark[n] = stamp;
                level[n] = level[c] + 1;
                out.push_back(n);
            }
        }
        int far = out.back();
        for (size_t h = out.size(); h-- > first;) {
            if (level[out[h]] != level[out.back()]) break;
            if (deg[out[h]] < deg[far]) far = out[h];
        }
        return far;
    };
    int stamp = 0;
    std::vector<int32_t> tmp;
    for (int c0 = 0; c0 < nf; ++c0) {
        if (mark[c0] == -2) continue;
        int root = c0;
        for (int rep = 0; rep < 3; ++rep) {   // pseudo-peripheral start: walk to the far end a few times
            tmp.clear();
            const int far = bfs(root, stamp++, tmp);
            if (far == root) break;
            root = far;
        }
        tmp.clear();
        bfs(root, stamp++, tmp);
        for (int c : tmp) { mark[c] = -2; order.push_back(c); }
    }
    std::reverse(order.begin(), order.end());
}

// Order along a chain whose co-visibility windows are irregular: every edge gets a LENGTH, the Jaccard distance of the
// two cameras' neighbourhoods (cameras next to each other on the trajectory share nearly all their neighbours, cameras a
// window apart share few), and the cameras are sorted by (distance from one end - distance from the other end) of the
// weighted shortest paths.  Breadth-first levels (Cuthill-McKee) only resolve the position to a window width, which
// doubles the bandwidth of such a chain; this coordinate resolves it inside the window.  Ends: the two ends of a
// double sweep (farthest camera from camera 0, farthest from that).  One connected component only (else false).
inline bool bcr_chain_order(int nf, const int64_t *keys, size_t nk, std::vector<int32_t> &order) {
    std::vector<int32_t> row(nf + 1, 0);
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        if (i != j) { row[i + 1]++; row[j + 1]++; }
    }
    for (int c = 0; c < nf; ++c) row[c + 1] += row[c];
    std::vector<int32_t> adj(row[nf]), fill(row.begin(), row.end() - 1);
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        if (i != j) { adj[fill[i]++] = j; adj[fill[j]++] = i; }
    }
    for (int c = 0; c < nf; ++c) std::sort(adj.begin() + row[c], adj.begin() + row[c + 1]);
    std::vector<double> len(row[nf]);
    for (int c = 0; c < nf; ++c)
        for (int e = row[c]; e < row[c + 1]; ++e) {
            const int n = adj[e];
            int a = row[c], b = row[n], common = 2;   // (the two cameras themselves: closed neighbourhoods)
            while (a < row[c + 1] && b < row[n + 1]) {
                if (adj[a] == adj[b]) { ++common; ++a; ++b; }
                else if (adj[a] < adj[b]) ++a;
                else ++b;
            }
            const int uni = (row[c + 1] - row[c] + 1) + (row[n + 1] - row[n] + 1) - common;
            len[e] = 1.0 - (double)common / (double)uni + 1e-3;
        }
    auto dijkstra = [&](int root, std::vector<double> &dist) {
        dist.assign(nf, 1e300);
        typedef std::pair<double, int> Item;
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
        dist[root] = 0.0;
        pq.push(Item(0.0, root));
        int far = root;
        while (!pq.empty()) {
            const Item it = pq.top();
            pq.pop();
            if (it.first > dist[it.second]) continue;
            far = it.second;
            for (int e = row[it.second]; e < row[it.second + 1]; ++e) {
                const double d = it.first + len[e];
                if (d < dist[adj[e]]) { dist[adj[e]] = d; pq.push(Item(d, adj[e])); }
            }
        }
        return far;
    };
    std::vector<double> d0, d1, d2;
    const int end1 = dijkstra(0, d0);
    for (int c = 0; c < nf; ++c)
        if (d0[c] >= 1e300) return false;   // not connected
    const int end2 = dijkstra(end1, d1);
    dijkstra(end2, d2);
    order.resize(nf);
    for (int c = 0; c < nf; ++c) order[c] = c;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        const double xa = d1[a] - d2[a], xb = d1[b] - d2[b];
        return xa != xb ? xa < xb : a < b;
    });
    return true;
}

// Plans the elimination for the pattern `keys` (sorted unique (i << 32 | j), i <= j, free-camera indices) with the
// cameras taken in the order `order`.
constexpr int BCR_MAX_BAND = 21;   // cameras per block of 128 scalars
inline bool bcr_plan_with_order(int nf, const int64_t *keys, size_t nk, const std::vector<int32_t> &order, BcrPlan &P,
                                bool allow_rest = false) {
    P = BcrPlan{};
    if (nf < 2) return false;
    std::vector<int32_t> pos(nf);
    for (int k = 0; k < nf; ++k) pos[order[k]] = k;
    int near_b = 0, wrap_b = 0;
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff), d = std::abs(pos[j] - pos[i]);
        if (d == 0) continue;
        // (with a remainder allowed, pairs further apart than any block could hold do not set the bandwidth)
        if (allow_rest && std::min(d, nf - d) > BCR_MAX_BAND) continue;
        if (2 * (int64_t)d <= nf) near_b = std::max(near_b, d);
        else wrap_b = std::max(wrap_b, nf - d);
    }
    const bool cyc = wrap_b > 0;
    const int band = std::max(1, std::max(near_b, wrap_b));
    int m = 0, nt = 0;
    std::vector<int32_t> start;
    for (int cand : {2, 4, 6, 8}) {
        const int fit = 16 * cand / 6;   // cameras that fit one block
        if (fit < band) continue;
        const int mm = (nf + fit - 1) / fit;
        if (!cyc) {
            start.assign(mm + 1, 0);
            for (int b = 0; b <= mm; ++b) start[b] = std::min(nf, b * fit);
        } else {
            // every block of a ring separates its two neighbours: all of them at least `band` cameras
            if (mm < 3 || nf / mm < band) continue;
            const int base = nf / mm, rem = nf % mm;
            start.assign(mm + 1, 0);
            for (int b = 0; b < mm; ++b) start[b + 1] = start[b] + base + (b < rem ? 1 : 0);
        }
        m = mm;
        nt = cand;
        break;
    }
    if (nt == 0 || m < 2) return false;
    P.nf = nf; P.m = m; P.nt = nt; P.nb = 16 * nt; P.band = band; P.cyclic = cyc;
    P.blk_start = start;
    const int cbt = P.nb / 6;
    P.cam_blk.resize(nf);
    P.pos_f.assign((size_t)m * cbt, -1);
    for (int b = 0; b < m; ++b)
        for (int k = start[b]; k < start[b + 1]; ++k) {
            P.cam_blk[order[k]] = (b << 8) | (k - start[b]);
            P.pos_f[(size_t)b * cbt + (k - start[b])] = order[k];
        }
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        const int bi = P.cam_blk[i] >> 8, bj = P.cam_blk[j] >> 8, lo = std::min(bi, bj), hi = std::max(bi, bj);
        if (!(hi == lo || hi == lo + 1 || (cyc && lo == 0 && hi == m - 1))) {
            if (!allow_rest) return false;
            ++P.n_rest;
        }
    }
    // pass 1: who is eliminated when, next to whom
    struct Lv { int k, u, v; };
    std::vector<std::vector<Lv>> lv;
    std::vector<int> act(m), etime(m, 0);
    for (int b = 0; b < m; ++b) act[b] = b;
    bool ring = cyc;
    int t = 0;
    while (act.size() > 1) {
        const int p = (int)act.size();
        std::vector<Lv> es;
        std::vector<int> sv;
        for (int pos = 0; pos < p; ++pos) {
            if (pos & 1) {
                const int v = pos + 1 < p ? act[pos + 1] : (ring ? act[0] : -1);
                es.push_back({act[pos], act[pos - 1], v});
                etime[act[pos]] = t++;
            } else {
                sv.push_back(act[pos]);
            }
        }
        lv.push_back(es);
        act.swap(sv);
        if (ring && act.size() < 3) ring = false;   // two survivors: their two couplings are one block
    }
    lv.push_back({Lv{act[0], -1, -1}});
    etime[act[0]] = t++;
    // pass 2: coupling slots (rows = the block eliminated later, columns = the earlier one) and the update lists
    typedef std::pair<int, int> Edge;
    auto edge = [](int a, int b) { return Edge(std::min(a, b), std::max(a, b)); };
    std::map<Edge, int> edges;
    P.edge_slot.assign(m, -1);
    for (int b = 0; b + 1 < m; ++b) {
        edges[edge(b, b + 1)] = b;
        P.edge_slot[b] = b | ((etime[b + 1] > etime[b]) ? (1 << 30) : 0);
    }
    P.n_init_slots = m - 1;
    if (cyc) {
        edges[edge(0, m - 1)] = m - 1;
        P.edge_slot[m - 1] = (m - 1) | ((etime[m - 1] > etime[0]) ? (1 << 30) : 0);
        P.n_init_slots = m;
    }
    P.n_slots = P.n_init_slots;
    P.lvl_elim.push_back(0);
    P.lvl_op.push_back(0);
    for (const auto &es : lv) {
        std::map<int, int> diag_op;
        std::map<Edge, int> coup_op;
        auto add_diag = [&](int dst, int slot, int k) {
            auto it = diag_op.find(dst);
            if (it == diag_op.end()) {
                diag_op[dst] = (int)P.ops.size();
                P.ops.push_back(BcrOp{0, dst, slot, -1, -1, -1, k, -1, 0});
            } else {
                BcrOp &op = P.ops[it->second];
                op.a1 = slot;
                op.k1 = k;
            }
        };
        for (const Lv &e : es) {
            BcrElim el{e.k, -1, -1, e.u, e.v};
            if (e.u >= 0) el.gu = edges.at(edge(e.u, e.k));
            if (e.v >= 0) el.gv = edges.at(edge(e.k, e.v));
            P.elims.push_back(el);
            if (e.u >= 0) add_diag(e.u, el.gu, e.k);
            if (e.v >= 0) add_diag(e.v, el.gv, e.k);
            if (e.u >= 0 && e.v >= 0) {
                const bool u_later = etime[e.u] > etime[e.v];
                const int ar = u_later ? el.gu : el.gv, ac = u_later ? el.gv : el.gu;   // rows: the later block
                const Edge uv = edge(e.u, e.v);
                auto it = coup_op.find(uv);
                if (it != coup_op.end()) {
                    BcrOp &op = P.ops[it->second];
                    op.a1 = ar;
                    op.b1 = ac;
                } else {
                    int init = 0, slot;
                    auto ex = edges.find(uv);
                    if (ex != edges.end()) slot = ex->second;
                    else { slot = P.n_slots++; edges[uv] = slot; init = 1; }
                    coup_op[uv] = (int)P.ops.size();
                    P.ops.push_back(BcrOp{1, slot, ar, ac, -1, -1, -1, -1, init});
                }
            }
            if (e.u >= 0) edges.erase(edge(e.u, e.k));
            if (e.v >= 0) edges.erase(edge(e.k, e.v));
        }
        P.lvl_elim.push_back((int32_t)P.elims.size());
        P.lvl_op.push_back((int32_t)P.ops.size());
    }
    // where every 6x6 block of the diagonal blocks / initial coupling slots comes from
    P.src.assign((size_t)(m + P.n_init_slots) * cbt * cbt, BCR_SRC_ZERO);
    for (int b = 0; b < m; ++b)   // padding cameras: unit diagonal
        for (int o = start[b + 1] - start[b]; o < cbt; ++o) P.src[((size_t)b * cbt + o) * cbt + o] = BCR_SRC_UNIT;
    for (size_t q = 0; q < nk; ++q) {
        // S block q = A[camera i rows][camera j columns], i <= j
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        const int bi = P.cam_blk[i] >> 8, oi = P.cam_blk[i] & 255, bj = P.cam_blk[j] >> 8, oj = P.cam_blk[j] & 255;
        if (!(bi == bj || std::abs(bi - bj) == 1 || (cyc && std::min(bi, bj) == 0 && std::max(bi, bj) == m - 1))) continue;   // remainder
        if (i == j) {
            P.src[((size_t)bi * cbt + oi) * cbt + oi] = (int32_t)q | BCR_SRC_SYMMETRISE;
        } else if (bi == bj) {   // lower triangle of the diagonal block: the row is the camera further down the block
            if (oj > oi) P.src[((size_t)bi * cbt + oj) * cbt + oi] = (int32_t)q | BCR_SRC_TRANSPOSED;
            else P.src[((size_t)bi * cbt + oi) * cbt + oj] = (int32_t)q;
        } else {
            const int lo = std::min(bi, bj), hi = std::max(bi, bj);
            const int es = P.edge_slot[hi == lo + 1 ? lo : m - 1], slot = es & 0x3fffffff;
            const int rows = (es >> 30 & 1) ? hi : lo;   // the block whose cameras are the rows of the slot
            if (bi == rows) P.src[((size_t)(m + slot) * cbt + oi) * cbt + oj] = (int32_t)q;
            else P.src[((size_t)(m + slot) * cbt + oj) * cbt + oi] = (int32_t)q | BCR_SRC_TRANSPOSED;
        }
    }
    // pending right-hand-side updates per block (k_bcr_fwd)
    {
        std::vector<std::vector<int32_t>> pl(m);
        for (const BcrOp &op : P.ops) {
            if (op.kind != 0) continue;
            pl[op.dst].push_back(op.a0); pl[op.dst].push_back(op.k0);
            if (op.a1 >= 0) { pl[op.dst].push_back(op.a1); pl[op.dst].push_back(op.k1); }
        }
        P.pend_ptr.assign(m + 1, 0);
        for (int b = 0; b < m; ++b) P.pend_ptr[b + 1] = P.pend_ptr[b] + (int32_t)pl[b].size() / 2;
        P.pend.clear();
        for (int b = 0; b < m; ++b) P.pend.insert(P.pend.end(), pl[b].begin(), pl[b].end());
    }
    // forward tasks per level (k_bcr_fwd)
    {
        const int nl = (int)P.lvl_elim.size() - 1;
        std::vector<char> touched(m, 0);
        P.lvl_fwd.assign(1, 0);
        P.fwd.clear();
        for (int l = 0; l < nl; ++l) {
            std::vector<char> elim_now(m, 0), listed(m, 0);
            for (int e = P.lvl_elim[l]; e < P.lvl_elim[l + 1]; ++e) elim_now[P.elims[e].k] = 1;
            if (l > 0)
                for (int o = P.lvl_op[l - 1]; o < P.lvl_op[l]; ++o) {
                    const BcrOp &op = P.ops[o];
                    if (op.kind != 0) continue;
                    const int32_t fl = (touched[op.dst] ? 0 : 1) | (elim_now[op.dst] ? 2 : 0);
                    P.fwd.insert(P.fwd.end(), {op.dst, op.a0, op.k0, op.a1, op.k1, fl});
                    touched[op.dst] = 1;
                    listed[op.dst] = 1;
                }
            for (int e = P.lvl_elim[l]; e < P.lvl_elim[l + 1]; ++e) {
                const int k = P.elims[e].k;
                if (!listed[k]) P.fwd.insert(P.fwd.end(), {k, -1, -1, -1, -1, 2 | (touched[k] ? 0 : 4)});
            }
            P.lvl_fwd.push_back((int32_t)(P.fwd.size() / 6));
        }
    }
    P.ok = true;
    return true;
}

// Plans the elimination for the pattern `keys`: the cameras in index order first (chains and rings as the caller numbered
// them), else along the chain coordinate of bcr_chain_order, else in reverse Cuthill-McKee order.  false: no order tried makes the pattern a chain / ring with a half
// bandwidth of at most 21 cameras (or it is too small to bother).
// allow_rest: when no order makes the WHOLE pattern a chain / ring, accept a plan for its in-band part (cameras in
// index order, else along the chain coordinate) if that part holds at least an eighth of the off-diagonal blocks -- the
// factorisation of the in-band part then preconditions a CG iteration on the whole system (P.n_rest > 0).  (A revisited
// landmark couples every camera of its first window with every camera of the second: far-apart blocks are many, each
// carried by a point or two, while an in-band block is carried by dozens.  Whether the preconditioner is any good shows
// at run time -- do_bcr_pcg gives it up when CG does not converge quickly.)
inline bool bcr_make_plan(int nf, const int64_t *keys, size_t nk, BcrPlan &P, bool allow_rest = false) {
    if (nf < 2) { P = BcrPlan{}; return false; }
    std::vector<int32_t> order(nf), chain;
    for (int c = 0; c < nf; ++c) order[c] = c;
    if (bcr_plan_with_order(nf, keys, nk, order, P)) return true;
    const bool have_chain = bcr_chain_order(nf, keys, nk, chain);
    if (have_chain && bcr_plan_with_order(nf, keys, nk, chain, P)) {
        P.reordered = true;
        return true;
    }
    std::vector<int32_t> rcm;
    bcr_rcm_order(nf, keys, nk, rcm);
    if (bcr_plan_with_order(nf, keys, nk, rcm, P)) {
        P.reordered = true;
        return true;
    }
    if (!allow_rest || nf < 2 * BCR_MAX_BAND) return false;
    const int64_t n_off = (int64_t)nk - nf;   // (every diagonal block is in the pattern)
    if (bcr_plan_with_order(nf, keys, nk, order, P, true) && P.n_rest <= 7 * (n_off - P.n_rest)) return true;
    if (have_chain && bcr_plan_with_order(nf, keys, nk, chain, P, true) && P.n_rest <= 7 * (n_off - P.n_rest)) {
        P.reordered = true;
        return true;
    }
    P = BcrPlan{};
    return false;
}

// m diagonal blocks D_k, the coupling slots, and m more blocks for the factors L_k: D_k stays READ-ONLY while a level is
// being factored -- both workgroups of an eliminated block load it, and nothing orders the second one's load before
// the first one's write-back (they may not even be resident together), so the factor goes to a block of its own
// (and m more for the explicit inverses X_k = L_k^-1 of the preconditioner sweeps, k_bcr_inv)
inline size_t bcr_matrix_doubles(const BcrPlan &P) { return (size_t)(3 * P.m + P.n_slots) * P.nb * P.nb; }
inline size_t bcr_vec_doubles(const BcrPlan &P) { return (size_t)2 * P.m * P.nb; }   // block vectors + the forward sweep's running sums
inline size_t bcr_linv_doubles(const BcrPlan &P) { return (size_t)P.m * P.nt * 256; }
#ifndef BA_BCR_BACK_THREADS
#define BA_BCR_BACK_THREADS 1024   // threads of k_bcr_back: 256 / 512 / 1024 measured, 0.439 / 0.408 / 0.390 ms per solve (Venice shape)
#endif
inline size_t bcr_back_lds_bytes(int nt) {
    const int nb = 16 * nt;
    return ((size_t)nb * nb + (size_t)nt * 256 + (size_t)(BA_BCR_BACK_THREADS / nb) * nb + nb + 16) * sizeof(double);
}
inline size_t bcr_factor_lds_bytes(int nt) {
    const int nb = 16 * nt;
    return ((size_t)(nb + 1) * (nb + 1) + 16 * 17 + (size_t)nt * 16 * 17) * sizeof(double);
}

// ------------------------------------------------------------------------------------------------ assembly
// One pass writes EVERY element of the diagonal blocks and the initial coupling slots -- from its 6x6 block of S (upper
// blocks, row-major), zero, or the unit diagonal of the padding -- plus the right-hand side b = -g per block and the pivot
// flag.  (A memset of the blocks followed by a scatter of S took two fills and two kernels with their launch gaps: 60 us
// in front of a 400 us solve.)
__global__ void k_bcr_gather(int n_blocks, int m, int NB, const int32_t *__restrict__ src, const double *__restrict__ S,
                             const int32_t *__restrict__ pos_f, const double *__restrict__ g, double *__restrict__ DG,
                             double *__restrict__ bv, int *__restrict__ info) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nb2 = (size_t)NB * NB;
    if (t == 0) *info = 0;
    if (t < (size_t)m * NB) {
        const int b = (int)(t / NB), o = (int)(t - (size_t)b * NB), cb = NB / 6;
        const int f = o < 6 * cb ? pos_f[b * cb + o / 6] : -1;
        bv[t] = f >= 0 ? -g[(size_t)6 * f + o % 6] : 0.0;
    }
    if (t >= (size_t)n_blocks * nb2) return;
    const int blk = (int)(t / nb2), e = (int)(t - blk * nb2), r = e / NB, c = e - r * NB;
    const int cbt = NB / 6, ro = r / 6, a = r - ro * 6, co = c / 6, b = c - co * 6;
    double v = 0.0;
    if (ro >= cbt || co >= cbt) {
        v = (r == c && blk < m) ? 1.0 : 0.0;   // (rows / columns behind the last whole camera)
    } else {
        const int32_t sc = src[((size_t)blk * cbt + ro) * cbt + co];
        if (sc == BCR_SRC_UNIT) {
            v = a == b ? 1.0 : 0.0;
        } else if (sc >= 0) {
            const double *sb = S + (size_t)(sc & BCR_SRC_MASK) * 36;
            if (sc & BCR_SRC_SYMMETRISE) v = 0.5 * (sb[a * 6 + b] + sb[b * 6 + a]);
            else if (sc & BCR_SRC_TRANSPOSED) v = sb[b * 6 + a];
            else v = sb[a * 6 + b];
        }
    }
    DG[t] = v;
}

// ------------------------------------------------------------------------------------------------ factor
#define BA_BCR_DWAVES 4
#ifndef BA_BCR_UPD_WAVES
#define BA_BCR_UPD_WAVES 1   // waves (= output tiles) per workgroup of k_bcr_update: 1, 2, 4, 8, 16 measured -- 0.406, 0.415, 0.427, 0.442,
                           // 0.474 ms per solve on the Venice shape (a workgroup holds its slots until its last wave is done)
#endif
#ifdef BA_BCR_TIMING   // tools/bcr_bench.hip: cycle stamps of workgroup 0 (wave, stamp)
__device__ long long g_bcr_t[16 * 40];
#define BCR_STAMP(k) do { if (blockIdx.x == 0 && lane == 0) g_bcr_t[wv * 40 + (k)] = clock64(); } while (0)
#else
#define BCR_STAMP(k) do { } while (0)
#endif
// compile-time loop (the step index selects accumulator registers: the loop has to be unrolled whatever its size)
template <int I, int N, class F>
__device__ __forceinline__ void bcr_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        bcr_static_for<I + 1, N>(f);
    }
}
template <int NT>
__global__ __launch_bounds__(64 * (NT + BA_BCR_DWAVES)) void k_bcr_factor(const BcrElim *__restrict__ elims, const double *__restrict__ D,
                                                                         double *__restrict__ Lf, double *__restrict__ G,
                                                                         double *__restrict__ bv, double *__restrict__ Linv16,
                                                                         int *__restrict__ info) {
    constexpr int NB = 16 * NT, LDD = NB + 1, NTHREADS = 64 * (NT + BA_BCR_DWAVES);
    extern __shared__ double bcr_smem[];
    double *sD = bcr_smem;                    // (NB + 1) x LDD: D_k, row NB = right-hand side
    double *sLinv = sD + (NB + 1) * LDD;      // 16 x 17: inverse of the current diagonal tile's factor
    double *sStg = sLinv + 16 * 17;           // NT x (16 x 17): per-strip staging tile
    const BcrElim el = elims[blockIdx.x >> 1];
    const int side = blockIdx.x & 1;
    const int slot = side ? el.gv : el.gu;
    if (side == 1 && slot < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *Dk = D + (size_t)el.k * NB * NB;
    for (int e = tid; e < NB * NB; e += NTHREADS) {
        const int r = e / NB, c = e - r * NB;
        sD[r * LDD + c] = Dk[e];
    }
    for (int e = tid; e < NB; e += NTHREADS) sD[NB * LDD + e] = side == 0 ? bv[(size_t)el.k * NB + e] : 0.0;
    const bool strip = wv >= BA_BCR_DWAVES && slot >= 0;
    const int sr = wv - BA_BCR_DWAVES;
    double *Gs = G + (size_t)(slot >= 0 ? slot : 0) * NB * NB;
    double *stg = sStg + (sr >= 0 ? sr : 0) * 16 * 17;
    mfma_d4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    if (strip) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[c][v] = Gs[(size_t)(sr * 16 + l4 + 4 * v) * NB + c * 16 + l15];
    }
    BCR_STAMP(0);
    __syncthreads();
    BCR_STAMP(1);
    int bad = 0;
    double aW[4] = {0.0, 0.0, 0.0, 0.0};
    // one 16x16 tile task of the D-waves: (i, s) <- (i, s) Linv^T (panel), or (i, j) -= (i, s) (j, s)^T (trailing);
    // tile row NT is the right-hand-side row (sixteen copies of it as the MFMA operand, row 0 of the result kept)
    auto panel_tile = [&](int i, int s) {
        const int arow = i < NT ? 16 * i + l15 : NB;
        mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
        double aq[4], bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { aq[q] = lds_f64(&sD[arow * LDD + 16 * s + 4 * q + l4]); bq[q] = lds_f64(&sLinv[l15 * 17 + 4 * q + l4]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], w, 0, 0, 0);
        if (i < NT) {
#pragma unroll
            for (int v = 0; v < 4; ++v) sD[(16 * i + l4 + 4 * v) * LDD + 16 * s + l15] = w[v];
        } else if (l4 == 0) {
            sD[NB * LDD + 16 * s + l15] = w[0];
        }
    };
    auto trailing_tile = [&](int i, int j, int s) {
        const int arow = i < NT ? 16 * i + l15 : NB;
        double aq[4], bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            aq[q] = -lds_f64(&sD[arow * LDD + 16 * s + 4 * q + l4]);
            bq[q] = lds_f64(&sD[(16 * j + l15) * LDD + 16 * s + 4 * q + l4]);
        }
        mfma_d4 w;
        if (i < NT) {
#pragma unroll
            for (int v = 0; v < 4; ++v) w[v] = sD[(16 * i + l4 + 4 * v) * LDD + 16 * j + l15];
        } else {
            w = mfma_d4{sD[NB * LDD + 16 * j + l15], 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], w, 0, 0, 0);
        if (i < NT) {
#pragma unroll
            for (int v = 0; v < 4; ++v) sD[(16 * i + l4 + 4 * v) * LDD + 16 * j + l15] = w[v];
        } else if (l4 == 0) {
            sD[NB * LDD + 16 * j + l15] = w[0];
        }
    };
    // Two barriers per 16-column step s: B1 releases the factor + inverse of tile (s, s), B2 the panel (i, s).
    // Wave 0 owns the serial chain and nothing else: factor (s, s) | B1 | its own panel tile (s+1, s) and the update of
    // (s+1, s+1) | B2 | factor (s+1, s+1) ... -- the next diagonal tile is being factored while the other waves still
    // apply step s.  (A code path of its own also keeps the other waves' addresses out of its scalar registers.)
    if (wv == 0) {
        bcr_static_for<0, NT>([&](auto step_c) {
            constexpr int s = decltype(step_c)::value;
            if (bcr_potrf16(&sD[(16 * s + l15) * LDD + 16 * s], sLinv, side == 0 ? Linv16 + ((size_t)el.k * NT + s) * 256 : nullptr, lane))
                bad = 1;
            BCR_STAMP(2 + 4 * s);
            __syncthreads();
            BCR_STAMP(3 + 4 * s);
            panel_tile(s + 1, s);   // (s + 1 == NT: the right-hand-side row)
            if (s + 1 < NT) {
                band_wave_sync();
                trailing_tile(s + 1, s + 1, s);
            }
            BCR_STAMP(4 + 4 * s);
            __syncthreads();
            BCR_STAMP(5 + 4 * s);
        });
    } else {
        bcr_static_for<0, NT>([&](auto step_c) {
            constexpr int s = decltype(step_c)::value;
            BCR_STAMP(2 + 4 * s);
            __syncthreads();
            BCR_STAMP(3 + 4 * s);
            if (wv < BA_BCR_DWAVES) {
#pragma unroll
                for (int i = s + 2; i <= NT; ++i)
                    if ((i - s - 2) % (BA_BCR_DWAVES - 1) == wv - 1) panel_tile(i, s);
            } else if (strip) {
#pragma unroll
                for (int v = 0; v < 4; ++v) stg[(l4 + 4 * v) * 17 + l15] = acc[s][v];
                band_wave_sync();
                double aq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) aq[q] = lds_f64(&stg[l15 * 17 + 4 * q + l4]);
                mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], lds_f64(&sLinv[l15 * 17 + 4 * q + l4]), w, 0, 0, 0);
                acc[s] = w;
                band_wave_sync();
#pragma unroll
                for (int v = 0; v < 4; ++v) stg[(l4 + 4 * v) * 17 + l15] = w[v];
                band_wave_sync();
#pragma unroll
                for (int q = 0; q < 4; ++q) aW[q] = -lds_f64(&stg[l15 * 17 + 4 * q + l4]);
            }
            BCR_STAMP(4 + 4 * s);
            __syncthreads();
            BCR_STAMP(5 + 4 * s);
            if (s + 1 < NT) {
                if (wv < BA_BCR_DWAVES) {
                    int task = 0;
#pragma unroll
                    for (int i = s + 1; i <= NT; ++i) {
#pragma unroll
                        for (int j = s + 1; j < NT; ++j) {
                            if (j > i || (i == s + 1 && j == s + 1)) continue;   // (wave 0 has done (s+1, s+1))
                            if (wv == 1 + task % (BA_BCR_DWAVES - 1)) trailing_tile(i, j, s);
                            ++task;
                        }
                    }
                } else if (strip) {
#pragma unroll
                    for (int c = s + 1; c < NT; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[q], lds_f64(&sD[(16 * c + l15) * LDD + 16 * s + 4 * q + l4]), acc[c], 0, 0, 0);
                }
            }
        });
    }
    BCR_STAMP(2 + 4 * NT);
    __syncthreads();
    BCR_STAMP(3 + 4 * NT);
    if (strip) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) Gs[(size_t)(sr * 16 + l4 + 4 * v) * NB + c * 16 + l15] = acc[c][v];
    }
    if (side == 0) {
        // the factor goes to Lf, NOT back into D_k: the other workgroup of this block may still have to load D_k
        double *Lk = Lf + (size_t)el.k * NB * NB;
        for (int e = tid; e < NB * NB; e += NTHREADS) {
            const int r = e / NB, c = e - r * NB;
            if (c <= r) Lk[e] = sD[r * LDD + c];
        }
        for (int e = tid; e < NB; e += NTHREADS) bv[(size_t)el.k * NB + e] = sD[NB * LDD + e];
    }
    // (both sides factor the same D_k and meet the same pivots; either one may raise the flag)
    if (wv == 0 && lane == 0 && bad) atomicCAS(info, 0, el.k + 1);
    BCR_STAMP(4 + 4 * NT);
}

// ------------------------------------------------------------------------------------------------ update
// One wave per 16x16 output tile (plus one per 16 rows of the right-hand side), operands straight from L2 / memory.
// (Measured: fatter waves -- a 16-row strip times half the tile columns per wave, operand rows loaded once -- are
// slower, 10 instead of 5.5 us on the upper levels: what an update waits for is the latency of its first loads, the
// blocks were written by other XCDs a kernel ago, and a short wave exposes it once.)
template <int NT>
__global__ __launch_bounds__(64 * BA_BCR_UPD_WAVES) void k_bcr_update(int n_ops, const BcrOp *__restrict__ ops, double *__restrict__ D,
                                                    double *__restrict__ G, double *__restrict__ bv) {
    constexpr int NB = 16 * NT, KQ = NB / 4, TASKS = NT * NT + NT, WGS = (TASKS + BA_BCR_UPD_WAVES - 1) / BA_BCR_UPD_WAVES;
    // workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own: renumbered so that the workgroups of
    // one update (they share its operand blocks) run on ONE XCD
    const int per_xcd = (n_ops * WGS + 7) / 8;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wg >= n_ops * WGS) return;
    const BcrOp op = ops[wg / WGS];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int task = (wg % WGS) * BA_BCR_UPD_WAVES + (tid >> 6);
    if (task >= TASKS) return;
    const size_t nb2 = (size_t)NB * NB;
    if (task < NT * NT) {
        const int ti = task / NT, tj = task - ti * NT;
        if (op.kind == 0 && tj > ti) return;
        double *dst = (op.kind == 0 ? D + op.dst * nb2 : G + op.dst * nb2) + (size_t)(ti * 16 + l4) * NB + tj * 16 + l15;
        double old[4] = {0.0, 0.0, 0.0, 0.0};
        if (!op.init) {
#pragma unroll
            for (int v = 0; v < 4; ++v) old[v] = dst[(size_t)4 * v * NB];
        }
        mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sa = q ? op.a1 : op.a0, sb = op.kind == 0 ? sa : (q ? op.b1 : op.b0);
            if (sa < 0) continue;
            // lane (row, l4) takes doubles 8 j + 2 l4, + 1 of its row: four lanes read 64 contiguous bytes
            const double *pa = G + sa * nb2 + (size_t)(ti * 16 + l15) * NB + l4 * 2;
            const double *pb = G + sb * nb2 + (size_t)(tj * 16 + l15) * NB + l4 * 2;
            double a[KQ], b[KQ];
#pragma unroll
            for (int k = 0; k < KQ; ++k) { a[k] = pa[(k >> 1) * 8 + (k & 1)]; b[k] = pb[(k >> 1) * 8 + (k & 1)]; }
#pragma unroll
            for (int k = 0; k < KQ; ++k) w = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[k], w, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) dst[(size_t)4 * v * NB] = old[v] - w[v];
    } else {
        if (op.kind != 0) return;
        const int ti = task - NT * NT;   // rows 16 ti .. of b[dst]
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sa = q ? op.a1 : op.a0, ky = q ? op.k1 : op.k0;
            if (sa < 0) continue;
            const double *pa = G + sa * nb2 + (size_t)(ti * 16 + l15) * NB + l4 * 2;
            const double *py = bv + (size_t)ky * NB + l4 * 2;
#pragma unroll
            for (int k = 0; k < KQ; ++k) s = fma(pa[(k >> 1) * 8 + (k & 1)], py[(k >> 1) * 8 + (k & 1)], s);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (l4 == 0) bv[(size_t)op.dst * NB + ti * 16 + l15] -= s;
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int NT>
__global__ __launch_bounds__(BA_BCR_BACK_THREADS) void k_bcr_back(const BcrElim *__restrict__ elims, const double *__restrict__ Lf,
                                                  const double *__restrict__ G, double *__restrict__ bv,
                                                  const double *__restrict__ Linv16, const int32_t *__restrict__ pos_f,
                                                  double *__restrict__ xout, const double *__restrict__ done) {
    constexpr int NB = 16 * NT, PARTS = BA_BCR_BACK_THREADS / NB, RP = (NB + PARTS - 1) / PARTS;
    if (done && *done != 0.0) return;   // (uniform: queued behind a converged PCG iteration)
    extern __shared__ double bcr_smem[];
    double *sL = bcr_smem;             // NB x NB: L_k (every global load of the kernel is issued up front: the blocked
    double *sLi = sL + NB * NB;        // NT x 256: inverses of its diagonal tiles     substitution below runs out of LDS)
    double *sred = sLi + NT * 256;     // PARTS x NB
    double *srhs = sred + PARTS * NB;  // NB
    double *sx = srhs + NB;            // 16
    const BcrElim el = elims[blockIdx.x];
    const int tid = threadIdx.x;
    const size_t nb2 = (size_t)NB * NB;
    // (every trip to memory starts here, not behind a barrier)
    const double yk = tid < NB ? bv[(size_t)el.k * NB + tid] : 0.0;
    const int fcam = tid < 6 * (NB / 6) ? pos_f[el.k * (NB / 6) + tid / 6] : -1;   // the camera of unknown tid of this block
    {
        const double *L = Lf + el.k * nb2, *li = Linv16 + (size_t)el.k * NT * 256;
        for (int e = tid; e < NB * NB; e += BA_BCR_BACK_THREADS) sL[e] = L[e];
        for (int e = tid; e < NT * 256; e += BA_BCR_BACK_THREADS) sLi[e] = li[e];
    }
    if (tid < PARTS * NB) {
        const int part = tid / NB, c = tid - part * NB;
        const int r0 = part * RP;
        double s = 0.0;
        // (compile-time trip count: all RP loads of a neighbour are in flight at once -- the blocks were written by other
        // XCDs, so each load is a trip to memory, and a loop with a run-time bound took them a few at a time)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = q ? el.gv : el.gu;
            if (g < 0) continue;
            const double *W = G + g * nb2 + (size_t)r0 * NB + c, *xn = bv + (size_t)(q ? el.v : el.u) * NB + r0;
            double w[RP], xr[RP];
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const bool in = r0 + i < NB;
                w[i] = in ? W[(size_t)i * NB] : 0.0;
                xr[i] = in ? xn[i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < RP; ++i) s = fma(w[i], xr[i], s);
        }
        sred[tid] = s;
    }
    __syncthreads();
    if (tid < NB) {
        double t = yk;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) t -= sred[p * NB + tid];
        srhs[tid] = t;
    }
    __syncthreads();
    for (int s = NT - 1; s >= 0; --s) {
        if (tid < 16) {
            const double *li = sLi + s * 256;   // x_s = Linv_s^T rhs_s
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum = fma(li[r * 16 + tid], srhs[16 * s + r], sum);
            sx[tid] = sum;
        }
        __syncthreads();
        if (tid < 16 * s) {
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum = fma(sL[(16 * s + i) * NB + tid], sx[i], sum);
            srhs[tid] -= sum;
        } else if (tid < 16 * s + 16) {
            srhs[tid] = sx[tid - 16 * s];
        }
        __syncthreads();
    }
    if (tid < NB) {
        bv[(size_t)el.k * NB + tid] = srhs[tid];
        if (fcam >= 0) xout[(size_t)6 * fcam + tid % 6] = srhs[tid];
    }
}

// ------------------------------------------------------------------------------------------------ the factorisation as a preconditioner
// z = M^-1 r for a NEW right-hand side through the stored factors (do_bcr_pcg: CG on the whole S, preconditioned by the
// in-band part M).  A CG iteration is as fast as this sweep, and the sweep is 2 (levels) dependent launches, so each
// launch must be short: with the explicit inverses X_k = L_k^-1 (k_bcr_inv, once per factorisation) every step of the
// sweep is a dense matrix-vector product streamed from L2 -- no triangular substitution, no LDS staging of a factor:
//     forward   y_k = X_k (r_k - sum over the updates the block has received, W[slot] y_j: the plan's pending list)
//     backward  x_k = X_k^T (y_k - W_u^T x_u - W_v^T x_v)
// Every sum has a fixed order (lanes over columns, DPP tree, then the waves' partials in wave order), so M^-1 is a fixed
// linear operator: bit-identical from run to run.  `done` (may be null): the PCG's convergence flag -- a launch queued
// behind a converged iteration returns at once.

// X_k = L_k^-1, full lower-triangular block, for every block of the plan in ONE launch: wave j of a workgroup owns block
// column j (16 columns) and walks down it tile by tile, X_jj = the diagonal tile's inverse (from the factorisation),
// X_ij = -X_ii sum_{k = j .. i-1} L_ik X_kj -- the recurrence of k_big_potrf (ba_dense_big.h), operands L_ik straight from L2,
// the column's finished tiles in LDS; block columns do not depend on each other: no workgroup barrier.
template <int NT>
__global__ __launch_bounds__(64 * NT) void k_bcr_inv(const double *__restrict__ Lf, const double *__restrict__ Linv16, double *__restrict__ Xf) {
    constexpr int NB = 16 * NT;
    __shared__ double sX[NT * (NT + 1) / 2 * 272];   // tile (i, j), i >= j, of column j at (j NT - j (j - 1) / 2 + (i - j)) * 272, 16 x 17
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *L = Lf + (size_t)k * NB * NB, *li = Linv16 + (size_t)k * NT * 256;
    double *X = Xf + (size_t)k * NB * NB;
    double *col = sX + (j * NT - j * (j - 1) / 2) * 272;
    {
        double *t = col;   // X_jj
#pragma unroll
        for (int v = 0; v < 4; ++v) t[(l4 + 4 * v) * 17 + l15] = li[j * 256 + (l4 + 4 * v) * 16 + l15];
    }
    band_wave_sync();
    for (int i = j + 1; i < NT; ++i) {
        mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
        for (int kk = j; kk < i; ++kk) {
            const double *xt = col + (kk - j) * 272;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w = __builtin_amdgcn_mfma_f64_16x16x4f64(L[(size_t)(16 * i + l15) * NB + 16 * kk + 4 * q + l4], xt[(4 * q + l4) * 17 + l15], w, 0, 0, 0);
        }
        mfma_d4 x = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(-li[i * 256 + l15 * 16 + 4 * q + l4], w[q], x, 0, 0, 0);
        double *t = col + (i - j) * 272;
#pragma unroll
        for (int v = 0; v < 4; ++v) t[(l4 + 4 * v) * 17 + l15] = x[v];
        band_wave_sync();
    }
    // the block column to memory, zeros above the diagonal tile (the sweeps read whole rows)
    for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = l4 + 4 * v;
            double val = 0.0;
            if (i >= j) val = col[(i - j) * 272 + r * 17 + l15];
            if (i == j && l15 > r) val = 0.0;
            X[(size_t)(16 * i + r) * NB + 16 * j + l15] = val;
        }
    }
}

#define BA_BCR_SWEEP_THREADS 1024
// One forward level: task t = (block d, up to two updates of the previous level, flags) -- acc_d (+)= W[a0] y[k0] + W[a1] y[k1],
// and for a block eliminated at this level y_d = X_d (r_d - acc_d).  Every workgroup streams at most three 128 x 128
// matrices: the launch is as short as the shortest chain allows, whatever the level.
template <int NT>
__global__ __launch_bounds__(BA_BCR_SWEEP_THREADS) void k_bcr_fwd(const int32_t *__restrict__ tasks, const double *__restrict__ Xf,
                                                 const double *__restrict__ G, double *__restrict__ bv, double *__restrict__ bacc,
                                                 const int32_t *__restrict__ pos_f, const double *__restrict__ rin,
                                                 const double *__restrict__ done) {
    constexpr int NB = 16 * NT, WAVES = BA_BCR_SWEEP_THREADS / 64, RPW = (NB + WAVES - 1) / WAVES, CPL = (NB + 63) / 64;
    __shared__ double sb[NB];
    if (done && *done != 0.0) return;
    const int32_t *tk = tasks + 6 * blockIdx.x;
    const int d = tk[0], flags = tk[5];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t nb2 = (size_t)NB * NB;
    // wave wv takes rows wv, wv + WAVES, ...; lanes across the columns; one DPP reduction per row
    const bool solve = (flags & 2) != 0;
    // every trip to memory of the launch is started here (a launch is a chain of dependent round trips otherwise: the
    // running sums and the right-hand side behind the reductions, the rows of X behind the barrier)
    const double *X = Xf + d * nb2;
    double xw[RPW][CPL], old[RPW], rr[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wv + WAVES * i, rc = r < NB ? r : 0;
        old[i] = (!(flags & 1) && !(flags & 4)) ? bacc[(size_t)d * NB + rc] : 0.0;
        rr[i] = 0.0;
        if (solve) {
            const int fcam = rc < 6 * (NB / 6) ? pos_f[d * (NB / 6) + rc / 6] : -1;
            rr[i] = fcam >= 0 ? rin[(size_t)6 * fcam + rc % 6] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) xw[i][c] = (solve && r < NB && lane + 64 * c < NB) ? X[(size_t)r * NB + lane + 64 * c] : 0.0;
    }
    double acc[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) acc[i] = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int slot = tk[1 + 2 * q], src = tk[2 + 2 * q];
        if (slot < 0) continue;
        const double *W = G + (size_t)slot * nb2, *y = bv + (size_t)src * NB;
        double yv[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) yv[c] = lane + 64 * c < NB ? y[lane + 64 * c] : 0.0;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wv + WAVES * i;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (r < NB && lane + 64 * c < NB) acc[i] = fma(W[(size_t)r * NB + lane + 64 * c], yv[c], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wv + WAVES * i;
        const double s = wave_sum_dpp(acc[i]) + old[i];     // (old: the earlier levels' updates, zero when there are none)
        if (lane == 0 && r < NB) {
            if (!(flags & 4) && !solve) bacc[(size_t)d * NB + r] = s;
            if (solve) sb[r] = rr[i] - ((flags & 4) ? 0.0 : s);
        }
    }
    if (!solve) return;   // (uniform: the task's flags)
    __syncthreads();
    // y = X b (X lower triangular, zeros stored above the diagonal)
    double bc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) bc[c] = lane + 64 * c < NB ? sb[lane + 64 * c] : 0.0;
    double part[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        part[i] = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) part[i] = fma(xw[i][c], bc[c], part[i]);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wv + WAVES * i;
        const double s = wave_sum_dpp(part[i]);
        if (lane == 0 && r < NB) bv[(size_t)d * NB + r] = s;
    }
}

// x_k = X_k^T (y_k - W_u^T x_u - W_v^T x_v): both products are TRANSPOSED ones -- lanes across the columns (coalesced rows),
// the waves split the rows, their partial sums are joined through LDS in wave order
template <int NT>
__global__ __launch_bounds__(BA_BCR_SWEEP_THREADS) void k_bcr_backx(const BcrElim *__restrict__ elims, const double *__restrict__ Xf,
                                                   const double *__restrict__ G, double *__restrict__ bv,
                                                   const int32_t *__restrict__ pos_f, double *__restrict__ xout,
                                                   const double *__restrict__ done) {
    constexpr int NB = 16 * NT, WAVES = BA_BCR_SWEEP_THREADS / 64, RPW = (NB + WAVES - 1) / WAVES, CPL = (NB + 63) / 64;
    __shared__ double sred[WAVES][NB];
    __shared__ double st[NB];
    if (done && *done != 0.0) return;
    const BcrElim el = elims[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t nb2 = (size_t)NB * NB;
    const double yk = tid < NB ? bv[(size_t)el.k * NB + tid] : 0.0;
    const int fcam = tid < 6 * (NB / 6) ? pos_f[el.k * (NB / 6) + tid / 6] : -1;
    const double *X = Xf + el.k * nb2;
    double xw[RPW][CPL];   // (the second product's rows, requested up front)
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int r = wv + WAVES * i;
            xw[i][c] = (r < NB && lane + 64 * c < NB) ? X[(size_t)r * NB + lane + 64 * c] : 0.0;
        }
    double part[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) part[c] = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = q ? el.gv : el.gu;
        if (g < 0) continue;
        const double *W = G + g * nb2, *xn = bv + (size_t)(q ? el.v : el.u) * NB;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wv + WAVES * i;
            const double xr = r < NB ? xn[r] : 0.0;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (r < NB && lane + 64 * c < NB) part[c] = fma(W[(size_t)r * NB + lane + 64 * c], xr, part[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c)
        if (lane + 64 * c < NB) sred[wv][lane + 64 * c] = part[c];
    __syncthreads();
    if (tid < NB) {
        double t = yk;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t -= sred[w][tid];
        st[tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPL; ++c) part[c] = 0.0;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wv + WAVES * i;
        const double tr = r < NB ? st[r] : 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) part[c] = fma(xw[i][c], tr, part[c]);
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c)
        if (lane + 64 * c < NB) sred[wv][lane + 64 * c] = part[c];
    __syncthreads();
    if (tid < NB) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) x += sred[w][tid];
        bv[(size_t)el.k * NB + tid] = x;
        if (fcam >= 0) xout[(size_t)6 * fcam + tid % 6] = x;
    }
}

// ------------------------------------------------------------------------------------------------ launch
struct BcrDev {
    const BcrElim *elims;
    const BcrOp *ops;
    const int32_t *src, *pos_f;
    const int32_t *fwd;               // forward tasks of a new right-hand side (k_bcr_fwd), BcrPlan::fwd
    double *bacc;                     // [m * nb] running sums of the forward sweep
    double *DG;      // bcr_matrix_doubles: m diagonal blocks, the coupling slots, then the m factors L_k
    double *bv;      // bcr_vec_doubles
    double *linv;    // bcr_linv_doubles
    int *info;
};

template <int NT>
inline hipError_t bcr_set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bcr_factor<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bcr_factor_lds_bytes(NT));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_bcr_back<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bcr_back_lds_bytes(NT));
}
inline hipError_t bcr_prepare_device(int nt) {   // once per device before the first launch (LDS beyond 64 KB)
    switch (nt) {
        case 2: return bcr_set_attr<2>();
        case 4: return bcr_set_attr<4>();
        case 6: return bcr_set_attr<6>();
        default: return bcr_set_attr<8>();
    }
}

template <int NT>
inline hipError_t bcr_solve_launch_t(hipStream_t st, const BcrPlan &P, const BcrDev &d, const double *S, const double *g, double *x) {
    constexpr int NB = 16 * NT;
    const size_t nb2 = (size_t)NB * NB;
    double *D = d.DG, *G = d.DG + (size_t)P.m * nb2, *Lf = G + (size_t)P.n_slots * nb2;
    {
        const int n_blocks = P.m + P.n_init_slots;
        const size_t n_el = (size_t)n_blocks * nb2;
        hipLaunchKernelGGL(k_bcr_gather, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, st, n_blocks, P.m, NB, d.src, S, d.pos_f, g,
                           d.DG, d.bv, d.info);
    }
    const size_t lds = bcr_factor_lds_bytes(NT);
    constexpr int WGS = (NT * NT + NT + BA_BCR_UPD_WAVES - 1) / BA_BCR_UPD_WAVES;
    const int nl = P.levels();
    for (int l = 0; l < nl; ++l) {
        const int ne_l = P.lvl_elim[l + 1] - P.lvl_elim[l], no_l = P.lvl_op[l + 1] - P.lvl_op[l];
        hipLaunchKernelGGL(k_bcr_factor<NT>, dim3(2 * ne_l), dim3(64 * (NT + BA_BCR_DWAVES)), lds, st, d.elims + P.lvl_elim[l], D, Lf, G,
                           d.bv, d.linv, d.info);
        if (no_l > 0)
            hipLaunchKernelGGL(k_bcr_update<NT>, dim3(8 * ((no_l * WGS + 7) / 8)), dim3(64 * BA_BCR_UPD_WAVES), 0, st, no_l, d.ops + P.lvl_op[l], D, G, d.bv);
    }
    for (int l = nl - 1; l >= 0; --l) {
        const int ne_l = P.lvl_elim[l + 1] - P.lvl_elim[l];
        hipLaunchKernelGGL(k_bcr_back<NT>, dim3(ne_l), dim3(BA_BCR_BACK_THREADS), bcr_back_lds_bytes(NT), st, d.elims + P.lvl_elim[l], Lf, G, d.bv, d.linv, d.pos_f, x, (const double *)nullptr);
    }
    return hipGetLastError();
}

// X_k = L_k^-1 for every block, after bcr_solve_launch (once per factorisation that is going to be used as a preconditioner)
template <int NT>
inline hipError_t bcr_invert_launch_t(hipStream_t st, const BcrPlan &P, const BcrDev &d) {
    const size_t nb2 = (size_t)16 * NT * 16 * NT;
    double *G = d.DG + (size_t)P.m * nb2, *Lf = G + (size_t)P.n_slots * nb2, *Xf = Lf + (size_t)P.m * nb2;
    hipLaunchKernelGGL(k_bcr_inv<NT>, dim3(P.m), dim3(64 * NT), 0, st, Lf, d.linv, Xf);
    return hipGetLastError();
}
inline hipError_t bcr_invert_launch(hipStream_t st, const BcrPlan &P, const BcrDev &d) {
    switch (P.nt) {
        case 2: return bcr_invert_launch_t<2>(st, P, d);
        case 4: return bcr_invert_launch_t<4>(st, P, d);
        case 6: return bcr_invert_launch_t<6>(st, P, d);
        case 8: return bcr_invert_launch_t<8>(st, P, d);
    }
    return hipErrorInvalidValue;
}

// z = M^-1 r through the factors of the last bcr_solve_launch + bcr_invert_launch (r, z: vectors over the free cameras)
template <int NT>
inline hipError_t bcr_apply_launch_t(hipStream_t st, const BcrPlan &P, const BcrDev &d, const double *r, double *z, const double *done) {
    constexpr int NB = 16 * NT;
    const size_t nb2 = (size_t)NB * NB;
    double *G = d.DG + (size_t)P.m * nb2, *Xf = G + (size_t)(P.n_slots + P.m) * nb2;
    const int nl = P.levels();
    for (int l = 0; l < nl; ++l) {
        const int nt_l = P.lvl_fwd[l + 1] - P.lvl_fwd[l];
        hipLaunchKernelGGL(k_bcr_fwd<NT>, dim3(nt_l), dim3(BA_BCR_SWEEP_THREADS), 0, st, d.fwd + 6 * (size_t)P.lvl_fwd[l], Xf, G, d.bv, d.bacc,
                           d.pos_f, r, done);
    }
    for (int l = nl - 1; l >= 0; --l) {
        const int ne_l = P.lvl_elim[l + 1] - P.lvl_elim[l];
        hipLaunchKernelGGL(k_bcr_backx<NT>, dim3(ne_l), dim3(BA_BCR_SWEEP_THREADS), 0, st, d.elims + P.lvl_elim[l], Xf, G, d.bv, d.pos_f, z, done);
    }
    return hipGetLastError();
}
inline hipError_t bcr_apply_launch(hipStream_t st, const BcrPlan &P, const BcrDev &d, const double *r, double *z, const double *done) {
    switch (P.nt) {
        case 2: return bcr_apply_launch_t<2>(st, P, d, r, z, done);
        case 4: return bcr_apply_launch_t<4>(st, P, d, r, z, done);
        case 6: return bcr_apply_launch_t<6>(st, P, d, r, z, done);
        case 8: return bcr_apply_launch_t<8>(st, P, d, r, z, done);
    }
    return hipErrorInvalidValue;
}

// x = -S^-1 g; *d.info (device) = 0 on success, else 1 + the block whose factorisation met a non-positive pivot
inline hipError_t bcr_solve_launch(hipStream_t st, const BcrPlan &P, const BcrDev &d, const double *S, const double *g, double *x) {
    switch (P.nt) {
        case 2: return bcr_solve_launch_t<2>(st, P, d, S, g, x);
        case 4: return bcr_solve_launch_t<4>(st, P, d, S, g, x);
        case 6: return bcr_solve_launch_t<6>(st, P, d, S, g, x);
        case 8: return bcr_solve_launch_t<8>(st, P, d, S, g, x);
    }
    return hipErrorInvalidValue;
}

}  // namespace ba
