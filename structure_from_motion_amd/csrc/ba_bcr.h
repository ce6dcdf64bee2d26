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
//   k_bcr_factor  two workgroups per eliminated block (one per neighbour, both factor D_k): D_k lives in LDS and is
//                 factored right-looking in 16-column steps -- ONE wave factors the 16x16 diagonal tile in registers (a
//                 row per lane, readlane broadcasts) and inverts it, three more waves do panel and trailing tiles on the
//                 fp64 matrix cores (v_mfma_f64_16x16x4_f64); the neighbour's coupling block rides along as NT
//                 16-row strips, one wave each, whose tiles never leave the accumulators (a strip visits LDS only to
//                 turn an accumulator tile into an MFMA operand); the right-hand side is one more row of D_k.
//   k_bcr_update  one wave per 16x16 output tile, operands straight from L2: lane (row, q) loads a CONTIGUOUS quarter
//                 of its row (the k-index of an MFMA lane is free as long as both operands agree), so a wave reads whole
//                 cache lines and needs no LDS.
//   k_bcr_back    one workgroup per block of the level.
// The schedule (which block, which neighbours, which coupling slot, per level) is planned on the host once per pattern
// (bcr_make_plan) and executed from tables.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <type_traits>
#include <utility>
#include <vector>

#include "ba_banded.h"  // band_readlane, band_rsqrt, band_wave_sync
#include "ba_dense.h"   // mfma_d4

namespace ba {

struct BcrElim { int32_t k, gu, gv, u, v; };                       // block, coupling slots (-1: none), neighbour blocks
struct BcrOp { int32_t kind, dst, a0, b0, a1, b1, k0, k1, init; };  // kind 0: D[dst] -= W[a0] W[a0]^T (+ a1), b[dst] -= W[a0] y[k0] (+ ..)
                                                                    // kind 1: G[dst] (init: = 0) -= W[a0] W[b0]^T (+ W[a1] W[b1]^T)
struct BcrPlan {
    bool ok = false, cyclic = false;
    int nf = 0, m = 0, nt = 0, nb = 0, band = 0, n_slots = 0, n_init_slots = 0;
    std::vector<int32_t> blk_start;  // [m + 1] first free camera of each block
    std::vector<int32_t> cam_blk;    // [nf] block << 8 | offset inside the block
    std::vector<int32_t> edge_slot;  // [m] slot of the coupling between block b and b + 1 (entry m - 1: the ring's corner
                                     // block, between 0 and m - 1) | bit 30: rows belong to the higher block; -1: none
    std::vector<int32_t> lvl_elim, lvl_op;  // [levels + 1] offsets into elims / ops
    std::vector<BcrElim> elims;
    std::vector<BcrOp> ops;
    int levels() const { return (int)lvl_elim.size() - 1; }
};

// Plans the elimination for the pattern `keys` (sorted unique (i << 32 | j), i <= j, free-camera indices).  false: the
// pattern is not a chain / ring with a half bandwidth of at most 21 cameras (or too small to bother).
inline bool bcr_make_plan(int nf, const int64_t *keys, size_t nk, BcrPlan &P) {
    P = BcrPlan{};
    if (nf < 2) return false;
    int near_b = 0, wrap_b = 0;
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff), d = j - i;
        if (d <= 0) continue;
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
    P.cam_blk.resize(nf);
    for (int b = 0; b < m; ++b)
        for (int c = start[b]; c < start[b + 1]; ++c) P.cam_blk[c] = (b << 8) | (c - start[b]);
    for (size_t q = 0; q < nk; ++q) {
        const int i = (int)(keys[q] >> 32), j = (int)(keys[q] & 0xffffffff);
        const int bi = P.cam_blk[i] >> 8, bj = P.cam_blk[j] >> 8;
        if (!(bj == bi || bj == bi + 1 || (cyc && bi == 0 && bj == m - 1))) return false;
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
    P.ok = true;
    return true;
}

inline size_t bcr_matrix_doubles(const BcrPlan &P) { return (size_t)(P.m + P.n_slots) * P.nb * P.nb; }
inline size_t bcr_vec_doubles(const BcrPlan &P) { return (size_t)P.m * P.nb; }
inline size_t bcr_linv_doubles(const BcrPlan &P) { return (size_t)P.m * P.nt * 256; }
inline size_t bcr_factor_lds_bytes(int nt) {
    const int nb = 16 * nt;
    return ((size_t)(nb + 1) * (nb + 1) + 16 * 17 + (size_t)nt * 16 * 17) * sizeof(double);
}

// ------------------------------------------------------------------------------------------------ assembly
// S (upper 6x6 blocks, keys (i << 32 | j), i <= j) -> diagonal blocks D (lower triangle) and the initial coupling slots
__global__ void k_bcr_assemble(int n_upper, const long long *__restrict__ keys, const double *__restrict__ S,
                               const int32_t *__restrict__ cam_blk, const int32_t *__restrict__ edge_slot, int m, int NB,
                               double *__restrict__ D, double *__restrict__ G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_upper * 36) return;
    const int k = e / 36, ab = e - k * 36, a = ab / 6, b = ab - a * 6;
    const int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffffLL);
    const int ci = cam_blk[i], cj = cam_blk[j];
    const int bi = ci >> 8, oi = ci & 255, bj = cj >> 8, oj = cj & 255;
    const double v = S[e];
    const size_t nb2 = (size_t)NB * NB;
    if (bi == bj) {
        if (i == j) {
            if (a >= b) D[bi * nb2 + (size_t)(oi * 6 + a) * NB + oi * 6 + b] = 0.5 * (v + S[(size_t)k * 36 + b * 6 + a]);
        } else {
            D[bi * nb2 + (size_t)(oj * 6 + b) * NB + oi * 6 + a] = v;   // lower block (j, i) = upper block transposed
        }
    } else {
        const int es = edge_slot[bj == bi + 1 ? bi : m - 1];   // camera i sits in the lower-numbered block of the edge
        double *g = G + (size_t)(es & 0x3fffffff) * nb2;
        if (es >> 30 & 1) g[(size_t)(oj * 6 + b) * NB + oi * 6 + a] = v;
        else g[(size_t)(oi * 6 + a) * NB + oj * 6 + b] = v;
    }
}

// right-hand side b = -g per block, unit diagonal and zero right-hand side in the padding of each block
__global__ void k_bcr_init(int m, int NB, const int32_t *__restrict__ blk_start, const double *__restrict__ g,
                           double *__restrict__ D, double *__restrict__ bv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * NB) return;
    const int b = t / NB, o = t - b * NB, sz = 6 * (blk_start[b + 1] - blk_start[b]);
    if (o < sz) {
        bv[t] = -g[(size_t)6 * blk_start[b] + o];
    } else {
        bv[t] = 0.0;
        D[(size_t)b * NB * NB + (size_t)o * NB + o] = 1.0;
    }
}

// ------------------------------------------------------------------------------------------------ factor
#define BA_BCR_DWAVES 4
// compile-time loop (the step index selects accumulator registers: the loop has to be unrolled whatever its size)
template <int I, int N, class F>
__device__ __forceinline__ void bcr_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        bcr_static_for<I + 1, N>(f);
    }
}
template <int NT>
__global__ __launch_bounds__(64 * (NT + BA_BCR_DWAVES)) void k_bcr_factor(const BcrElim *__restrict__ elims, double *__restrict__ D,
                                                                         double *__restrict__ G, double *__restrict__ bv,
                                                                         double *__restrict__ Linv16, int *__restrict__ info) {
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
    double *Dk = D + (size_t)el.k * NB * NB;
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
    __syncthreads();
    int bad = 0;
    double aW[4] = {0.0, 0.0, 0.0, 0.0};
    bcr_static_for<0, NT>([&](auto step_c) {
        constexpr int s = decltype(step_c)::value;
        // ---- A: wave 0 factors the diagonal tile (s, s) and inverts its factor (lanes 16.. mirror lanes 0..15)
        if (wv == 0) {
            double row[16], invd[16];
            double *dt = &sD[(16 * s + l15) * LDD + 16 * s];
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = dt[c];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double d = band_readlane(row[j], j);
                if (!(d > 0.0)) bad = 1;
                const double inv = band_rsqrt(d > 0.0 ? d : 1.0);
                invd[j] = inv;
                row[j] *= inv;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) row[c] = fma(-row[j], band_readlane(row[j], c), row[c]);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c <= l15) dt[c] = row[c];
            }
            // X = L^-1, lane = column: x_r = (delta_rc - sum_{k<r} L[r][k] x_k) / L[r][r]
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double sum = (r == l15) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < r; ++k) sum = fma(-band_readlane(row[k], r), x[k], sum);
                x[r] = sum * invd[r];
            }
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sLinv[r * 17 + l15] = x[r];
                if (side == 0) {
                    double *li = Linv16 + ((size_t)el.k * NT + s) * 256;
#pragma unroll
                    for (int r = 0; r < 16; ++r) li[r * 16 + l15] = x[r];
                }
            }
        }
        __syncthreads();
        // ---- B: panel tiles (i, s) <- (i, s) Linv^T; the strips turn their tile s into W_s
        if (wv < BA_BCR_DWAVES) {
            double bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = sLinv[l15 * 17 + 4 * q + l4];
#pragma unroll
            for (int i = s + 1; i <= NT; ++i) {
                if ((i - s - 1) % BA_BCR_DWAVES != wv) continue;
                const int arow = i < NT ? 16 * i + l15 : NB;   // (tile NT: sixteen copies of the right-hand-side row)
                mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
                double aq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) aq[q] = sD[arow * LDD + 16 * s + 4 * q + l4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], bq[q], w, 0, 0, 0);
                if (i < NT) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) sD[(16 * i + l4 + 4 * v) * LDD + 16 * s + l15] = w[v];
                } else if (l4 == 0) {
                    sD[NB * LDD + 16 * s + l15] = w[0];
                }
            }
        } else if (strip) {
#pragma unroll
            for (int v = 0; v < 4; ++v) stg[(l4 + 4 * v) * 17 + l15] = acc[s][v];
            band_wave_sync();
            double aq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) aq[q] = stg[l15 * 17 + 4 * q + l4];
            mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) w = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], sLinv[l15 * 17 + 4 * q + l4], w, 0, 0, 0);
            acc[s] = w;
            band_wave_sync();
#pragma unroll
            for (int v = 0; v < 4; ++v) stg[(l4 + 4 * v) * 17 + l15] = w[v];
            band_wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) aW[q] = -stg[l15 * 17 + 4 * q + l4];
        }
        __syncthreads();
        // ---- C: trailing tiles (i, j) -= (i, s) (j, s)^T.  Wave 0 takes (s+1, s+1) only and goes on to factor it while
        //         the other waves finish (no barrier between C and the next A)
        if (s + 1 < NT) {
            if (wv < BA_BCR_DWAVES) {
                int task = 0;
#pragma unroll
                for (int i = s + 1; i <= NT; ++i) {
#pragma unroll
                    for (int j = s + 1; j < NT; ++j) {
                        if (j > i) continue;
                        bool mine;
                        if (i == s + 1 && j == s + 1) mine = wv == 0;
                        else { mine = wv == 1 + task % (BA_BCR_DWAVES - 1); ++task; }
                        if (!mine) continue;
                        const int arow = i < NT ? 16 * i + l15 : NB;
                        double aq[4], bq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            aq[q] = -sD[arow * LDD + 16 * s + 4 * q + l4];
                            bq[q] = sD[(16 * j + l15) * LDD + 16 * s + 4 * q + l4];
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
                    }
                }
            } else if (strip) {
#pragma unroll
                for (int c = s + 1; c < NT; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[q], sD[(16 * c + l15) * LDD + 16 * s + 4 * q + l4], acc[c], 0, 0, 0);
            }
        }
    });
    __syncthreads();
    if (strip) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) Gs[(size_t)(sr * 16 + l4 + 4 * v) * NB + c * 16 + l15] = acc[c][v];
    }
    if (side == 0) {
        for (int e = tid; e < NB * NB; e += NTHREADS) {
            const int r = e / NB, c = e - r * NB;
            if (c <= r) Dk[e] = sD[r * LDD + c];
        }
        for (int e = tid; e < NB; e += NTHREADS) bv[(size_t)el.k * NB + e] = sD[NB * LDD + e];
        if (wv == 0 && lane == 0 && bad) atomicCAS(info, 0, el.k + 1);
    }
}

// ------------------------------------------------------------------------------------------------ update
template <int NT>
__global__ __launch_bounds__(256) void k_bcr_update(const BcrOp *__restrict__ ops, double *__restrict__ D, double *__restrict__ G,
                                                    double *__restrict__ bv) {
    constexpr int NB = 16 * NT, KQ = NB / 4, TASKS = NT * NT + NT, WGS = (TASKS + 3) / 4;
    const BcrOp op = ops[blockIdx.x / WGS];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int task = (blockIdx.x % WGS) * 4 + (tid >> 6);
    if (task >= TASKS) return;
    const size_t nb2 = (size_t)NB * NB;
    if (task < NT * NT) {
        const int ti = task / NT, tj = task - ti * NT;
        if (op.kind == 0 && tj > ti) return;
        mfma_d4 w = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sa = q ? op.a1 : op.a0, sb = op.kind == 0 ? sa : (q ? op.b1 : op.b0);
            if (sa < 0) continue;
            const double *pa = G + sa * nb2 + (size_t)(ti * 16 + l15) * NB + l4 * KQ;
            const double *pb = G + sb * nb2 + (size_t)(tj * 16 + l15) * NB + l4 * KQ;
            double a[KQ], b[KQ];
#pragma unroll
            for (int k = 0; k < KQ; ++k) { a[k] = pa[k]; b[k] = pb[k]; }
#pragma unroll
            for (int k = 0; k < KQ; ++k) w = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[k], w, 0, 0, 0);
        }
        double *dst = (op.kind == 0 ? D + op.dst * nb2 : G + op.dst * nb2) + (size_t)(ti * 16 + l4) * NB + tj * 16 + l15;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double old = op.init ? 0.0 : dst[(size_t)4 * v * NB];
            dst[(size_t)4 * v * NB] = old - w[v];
        }
    } else {
        if (op.kind != 0) return;
        const int ti = task - NT * NT;   // rows 16 ti .. of b[dst]
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sa = q ? op.a1 : op.a0, ky = q ? op.k1 : op.k0;
            if (sa < 0) continue;
            const double *pa = G + sa * nb2 + (size_t)(ti * 16 + l15) * NB + l4 * KQ;
            const double *py = bv + (size_t)ky * NB + l4 * KQ;
#pragma unroll
            for (int k = 0; k < KQ; ++k) s = fma(pa[k], py[k], s);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (l4 == 0) bv[(size_t)op.dst * NB + ti * 16 + l15] -= s;
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int NT>
__global__ __launch_bounds__(512) void k_bcr_back(const BcrElim *__restrict__ elims, const double *__restrict__ D,
                                                  const double *__restrict__ G, double *__restrict__ bv,
                                                  const double *__restrict__ Linv16, const int32_t *__restrict__ blk_start,
                                                  double *__restrict__ xout) {
    constexpr int NB = 16 * NT, PARTS = 512 / NB, RP = (NB + PARTS - 1) / PARTS;
    __shared__ double sred[PARTS * NB];
    __shared__ double srhs[NB];
    __shared__ double sx[16];
    const BcrElim el = elims[blockIdx.x];
    const int tid = threadIdx.x;
    const size_t nb2 = (size_t)NB * NB;
    if (tid < PARTS * NB) {
        const int part = tid / NB, c = tid - part * NB;
        const int r0 = part * RP, r1 = r0 + RP < NB ? r0 + RP : NB;
        double s = 0.0;
        if (el.gu >= 0) {
            const double *W = G + el.gu * nb2, *xu = bv + (size_t)el.u * NB;
            for (int r = r0; r < r1; ++r) s = fma(W[(size_t)r * NB + c], xu[r], s);
        }
        if (el.gv >= 0) {
            const double *W = G + el.gv * nb2, *xv = bv + (size_t)el.v * NB;
            for (int r = r0; r < r1; ++r) s = fma(W[(size_t)r * NB + c], xv[r], s);
        }
        sred[tid] = s;
    }
    __syncthreads();
    if (tid < NB) {
        double t = bv[(size_t)el.k * NB + tid];
#pragma unroll
        for (int p = 0; p < PARTS; ++p) t -= sred[p * NB + tid];
        srhs[tid] = t;
    }
    __syncthreads();
    const double *L = D + el.k * nb2;
    for (int s = NT - 1; s >= 0; --s) {
        if (tid < 16) {
            const double *li = Linv16 + ((size_t)el.k * NT + s) * 256;   // x_s = Linv_s^T rhs_s
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum = fma(li[r * 16 + tid], srhs[16 * s + r], sum);
            sx[tid] = sum;
        }
        __syncthreads();
        if (tid < 16 * s) {
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum = fma(L[(size_t)(16 * s + i) * NB + tid], sx[i], sum);
            srhs[tid] -= sum;
        } else if (tid < 16 * s + 16) {
            srhs[tid] = sx[tid - 16 * s];
        }
        __syncthreads();
    }
    if (tid < NB) {
        bv[(size_t)el.k * NB + tid] = srhs[tid];
        const int sz = 6 * (blk_start[el.k + 1] - blk_start[el.k]);
        if (tid < sz) xout[(size_t)6 * blk_start[el.k] + tid] = srhs[tid];
    }
}

// ------------------------------------------------------------------------------------------------ launch
struct BcrDev {
    const BcrElim *elims;
    const BcrOp *ops;
    const int32_t *cam_blk, *edge_slot, *blk_start;
    double *DG;      // bcr_matrix_doubles: m diagonal blocks, then the coupling slots
    double *bv;      // bcr_vec_doubles
    double *linv;    // bcr_linv_doubles
    int *info;
};

template <int NT>
inline hipError_t bcr_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_bcr_factor<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bcr_factor_lds_bytes(NT));
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
inline hipError_t bcr_solve_launch_t(hipStream_t st, const BcrPlan &P, const BcrDev &d, int n_upper, const long long *keys,
                                     const double *S, const double *g, double *x) {
    constexpr int NB = 16 * NT;
    const size_t nb2 = (size_t)NB * NB;
    double *D = d.DG, *G = d.DG + (size_t)P.m * nb2;
    hipError_t e;
    if ((e = hipMemsetAsync(d.DG, 0, (size_t)(P.m + P.n_init_slots) * nb2 * sizeof(double), st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(d.info, 0, sizeof(int), st)) != hipSuccess) return e;
    const int ne = n_upper * 36;
    hipLaunchKernelGGL(k_bcr_assemble, dim3((ne + 255) / 256), dim3(256), 0, st, n_upper, keys, S, d.cam_blk, d.edge_slot, P.m, NB, D, G);
    hipLaunchKernelGGL(k_bcr_init, dim3((P.m * NB + 255) / 256), dim3(256), 0, st, P.m, NB, d.blk_start, g, D, d.bv);
    const size_t lds = bcr_factor_lds_bytes(NT);
    constexpr int WGS = (NT * NT + NT + 3) / 4;
    const int nl = P.levels();
    for (int l = 0; l < nl; ++l) {
        const int ne_l = P.lvl_elim[l + 1] - P.lvl_elim[l], no_l = P.lvl_op[l + 1] - P.lvl_op[l];
        hipLaunchKernelGGL(k_bcr_factor<NT>, dim3(2 * ne_l), dim3(64 * (NT + BA_BCR_DWAVES)), lds, st, d.elims + P.lvl_elim[l], D, G,
                           d.bv, d.linv, d.info);
        if (no_l > 0) hipLaunchKernelGGL(k_bcr_update<NT>, dim3(no_l * WGS), dim3(256), 0, st, d.ops + P.lvl_op[l], D, G, d.bv);
    }
    for (int l = nl - 1; l >= 0; --l) {
        const int ne_l = P.lvl_elim[l + 1] - P.lvl_elim[l];
        hipLaunchKernelGGL(k_bcr_back<NT>, dim3(ne_l), dim3(512), 0, st, d.elims + P.lvl_elim[l], D, G, d.bv, d.linv, d.blk_start, x);
    }
    return hipGetLastError();
}

// x = -S^-1 g; *d.info (device) = 0 on success, else 1 + the block whose factorisation met a non-positive pivot
inline hipError_t bcr_solve_launch(hipStream_t st, const BcrPlan &P, const BcrDev &d, int n_upper, const long long *keys,
                                   const double *S, const double *g, double *x) {
    switch (P.nt) {
        case 2: return bcr_solve_launch_t<2>(st, P, d, n_upper, keys, S, g, x);
        case 4: return bcr_solve_launch_t<4>(st, P, d, n_upper, keys, S, g, x);
        case 6: return bcr_solve_launch_t<6>(st, P, d, n_upper, keys, S, g, x);
        case 8: return bcr_solve_launch_t<8>(st, P, d, n_upper, keys, S, g, x);
    }
    return hipErrorInvalidValue;
}

}  // namespace ba
