// Host-side symbolic analysis (see ba_symbolic.h).  Plain C++: also compiled by g++ for the
// CPU-only structure tests (ba_analyze needs no GPU).
#include "ba_symbolic.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "../../include/ba_hip.h"

namespace ba {

namespace {
// fn(lo, hi) over [0, n) on up to 8 host threads (large n only).  Used where the result does not depend on the split.
template <typename F>
void parallel_for(int64_t n, F fn) {
    const int T = (int)std::min<int64_t>(8, n / 100000);   // (threads cost tens of microseconds to start, and the counting pass
                                                           // uses atomics: 100 k items each at least)
    if (T < 2) {
        fn((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T); });
    for (auto &x : th) x.join();
}
}  // namespace

int analyze_structure(int n_cams, int n_points, int64_t n_obs, const uint8_t *cam_fixed,
                      const int32_t *obs_cam, const int32_t *obs_pt, Symbolic &s, const char **err, bool deterministic, bool rows) {
    const bool tm_on = getenv("BA_TIME") != nullptr;
    auto tm_t0 = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) { if (tm_on) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[symbolic] %-22s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tm_t0).count()); tm_t0 = t; } };
    s = Symbolic();
    s.deterministic = deterministic;
    s.rows = rows;
    // longest track on the chunk path (longer: heavy point).  The atomic-free build has no path for a camera without a
    // local slot, so there a point must fit the slots of one chunk.
    const int max_chunk_track = (deterministic && !rows) ? CHUNK_CAMS_WIDE : CHUNK_OBS;
    s.n_cams = n_cams;
    s.n_points = n_points;
    s.n_obs = n_obs;
    s.cam_free.assign(n_cams, -1);
    for (int c = 0; c < n_cams; ++c)
        if (!cam_fixed[c]) {
            s.cam_free[c] = s.n_free++;
            s.free_cam.push_back(c);
        }
    // --- observations per point, validity -------------------------------------------------
    // (one pass on host threads: count, smallest and largest camera per point -- integer atomics, so the result does not
    // depend on the interleaving)
    std::vector<int32_t> cnt(n_points + 1, 0);
    std::vector<int32_t> cmin(n_points, INT32_MAX), cmax(n_points, -1);
    {
        int bad = 0;
        if (n_obs < 200000) {   // one thread (parallel_for's own rule): plain updates, a third of the atomic pass's time
            for (int64_t k = 0; k < n_obs; ++k) {
                const int32_t c = obs_cam[k], p = obs_pt[k];
                if (c < 0 || c >= n_cams || p < 0 || p >= n_points) { bad = 1; continue; }
                ++cnt[p];
                if (c < cmin[p]) cmin[p] = c;
                if (c > cmax[p]) cmax[p] = c;
            }
        } else
        parallel_for(n_obs, [&](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; ++k) {
                const int32_t c = obs_cam[k], p = obs_pt[k];
                if (c < 0 || c >= n_cams || p < 0 || p >= n_points) {
                    __atomic_store_n(&bad, 1, __ATOMIC_RELAXED);
                    continue;
                }
                __atomic_fetch_add(&cnt[p], 1, __ATOMIC_RELAXED);
                int32_t cur = __atomic_load_n(&cmin[p], __ATOMIC_RELAXED);
                while (c < cur && !__atomic_compare_exchange_n(&cmin[p], &cur, c, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                cur = __atomic_load_n(&cmax[p], __ATOMIC_RELAXED);
                while (c > cur && !__atomic_compare_exchange_n(&cmax[p], &cur, c, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
        });
        if (bad) {
            *err = "observation index out of range";
            return BA_ERR_INVALID;
        }
    }
    tick("count");
    // --- point order: observed points sorted by (min camera, max camera), unobserved last ----
    // (two stable counting passes, by max camera then by min camera: O(Np + Nc), keeps index order among ties)
    s.pt_order.resize(n_points);
    {
        std::vector<int32_t> tmp(n_points), head(n_cams + 2);
        int n_obs_pts = 0, n_heavy = 0;  // (n_obs_pts: the points that go into chunks)
        for (int i = 0; i < n_points; ++i)
            if (cnt[i] > 0 && cnt[i] <= max_chunk_track) tmp[n_obs_pts++] = i;
        int n_un = n_obs_pts;
        for (int i = 0; i < n_points; ++i)
            if (cnt[i] > max_chunk_track) s.pt_order[n_un++] = i, ++n_heavy;  // heavy points: own kernels, after the chunked ones
        for (int i = 0; i < n_points; ++i)
            if (cnt[i] == 0) s.pt_order[n_un++] = i;  // unobserved points last, in index order
        s.n_pts_chunked = n_obs_pts;
        s.n_heavy = n_heavy;
        auto pass = [&](const std::vector<int32_t> &key, const int32_t *src, int32_t *dst) {
            std::fill(head.begin(), head.end(), 0);
            for (int i = 0; i < n_obs_pts; ++i) head[key[src[i]] + 1]++;
            for (int c = 0; c <= n_cams; ++c) head[c + 1] += head[c];
            for (int i = 0; i < n_obs_pts; ++i) dst[head[key[src[i]]]++] = src[i];
        };
        std::vector<int32_t> tmp2(n_obs_pts);
        pass(cmax, tmp.data(), tmp2.data());
        pass(cmin, tmp2.data(), s.pt_order.data());
    }
    tick("point sort");
    std::vector<int32_t> pt_inv(n_points);
    s.n_pts_obs = 0;
    for (int i = 0; i < n_points; ++i) {
        pt_inv[s.pt_order[i]] = i;
        if (cnt[s.pt_order[i]] > 0) s.n_pts_obs = i + 1;
    }
    // --- observations sorted by (internal point, camera) --------------------------------
    std::vector<int64_t> pstart(n_points + 1, 0);
    for (int i = 0; i < n_points; ++i) pstart[i + 1] = pstart[i] + cnt[s.pt_order[i]];
    s.obs_order.resize(n_obs);
    {
        // (scatter by point on host threads -- the order inside a point's run is settled by the sort that follows, whose
        // key (camera, caller's index) is total)
        std::vector<int64_t> fill(pstart.begin(), pstart.end() - 1);
        if (n_obs < 200000) {
            for (int64_t k = 0; k < n_obs; ++k) s.obs_order[fill[pt_inv[obs_pt[k]]]++] = k;
        } else
        parallel_for(n_obs, [&](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; ++k) s.obs_order[__atomic_fetch_add(&fill[pt_inv[obs_pt[k]]], (int64_t)1, __ATOMIC_RELAXED)] = k;
        });
        parallel_for(s.n_pts_obs, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i)
                std::sort(s.obs_order.begin() + pstart[i], s.obs_order.begin() + pstart[i + 1],
                          [&](int64_t a, int64_t b) { return obs_cam[a] != obs_cam[b] ? obs_cam[a] < obs_cam[b] : a < b; });
        });
    }
    tick("obs order");
    s.o_cam.resize(n_obs);
    s.o_pt.resize(n_obs);
    s.o_lcam.resize(n_obs);
    parallel_for(n_obs, [&](int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            s.o_cam[k] = obs_cam[s.obs_order[k]];
            s.o_pt[k] = pt_inv[obs_pt[s.obs_order[k]]];
        }
    });
    double track_sum = 0;
    for (int i = 0; i < s.n_pts_obs; ++i) {
        int L = (int)(pstart[i + 1] - pstart[i]);
        s.max_track = std::max(s.max_track, L);
        track_sum += L;
    }
    tick("obs gather");
    // Global per-block entry lists (Symbolic::rw_*) of the points [pt_lo, pt_hi): the rows mode builds them for every point,
    // the chunk mode for its heavy points (whose pairs no chunk can hold).
    auto build_row_lists = [&](const int pt_lo, const int pt_hi) {
        // entries by block row: observations inside a point are sorted by camera, and free indices grow with the camera
        // index, so (a <= c) gives fa <= fc: the entry belongs to the upper block (fa, fc) as T_a T_c^T
        const int nf = s.n_free;
        std::vector<int64_t> rstart((size_t)nf + 1, 0);
        for (int i = pt_lo; i < pt_hi; ++i) {
            const int64_t b = pstart[i], e = pstart[i + 1];
            for (int64_t a = b; a < e; ++a) {
                const int fa = s.cam_free[s.o_cam[a]];
                if (fa < 0) continue;
                for (int64_t c = a; c < e; ++c) {
                    const int fc = s.cam_free[s.o_cam[c]];
                    if (fc < 0) continue;
                    rstart[fa + 1] += (c != a && fc == fa) ? 2 : 1;
                }
            }
        }
        for (int f = 0; f < nf; ++f) rstart[f + 1] += rstart[f];
        const int64_t n_ent = rstart[nf];
        struct Ent { int32_t fc, oi, oj; };
        std::vector<Ent> ents((size_t)n_ent);
        {
            std::vector<int64_t> fill(rstart.begin(), rstart.end() - 1);
            for (int i = pt_lo; i < pt_hi; ++i) {
                const int64_t b = pstart[i], e = pstart[i + 1];
                for (int64_t a = b; a < e; ++a) {
                    const int fa = s.cam_free[s.o_cam[a]];
                    if (fa < 0) continue;
                    for (int64_t c = a; c < e; ++c) {
                        const int fc = s.cam_free[s.o_cam[c]];
                        if (fc < 0) continue;
                        ents[fill[fa]++] = Ent{fc, (int32_t)a, (int32_t)c};
                        if (c != a && fc == fa) ents[fill[fa]++] = Ent{fc, (int32_t)c, (int32_t)a};  // one camera, the point twice
                    }
                }
            }
        }
        tick("row entries");
        // each row sorted by column (stable: point order inside a block), rows dealt to host threads by entry count
        {
            const int T = (int)std::max<int64_t>(1, std::min<int64_t>(8, n_ent / 200000));
            std::vector<std::thread> th;
            auto work = [&](int t) {
                const int64_t lo = n_ent * t / T, hi = n_ent * (t + 1) / T;
                int f0 = (int)(std::lower_bound(rstart.begin(), rstart.end(), lo) - rstart.begin());
                int f1 = (int)(std::lower_bound(rstart.begin(), rstart.end(), hi) - rstart.begin());
                if (t == T - 1) f1 = nf;
                for (int f = f0; f < f1 && f < nf; ++f)
                    std::stable_sort(ents.begin() + rstart[f], ents.begin() + rstart[f + 1], [](const Ent &x, const Ent &y) { return x.fc < y.fc; });
            };
            if (T == 1) work(0);
            else {
                for (int t = 0; t < T; ++t) th.emplace_back(work, t);
                for (auto &x : th) x.join();
            }
        }
        tick("row sort");
        s.rw_ent.resize((size_t)n_ent * 2);
        s.rw_ptr.clear();
        for (int f = 0; f < nf; ++f)
            for (int64_t k = rstart[f]; k < rstart[f + 1]; ++k) {
                if (k == rstart[f] || ents[k].fc != ents[k - 1].fc) {
                    s.rw_keys.push_back(((int64_t)f << 32) | (uint32_t)ents[k].fc);
                    s.rw_ptr.push_back(k);
                }
                s.rw_ent[2 * k] = ents[k].oi;
                s.rw_ent[2 * k + 1] = ents[k].oj;
            }
        s.rw_ptr.push_back(n_ent);
        if (s.rows) s.n_pair_real += n_ent;
        tick("row blocks");
    };
    if (s.rows) {
        // --- rows mode: chunks of whole points by observation / point count only; the Schur entries are global ------
        s.ck_obs.push_back(0);
        s.ck_pt.push_back(0);
        int cur_obs = 0, cur_pts = 0;
        for (int i = 0; i < s.n_pts_chunked; ++i) {
            const int L = (int)(pstart[i + 1] - pstart[i]);
            if (cur_pts > 0 && (cur_obs + L > CHUNK_OBS || cur_pts + 1 > CHUNK_PTS)) {
                s.ck_obs.push_back((int32_t)pstart[i]);
                s.ck_pt.push_back(i);
                cur_obs = cur_pts = 0;
            }
            cur_obs += L;
            cur_pts += 1;
        }
        if (cur_pts > 0) {
            s.ck_obs.push_back((int32_t)pstart[s.n_pts_chunked]);
            s.ck_pt.push_back(s.n_pts_chunked);
        }
        const size_t nck = s.ck_obs.size() - 1;
        s.ck_cam.assign(nck + 1, 0);
        s.ck_pair.assign(nck + 1, 0);
        s.ck_pe.assign(nck + 1, 0);
        s.ck_pgrp.assign(nck, 0);
        parallel_for(pstart[s.n_pts_chunked], [&](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; ++k) s.o_lcam[k] = s.cam_free[s.o_cam[k]] < 0 ? -2 : -1;
        });
        tick("chunking (rows)");
        build_row_lists(0, s.n_pts_chunked + s.n_heavy);
    } else {
    // --- greedy chunking -----------------------------------------------------------------
    // The point range is cut into up to CHUNK_THREADS parts (balanced by observations) that are chunked independently on
    // host threads and concatenated: a chunk never spans a seam, nothing else differs from one sequential sweep.  The
    // number of parts depends on the problem size only, not on the machine, so the chunking -- and with it the order of
    // every floating-point sum of the engine -- is the same everywhere.
    struct Part {
        std::vector<int32_t> ck_obs, ck_pt, ck_cam, ck_cam_list, ck_pair, ck_pgrp;
        std::vector<int64_t> ck_pe, keys;
        std::vector<uint32_t> pair_entries;
        int64_t n_pair_real = 0, n_direct_pairs = 0, n_direct_cam_obs = 0;
        int max_chunk_cams = 0, max_chunk_pairs = 0;
    };
    auto chunk_range = [&](const int p0, const int p1, Part &P) {
        std::vector<int32_t> cam_stamp(std::max(s.n_free, 1), -1), cam_slot(std::max(s.n_free, 1), 0);
        // local pair ids of the open chunk: a (slot, slot) table for cameras with an LDS slot, a hash map for the rest
        std::unordered_map<int64_t, int32_t> pair_map;
        std::vector<int32_t> pair_lut((size_t)CHUNK_CAMS_WIDE * CHUNK_CAMS_WIDE, -1), lut_touched;
        std::vector<int32_t> pair_cnt;  // entries per local pair of the open chunk
        std::vector<uint32_t> sorted;   // scratch of close_chunk
        std::vector<int32_t> pair_off;
        int cur_groups = 0, cur_pairs = 0;
        auto find_pair = [&](int fa, int fc, int chunk_now) -> int32_t {
            if (cam_stamp[fa] != chunk_now || cam_stamp[fc] != chunk_now) return -1;
            const int sa = cam_slot[fa], sc = cam_slot[fc];
            if (sa >= 0 && sc >= 0) return pair_lut[(size_t)sa * CHUNK_CAMS_WIDE + sc];
            auto it = pair_map.find(((int64_t)fa << 32) | (uint32_t)fc);
            return it == pair_map.end() ? -1 : it->second;
        };
        std::vector<int64_t> &ck_pair_key = P.keys;  // parallel to ck_pair_slot
        P.ck_obs.push_back((int32_t)pstart[p0]);
        P.ck_pt.push_back(p0);
        P.ck_cam.push_back(0);
        P.ck_pair.push_back(0);
        P.ck_pe.push_back(0);
        int chunk = 0, cur_obs = 0, cur_pts = 0, cur_cams = 0;
        int64_t obs_base = pstart[p0];
        {
            double est = 0;  // pair entries: sum of L (L + 1) / 2, plus the padding of the groups
            for (int i = p0; i < p1; ++i) {
                const double L = (double)(pstart[i + 1] - pstart[i]);
                est += 0.5 * L * (L + 1);
            }
            P.pair_entries.reserve((size_t)(est * 1.25) + 4096);
        }
        auto close_chunk = [&](int next_pt, int64_t next_obs) {
            P.ck_obs.push_back((int32_t)next_obs);
            P.ck_pt.push_back(next_pt);
            P.ck_cam.push_back((int32_t)P.ck_cam_list.size());
            P.ck_pair.push_back((int32_t)ck_pair_key.size());
            // Entries of a chunk, sorted by local pair, are cut into GROUPS of PAIR_RUN entry words of ONE camera pair
            // (null-padded).  Lane pair g of the build kernel reduces group g to one 6x6 partial block in registers; the
            // partial blocks of a pair (groups ck_pgrp[l] .. ck_pgrp[l+1]) are summed when the chunk is flushed, so the
            // Schur phase needs no LDS atomics and no LDS tile per pair.  Groups beyond the first CHUNK_OBS / 2 of a
            // chunk (second "round": long tracks, many small pairs) go to HBM with atomics straight from registers.
            {
                const size_t b0 = (size_t)P.ck_pe.back(), ne = P.pair_entries.size() - b0;
                // counting sort by local pair (stable: entries of a pair stay in point order)
                pair_off.assign(cur_pairs + 1, 0);
                for (int l = 0; l < cur_pairs; ++l) pair_off[l + 1] = pair_off[l] + pair_cnt[l];
                sorted.resize(ne);
                for (size_t k = 0; k < ne; ++k) {
                    const uint32_t ent = P.pair_entries[b0 + k];
                    sorted[pair_off[ent >> 16]++] = ent;
                }
                P.n_pair_real += (int64_t)ne;
                P.pair_entries.resize(b0);
                int32_t n_groups = 0;
                size_t i = 0;
                for (int l = 0; l < cur_pairs; ++l) {
                    const size_t j = i + pair_cnt[l];
                    P.ck_pgrp.push_back(n_groups);
                    // The pair's entries (in point order) are DEALT over the lanes of its groups -- lane 2 g + h of the build
                    // kernel takes words 4 h .. 4 h + 3 of group g -- so that at every step neighbouring lanes read the T rows
                    // of neighbouring points: rows a track length apart fall into different LDS banks, where a contiguous cut
                    // (eight consecutive points per group) put the lanes of one pair 8 tracks apart = the same banks.
                    const size_t cnt = pair_cnt[l], G = (cnt + PAIR_RUN - 1) / PAIR_RUN, slots = 2 * G;
                    const size_t w0 = P.pair_entries.size();
                    P.pair_entries.resize(w0 + G * PAIR_RUN, PAIR_NULL);
                    for (size_t k = 0; k < cnt; ++k) {
                        const size_t slot = k % slots, step = k / slots;  // step < PAIR_RUN / 2
                        P.pair_entries[w0 + (slot >> 1) * PAIR_RUN + (slot & 1) * (PAIR_RUN / 2) + step] = sorted[i + k];
                    }
                    n_groups += (int32_t)G;
                    i = j;
                }
                P.ck_pgrp.push_back(n_groups);
                const size_t per_round = CHUNK_OBS / 2;
                const size_t rounds = ((size_t)n_groups + per_round - 1) / per_round;
                P.pair_entries.resize(b0 + rounds * per_round * PAIR_RUN, PAIR_NULL);
                if (rounds > 1) P.n_direct_pairs += n_groups - (int32_t)per_round;
            }
            P.ck_pe.push_back((int64_t)P.pair_entries.size());
            P.max_chunk_cams = std::max(P.max_chunk_cams, cur_cams);
            P.max_chunk_pairs = std::max(P.max_chunk_pairs, cur_pairs);
            pair_map.clear();
            for (int32_t t : lut_touched) pair_lut[t] = -1;
            lut_touched.clear();
            pair_cnt.clear();
            cur_groups = 0;
            cur_pairs = 0;
            ++chunk;
            cur_obs = cur_pts = cur_cams = 0;
            obs_base = next_obs;
        };
        std::vector<int32_t> fidx;
        for (int i = p0; i < p1; ++i) {
            const int64_t b = pstart[i], e = pstart[i + 1];
            const int L = (int)(e - b);
            fidx.resize(L);
            for (int a = 0; a < L; ++a) fidx[a] = s.cam_free[s.o_cam[b + a]];
            bool cut = cur_obs + L > CHUNK_OBS || cur_pts + 1 > CHUNK_PTS;
            if (!cut && deterministic && cur_cams + L > CHUNK_CAMS_WIDE) {
                // (atomic-free build: every camera of a chunk needs a local slot)
                int newc = 0;
                for (int a = 0; a < L; ++a) {
                    if (fidx[a] < 0) continue;
                    bool seen = cam_stamp[fidx[a]] == chunk;
                    for (int a2 = 0; a2 < a && !seen; ++a2) seen = fidx[a2] == fidx[a];
                    if (!seen) ++newc;
                }
                if (cur_cams + newc > CHUNK_CAMS_WIDE) cut = true;
            }
            if (!cut && cur_obs >= CHUNK_OBS / 4) {
                // soft cut: would this point overflow the LDS tiles?
                int newc = 0, newp = 0;
                for (int a = 0; a < L; ++a) {
                    if (fidx[a] < 0) continue;
                    bool seen = cam_stamp[fidx[a]] == chunk;
                    for (int a2 = 0; a2 < a && !seen; ++a2) seen = fidx[a2] == fidx[a];
                    if (!seen) ++newc;
                    for (int c = a; c < L; ++c) {
                        if (fidx[c] < 0) continue;
                        const int32_t lp = find_pair(fidx[a], fidx[c], chunk);  // (duplicates inside the point over-count: harmless)
                        if (lp < 0 || pair_cnt[lp] % PAIR_RUN == 0) ++newp;  // opens a new group
                    }
                }
                if (cur_cams + newc > CHUNK_CAMS || cur_groups + newp > CHUNK_OBS / 2) cut = true;
            }
            if (cut && cur_pts > 0) close_chunk(i, b);
            for (int a = 0; a < L; ++a) {
                int f = fidx[a];
                int16_t slot = -2;
                if (f >= 0) {
                    if (cam_stamp[f] != chunk) {
                        cam_stamp[f] = chunk;
                        if (cur_cams < CHUNK_CAMS_WIDE) {
                            cam_slot[f] = cur_cams++;
                            P.ck_cam_list.push_back(f);
                        } else {
                            cam_slot[f] = -1;
                        }
                    }
                    slot = (int16_t)cam_slot[f];
                    if (slot < 0) {
                        slot = -1;
                        P.n_direct_cam_obs++;
                    }
                }
                s.o_lcam[b + a] = slot;
            }
            for (int a = 0; a < L; ++a) {
                if (fidx[a] < 0) continue;
                for (int c = a; c < L; ++c) {
                    if (fidx[c] < 0) continue;
                    int32_t lp = find_pair(fidx[a], fidx[c], chunk);
                    if (lp < 0) {
                        lp = cur_pairs++;
                        const int64_t key = ((int64_t)fidx[a] << 32) | (uint32_t)fidx[c];
                        const int sa = cam_slot[fidx[a]], sc = cam_slot[fidx[c]];
                        if (sa >= 0 && sc >= 0) {
                            pair_lut[(size_t)sa * CHUNK_CAMS_WIDE + sc] = lp;
                            lut_touched.push_back(sa * CHUNK_CAMS_WIDE + sc);
                        } else {
                            pair_map.emplace(key, lp);
                        }
                        ck_pair_key.push_back(key);
                    }
                    const uint32_t oi = (uint32_t)(b + a - obs_base), oj = (uint32_t)(b + c - obs_base);
                    if ((size_t)lp >= pair_cnt.size()) pair_cnt.resize(lp + 1, 0);
                    P.pair_entries.push_back(oi | (oj << 8) | ((uint32_t)lp << 16));
                    if (pair_cnt[lp]++ % PAIR_RUN == 0) ++cur_groups;
                    if (c != a && fidx[c] == fidx[a]) {  // same camera observing the point twice
                        P.pair_entries.push_back(oj | (oi << 8) | ((uint32_t)lp << 16));
                        if (pair_cnt[lp]++ % PAIR_RUN == 0) ++cur_groups;
                    }
                }
            }
            cur_obs += L;
            cur_pts += 1;
        }
        if (cur_pts > 0) close_chunk(p1, pstart[p1]);
    };
    constexpr int CHUNK_THREADS = 8;
    const int n_parts = std::max(1, std::min(CHUNK_THREADS, s.n_pts_chunked / 4000));   // (the reference's own sizes: 4-5 parts)
    std::vector<Part> parts(n_parts);
    {
        std::vector<int> cut(n_parts + 1, 0);
        cut[n_parts] = s.n_pts_chunked;
        const int64_t total = pstart[s.n_pts_chunked];
        for (int t = 1; t < n_parts; ++t)
            cut[t] = (int)(std::lower_bound(pstart.begin(), pstart.begin() + s.n_pts_chunked + 1, total * t / n_parts) - pstart.begin());
        for (int t = 1; t <= n_parts; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
        if (n_parts == 1) {
            chunk_range(cut[0], cut[1], parts[0]);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_parts; ++t) th.emplace_back([&, t] { chunk_range(cut[t], cut[t + 1], parts[t]); });
            for (auto &x : th) x.join();
        }
    }
    tick("chunking (parts)");
    std::vector<int64_t> ck_pair_key;
    s.ck_obs.push_back(0);
    s.ck_pt.push_back(0);
    s.ck_cam.push_back(0);
    s.ck_pair.push_back(0);
    s.ck_pe.push_back(0);
    {
        size_t n_pe = 0, n_keys = 0, n_cl = 0, n_pg = 0, n_ck = 0;
        for (const Part &P : parts) { n_pe += P.pair_entries.size(); n_keys += P.keys.size(); n_cl += P.ck_cam_list.size(); n_pg += P.ck_pgrp.size(); n_ck += P.ck_obs.size(); }
        s.pair_entries.reserve(n_pe + (size_t)(CHUNK_OBS / 2) * PAIR_RUN);
        ck_pair_key.reserve(n_keys); s.ck_cam_list.reserve(n_cl); s.ck_pgrp.reserve(n_pg);
        s.ck_obs.reserve(n_ck); s.ck_pt.reserve(n_ck); s.ck_cam.reserve(n_ck); s.ck_pair.reserve(n_ck); s.ck_pe.reserve(n_ck);
    }
    for (Part &P : parts) {
        const int32_t cl0 = (int32_t)s.ck_cam_list.size(), k0 = (int32_t)ck_pair_key.size();
        const int64_t pe0 = (int64_t)s.pair_entries.size();
        for (size_t c = 1; c < P.ck_obs.size(); ++c) {
            s.ck_obs.push_back(P.ck_obs[c]);
            s.ck_pt.push_back(P.ck_pt[c]);
            s.ck_cam.push_back(P.ck_cam[c] + cl0);
            s.ck_pair.push_back(P.ck_pair[c] + k0);
            s.ck_pe.push_back(P.ck_pe[c] + pe0);
        }
        s.ck_cam_list.insert(s.ck_cam_list.end(), P.ck_cam_list.begin(), P.ck_cam_list.end());
        ck_pair_key.insert(ck_pair_key.end(), P.keys.begin(), P.keys.end());
        s.ck_pgrp.insert(s.ck_pgrp.end(), P.ck_pgrp.begin(), P.ck_pgrp.end());
        s.pair_entries.insert(s.pair_entries.end(), P.pair_entries.begin(), P.pair_entries.end());
        s.n_pair_real += P.n_pair_real;
        s.n_direct_pairs += P.n_direct_pairs;
        s.n_direct_cam_obs += P.n_direct_cam_obs;
        s.max_chunk_cams = std::max(s.max_chunk_cams, P.max_chunk_cams);
        s.max_chunk_pairs = std::max(s.max_chunk_pairs, P.max_chunk_pairs);
        P = Part();
    }
    s.raw_keys.swap(ck_pair_key);
    tick("chunking");
    }   // (chunk mode)
    // --- heavy points (more observations than a chunk holds): one workgroup each (k_build_heavy), every (a <= c)
    // pair of the point's observations gets its block slot from a triangular list
    // (the deterministic chunk mode keeps a triangular key list per heavy point for k_build_heavy's slab rows; every other
    // mode takes the heavy points' pairs through the global entry lists: k_lin_heavy + k_rows_*)
    const bool heavy_lists = s.deterministic && !s.rows;
    s.hv_obs.push_back(pstart[s.n_pts_chunked]);
    s.hv_pair.push_back(0);
    for (int i = s.n_pts_chunked; i < s.n_pts_chunked + s.n_heavy; ++i) {
        const int64_t b = pstart[i], e = pstart[i + 1];
        const int L = (int)(e - b);
        for (int a = 0; a < L; ++a) {
            const int fa = s.cam_free[s.o_cam[b + a]];
            s.o_lcam[b + a] = fa < 0 ? -2 : -1;
            for (int c = a; heavy_lists && c < L; ++c) {
                const int fc = s.cam_free[s.o_cam[b + c]];
                int64_t key = -1;
                if (fa >= 0 && fc >= 0) {
                    key = ((int64_t)std::min(fa, fc) << 32) | (uint32_t)std::max(fa, fc);
                    if (fa > fc) key |= HV_KEY_TRANSPOSED;
                    if (fa == fc && a != c) key |= HV_KEY_TWICE;
                }
                s.hv_keys.push_back(key);
            }
        }
        s.hv_obs.push_back(e);
        s.hv_pair.push_back((int64_t)s.hv_keys.size());
    }
    if (!s.rows && !heavy_lists && s.n_heavy > 0) build_row_lists(s.n_pts_chunked, s.n_pts_chunked + s.n_heavy);
    tick("heavy points");
    if (!s.rows) {
        long long groups = 0, pairs = 0;
        for (size_t c = 0; c + 1 < s.ck_pair.size(); ++c) {
            const int np_ = s.ck_pair[c + 1] - s.ck_pair[c];
            groups += s.ck_pgrp[s.ck_pair[c] + c + np_];
            pairs += np_;
        }
        s.n_groups = groups;
        if (tm_on) fprintf(stderr, "[symbolic] chunks %zu, pairs %lld, groups %lld, real entries %lld (fill %.1f %% of the groups, %.1f %% of the padded rounds)\n",
                s.ck_obs.size() - 1, pairs, groups, (long long)s.n_pair_real, 100.0 * s.n_pair_real / (groups * (double)PAIR_RUN),
                100.0 * s.n_pair_real / (double)(s.pair_entries.size() - (size_t)(CHUNK_OBS / 2) * PAIR_RUN));
    }
    // (the build kernel prefetches one round of entries unconditionally: readable padding behind the last chunk)
    s.pair_entries.resize(s.pair_entries.size() + (size_t)(CHUNK_OBS / 2) * PAIR_RUN, PAIR_NULL);
    // slots are resolved by finalize_pattern (the pattern may be a union over ranks)
    s.ck_pair_slot.assign(s.raw_keys.size(), -1);
    s.mean_track = s.n_pts_obs ? track_sum / s.n_pts_obs : 0.0;
    return BA_OK;
}

void finalize_pattern(Symbolic &s, const std::vector<int64_t> &extra_keys) {
    const std::vector<int64_t> &raw = s.raw_keys;  // per-chunk keys, same order as ck_pair_slot
    std::vector<int64_t> keys(raw);
    keys.insert(keys.end(), extra_keys.begin(), extra_keys.end());
    for (int64_t k : s.hv_keys)
        if (k >= 0) keys.push_back(k & HV_KEY_MASK);
    keys.insert(keys.end(), s.rw_keys.begin(), s.rw_keys.end());
    for (int f = 0; f < s.n_free; ++f) keys.push_back(((int64_t)f << 32) | (uint32_t)f);  // every diagonal block
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    s.upper_keys = keys;
    for (size_t k = 0; k < raw.size(); ++k)
        s.ck_pair_slot[k] = (int32_t)(std::lower_bound(keys.begin(), keys.end(), raw[k]) - keys.begin());
    s.rw_slot.resize(s.rw_keys.size());
    for (size_t k = 0; k < s.rw_keys.size(); ++k)
        s.rw_slot[k] = (int32_t)(std::lower_bound(keys.begin(), keys.end(), s.rw_keys[k]) - keys.begin());
    s.hv_pair_slot.assign(s.hv_keys.size(), -1);
    for (size_t k = 0; k < s.hv_keys.size(); ++k) {
        const int64_t key = s.hv_keys[k];
        if (key < 0) continue;
        int32_t slot = (int32_t)(std::lower_bound(keys.begin(), keys.end(), key & HV_KEY_MASK) - keys.begin());
        if (key & HV_KEY_TRANSPOSED) slot |= HV_SLOT_TRANSPOSED;
        if (key & HV_KEY_TWICE) slot |= HV_SLOT_TWICE;
        s.hv_pair_slot[k] = slot;
    }
    // full CSR view
    const int nf = s.n_free;
    s.diag_slot.assign(nf, -1);
    s.band = 0;
    for (size_t k = 0; k < keys.size(); ++k)
        s.band = std::max(s.band, (int)(keys[k] & 0xffffffff) - (int)(keys[k] >> 32));
    std::vector<int32_t> rowcnt(nf + 1, 0);
    for (size_t k = 0; k < keys.size(); ++k) {
        int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffff);
        rowcnt[i]++;
        if (i != j) rowcnt[j]++;
        else s.diag_slot[i] = (int32_t)k;
    }
    s.bsr_row.assign(nf + 1, 0);
    for (int i = 0; i < nf; ++i) s.bsr_row[i + 1] = s.bsr_row[i] + rowcnt[i];
    s.bsr_col.assign(s.bsr_row[nf], 0);
    s.bsr_src.assign(s.bsr_row[nf], 0);
    std::vector<int32_t> fill(s.bsr_row.begin(), s.bsr_row.end() - 1);
    // lower entries of row j come from keys (i<j, j) which sort by i: emit them first, then upper
    for (size_t k = 0; k < keys.size(); ++k) {
        int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffff);
        if (i != j) {
            int pos = fill[j]++;
            s.bsr_col[pos] = i;
            s.bsr_src[pos] = (int32_t)((uint32_t)k | 0x80000000u);
        }
    }
    for (size_t k = 0; k < keys.size(); ++k) {
        int i = (int)(keys[k] >> 32), j = (int)(keys[k] & 0xffffffff);
        int pos = fill[i]++;
        s.bsr_col[pos] = j;
        s.bsr_src[pos] = (int32_t)k;
    }
}

// Contribution tables of the atomic-free build (ba_symbolic.h); after finalize_pattern.
void build_det_tables(const Symbolic &s, DetTables &t) {
    t = DetTables();
    const int n_chunks = (int)s.ck_obs.size() - 1;
    const int half = CHUNK_OBS / 2;
    t.n_cl = (int64_t)s.ck_cam_list.size();
    t.n_raw = (int64_t)s.raw_keys.size();
    t.ck_lg.assign(n_chunks + 1, 0);
    for (int c = 0; c < n_chunks; ++c) {
        const int np_ = s.ck_pair[c + 1] - s.ck_pair[c];
        const int n_groups = s.ck_pgrp[s.ck_pair[c] + c + np_];
        t.ck_lg[c + 1] = t.ck_lg[c] + std::max(0, n_groups - half);
    }
    t.n_lg = t.ck_lg[n_chunks];
    const int64_t n_hobs = s.hv_obs.empty() ? 0 : s.hv_obs.back() - s.hv_obs.front();
    t.cam_rows = t.n_cl + n_hobs;
    t.blk_rows = t.n_raw + t.n_lg + (int64_t)s.hv_keys.size();
    // ---- cameras
    const int nf = s.n_free;
    t.c_ptr.assign(nf + 1, 0);
    for (int64_t r = 0; r < t.n_cl; ++r) t.c_ptr[s.ck_cam_list[r] + 1]++;
    for (int64_t o = 0; o < n_hobs; ++o) {
        const int f = s.cam_free[s.o_cam[s.hv_obs.front() + o]];
        if (f >= 0) t.c_ptr[f + 1]++;
    }
    for (int f = 0; f < nf; ++f) t.c_ptr[f + 1] += t.c_ptr[f];
    t.c_idx.resize(t.c_ptr[nf]);
    {
        std::vector<int32_t> fill(t.c_ptr.begin(), t.c_ptr.end() - 1);
        for (int64_t r = 0; r < t.n_cl; ++r) t.c_idx[fill[s.ck_cam_list[r]]++] = (int32_t)r;
        for (int64_t o = 0; o < n_hobs; ++o) {
            const int f = s.cam_free[s.o_cam[s.hv_obs.front() + o]];
            if (f >= 0) t.c_idx[fill[f]++] = (int32_t)(t.n_cl + o);
        }
    }
    // ---- blocks: (destination slot, slab row | transposed flag) in ascending row order per slot
    const size_t nu = s.upper_keys.size();
    t.s_ptr.assign(nu + 1, 0);
    auto each = [&](auto &&emit) {
        for (int c = 0; c < n_chunks; ++c) {
            const int qb = s.ck_pair[c], np_ = s.ck_pair[c + 1] - qb;
            const int32_t *pg = &s.ck_pgrp[qb + c];
            for (int l = 0; l < np_ && l < half; ++l)
                if (pg[l] < std::min(pg[l + 1], half)) emit(s.ck_pair_slot[qb + l], (uint32_t)(qb + l));
        }
        for (int c = 0; c < n_chunks; ++c) {
            const int qb = s.ck_pair[c], np_ = s.ck_pair[c + 1] - qb;
            const int32_t *pg = &s.ck_pgrp[qb + c];
            for (int l = 0; l < np_; ++l)
                for (int g = std::max(pg[l], half); g < pg[l + 1]; ++g)
                    emit(s.ck_pair_slot[qb + l], (uint32_t)(t.n_raw + t.ck_lg[c] + (g - half)));
        }
        for (size_t k = 0; k < s.hv_pair_slot.size(); ++k) {
            const int32_t sw = s.hv_pair_slot[k];
            if (sw < 0) continue;
            const uint32_t row = (uint32_t)(t.n_raw + t.n_lg + (int64_t)k);
            const bool tr = (sw & HV_SLOT_TRANSPOSED) != 0, twice = (sw & HV_SLOT_TWICE) != 0;
            if (twice) {   // one camera observing the point twice: both orders land in its diagonal block
                emit(sw & HV_SLOT_MASK, row);
                emit(sw & HV_SLOT_MASK, row | 0x80000000u);
            } else {
                emit(sw & HV_SLOT_MASK, tr ? (row | 0x80000000u) : row);
            }
        }
    };
    each([&](int32_t slot, uint32_t) { t.s_ptr[slot + 1]++; });
    for (size_t k = 0; k < nu; ++k) t.s_ptr[k + 1] += t.s_ptr[k];
    t.s_idx.resize(t.s_ptr[nu]);
    std::vector<int32_t> fill(t.s_ptr.begin(), t.s_ptr.end() - 1);
    each([&](int32_t slot, uint32_t row) { t.s_idx[fill[slot]++] = row; });
}

}  // namespace ba
