// Developer: the HOST-only entry points (ba_analyze = the symbolic analysis, ba_bcr_plan = the cyclic-reduction planner) under
// AddressSanitizer + UBSan on random inputs.  No GPU call is made.  Built and run by tools/host_asan.sh (CPU container).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "../include/ba_hip.h"
#include "../structure_from_motion_amd/csrc/ba_symbolic.h"

int main(int argc, char **argv) {
    const int n_cases = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937_64 rng(12345);
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    int plans = 0, analyses = 0, rows_analyses = 0, rejected = 0;
    for (int cs = 0; cs < n_cases; ++cs) {
        // ---- a random problem: chain / ring / random visibility, odd sizes, unobserved points, fixed cameras anywhere
        const int nc = uni(1, cs % 7 == 0 ? 900 : 120), np = uni(1, 3000), kind = uni(0, 3);
        std::vector<double> pose((size_t)nc * 6, 0.1), intr((size_t)nc * 4, 500.0), pts((size_t)np * 3, 1.0), xy;
        std::vector<uint8_t> fixed(nc, 0);
        for (int i = 0; i < nc; ++i) fixed[i] = (uni(0, 9) == 0) || i == 0;
        if (cs % 11 == 0) std::fill(fixed.begin(), fixed.end(), 1);   // no free camera at all
        std::vector<int32_t> oc, op;
        for (int p = 0; p < np; ++p) {
            if (uni(0, 19) == 0) continue;   // unobserved point
            int L = kind == 3 && uni(0, 200) == 0 ? uni(257, 400) : uni(1, 12);   // (a few heavy points)
            if (L > nc) L = nc;
            std::set<int> cams;
            const int start = uni(0, nc - 1);
            for (int a = 0; a < L; ++a) {
                int c = kind == 0 ? std::min(start + a, nc - 1) : kind == 1 ? (start + a) % nc : uni(0, nc - 1);
                if (kind == 3 && L > 256) c = a % nc;
                if (kind != 3 || L <= 256) { if (!cams.insert(c).second && uni(0, 3)) continue; }   // (duplicates allowed sometimes)
                oc.push_back(c); op.push_back(p);
            }
        }
        // camera-major order as the reference marshals it, or shuffled
        if (cs % 3) {
            std::vector<size_t> idx(oc.size());
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
            std::shuffle(idx.begin(), idx.end(), rng);
            std::vector<int32_t> a(oc.size()), b(oc.size());
            for (size_t i = 0; i < idx.size(); ++i) { a[i] = oc[idx[i]]; b[i] = op[idx[i]]; }
            oc.swap(a); op.swap(b);
        }
        xy.assign(oc.size() * 2 + 2, 3.0);
        if (cs % 13 == 5 && !oc.empty()) oc[uni(0, (int)oc.size() - 1)] = nc + uni(0, 3);   // bad index: must be refused
        BaProblem P{};
        P.n_cams = nc; P.n_points = np; P.n_obs = (int64_t)oc.size();
        P.cam_pose = pose.data(); P.cam_intr = intr.data(); P.cam_fixed = fixed.data(); P.points = pts.data();
        P.obs_cam = oc.data(); P.obs_pt = op.data(); P.obs_xy = xy.data(); P.memory = BA_MEM_HOST;
        BaAnalysis A{};
        const int e = ba_analyze(&P, &A);
        if (e == BA_OK) ++analyses; else ++rejected;
        if (e == BA_OK) {
            // the global per-block entry lists of the analysis (rows mode: every point; chunk mode: the heavy points only):
            // every entry inside its block, blocks ascending, slots resolved
            for (int rows = 0; rows < 2; ++rows) {
                ba::Symbolic s;
                const char *msg = "";
                if (ba::analyze_structure(nc, np, (int64_t)oc.size(), fixed.data(), oc.data(), op.data(), s, &msg, false, rows != 0) != BA_OK) { fprintf(stderr, "rows analysis refused: %s\n", msg); return 1; }
                ba::finalize_pattern(s, {});
                if (s.rw_keys.empty() && s.rw_ptr.empty()) continue;   // (chunk mode without heavy points)
                if (s.rw_ptr.size() != s.rw_keys.size() + 1 || s.rw_slot.size() != s.rw_keys.size() || (int64_t)s.rw_ent.size() != 2 * s.rw_ptr.back()) { fprintf(stderr, "rows tables: sizes\n"); return 1; }
                int64_t want = 0;   // entries the lists must hold: pairs (a <= c) of free observations of the covered points
                for (int64_t o = 0; o < (int64_t)oc.size(); ++o) {
                    if (s.cam_free[s.o_cam[o]] < 0 || (!rows && s.o_pt[o] < s.n_pts_chunked)) continue;
                    for (int64_t o2 = o; o2 < (int64_t)oc.size() && s.o_pt[o2] == s.o_pt[o]; ++o2)
                        if (s.cam_free[s.o_cam[o2]] >= 0) want += (o2 != o && s.o_cam[o2] == s.o_cam[o]) ? 2 : 1;
                }
                if (want != s.rw_ptr.back()) { fprintf(stderr, "rows tables: %lld entries, expected %lld\n", (long long)s.rw_ptr.back(), (long long)want); return 1; }
                for (size_t k = 0; k < s.rw_keys.size(); ++k) {
                    if (k && s.rw_keys[k] <= s.rw_keys[k - 1]) { fprintf(stderr, "rows tables: keys not ascending\n"); return 1; }
                    if (s.rw_ptr[k + 1] <= s.rw_ptr[k]) { fprintf(stderr, "rows tables: empty block\n"); return 1; }
                    const int fi = (int)(s.rw_keys[k] >> 32), fj = (int)(s.rw_keys[k] & 0xffffffff);
                    if (s.upper_keys[s.rw_slot[k]] != s.rw_keys[k]) { fprintf(stderr, "rows tables: slot\n"); return 1; }
                    for (int64_t q = s.rw_ptr[k]; q < s.rw_ptr[k + 1]; ++q) {
                        const int32_t oi = s.rw_ent[2 * q], oj = s.rw_ent[2 * q + 1];
                        if (oi < 0 || oj < 0 || oi >= (int64_t)oc.size() || oj >= (int64_t)oc.size() || s.o_pt[oi] != s.o_pt[oj] ||
                            s.cam_free[s.o_cam[oi]] != fi || s.cam_free[s.o_cam[oj]] != fj || (!rows && s.o_pt[oi] < s.n_pts_chunked)) { fprintf(stderr, "rows tables: entry\n"); return 1; }
                    }
                }
            }
            ++rows_analyses;
        }
        // ---- a random pattern for the planner: band / ring / band + chords / random, with a random relabelling
        const int nf = uni(1, cs % 5 == 0 ? 2500 : 400), band = uni(0, 26), pk = uni(0, 3);
        std::vector<int> lab(nf);
        for (int i = 0; i < nf; ++i) lab[i] = i;
        if (cs % 2) std::shuffle(lab.begin(), lab.end(), rng);
        std::set<std::pair<int, int>> pr;
        for (int i = 0; i < nf; ++i) {
            pr.insert({lab[i], lab[i]});
            for (int d = 1; d <= band; ++d) {
                if (uni(0, 9) == 0) continue;
                int j = i + d;
                if (j >= nf) { if (pk != 1) continue; j %= nf; }
                if (j == i) continue;
                pr.insert({std::min(lab[i], lab[j]), std::max(lab[i], lab[j])});
            }
        }
        if (pk == 2) for (int k = 0; k < 3; ++k) { int a = uni(0, nf - 1), b = uni(0, nf - 1); pr.insert({std::min(a, b), std::max(a, b)}); }
        if (pk == 3) for (int k = 0; k < nf * 4; ++k) { int a = uni(0, nf - 1), b = uni(0, nf - 1); pr.insert({std::min(a, b), std::max(a, b)}); }
        std::vector<int32_t> pairs;
        for (auto &q : pr) { pairs.push_back(q.first); pairs.push_back(q.second); }
        int64_t need = 0;
        if (ba_bcr_plan(nf, pairs.data(), (int64_t)pr.size(), nullptr, 0, &need) != BA_OK) { ++rejected; continue; }
        std::vector<int32_t> out((size_t)need);
        if (ba_bcr_plan(nf, pairs.data(), (int64_t)pr.size(), out.data(), need, &need) != BA_OK) { ++rejected; continue; }
        plans += out[0];
    }
    printf("cases %d: analyses ok %d (rows mode checked %d), refused %d, patterns with a schedule %d\n", n_cases, analyses, rows_analyses, rejected, plans);
    return 0;
}
